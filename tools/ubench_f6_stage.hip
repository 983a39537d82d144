// Development: would the as-coded evaluation gain from carrying the LOW point plane (|l| <= 2^-12 |x|) on the MX-scaled FP6 path instead of
// seven more float16 MFMAs?  Stage-shaped loop per wave, every CU busy: the microbenchmark says yes (-12 % time, through the CLOCK: the cycles
// barely move), the kernel said no (profiles/r04_compat_eval_experiments.txt, section 4).
// stage-shaped loop: a block of MFMAs then 272 vector instructions of the as-coded epilogue's mix; A: 15 x 32x32x16 f16; B: 8 f16 + 2 scaled fp6 32x32x64; C: 8 f16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
struct Stamp { unsigned long long cyc, rt; };
__device__ __forceinline__ void vec16(float (&v)[32], float av, float bv, int base) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float &x = v[(base + i) & 31];
        const float y = v[(base + i + 5) & 31], z = v[(base + i + 11) & 31];
        const int k = i < 7 ? 0 : (i < 11 ? 1 : (i < 13 ? 3 : (i < 15 ? 4 : 5)));
        if (k == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
        if (k == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(x) : "v"(av), "v"(bv));
        if (k == 3) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(av));
        if (k == 4) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(y));
        if (k == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    }
}
template <int WHAT>
__global__ __launch_bounds__(1024) void k(float *out, Stamp *st, int iters, const float *rnd) {
    f32x16 acc;
    float v[32];
    f16x8 a, b;
    i32x8 a6, b6;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        a[c] = (_Float16)rnd[(threadIdx.x * 8 + c) & 4095];
        b[c] = (_Float16)rnd[(threadIdx.x * 8 + c + 2048) & 4095];
        a6[c] = (int)(rnd[(threadIdx.x * 8 + c + 100) & 4095] * 4e9f);
        b6[c] = (int)(rnd[(threadIdx.x * 8 + c + 300) & 4095] * 4e9f);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = rnd[(threadIdx.x + 64 * i) & 4095];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = rnd[(threadIdx.x + r) & 4095];
    float av = 1.0f + rnd[threadIdx.x & 4095] * 1e-3f, bv = rnd[(threadIdx.x + 7) & 4095] * 1e-3f;
    asm volatile("" : "+v"(av), "+v"(bv), "+v"(a), "+v"(b), "+v"(a6), "+v"(b6));
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        constexpr int NF16 = WHAT == 0 ? 15 : 8;
#pragma unroll
        for (int i = 0; i < NF16; ++i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        if (WHAT == 1) {
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a6, b6, acc, 2, 2, 0, 127, 0, 127);
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a6, b6, acc, 2, 2, 0, 127, 0, 127);
            asm volatile("" : "+v"(acc));
        }
#pragma unroll
        for (int q = 0; q < 17; ++q) vec16(v, av, bv, 16 * q);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += v[i];
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = {c1 - c0, r1 - r0};
}
template <int WHAT>
void run(float *out, Stamp *st, const float *rnd, int wps, const char *name) {
    const int iters = 4000, threads = 256 * wps;
    hipLaunchKernelGGL((k<WHAT>), dim3(256), dim3(threads), 0, 0, out, st, 200, rnd);
    hipLaunchKernelGGL((k<WHAT>), dim3(256), dim3(threads), 0, 0, out, st, iters, rnd);
    hipDeviceSynchronize();
    std::vector<Stamp> h(256 * 4 * wps);
    hipMemcpy(h.data(), st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> c, n;
    for (auto &t : h) { c.push_back((double)t.cyc / iters); n.push_back((double)t.rt * 10.0 / iters); }
    std::sort(c.begin(), c.end()); std::sort(n.begin(), n.end());
    printf("  %-44s %d waves/SIMD: %8.1f cycles %8.1f ns per stage  (%.2f GHz)\n", name, wps, c[c.size() / 2], n[n.size() / 2], c[c.size() / 2] / n[n.size() / 2]);
}
int main() {
    float *out, *rnd, h[4096];
    Stamp *st;
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&st, 256 * 16 * sizeof(Stamp)); hipMalloc(&rnd, sizeof(h));
    hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
    for (int wps = 3; wps <= 4; ++wps) {
        run<0>(out, st, rnd, wps, "15 f16 MFMAs + 272 vector");
        run<1>(out, st, rnd, wps, "8 f16 + 2 scaled fp6 32x32x64 + 272 vector");
        run<2>(out, st, rnd, wps, "8 f16 MFMAs + 272 vector");
    }
    return 0;
}
