// Development: does a v_mfma_f32_32x32x16_f16 overlap with the vector instructions that follow it in the same wave?  One MFMA per N vector
// instructions of one kind (random operands: realistic switching activity), against the MFMAs alone and the vector instructions alone.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND: 0 v_mul_f32, 1 v_fma_mix_f32, 2 v_cvt_pk_f16_f32, 3 v_exp_f32, 4 v_fma_f32 (3 distinct sources), 5 v_sub_f32
// WHAT: 0 both, 1 MFMA only, 2 vector only.   N = vector instructions per MFMA (64 vector instructions per iteration)
template <int KIND, int WHAT, int N>
__global__ __launch_bounds__(1024) void k(float *out, int iters, const float *rnd) {
    f32x16 acc[2];
    float v[32];
    f16x8 a, b;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        a[c] = (_Float16)rnd[(threadIdx.x * 8 + c) & 4095];
        b[c] = (_Float16)rnd[(threadIdx.x * 8 + c + 2048) & 4095];
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = rnd[(threadIdx.x + 64 * i) & 4095];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = rnd[(threadIdx.x + r) & 4095];
    float av = 1.0f + rnd[threadIdx.x & 4095] * 1e-3f, bv = rnd[(threadIdx.x + 7) & 4095] * 1e-3f;
    asm volatile("" : "+v"(av), "+v"(bv), "+v"(a), "+v"(b));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (WHAT != 2 && i % N == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[(i / N) & 1]) : "v"(a), "v"(b));
            if (WHAT != 1) {
                float &x = v[i & 31];
                if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(av));
                if (KIND == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(x) : "v"(av), "v"(bv));
                if (KIND == 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(av));
                if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                if (KIND == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(v[(i + 1) & 31]), "v"(v[(i + 7) & 31]));
                if (KIND == 5) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(v[(i + 3) & 31]));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += v[i];
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static int g_blocks = 256;
template <int KIND, int WHAT, int N>
float run(float *out, const float *rnd, int threads) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, WHAT, N>), dim3(g_blocks), dim3(threads), 0, 0, out, 100, rnd);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, WHAT, N>), dim3(g_blocks), dim3(threads), 0, 0, out, iters, rnd);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters;   // ns per iteration (64 vector instructions and 64 / N MFMAs per wave)
}
template <int KIND, int N>
void line(float *out, const float *rnd, const char *what) {
    printf("  %-22s 1 MFMA per %2d:", what, N);
    for (int threads = 256; threads <= 1024; threads *= 2) {
        const float both = run<KIND, 0, N>(out, rnd, threads), m = run<KIND, 1, N>(out, rnd, threads), v = run<KIND, 2, N>(out, rnd, threads);
        printf("   %dw: both %6.1f  mfma %6.1f  vector %6.1f  (sum %6.1f)", threads / 256, both, m, v, m + v);
    }
    printf("\n");
}
int main(int argc, char **argv) {
    if (argc > 1) g_blocks = atoi(argv[1]);
    float *out, *rnd, h[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    (void)hipMalloc(&rnd, sizeof(h));
    (void)hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
    printf("ns per iteration of 64 vector instructions (+ 64/N v_mfma_f32_32x32x16_f16) per wave, %d workgroups (one per CU), 1 / 2 / 4 waves per SIMD\n", g_blocks);
    line<0, 16>(out, rnd, "v_mul_f32");
    line<0, 8>(out, rnd, "v_mul_f32");
    line<4, 16>(out, rnd, "v_fma_f32 (3 sources)");
    line<4, 8>(out, rnd, "v_fma_f32 (3 sources)");
    line<1, 16>(out, rnd, "v_fma_mix_f32");
    line<1, 8>(out, rnd, "v_fma_mix_f32");
    line<2, 16>(out, rnd, "v_cvt_pk_f16_f32");
    line<2, 8>(out, rnd, "v_cvt_pk_f16_f32");
    line<3, 16>(out, rnd, "v_exp_f32");
    line<3, 8>(out, rnd, "v_exp_f32");
    line<5, 16>(out, rnd, "v_sub_f32");
    return 0;
}
