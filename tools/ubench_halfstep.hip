// Microbenchmark: the inner half step of gp_eval_f16_kernel with register-only operands (no LDS, no global loads):
// 16 v_mfma_f32_16x16x32_f16 interleaved with 8 pair evaluations (20 VALU each).  BITS: 1 = MFMAs, 2 = epilogue.
// VAR: 0 as in the kernel (MFMA between the halves of a pair evaluation), 1 = all MFMAs first, then the epilogue,
//      2 = 32x32x16 MFMAs (8 instead of 16, same flops)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BITS, int VAR>
__global__ __launch_bounds__(512, 2) void k(float *out, const float *in, int iters, unsigned long long *clk) {
    const int t = threadIdx.x;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    h16x8 xh[2][4], xl[2][4], bh[4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                xh[p][s][c] = (_Float16)in[(t * 7 + p * 31 + s * 5 + c) & 1023];
                xl[p][s][c] = (_Float16)in[(t * 3 + p * 17 + s * 11 + c) & 1023];
                bh[s][c] = (_Float16)in[(t * 5 + s * 13 + c) & 1023];
            }
    float q[12], nx[2][4], sx[2][4], tx[2][4], au[2][4], at[2][4], ad[2][4], al[2][4];
#pragma unroll
    for (int i = 0; i < 12; ++i) q[i] = in[(t + i * 37) & 1023];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            nx[p][i] = in[(t + p * 4 + i) & 1023];
            sx[p][i] = in[(t + 100 + p * 4 + i) & 1023];
            tx[p][i] = in[(t + 200 + p * 4 + i) & 1023];
            au[p][i] = at[p][i] = ad[p][i] = al[p][i] = 0.0f;
        }
    float k1 = in[5] * 1e-3f, k2 = in[6] * 1e-3f;
    asm volatile("" : "+v"(k1), "+v"(k2));
    const f32x4 zero4 = {0, 0, 0, 0};
    f32x4 accC[2] = {zero4, zero4}, accN[2], accM[2];
    f32x16 big[2];
    for (int it = 0; it < iters; ++it) {
        accN[0] = accN[1] = accM[0] = accM[1] = zero4;
        auto mfma_one = [&](int m) {
            if (!(BITS & 1)) return;
            if (VAR == 2) {
                if (m & 1) return;
                const int s = m / 4, p = (m / 2) % 2;
                big[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[p][s], bh[s], s == 0 ? (f32x16){0} : big[p], 0, 0, 0);
                return;
            }
            const int s = m / 4, p = (m / 2) % 2, w = m % 2;
            if (w == 0) accN[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[p][s], bh[s], s == 0 ? zero4 : accN[p], 0, 0, 0);
            else accM[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[p][s], bh[s], s == 0 ? zero4 : accM[p], 0, 0, 0);
        };
        if (VAR == 1) {
#pragma unroll
            for (int m = 0; m < 16; ++m) mfma_one(m);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int p = e / 4, i = e % 4;
            if (VAR != 1) { mfma_one(2 * e); __builtin_amdgcn_sched_barrier(0); }
            if (BITS & 2) {
                const float L0 = accC[p][i] + nx[p][i];
                const float pp = tx[p][i] - q[1], ss = sx[p][i] - q[0];
                const float kap = __builtin_amdgcn_exp2f(fmaf(L0, k1, k2));
                const float L = fmaf(-pp, pp, L0);
                const float E = fmaf(q[5], ss, fmaf(q[4], pp, fmaf(q[3], L, q[2])));
                if (VAR != 1) { mfma_one(2 * e + 1); __builtin_amdgcn_sched_barrier(0); }
                au[p][i] = fmaf(kap, E, au[p][i]);
                at[p][i] = fmaf(kap, fmaf(-pp, E, q[6]), at[p][i]);
                ad[p][i] = fmaf(kap, fmaf(-ss, E, fmaf(q[7], ss, q[8])), ad[p][i]);
                al[p][i] = fmaf(kap, fmaf(L, E, fmaf(q[9], L, fmaf(q[10], ss, q[11]))), al[p][i]);
            } else {
                if (VAR != 1) { mfma_one(2 * e + 1); __builtin_amdgcn_sched_barrier(0); }
                au[p][i] += accC[p][i];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (VAR == 2) accC[p][i] = fmaf(big[p][i + 4], 0x1p-11f, big[p][i]);
                else accC[p][i] = fmaf(accM[p][i], 0x1p-11f, accN[p][i]);
            }
    }
    float s = 0;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) s += au[p][i] + at[p][i] + ad[p][i] + al[p][i];
    out[blockIdx.x * blockDim.x + t] = s;
    if (blockIdx.x == 7 && t == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - w0; }
}

static unsigned long long *g_clk;
template <int BITS, int VAR>
void run(float *out, const float *in, const char *what) {
    const int iters = 20000;
    printf("  %-44s", what);
    for (int blocks = 256; blocks <= 512; blocks *= 2) {   // 512 threads per block: 2 or 4 waves per SIMD
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL((k<BITS, VAR>), dim3(blocks), dim3(512), 0, 0, out, in, 10, g_clk);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<BITS, VAR>), dim3(blocks), dim3(512), 0, 0, out, in, iters, g_clk);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        (void)hipMemcpy(h, g_clk, 16, hipMemcpyDeviceToHost);
        printf("  %dw/SIMD: %7.3f ms, clock %4.0f MHz, %6.1f cyc/iter", blocks / 128, ms, h[0] / (h[1] / 100.0), (double)h[0] / iters);
    }
    printf("\n");
}

int main() {
    float *out, *in;
    (void)hipMalloc(&out, 512 * 512 * 4);
    (void)hipMalloc(&in, 4096);
    (void)hipMalloc(&g_clk, 16);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.0f - 0.5f;
    (void)hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    printf("per iteration = one half step by every resident wave; clock = s_memtime ticks per 100 MHz wall tick of one wave\n");
    run<1, 0>(out, in, "16 MFMA 16x16x32 only");
    run<2, 0>(out, in, "8 pair evaluations only");
    run<3, 0>(out, in, "interleaved (kernel order)");
    run<3, 1>(out, in, "MFMAs first, then epilogue");
    run<1, 2>(out, in, "8 MFMA 32x32x16 only");
    run<3, 2>(out, in, "interleaved, 32x32x16");
    return 0;
}
