// Development: which ingredient of the eval kernels' instruction stream keeps a 32x32x16 MFMA from running under the vector instructions
// that follow it?  One MFMA per 16 v_mul_f32, with (LDSA) the A operand of each MFMA read from LDS one step ahead, (CHAIN) all MFMAs
// accumulating into one register set, (EXPS) four of the sixteen vector instructions replaced by v_exp_f32, (DSRD) two extra LDS reads per step
// whose results the vector instructions consume.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WHAT, bool LDSA, bool CHAIN, bool EXPS, bool DSRD, int VK = 0>
__global__ __launch_bounds__(1024) void k(float *out, int iters, const float *rnd) {
    __shared__ f32x4 lds[16 * 64];
    f32x16 acc[2];
    float v[32];
    f16x8 a[2], b;
    for (int i = threadIdx.x; i < 16 * 64; i += blockDim.x) lds[i] = (f32x4){rnd[i & 4095], rnd[(i + 1) & 4095], rnd[(i + 2) & 4095], rnd[(i + 3) & 4095]};
    __syncthreads();
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        a[0][c] = a[1][c] = (_Float16)rnd[(threadIdx.x * 8 + c) & 4095];
        b[c] = (_Float16)rnd[(threadIdx.x * 8 + c + 2048) & 4095];
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = rnd[(threadIdx.x + 64 * i) & 4095];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = rnd[(threadIdx.x + r) & 4095];
    float av = 1.0f + rnd[threadIdx.x & 4095] * 1e-3f;
    asm volatile("" : "+v"(av), "+v"(b));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {          // 4 steps per iteration: 1 MFMA + 16 vector instructions each
            if (WHAT != 2) {
                if (LDSA) {
                    f32x4 t = lds[((it * 4 + s + 1) & 15) * 64 + lane];
                    a[(s + 1) & 1] = __builtin_bit_cast(f16x8, t);
                }
                f32x16 &d = acc[CHAIN ? 0 : (s & 1)];
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s & 1], b, d, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (WHAT != 1) {
                float c0 = av, c1 = av;
                if (DSRD) {
                    f32x4 t = lds[((it * 4 + s) & 15) * 64 + ((lane * 5) & 63)];
                    c0 = t.x;
                    c1 = t.z;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float &x = v[(s * 8 + i) & 31];
                    if (EXPS && i < 4) x = __builtin_amdgcn_exp2f(x);
                    else if (VK == 1) asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(x) : "v"(c0), "v"(c1));      // float32 x float16 + float32
                    else if (VK == 2) asm("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(c1));
                    else if (VK == 3) asm("v_exp_f32 %0, %0" : "+v"(x));
                    else x = x * ((i & 1) ? c1 : c0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float sum = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) sum += v[i];
    for (int r = 0; r < 16; ++r) sum += acc[0][r] + acc[1][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
static int g_blocks = 256;
template <int WHAT, bool LDSA, bool CHAIN, bool EXPS, bool DSRD, int VK = 0>
float run(float *out, const float *rnd, int threads) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<WHAT, LDSA, CHAIN, EXPS, DSRD, VK>), dim3(g_blocks), dim3(threads), 0, 0, out, 100, rnd);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<WHAT, LDSA, CHAIN, EXPS, DSRD, VK>), dim3(g_blocks), dim3(threads), 0, 0, out, iters, rnd);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters / 4;   // ns per step (1 MFMA + 16 vector instructions per wave)
}
template <bool LDSA, bool CHAIN, bool EXPS, bool DSRD, int VK = 0>
void line(float *out, const float *rnd, const char *what) {
    printf("  %-44s", what);
    for (int threads = 256; threads <= 1024; threads *= 2) {
        const float both = run<0, LDSA, CHAIN, EXPS, DSRD, VK>(out, rnd, threads), m = run<1, LDSA, CHAIN, EXPS, DSRD, VK>(out, rnd, threads),
                    v = run<2, LDSA, CHAIN, EXPS, DSRD, VK>(out, rnd, threads);
        printf("   %dw: both %5.1f mfma %5.1f vec %5.1f", threads / 256, both, m, v);
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
    printf("ns per step (1 v_mfma_f32_32x32x16_f16 + 16 vector instructions) per wave, %d workgroups (one per CU), 1 / 2 / 4 waves per SIMD\n", g_blocks);
    line<false, false, false, false>(out, rnd, "registers, two accumulators, v_mul");
    line<false, true, false, false>(out, rnd, "registers, ONE accumulator chain");
    line<true, false, false, false>(out, rnd, "A operand from LDS one step ahead");
    line<true, true, false, false>(out, rnd, "A from LDS, one chain");
    line<true, true, true, false>(out, rnd, "A from LDS, one chain, 4 of 16 are v_exp");
    line<true, true, false, true>(out, rnd, "A from LDS, one chain, vector operands from LDS");
    line<true, true, true, true>(out, rnd, "A from LDS, one chain, v_exp, operands from LDS");
    line<true, true, false, false, 1>(out, rnd, "A from LDS, one chain, 16 v_fma_mix_f32");
    line<true, true, false, false, 2>(out, rnd, "A from LDS, one chain, 16 v_cvt_pk_f16_f32");
    line<true, true, false, false, 3>(out, rnd, "A from LDS, one chain, 16 v_exp_f32");
    return 0;
}
