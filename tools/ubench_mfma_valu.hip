// Microbenchmark (superseded by ubench_valu_rate.hip / ubench_valu_forms.hip): do v_mfma_f32_32x32x2_f32 and f32 VALU
// co-execute on a gfx950 SIMD?  NOTE: the "VALU" loops below are plain C++ and hipcc SLP-packs them into v_pk_fma_f32,
// which never overlaps with MFMA -- the conclusions first drawn from this file held for packed f32 only.
// Variants (1024-thread grid x 256 CUs, N iterations each):
//   0: MFMA only            (2 independent accumulators, 8 MFMAs / iter)
//   1: VALU only            (32 independent v_fma_f32 / iter)
//   2: both, interleaved in ONE wave (1 MFMA : 4 VALU)
//   3: MFMA-only waves and VALU-only waves sharing a SIMD (waves 0-3 MFMA, 4-7 VALU of a 512-thread block)
//   4: bf16 MFMA 32x32x16 only;  5: bf16 MFMA + VALU interleaved in one wave
//   6: bf16-MFMA-only waves (0-3) and VALU-only waves (4-7) sharing a SIMD;  7: like 6 with 2 VALU waves per SIMD (waves 4-11)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(768) void k(float *out, int iters, float a, float b, unsigned long long *clk = nullptr) {
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    f32x16 acc0 = {0}, acc1 = {0};
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3f + i;
    s16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8};
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wv < 4);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wv >= 4);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 6 || MODE == 7) {
            if (wv < 4) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc0, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 32; ++i) v[i] = __builtin_fmaf(v[i], a, b);
            }
        } else if (MODE == 4 || MODE == 5) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc0, 0, 0, 0);
                if (MODE == 5) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[4 * j + q] = __builtin_fmaf(v[4 * j + q], a, b);
                }
            }
        } else if (do_m && do_v) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[4 * j + q] = __builtin_fmaf(v[4 * j + q], a, b);
            }
        } else if (do_m) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            }
        } else if (do_v) {
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = __builtin_fmaf(v[i], a, b);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {   // shader-clock cycles (s_memtime) and 100 MHz wall ticks of this wave
        clk[0] = __builtin_readcyclecounter() - c0;
        clk[1] = wall_clock64() - w0;
    }
}

static unsigned long long *g_clk;
static char g_note[96];
template <int MODE>
float run(float *out, int threads, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 100, 1.0001f, 1e-6f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 1e-6f, g_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, g_clk, sizeof h, hipMemcpyDeviceToHost);
    snprintf(g_note, sizeof g_note, "[%.1f memtime ticks/iter, memtime %.0f MHz]", (double)h[0] / iters, h[0] / (h[1] / 100.0));
    fprintf(stderr, "      mode %d x %d threads: %s\n", MODE, threads, g_note);
    return ms;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// VALU-only: 32 scalar v_fma_f32 per iteration vs 16 packed v_pk_fma_f32 (same flops)
template <int PACKED>
__global__ __launch_bounds__(1024) void kv(float *out, int iters, float a, float b) {
    f32x2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (f32x2){threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f - i};
    const f32x2 a2 = {a, a * 1.0001f}, b2 = {b, b * 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (PACKED) {
                v[i] = __builtin_elementwise_fma(v[i], a2, b2);
            } else {
                v[i].x = __builtin_fmaf(v[i].x, a2.x, b2.x);
                v[i].y = __builtin_fmaf(v[i].y, a2.y, b2.y);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int PACKED>
float runv(float *out, int threads, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kv<PACKED>, dim3(256), dim3(threads), 0, 0, out, 100, 1.0001f, 1e-6f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kv<PACKED>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 1e-6f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&g_clk, 16);
    const int it = 200000;
    printf("one wave per SIMD (256 threads/block, 1 block/CU)\n");
    printf("  0 mfma_f32 only      : %.3f ms\n", run<0>(out, 256, it));
    printf("  1 valu only (32 fma) : %.3f ms\n", run<1>(out, 256, it));
    printf("  2 interleaved 1 wave : %.3f ms\n", run<2>(out, 256, it));
    printf("  4 mfma_bf16 only     : %.3f ms\n", run<4>(out, 256, it));
    printf("  5 bf16+valu 1 wave   : %.3f ms\n", run<5>(out, 256, it));
    printf("two waves per SIMD (512 threads/block)\n");
    printf("  0 mfma_f32 only      : %.3f ms\n", run<0>(out, 512, it));
    printf("  1 valu only          : %.3f ms\n", run<1>(out, 512, it));
    printf("  3 mfma waves || valu waves : %.3f ms\n", run<3>(out, 512, it));
    printf("  2 interleaved, both waves  : %.3f ms\n", run<2>(out, 512, it));
    printf("  5 bf16+valu, both waves    : %.3f ms\n", run<5>(out, 512, it));
    printf("  4 mfma_bf16 only, both     : %.3f ms\n", run<4>(out, 512, it));
    printf("  6 bf16 mfma waves || valu waves (1+1 per SIMD) : %.3f ms\n", run<6>(out, 512, it));
    printf("three waves per SIMD (768 threads/block)\n");
    printf("  1 valu only                : %.3f ms\n", run<1>(out, 768, it));
    printf("  7 bf16 mfma wave || 2 valu waves per SIMD : %.3f ms\n", run<7>(out, 768, it));
    printf("VALU only, 32 flops-pairs per iteration: scalar v_fma_f32 x32 vs packed v_pk_fma_f32 x16\n");
    for (int th = 256; th <= 1024; th *= 2)
        printf("  %d waves/SIMD: scalar %.3f ms   packed %.3f ms\n", th / 256, runv<0>(out, th, it), runv<1>(out, th, it));
    return 0;
}
