// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 on a gfx950 SIMD (inline asm, so the optimiser cannot
// re-pack or scalarise), alone and interleaved with v_mfma_f32_32x32x16_f16 in the same wave.
//   MODE 0: 32 x v_fma_f32            MODE 1: 16 x v_pk_fma_f32 (same flops)   MODE 2: 32 x v_pk_fma_f32 (2x flops)
//   MODE 3: 8 MFMA only               MODE 4: 8 MFMA + 32 v_fma_f32 (1:4)       MODE 5: 8 MFMA + 32 v_pk_fma_f32 (1:4)
//   MODE 6: 32 x v_exp_f32            MODE 7: 8 MFMA + 16 v_pk_fma_f32 (1:2)
//   MODE 8: even waves 8 MFMA, odd waves 32 v_fma_f32 (separate waves sharing each SIMD)
//   MODE 9: 8 MFMA + 64 v_fma_f32 (1:8)   MODE 10: 64 v_fma_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float a, float b, unsigned long long *clk, const float *rnd = nullptr) {
    f32x16 acc0 = {0}, acc1 = {0};
    f32x2 v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = (f32x2){threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f - i};
    f32x2 a2 = {a, a}, b2 = {b, b};
    f16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8};
    if (rnd) {   // random operands: realistic switching activity in the matrix and vector pipes
#pragma unroll
        for (int c = 0; c < 8; ++c) ab[c] = (_Float16)rnd[(threadIdx.x * 8 + c) & 4095];
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = (f32x2){rnd[(threadIdx.x + 64 * i) & 4095], rnd[(threadIdx.x + 64 * i + 32) & 4095]};
        a2 = (f32x2){1.0f + rnd[threadIdx.x & 4095] * 1e-3f, 1.0f + rnd[(threadIdx.x + 1) & 4095] * 1e-3f};
        b2 = (f32x2){rnd[(threadIdx.x + 2) & 4095], rnd[(threadIdx.x + 3) & 4095]};
    }
    asm volatile("" : "+v"(a2), "+v"(b2), "+v"(ab));
    const unsigned long long c0 = __builtin_readcyclecounter();
    // waves w and w+4 of a workgroup share a SIMD: role by bit 2 of the wave id
    const bool mrole = (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 4) == 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 8) {
            if (mrole) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(acc1) : "v"(ab));
                    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(acc0) : "v"(ab));
                }
            } else {
#pragma unroll
                for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].x) : "v"(a2.x), "v"(b2.x));
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 9 || MODE == 10) {
                if (MODE == 9) {
                    if (j & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(acc1) : "v"(ab));
                    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(acc0) : "v"(ab));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[4 * j + q].x) : "v"(a2.x), "v"(b2.x));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[4 * j + q].y) : "v"(a2.x), "v"(b2.x));
                }
                continue;
            }
            if (MODE == 3 || MODE == 4 || MODE == 5 || MODE == 7) {
                if (j & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(acc1) : "v"(ab));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(acc0) : "v"(ab));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 4 * j + q;
                if (MODE == 0 || MODE == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].x) : "v"(a2.x), "v"(b2.x));
                if (MODE == 1 && (q & 1)) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a2), "v"(b2));
                if (MODE == 7 && (q & 1)) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a2), "v"(b2));
                if (MODE == 2 || MODE == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a2), "v"(b2));
                if (MODE == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i].x));
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += v[i].x + (MODE == 8 || MODE == 0 || MODE == 4 ? 0.0f : v[i].y);
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}

// MFMA-only waves and v_fma_f32-only waves sharing every SIMD (waves w and w+4 of a workgroup land on one SIMD)
__global__ __launch_bounds__(1024) void k8(float *out, int iters, float a, float b, unsigned long long *clk) {
    const bool mrole = (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 4) == 0;
    const unsigned long long c0 = __builtin_readcyclecounter();
    float s = 0;
    if (mrole) {
        f32x16 acc0 = {0}, acc1 = {0};
        f16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8};
        asm volatile("" : "+v"(ab));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(acc0) : "v"(ab));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(acc1) : "v"(ab));
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    } else {
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3f + i;
        asm volatile("" : "+v"(a), "+v"(b));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) s += v[i];
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) clk[threadIdx.x >> 8] = c1 - c0;
}

static unsigned long long *g_clk;
static const float *g_rnd;
static void run8(float *out, int threads, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k8, dim3(256), dim3(threads), 0, 0, out, 100, 1.0001f, 1e-6f, g_clk);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k8, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 1e-6f, g_clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    (void)hipMemcpy(h, g_clk, 16, hipMemcpyDeviceToHost);
    printf("  %-34s %d waves/SIMD: %8.3f ms  = %7.1f cycles/iter/SIMD @2.4GHz   (mfma wave 0: %.1f, valu wave 4: %.1f ticks/iter)\n",
           "mfma waves || v_fma_f32 waves", threads / 256, ms, ms * 1e-3 * 2.4e9 / iters, (double)h[0] / iters, (double)h[1] / iters);
}
template <int MODE>
void run(float *out, int threads, int iters, const char *what) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 100, 1.0001f, 1e-6f, g_clk);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 1e-6f, g_clk, g_rnd);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h;
    (void)hipMemcpy(&h, g_clk, 8, hipMemcpyDeviceToHost);
    // kernel-level cycles per iteration per SIMD assume 2.4 GHz (the wave-0 memtime column is that wave's own span)
    printf("  %-34s %d waves/SIMD: %8.3f ms  = %7.1f cycles/iter/SIMD @2.4GHz   (wave 0: %.1f ticks/iter)\n", what, threads / 256, ms,
           ms * 1e-3 * 2.4e9 / iters, (double)h / iters);
}

int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    (void)hipMalloc(&g_clk, 16);
    const int it = 100000;
    float *rnd, h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)(((unsigned)i * 2654435761u >> 8) & 0xFFFF) / 65536.0f - 0.5f;
    (void)hipMalloc(&rnd, sizeof h);
    (void)hipMemcpy(rnd, h, sizeof h, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; ++pass) {
        g_rnd = pass ? rnd : nullptr;
        printf(pass ? "RANDOM operands\n" : "CONSTANT operands\n");
    for (int th = 256; th <= 1024; th *= 2) {
        run<0>(out, th, it, "32 v_fma_f32");
        run<1>(out, th, it, "16 v_pk_fma_f32");
        run<2>(out, th, it, "32 v_pk_fma_f32");
        run<6>(out, th, it, "32 v_exp_f32");
        run<3>(out, th, it, "8 mfma_f16 32x32x16");
        run<4>(out, th, it, "8 mfma + 32 v_fma_f32");
        run<5>(out, th, it, "8 mfma + 32 v_pk_fma_f32");
        run<7>(out, th, it, "8 mfma + 16 v_pk_fma_f32");
        run<10>(out, th, it, "64 v_fma_f32");
        run<9>(out, th, it, "8 mfma + 64 v_fma_f32");
        if (th >= 512 && !pass) run8(out, th, it);
    }
    }
    return 0;
}
