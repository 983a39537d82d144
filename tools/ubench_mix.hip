// Development: the per-register instruction mix of the as-coded surrogate's epilogue (1 v_exp_f32, 11 plain float32, 3 v_cvt_pk_f16_f32, 10 v_fma_mix_f32,
// one LDS read of row constants) with ONE 32x32x16 MFMA per step placed at different points of the stream, with and without the dependencies of the real
// code (exp -> entries -> conversions -> sums).  Which placement lets the MFMA hide, and do the dependencies undo it?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// PLACE: 0 MFMA between the plain and the float16 part; 1 MFMA first; 2 MFMA last; 3 MFMA in the middle of the float16 part
// DEP: 0 independent registers; 1 the real chain (exp -> 2 muls -> fmas -> cvt_pk of those -> fma_mix of those into 4 accumulators)
// WHAT: 0 both, 1 MFMA only, 2 vector only
template <int PLACE, int DEP, int WHAT>
__global__ __launch_bounds__(256) void k(float *out, int iters, const float *rnd) {
    __shared__ f32x4 lds[32 * 64];
    f32x16 acc;
    f16x8 a[2], b;
    for (int i = threadIdx.x; i < 32 * 64; i += blockDim.x) lds[i] = (f32x4){rnd[i & 4095], rnd[(i + 1) & 4095], rnd[(i + 2) & 4095], rnd[(i + 3) & 4095]};
    __syncthreads();
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        a[0][c] = a[1][c] = (_Float16)rnd[(threadIdx.x * 8 + c) & 4095];
        b[c] = (_Float16)rnd[(threadIdx.x * 8 + c + 2048) & 4095];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = rnd[(threadIdx.x + r) & 4095];
    float lam[16], s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) lam[r] = -rnd[(threadIdx.x + 64 * r) & 4095] - 1.0f;
    const float px = rnd[threadIdx.x & 4095], py = rnd[(threadIdx.x + 9) & 4095];
    f32x4 rc = lds[lane];
    asm volatile("" : "+v"(b));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {          // one step = one accumulator register of the epilogue
            f32x4 rcn = lds[((it * 16 + r + 1) & 31) * 64 + ((lane * 5) & 63)];     // row constants of the next step
            auto mfma = [&]() {
                if (WHAT == 2) return;
                f32x4 t = lds[((it * 16 + r) & 31) * 64 + lane];
                a[(r + 1) & 1] = __builtin_bit_cast(f16x8, t);
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a[r & 1]), "v"(b));
            };
            float kap, e1, e2, e3, e4, e5, e6, pp, ss;
            unsigned k01, k23, k45;
            auto plain = [&]() {
                if (WHAT == 1) return;
                const float x = DEP ? lam[r] : lam[(r + 5) & 15];
                asm volatile("v_exp_f32 %0, %1" : "=v"(kap) : "v"(x));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(pp) : "v"(px), "v"(rc.x));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(ss) : "v"(py), "v"(rc.y));
                const float kk = DEP ? kap : rc.z;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(pp), "v"(kk));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e2) : "v"(ss), "v"(kk));
                const float f1 = DEP ? e1 : rc.w, f2 = DEP ? e2 : rc.x;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e4) : "v"(pp), "v"(f2));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e3) : "v"(py), "v"(kk));
                asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(e3) : "v"(pp), "v"(f1));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e5) : "v"(px), "v"(kk));
                asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(e5) : "v"(ss), "v"(f2));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e6) : "v"(rc.z), "v"(py));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(e6) : "v"(kk));
            };
            auto slow_a = [&]() {
                if (WHAT == 1) return;
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(k01) : "v"(kap), "v"(e1));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(k23) : "v"(e2), "v"(e3));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(s0) : "v"(rc.z), "v"(DEP ? k01 : (unsigned)lane));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(s0) : "v"(rc.w), "v"(DEP ? k01 : (unsigned)lane));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(s0) : "v"(rc.x), "v"(DEP ? k23 : (unsigned)lane));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(s1) : "v"(rc.z), "v"(DEP ? k01 : (unsigned)lane));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(s1) : "v"(rc.w), "v"(DEP ? k23 : (unsigned)lane));
            };
            auto slow_b = [&]() {
                if (WHAT == 1) return;
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(k45) : "v"(e4), "v"(e5));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(s1) : "v"(rc.x), "v"(DEP ? k45 : (unsigned)lane));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(s2) : "v"(rc.z), "v"(DEP ? k23 : (unsigned)lane));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(s2) : "v"(rc.w), "v"(DEP ? k45 : (unsigned)lane));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(s2) : "v"(rc.x), "v"(DEP ? k45 : (unsigned)lane));
                asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(k45) : "v"(e6), "v"(rc.y));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(s3) : "v"(rc.z), "v"(DEP ? k45 : (unsigned)lane));
            };
            if (PLACE == 1) mfma();
            plain();
            if (PLACE == 0) mfma();
            slow_a();
            if (PLACE == 3) mfma();
            slow_b();
            if (PLACE == 2) mfma();
            if (WHAT != 1 && DEP) lam[r] = lam[r] * 0.999f - 1e-3f * s3 * 0.0f;
            rc = rcn;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sum = s0 + s1 + s2 + s3;
    for (int r = 0; r < 16; ++r) sum += acc[r] + lam[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
static int g_blocks = 256;
template <int PLACE, int DEP, int WHAT>
float run(float *out, const float *rnd, int wgs_per_cu) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<PLACE, DEP, WHAT>), dim3(g_blocks * wgs_per_cu), dim3(256), 0, 0, out, 50, rnd);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<PLACE, DEP, WHAT>), dim3(g_blocks * wgs_per_cu), dim3(256), 0, 0, out, iters, rnd);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters / 16;   // ns per step
}
template <int PLACE, int DEP>
void line(float *out, const float *rnd, const char *what) {
    printf("  %-58s", what);
    for (int w = 1; w <= 3; ++w) {
        const float both = run<PLACE, DEP, 0>(out, rnd, w), m = run<PLACE, DEP, 1>(out, rnd, w), v = run<PLACE, DEP, 2>(out, rnd, w);
        printf("   %dw: both %5.1f mfma %5.1f vec %5.1f", w, both, m, v);
    }
    printf("\n");
}
int main(int argc, char **argv) {
    if (argc > 1) g_blocks = atoi(argv[1]);
    float *out, *rnd, h[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMalloc(&out, 256 * 3 * 256 * 4);
    (void)hipMalloc(&rnd, sizeof(h));
    (void)hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
    printf("ns per step (one accumulator register of the epilogue + 1 v_mfma_f32_32x32x16_f16) per wave; %d x (1, 2, 3) workgroups of 4 waves = 1 / 2 / 3 waves per SIMD\n", g_blocks);
    line<0, 0>(out, rnd, "independent registers, MFMA between plain and float16 part");
    line<1, 0>(out, rnd, "independent registers, MFMA first");
    line<2, 0>(out, rnd, "independent registers, MFMA last");
    line<3, 0>(out, rnd, "independent registers, MFMA inside the float16 part");
    line<0, 1>(out, rnd, "real dependencies, MFMA between plain and float16 part");
    line<1, 1>(out, rnd, "real dependencies, MFMA first");
    line<2, 1>(out, rnd, "real dependencies, MFMA last");
    line<3, 1>(out, rnd, "real dependencies, MFMA inside the float16 part");
    return 0;
}
