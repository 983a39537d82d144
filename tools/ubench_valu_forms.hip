// Microbenchmark: issue rate of f32 VALU forms on a gfx950 SIMD (64 independent instructions per iteration).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float a, float b) {
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3f + i;
    float av = a, bv = b;
    asm volatile("" : "+v"(av), "+v"(bv));
    typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    h16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8};
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x16 big[2] = {{0}, {0}};
    asm volatile("" : "+v"(ab));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(av), "v"(bv));
                if (MODE == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(av), "v"(bv));
                if (MODE == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(av));
                if (MODE == 3) asm volatile("v_add_f32 %0, %1, %0" : "+v"(v[i]) : "s"(a));
                if (MODE == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "s"(a), "v"(bv));
                if (MODE == 5) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(v[i]) : "s"(a));
                if (MODE == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(v[(i + 1) & 31]), "v"(v[(i + 7) & 31]));
                if (MODE == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(v[(i + 1) & 31]));
                if (MODE == 8) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                if (MODE == 9) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 3) & 31]));
                if (MODE == 14) {   // 32 x 32 -> 64 bit multiply-add (Philox rounds)
                    unsigned long long r;
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(v[i]), "v"(av) : "vcc");
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(v[i]) : "v"((unsigned)(r >> 32)), "v"((unsigned)r));
                }
                if (MODE == 15) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[i]) : "v"(av));
                if (MODE == 17) asm volatile("v_add_f32 %0, 0x3e0eaaaa, %0" : "+v"(v[i]));                    // 32-bit literal
                if (MODE == 19) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(v[i]));                           // inline constant
                if (MODE == 20) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(av));
                if (MODE == 21) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(av));
                if (MODE == 16) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(av));
                // float16 forms of the as-coded surrogate's epilogue (round 3)
                if (MODE == 22) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(v[i]) : "v"(av), "v"(bv));
                if (MODE == 23) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(v[i]) : "v"(av), "v"(bv));
                if (MODE == 24) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(av));
                if (MODE == 25) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[i]));
                if (MODE == 26) asm volatile("v_fma_mixlo_f16 %0, %0, %1, 0" : "+v"(v[i]) : "v"(av));
                if (MODE == 27) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(v[i]) : "v"(av), "v"(bv));
                if (MODE == 28) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(v[i]) : "v"(av));
                if (MODE == 29) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(v[i]) : "v"(av), "v"(bv));
                if (MODE == 30) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(v[i]) : "v"(av), "v"(bv));
                if (MODE == 31) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v[i]));
                if (MODE == 32) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
                if (MODE == 33) asm volatile("v_fma_f16 %0, %0, %1, %2" : "+v"(v[i]) : "v"(av), "v"(bv));
                if (MODE == 34 && i < 16) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(f32x2 *)&v[2 * i]) : "v"(*(f32x2 *)&v[2 * ((i + 5) & 15)]));
                if (MODE == 35 && i < 16) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(f32x2 *)&v[2 * i]) : "v"(*(f32x2 *)&v[2 * ((i + 5) & 15)]), "v"(*(f32x2 *)&v[2 * ((i + 9) & 15)]));
                if (MODE == 36) asm volatile("v_mul_f16 %0, %0, %1" : "+v"(v[i]) : "v"(av));
                if (MODE == 37) asm volatile("v_log_f32 %0, %0" : "+v"(v[i]));
                if (MODE == 38) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
                if (MODE == 39) {   // the epilogue's pattern: an exp every 8 plain instructions
                    if (i % 8 == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                    else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(av));
                }
                if (MODE >= 10 && MODE <= 13) {   // v_fma stream with an MFMA every 16 (10, 12) or 8 (11, 13) of them
                    const int every = (MODE & 1) ? 8 : 16;
                    if (i % every == 0) {
                        if (MODE <= 11) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, %0" : "+v"(acc[(i / every) & 3]) : "v"(ab));
                        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(big[(i / every) & 1]) : "v"(ab));
                    }
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(av), "v"(bv));
                }
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + big[i & 1][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(float *out, const char *what) {
    const int iters = 50000;
    printf("  %-36s", what);
    for (int threads = 256; threads <= 1024; threads *= 2) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 100, 1.0001f, 1e-6f);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 1e-6f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  %dw: %5.2f cyc/instr", threads / 256, ms * 1e-3 * 2.4e9 / iters / 64 / (threads / 256));
    }
    printf("\n");
}
int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    printf("cycles per wave64 instruction per SIMD at 2.4 GHz, 1 / 2 / 4 waves per SIMD\n");
    run<0>(out, "v_fma_f32 v,v,v,v (2 shared srcs)");
    run<1>(out, "v_fmac_f32 v,v,v");
    run<2>(out, "v_mul_f32 v,v,v");
    run<3>(out, "v_add_f32 v,s,v");
    run<4>(out, "v_fma_f32 v,v,s,v");
    run<5>(out, "v_fma_f32 v,v,s,1.0");
    run<6>(out, "v_fma_f32 v,v',v'',v (distinct)");
    run<7>(out, "v_mov_b32 v,v'");
    run<9>(out, "v_sub_f32 v,v,v'");
    run<8>(out, "v_exp_f32");
    run<17>(out, "v_add_f32 v, literal, v");
    run<19>(out, "v_add_f32 v, 1.0, v");
    run<20>(out, "v_cndmask_b32 v, v, v, vcc");
    run<21>(out, "v_xor_b32 v, v, v");
    run<14>(out, "v_mad_u64_u32 + v_xor_b32 (pair)");
    run<15>(out, "v_mul_hi_u32");
    run<16>(out, "v_mul_u32_u24");
    printf("float16 forms (round 3; v_pk_*_f32 lines: 32 instructions per iteration counted as 64, i.e. cycles per TWO results)\n");
    run<22>(out, "v_fma_mix_f32 (f32 x f16lo + f32)");
    run<23>(out, "v_fma_mix_f32 (f32 x f16hi + f32)");
    run<24>(out, "v_cvt_pk_f16_f32");
    run<25>(out, "v_cvt_f16_f32");
    run<26>(out, "v_fma_mixlo_f16");
    run<27>(out, "v_pk_fma_f16");
    run<28>(out, "v_pk_mul_f16");
    run<29>(out, "v_dot2_f32_f16");
    run<30>(out, "v_dot2c_f32_f16");
    run<31>(out, "v_cvt_f32_f16");
    run<32>(out, "v_exp_f16");
    run<33>(out, "v_fma_f16");
    run<36>(out, "v_mul_f16");
    run<34>(out, "v_pk_mul_f32 (per 2 results)");
    run<35>(out, "v_pk_fma_f32 (per 2 results)");
    run<37>(out, "v_log_f32");
    run<38>(out, "v_rcp_f32");
    run<39>(out, "1 v_exp_f32 per 7 v_mul_f32");
    printf("64 v_fma_f32 per iteration plus MFMAs (cycles per v_fma, MFMA time included)\n");
    run<10>(out, "+ 4 x mfma 16x16x32 (1 per 16)");
    run<11>(out, "+ 8 x mfma 16x16x32 (1 per 8)");
    run<12>(out, "+ 4 x mfma 32x32x16 (1 per 16)");
    run<13>(out, "+ 8 x mfma 32x32x16 (1 per 8)");
    return 0;
}
