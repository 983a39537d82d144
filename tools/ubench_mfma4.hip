// Development: v_mfma_f32_4x4x4_16b_f16 on gfx950 -- operand layout check (16 independent 4x4 blocks, lanes 4b..4b+3) and issue rate
// alone / between plain vector instructions.  Used to move the as-coded surrogate's entry * coefficient sums off the vector ALUs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(const float *A, const float *B, float *D) {
    // A[b][i][k], B[b][k][j] -> D[b][i][j]; assumed: lane = 4 b + i holds A[b][i][0..3]; lane = 4 b + j holds B[b][0..3][j]; lane 4 b + j holds D[b][0..3][j]
    const int lane = threadIdx.x, b = lane >> 2, q = lane & 3;
    h16x4 a, bb;
    for (int k = 0; k < 4; ++k) {
        a[k] = (_Float16)A[(b * 4 + q) * 4 + k];
        bb[k] = (_Float16)B[(b * 4 + k) * 4 + q];
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, bb, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(b * 4 + i) * 4 + q] = c[i];
}

template <int MODE>
__global__ __launch_bounds__(1024) void rate(float *out, int iters, float a) {
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3f + i;
    float av = a;
    h16x4 ha = {1, 2, 3, 4}, hb = {(_Float16)0.5f, 1, 2, 3};
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    asm volatile("" : "+v"(av), "+v"(ha), "+v"(hb));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if (MODE == 0) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(ha), "v"(hb));
                if (MODE == 1) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0" : "+v"(acc[i & 1]) : "v"(ha), "v"(hb));
                if (MODE == 2 || MODE == 3 || MODE == 4) {   // one MFMA per 4 / 2 / 8 v_mul_f32
                    const int every = MODE == 2 ? 4 : (MODE == 3 ? 2 : 8);
                    if (i % every == 0) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0" : "+v"(acc[(i / every) & 7]) : "v"(ha), "v"(hb));
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(av));
                }
                if (MODE == 5) {   // cvt_pk feeding the MFMA's A operand (the epilogue's dependency)
                    if (i % 4 == 0) {
                        h16x4 t;
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(((uint32_t *)&t)[0]) : "v"(v[i]), "v"(v[i + 1]));
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(((uint32_t *)&t)[1]) : "v"(v[i + 2]), "v"(v[i + 3]));
                        asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0" : "+v"(acc[(i / 4) & 7]) : "v"(t), "v"(hb));
                    }
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(av));
                }
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(float *out, const char *what, int per_iter) {
    const int iters = 50000;
    printf("  %-52s", what);
    for (int threads = 256; threads <= 1024; threads *= 2) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(rate<MODE>, dim3(256), dim3(threads), 0, 0, out, 100, 1.0001f);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(rate<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  %dw: %6.2f", threads / 256, ms * 1e-3 * 2.4e9 / iters / per_iter / (threads / 256));
    }
    printf("\n");
}
int main() {
    float hA[256], hB[256], hD[256], *A, *B, *D, *out;
    for (int i = 0; i < 256; ++i) {
        hA[i] = (float)((i * 7) % 13 - 6);
        hB[i] = (float)((i * 5) % 11 - 5) * 0.5f;
    }
    (void)hipMalloc(&A, 1024);
    (void)hipMalloc(&B, 1024);
    (void)hipMalloc(&D, 1024);
    (void)hipMalloc(&out, 256 * 1024 * 4);
    (void)hipMemcpy(A, hA, 1024, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, A, B, D);
    (void)hipMemcpy(hD, D, 1024, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int b = 0; b < 16; ++b)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double want = 0;
                for (int k = 0; k < 4; ++k) want += (double)hA[(b * 4 + i) * 4 + k] * hB[(b * 4 + k) * 4 + j];
                worst = fmax(worst, fabs(want - hD[(b * 4 + i) * 4 + j]));
            }
    printf("layout check (lane 4b+i: A[b][i][:]; lane 4b+j: B[b][:][j], D[b][:][j]): max |diff| = %g %s\n", worst, worst == 0 ? "OK" : "MISMATCH");
    printf("cycles at 2.4 GHz per SIMD, 1 / 2 / 4 waves per SIMD\n");
    run<0>(out, "4x4x4 f16 MFMA alone, 8 accumulators (per MFMA)", 64);
    run<1>(out, "4x4x4 f16 MFMA alone, 2 accumulators (per MFMA)", 64);
    run<2>(out, "1 MFMA per 4 v_mul_f32 (per v_mul)", 64);
    run<3>(out, "1 MFMA per 2 v_mul_f32 (per v_mul)", 64);
    run<4>(out, "1 MFMA per 8 v_mul_f32 (per v_mul)", 64);
    run<5>(out, "2 cvt_pk -> MFMA per 4 v_mul_f32 (per v_mul)", 64);
    return 0;
}
