// Development: operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with FP6 (e2m3) operands, probed with exact data (cdna_hip_programming.md:
// "other dtypes: check the map with exact integer data").  Found: lane l holds the 32 elements k = 32 (l >> 5) .. + 31 of row / column l & 31,
// element p at bits [6 p, 6 p + 6) of six dwords (dwords 6, 7 of the eight-register operand are not read); code = sign | exponent (2) | mantissa (3),
// value (1 + m / 8) 2^(e - 1), e = 0: m / 8; the E8M0 scale byte scales by 2^(s - 127).  Record: profiles/r04_compat_eval_experiments.txt, section 4.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// A: 32 rows x 64 k as 6-bit codes, B: 64 k x 32 cols.  lane l: row/col = l & 31, k block = 32 (l >> 5), element p at bits [6p, 6p + 6) of the 192-bit fragment
__global__ void probe(const uint8_t *Acode, const uint8_t *Bcode, float *D, int scale_a, int scale_b) {
    const int lane = threadIdx.x, rc = lane & 31, h = lane >> 5;
    uint32_t fa[8] = {0}, fb[8] = {0};
    for (int p = 0; p < 32; ++p) {
        const uint32_t ca = Acode[rc * 64 + 32 * h + p] & 63, cb = Bcode[(32 * h + p) * 32 + rc] & 63;
        const int bit = 6 * p;
        fa[bit >> 5] |= ca << (bit & 31);
        if ((bit & 31) > 26) fa[(bit >> 5) + 1] |= ca >> (32 - (bit & 31));
        fb[bit >> 5] |= cb << (bit & 31);
        if ((bit & 31) > 26) fb[(bit >> 5) + 1] |= cb >> (32 - (bit & 31));
    }
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (int)fa[i]; b[i] = (int)fb[i]; }
    a[6] = scale_a; a[7] = 0x5A5A5A5A + lane; b[6] = scale_b; b[7] = (int)0xDEADBEEF - lane;   // dwords 6, 7: not part of an FP6 operand
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.0f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, a[6], 0, b[6]);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + rc] = c[r];
}
static float dec(int c) {
    const int s = c >> 5, e = (c >> 3) & 3, m = c & 7;
    const float v = e == 0 ? m / 8.0f : (1.0f + m / 8.0f) * (float)(1 << (e - 1));
    return s ? -v : v;
}
int main() {
    uint8_t A[32 * 64], B[64 * 32];
    srand(3);
    for (int i = 0; i < 32 * 64; ++i) { A[i] = rand() & 63; B[i] = rand() & 63; }
    uint8_t *dA, *dB; float *dD, D[1024];
    hipMalloc(&dA, sizeof(A)); hipMalloc(&dB, sizeof(B)); hipMalloc(&dD, sizeof(D));
    hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
    for (int sa = 126; sa <= 128; ++sa) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, sa, 127);
        hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
        double worst = 0; int bad = 0;
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
            double ref = 0;
            for (int k = 0; k < 64; ++k) ref += (double)dec(A[r * 64 + k]) * dec(B[k * 32 + c]);
            ref *= sa == 126 ? 0.5 : (sa == 128 ? 2.0 : 1.0);
            const double e = fabs(D[r * 32 + c] - ref);
            if (e > worst) worst = e;
            if (e > 1e-3 * (1 + fabs(ref))) ++bad;
        }
        printf("scale_a %d: max |D - ref| %.3g, %d of 1024 wrong; D[0][0] = %g\n", sa, worst, bad, D[0]);
    }
    return 0;
}
