// Device RNG: Philox4x32-10 + Box-Muller normals, bit-reproducible.
//
// Replaces jax.random.normal / random.uniform of solvers/MLP.py:178,221 and
// solvers/MLP_full_history.py:99,133,138.  The normal transform is specified operation by operation in IEEE-754
// binary32: multiply, add, correctly-rounded sqrt, and fused multiply-add exactly where `fmaf` is written (this TU is
// built with -ffp-contract=off, so `a * b + c` is never fused behind the specification's back), with the Cephes
// single precision logf / sinf / cosf polynomials in Horner form -- any conforming implementation yields the same
// bits, which is what tests/test_gpu_rng.py checks against the independent NumPy statement (oracle/philox.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scasml {

constexpr uint32_t kQuadTau = 0x80000000u;  // counter word 0 reserved for the full-history time draw

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

// ln(k * 2^-24), k in [1, 2^24]
__device__ __forceinline__ float ln_u24(uint32_t k) {
    const float f = (float)k;  // exact
    const uint32_t bits = __float_as_uint(f);
    int e = (int)(bits >> 23) - 127;
    float m = __uint_as_float((bits & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float x = m - 1.0f;
    const float z = x * x;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, x, -1.1514610310e-1f);
    p = __builtin_fmaf(p, x, 1.1676998740e-1f);
    p = __builtin_fmaf(p, x, -1.2420140846e-1f);
    p = __builtin_fmaf(p, x, 1.4249322787e-1f);
    p = __builtin_fmaf(p, x, -1.6668057665e-1f);
    p = __builtin_fmaf(p, x, 2.0000714765e-1f);
    p = __builtin_fmaf(p, x, -2.4999993993e-1f);
    p = __builtin_fmaf(p, x, 3.3333331174e-1f);
    float y = x * z;
    y = y * p;
    const float fe = (float)(e - 24);
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = x + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    return r;
}

// (cos, sin) of the uniform angle encoded by the 24-bit integer k (see oracle/philox.py)
__device__ __forceinline__ void sincos_u24(uint32_t k, float &cc, float &ss) {
    const int quad = (int)(k >> 22);
    const int frac = (int)(k & 0x3FFFFFu);
    const float w = (float)(frac - (1 << 21)) + 0.5f;
    const float x = w * 0x1.921fb6p-22f;  // (pi/2) * 2^-22 rounded to binary32
    const float z = x * x;
    float s = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    s = __builtin_fmaf(s, z, -1.6666654611e-1f);
    s = s * z;
    s = __builtin_fmaf(s, x, x);
    float c = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    c = __builtin_fmaf(c, z, 4.166664568298827e-2f);
    c = __builtin_fmaf(c, z * z, __builtin_fmaf(-0.5f, z, 1.0f));
    // quadrant: (cos, sin) = (c, s), (-s, c), (-c, -s), (s, -c) for quad = 0..3.  Bit-select and sign-bit XOR instead of six
    // compare-selects (a v_cndmask reads its mask from SGPRs and issues at half rate): odd quadrants swap the two,
    // cos is negated in quadrants 1 and 2 (bit 1 of quad + 1), sin in quadrants 2 and 3 (bit 1 of quad).
    const uint32_t swap = (uint32_t)__builtin_amdgcn_sbfe((int)k, 22, 1);                  // bit 22 of k -> 0 or 0xFFFFFFFF
    const uint32_t cb = __float_as_uint(c), sb = __float_as_uint(s);
    const uint32_t cm = (sb & swap) | (cb & ~swap), sm = (cb & swap) | (sb & ~swap);
    cc = __uint_as_float(cm ^ (((k + 0x400000u) << 8) & 0x80000000u));
    ss = __uint_as_float(sm ^ ((k << 8) & 0x80000000u));
    (void)quad;
}

// Correctly rounded binary32 square root, spelled out so that it does not depend on compiler flags or
// on which lowering the optimiser happens to pick (a scalar sqrtf may legally get a 2.5-ulp expansion):
// v_sqrt_f32 is within 1 ulp; the two neighbours are tested with exact FMA residuals (the scheme of
// LLVM's correctly-rounded f32 sqrt lowering).  Valid for x = 0 and normal x, which is all Box-Muller
// produces (x = -2 ln u is 0 or >= 1.19e-7).
__device__ __forceinline__ float sqrt_rn(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float down = __uint_as_float(__float_as_uint(s) - 1u);
    const float up = __uint_as_float(__float_as_uint(s) + 1u);
    const float vp = __builtin_fmaf(-down, s, x);
    const float vs = __builtin_fmaf(-up, s, x);
    float r = vp <= 0.0f ? down : s;
    r = vs > 0.0f ? up : r;
    return r;
}

__device__ __forceinline__ void box_muller(uint32_t ra, uint32_t rb, float &n0, float &n1) {
    const uint32_t k1 = (ra >> 8) + 1u;
    const uint32_t k2 = rb >> 8;
    float t = -2.0f * ln_u24(k1);
    t = t < 0.0f ? 0.0f : t;
    const float rad = sqrt_rn(t);
    float c, s;
    sincos_u24(k2, c, s);
    n0 = rad * c;
    n1 = rad * s;
}

// four N(0,1) values: dims 4*quad .. 4*quad+3 of path-step `site` of root `root`
__device__ __forceinline__ float4 normal4(uint32_t quad, uint32_t site, uint32_t root, uint32_t stream,
                                          uint32_t k0, uint32_t k1) {
#if defined(SCASML_ABLATION) && SCASML_ABL_RNG == 1   // development: cost of Philox (a two-multiply mixer instead)
    const uint32_t h = (quad * 0x9E3779B9u) ^ (site * 0x85EBCA6Bu) ^ root ^ stream ^ k0 ^ k1;
    const u32x4 r = {h, h * 0xC2B2AE35u, h ^ 0x27D4EB2Fu, h + 0x165667B1u};
#else
    const u32x4 r = philox4x32_10(quad, site, root, stream, k0, k1);
#endif
    float4 n;
#if defined(SCASML_ABLATION) && SCASML_ABL_RNG == 2   // development: cost of the normal transform (a scale instead)
    n.x = (float)(int)r.x * 0x1p-31f; n.y = (float)(int)r.y * 0x1p-31f; n.z = (float)(int)r.z * 0x1p-31f; n.w = (float)(int)r.w * 0x1p-31f;
#else
    box_muller(r.x, r.y, n.x, n.y);
    box_muller(r.z, r.w, n.z, n.w);
#endif
    return n;
}

// U(0,1): ((r0 >> 9) + 0.5) * 2^-23
__device__ __forceinline__ float uniform_tau(uint32_t site, uint32_t root, uint32_t stream, uint32_t k0, uint32_t k1) {
    const u32x4 r = philox4x32_10(kQuadTau, site, root, stream, k0, k1);
    return ((float)(r.x >> 9) + 0.5f) * 1.1920928955078125e-07f;
}

}  // namespace scasml
