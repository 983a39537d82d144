// Device RNG: Philox4x32-10 + inverse-CDF normals, bit-reproducible.
//
// Replaces jax.random.normal / random.uniform of solvers/MLP.py:178,221 and
// solvers/MLP_full_history.py:99,133,138.  The normal transform is integer bit manipulation, one table row and three
// fused multiply-adds in IEEE-754 binary32 (this TU is built with -ffp-contract=off, so nothing is fused or unfused behind
// the specification's back) -- any conforming implementation yields the same bits, which is what tests/test_gpu_rng.py
// checks against the independent NumPy statement (oracle/philox.py).
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

// ---- N(0,1) by a table-driven inverse CDF ------------------------------------------------------------------------------------
// One normal per 32-bit word r:  k = r >> 8,  u = (k + 1/2) 2^-24,  folded to  p = min(u, 1 - u) = v 2^-25  with the odd integer
// v = 2 j + 1 < 2^24 (j = k, or its 23-bit complement when u > 1/2).  binary32(v) is exact: its exponent (24 octaves) and top five
// mantissa bits (32 segments per octave) select one of 768 rows (c0, c1, c2, c3), its low 18 mantissa bits are the coordinate s
// inside the segment, and  Phi^-1(p) = fma(fma(fma(c3, s, c2), s, c1), s, c0)  -- the cubic Hermite interpolant of Phi^-1 over the
// segment (normal_table.inc, made by tools/gen_normal_table.py; within 4.7e-7 -- one ulp at |x| > 4 -- of scipy's ndtri on every input).  The sign is
// that of u - 1/2.  Three fused multiply-adds on table constants: any conforming implementation yields the same bits
// (oracle/philox.py restates table and transform; tests/test_gpu_rng.py compares all 2^24 inputs).  |N| <= 5.42.
//
// Round 2 until here used Box-Muller with Cephes ln / sin / cos polynomials (32 vector instructions per normal against 17 and
// one LDS read): GENERATE was bound by them (1.34 ms; 0.89 ms with the transform ablated), 1.05 ms with the table.
constexpr int kNormalTableRows = SCASML_NORMAL_TABLE_ROWS;   // include/scasml_hip.h

__device__ const float4 kNormalTable[kNormalTableRows] = {
#include "normal_table.inc"
};

// the workgroup's LDS copy (12 KB).  Every kernel that draws normals calls normal_table_to_lds() first, all threads, before any
// thread may return.
__device__ __forceinline__ float4 *normal_table_lds() {
    __shared__ float4 t[kNormalTableRows];
    return t;
}
__device__ __forceinline__ void normal_table_to_lds() {
    float4 *t = normal_table_lds();
    for (int i = threadIdx.x; i < kNormalTableRows; i += blockDim.x) t[i] = kNormalTable[i];
    __syncthreads();
}

__device__ __forceinline__ float icdf_normal(uint32_t r, const float4 *tbl) {
    const uint32_t sgn = (uint32_t)((int32_t)r >> 31);              // all ones when u > 1/2
    const uint32_t j = ((r >> 8) ^ sgn) & 0x7FFFFFu;
    const uint32_t b = __float_as_uint((float)(2u * j + 1u));       // exact
    const float4 c = tbl[(b >> 18) - (127u << 5)];
    const float s = (float)(b & 0x3FFFFu);                          // exact
    const float x = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(c.w, s, c.z), s, c.y), s, c.x);
    return __uint_as_float(__float_as_uint(x) ^ (sgn & 0x80000000u));
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
    const float4 *tbl = normal_table_lds();
    n.x = icdf_normal(r.x, tbl);
    n.y = icdf_normal(r.y, tbl);
    n.z = icdf_normal(r.z, tbl);
    n.w = icdf_normal(r.w, tbl);
#endif
    return n;
}

// U(0,1): ((r0 >> 9) + 0.5) * 2^-23
__device__ __forceinline__ float uniform_tau(uint32_t site, uint32_t root, uint32_t stream, uint32_t k0, uint32_t k1) {
    const u32x4 r = philox4x32_10(kQuadTau, site, root, stream, k0, k1);
    return ((float)(r.x >> 9) + 0.5f) * 1.1920928955078125e-07f;
}

}  // namespace scasml
