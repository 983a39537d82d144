// The reference's surrogate AS CODED, evaluated on the matrix cores: the hot-path form of GP(compat="reference").
//
// What is evaluated (oracle/gp_compat.py; float64 statement in gp_compat.hip): for a point x and a collocation point y with
// right_vector entries (c0, cL, ct, cS) the four rows  L^x u_hat = c0 P[x][I] + cL P[x][lap] + ct P[x][dt] + cS P[x][div],
// x in {I, dt, div, lap}, of models/GP.py:326-411, 630-651, where every entry P is rounded to float16 (:43, 55-179) and the
// "Laplacian" entries are the 5-index Hutchinson sum on a cyclically shifted argument (:28-39, 87-105, 119-179).  With
// v' = (v_1, .., v_{d-1}, t, v_0) the sixteen entries live in three pair geometries:
//     al  r = x - y     kappa0, S0 = sum_{k<d} r_k, rt = t_x - t_y          the nine Laplacian-free entries and lap_x lap_y
//     ys  r = x - y'    kappa1, S1, rD1 = t_x - y_0                         lap_y of (kappa, dt_x kappa, div_x kappa)
//     xs  r = x' - y    kappa2, S2, rD2 = x_0 - t_y                         lap_x of (kappa, dt_y kappa, div_y kappa)
// and all three are the SAME x.y product against a cyclically shifted collocation row: |x - y'|^2 = |x|^2 + |y|^2 - 2 x.y',
// |x' - y|^2 = |x|^2 + |y|^2 - 2 x.y'' with y''_m = y_{m-1}.  So one set of point planes (fp16 h + l, as gp_eval_bf16.hip's
// fp16x2 mode with fp16-exact collocation points -- the reference's are) meets three sets of collocation planes, and the
// MFMA delivers the exponent  Lam_g = k1 a^2 |r_g|^2  (k1 = -log2(e)/(2a)) of each geometry directly.
//
// The Hutchinson sums need Q_g = sum_{j<5} r_{g,j}^2 over the five drawn components: a K = 5 product, folded with its two
// rank-one terms and the constant into ONE extra 32x32x16 fp16 MFMA per geometry that delivers
//     G_g = h (a^2 Q_g - 5 a),   h = d/5                                        [h sum_j (a^2 r_j^2 - a) of oracle/gp_compat.py]
// (point side: u_j = -2 h a^2 x_j as h + l, PX = h a^2 sum x_j^2 - 5 a h as h + l, 1; row side: y_j twice, 1, 1, PY = h a^2 sum y_j^2
// as h + l).  With R_g = sum_j r_{g,j} (separable: a per-point minus a per-row constant), s_g = a S_g:
//     lap_y kappa = G1 kappa1                 dt_x lap_y = -(a rD1) G1 kappa1        div_x lap_y = (2 h a^2 R1 - s1 G1) kappa1
//     lap_x kappa = G2 kappa2                 lap_x dt_y = +(a rD2) G2 kappa2        lap_x div_y = -(2 h a^2 R2 - s2 G2) kappa2
//     lap_x lap_y = (G0 (G0 - 4 a h) - 10 a^2 h^2) kappa0
// Every entry is then rounded with v_cvt_pk_f16_f32 (two per instruction) and enters its sum through v_fma_mix_f32, which
// reads the fp16 half directly: a rounding costs half a vector instruction.  A rounding decision can differ from the float64
// statement's where the float32 value lands within ~2^-20 of a float16 midpoint (about 1 entry in 500); tests bound the effect.
//
// Cost model: the launch time is vector time PLUS matrix time (the chip's clock follows the MFMA density), so the levers are fewer instructions of
// either kind, not their arrangement; the measurements and the rejected arrangements are in profiles/HISTORY.md (4.4).
//
// Structure (as gp_eval_bf16.hip): 4-wave workgroups, 32 points per wave held in VGPRs as two fp16 planes for the whole sweep;
// the unit of work is a STAGE = (collocation tile of 32 rows, geometry): [KS KiB A fragments | 1 KiB Q fragment | 1 KiB row
// constants], fetched two stages ahead into a ring of three LDS slots by LDS-DMA behind a counted vmcnt and one raw barrier per
// stage.  Per site kind (scasml_plan_site_kinds) a wave runs only the geometries its outputs need:
//     full (u, dt, div, lap -> eps_PDE)   al, ys, xs on domain tiles;  al, xs on boundary tiles
//     u only / u and div                  al, ys on domain tiles;      al on boundary tiles
// Template parameters: KS K-steps of 16, BPC workgroups per CU, R16 the per-entry float16 rounding (the reference's code; off = the geometry mode's
// factored epilogue, compat_epilogue_fact), PLANES float16 planes of the evaluation point (2: products exact to 2^-22; 1: the geometry mode's option).
#include <stdlib.h>
#include <type_traits>

#include "gp_common.hpp"
#include "gp_mfma16.hpp"

namespace scasml {

constexpr int kHutch = 5;   // models/GP.py:30

typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct CompatIdxF {
    int32_t i[kHutch];
};

struct GpCompatArgs {
    const float *points;      // n_inf x kp
    const float *model;       // [tile][geometry 0..2][(KS + 2) KiB]: scasml_gp_compat_pack_mfma
    int32_t idx[kHutch];
    float4 *out4;
    float *lap;
    int64_t n_inf;
    int32_t n_pad, kp, d, first_bdy_tile;
    float a, sigma, mu;
    int32_t eq_id;
    int32_t round_out;        // u_hat and eps_PDE leave as float16 values (predict / compute_PDE_loss .astype(float16), models/GP.py:671, 769)
    const uint8_t *site_kinds;
    int64_t rows_per_site;
    const int32_t *site_order;   // scasml_gp_eval_compat_site_list: the sites to evaluate, in launch order (a 32-row wavefront tile is one site); null: all rows
    int32_t n_listed;
};

// round two float32 to float16 (RNE) in one instruction; R16 = false keeps them (development / parity of the formulas)
template <bool R16>
struct Pair {
    using T = std::conditional_t<R16, h16x2, f32x2>;
    T v;
    __device__ __forceinline__ Pair(float a, float b) {
        if constexpr (R16) v = __builtin_convertvector((f32x2){a, b}, h16x2);
        else v = (f32x2){a, b};
    }
    __device__ __forceinline__ float lo() const { return (float)v.x; }
    __device__ __forceinline__ float hi() const { return (float)v.y; }
};

// An entry that feeds u_hat or div u_hat is rounded from its float32 VALUE in every form: without this the compiler folds a
// product whose only use is the rounding into v_fma_mixlo_f16, which rounds the exact product once, while the same entry in
// another form (paired with a second value) goes float32 -> v_cvt_pk_f16_f32, twice rounded -- one entry in ~8000 then differs by a
// float16 ulp between forms, and what a site consumes would depend on the form its workgroup ran.  (Entries that only the full
// form uses keep the fused instruction.)
__device__ __forceinline__ float pin(float x) {
    asm("" : "+v"(x));
    return x;
}

// one collocation tile's x.y' on the fp16 matrix cores: 2 MFMAs per K-step (point planes h, l against the one collocation plane).
// PLANES = 1 (the geometry mode's option, round16 bit 2): the high plane only -- the point's coordinates enter x.y' rounded to float16
// (|x|^2 stays float32: the low part of its column is the accumulator's start value acc0), half the MFMAs.
template <int KS, int PLANES>
__device__ __forceinline__ void compat_mfma_lam(const float4 *lds_a, const s16x8 (&xb)[PLANES][KS], f32x16 &acc, int lane, float acc0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc0;
    // two planes: ONE fragment register set, read step by step (the other waves of the SIMD cover the LDS latency): with the row constants
    // of the epilogue read the same way the as-coded kernel fits 128 registers at KS = 7, i.e. a fourth wave per SIMD (19.6 -> 19.2 ms;
    // 134 registers and three waves with only the registers saved: 19.7)
    constexpr int NA = PLANES == 2 ? 1 : 2;
    Frag a[NA];
    if constexpr (NA == 2) a[0].f = lds_a[lane];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int cur = NA == 2 ? (s & 1) : 0, nxt = cur ^ 1;
        if constexpr (NA == 2) {
            if (s + 1 < KS) a[nxt].f = lds_a[(s + 1) * 64 + lane];
        } else {
            a[0].f = lds_a[s * 64 + lane];
        }
        Frag b0;
        b0.v = xb[0][s];
        if constexpr (PLANES == 2) {
            Frag b1;
            b1.v = xb[1][s];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cur].h, b1.h, acc, 0, 0, 0);   // small terms first
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cur].h, b0.h, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Row constants of a stage: 32 rows x 8 floats (scasml_gp_compat_pack_mfma)
//   al: sy  ty  c0  cL | ct  cS  act adcS         sy = a sum_{k<d} y_k, ty = a t_y;            act = a ct, adcS = a d cS
//   ys: sy1 ay0 wy1 cL | -   -   -   -            sy1 = a sum_{k<d} y'_k, ay0 = a y_0, wy1 = 2 h a^2 sum_j y_{i_j+1}
//   xs: sy  ty  wy2 c0 | ct  cS  e0  cSw          wy2 = 2 h a^2 sum_j y_{i_j};                 e0 = c0 - ct ty - cS sy, cSw = cS wy2
// (the last two of al and xs serve the factored epilogue only)
struct CompatPoint {
    float sx, tx, ax0, sx2, wxa, wxb;   // a S_x, a t_x, a x_0, a (S_x - x_0 + t_x), 2 h a^2 sum_j x_{i_j+1}, 2 h a^2 sum_j x_{i_j}
};
struct CompatConsts {
    float a, ad, c4, c10;               // a, a d, 4 a h, 10 a^2 h^2
};

// FORM: 0 full, 1 u only, 2 u and div.  GEOM: 0 al, 1 ys, 2 xs.  BDY: tile of boundary (and padding) rows: cL = ct = cS = 0.
template <int GEOM, int FORM, bool BDY, bool R16>
__device__ __forceinline__ void compat_epilogue(const float *rows_lds, const f32x16 &lam, const f32x16 &G, int half, const CompatPoint &p,
                                                const CompatConsts &c, float &au, float &at, float &ad, float &al) {
    const float4 *cb = reinterpret_cast<const float4 *>(__builtin_assume_aligned(rows_lds + 4 * 8 * half, 16));   // row = (r&3) + 8 (r>>2) + 4 half
    constexpr int NQ = (BDY || GEOM == 1) ? 1 : 2;
    float4 q[1][NQ];
    auto fetch = [&](int r, float4 (&dst)[NQ]) {
        const int row = (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int i = 0; i < NQ; ++i) dst[i] = cb[row * 2 + i];
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cur = 0;
        fetch(r, q[0]);             // read where used, not one row ahead: eight registers fewer (see compat_mfma_lam)
        __builtin_amdgcn_sched_barrier(0);
        const float kap = pin(__builtin_amdgcn_exp2f(lam[r]));
        if constexpr (GEOM == 0) {
            const float vsy = q[cur][0].x, vty = q[cur][0].y, c0 = q[cur][0].z;
            if constexpr (BDY && FORM == 1) {
                const Pair<R16> k0(kap, 0.0f);
                au = fmaf(c0, k0.lo(), au);
            } else {
                const float pp = p.tx - vty, ss = p.sx - vsy;       // a r_t, a S
                const float e1 = pin(pp * kap), e2 = pin(ss * kap);   // P[I][dt] = -P[dt][I],  P[I][div] = -P[div][I]
                if constexpr (BDY) {                                // boundary rows: c0 only
                    const Pair<R16> k01(kap, e1), k2(e2, 0.0f);
                    au = fmaf(c0, k01.lo(), au);
                    if constexpr (FORM == 0) at = fmaf(-c0, k01.hi(), at);
                    if constexpr (FORM != 1) ad = fmaf(-c0, k2.lo(), ad);
                } else {
                    const float cL = q[cur][0].w, ct = q[cur][1].x, cS = q[cur][1].y;
                    if constexpr (FORM == 1) {
                        const Pair<R16> k01(kap, e1), k2(e2, 0.0f);
                        au = fmaf(cS, k2.lo(), fmaf(ct, k01.hi(), fmaf(c0, k01.lo(), au)));
                    } else {
                        const float e4 = pin(pp * e2);                          // a^2 r_t S kappa = -P[dt][div] = -P[div][dt]
                        const float e5 = pin(fmaf(-ss, e2, c.ad * kap));        // (a d - a^2 S^2) kappa = P[div][div]
                        if constexpr (FORM == 2) {
                            const Pair<R16> k01(kap, e1), k24(e2, e4), k5(e5, 0.0f);
                            au = fmaf(cS, k24.lo(), fmaf(ct, k01.hi(), fmaf(c0, k01.lo(), au)));
                            ad = fmaf(cS, k5.lo(), fmaf(-ct, k24.hi(), fmaf(-c0, k24.lo(), ad)));
                        } else {
                            const float e3 = fmaf(-pp, e1, c.a * kap);          // (a - a^2 r_t^2) kappa = P[dt][dt]
                            const float g0 = G[r];
                            const float e6 = fmaf(g0, g0 - c.c4, -c.c10) * kap; // lap_x lap_y
                            const Pair<R16> k01(kap, e1), k23(e2, e3), k45(e4, e5), k6(e6, 0.0f);
                            au = fmaf(cS, k23.lo(), fmaf(ct, k01.hi(), fmaf(c0, k01.lo(), au)));
                            at = fmaf(-cS, k45.lo(), fmaf(ct, k23.hi(), fmaf(-c0, k01.hi(), at)));
                            ad = fmaf(cS, k45.hi(), fmaf(-ct, k45.lo(), fmaf(-c0, k23.lo(), ad)));
                            al = fmaf(cL, k6.lo(), al);
                        }
                    }
                }
            }
        } else if constexpr (GEOM == 1) {       // lap_y rows, coefficient cL (domain tiles only)
            const float vsy1 = q[cur][0].x, vay0 = q[cur][0].y, vwy1 = q[cur][0].z, cL = q[cur][0].w;
            const float g1 = G[r];
            const float f1 = pin(g1 * kap);                                     // P[I][lap]
            if constexpr (FORM == 1) {
                const Pair<R16> k1(f1, 0.0f);
                au = fmaf(cL, k1.lo(), au);
            } else {
                const float m = fmaf(-(p.sx - vsy1), g1, p.wxb - vwy1);         // h mix1
                const float f3 = pin(m * kap);                                  // P[div][lap]
                if constexpr (FORM == 2) {
                    const Pair<R16> k13(f1, f3);
                    au = fmaf(cL, k13.lo(), au);
                    ad = fmaf(cL, k13.hi(), ad);
                } else {
                    const float f2 = (p.tx - vay0) * f1;                        // -P[dt][lap]
                    const Pair<R16> k12(f1, f2), k3(f3, 0.0f);
                    au = fmaf(cL, k12.lo(), au);
                    at = fmaf(-cL, k12.hi(), at);
                    ad = fmaf(cL, k3.lo(), ad);
                }
            }
        } else {                                // lap_x rows (full form only)
            const float vsy = q[cur][0].x, vty = q[cur][0].y, vwy2 = q[cur][0].z, c0 = q[cur][0].w;
            const float g2 = G[r];
            const float h1 = g2 * kap;                                          // P[lap][I]
            if constexpr (BDY) {
                const Pair<R16> k1(h1, 0.0f);
                al = fmaf(c0, k1.lo(), al);
            } else {
                const float ct = q[cur][1].x, cS = q[cur][1].y;
                const float h2 = (p.ax0 - vty) * h1;                            // P[lap][dt]
                const float m = fmaf(-(p.sx2 - vsy), g2, p.wxa - vwy2);         // h mix2
                const float h3 = m * kap;                                       // -P[lap][div]
                const Pair<R16> k12(h1, h2), k3(h3, 0.0f);
                al = fmaf(-cS, k3.lo(), fmaf(ct, k12.hi(), fmaf(c0, k12.lo(), al)));
            }
        }
    }
}

// The GEOMETRY mode (round16 bit 0 off; GP(compat="reference-geometry")): the same sixteen entries in the same three pair geometries with the
// same Hutchinson sums, WITHOUT the float16 rounding of every entry.  Nothing then stops the four sums from factoring per geometry the way
// gp_eval_bf16.hip's E does:   with  pp = a r_t,  ss = a S,  E = c0 + ct pp + cS ss
//     al   u += kappa0 E      dt += kappa0 (a ct - pp E)      div += kappa0 (a d cS - ss E)      lap += cL kappa0 (G0 (G0 - 4 a h) - 10 a^2 h^2)
//     ys   w = cL kappa1 G1:  u += w      dt -= (a t_x - a y_0) w      div += -s1 w + cL kappa1 (2 h a^2 R1)
//     xs   lap += kappa2 (G2 (e0 + ct a x_0 + cS sx2) - cS (wxa - wy2)),   e0 = c0 - ct ty - cS sy  (a row constant)
// 13 + exp, 9 + exp and 5 + exp vector instructions per pair against 27 + exp, 13 + exp, 13 + exp of the as-coded form.  The fit is the
// as-coded one (same Gram, same right_vector); what the mode gives up is the reference's rounding noise in the hot evaluation -- measured on the
// reference's own experiments: GP relative L2 moves by <= 8e-6, ScaSML by <= 3e-5 (profiles/r04_eval_rounding_study.txt).
template <int GEOM, int FORM, bool BDY>
__device__ __forceinline__ void compat_epilogue_fact(const float *rows_lds, const f32x16 &lam, const f32x16 &G, int half, const CompatPoint &p,
                                                     const CompatConsts &c, float &au, float &at, float &ad, float &al) {
    const float4 *cb = reinterpret_cast<const float4 *>(__builtin_assume_aligned(rows_lds + 4 * 8 * half, 16));   // row = (r&3) + 8 (r>>2) + 4 half
    // which float4 of a row's eight constants this case reads: al domain both, al boundary and ys the first, xs the second
    constexpr int Q0 = GEOM == 2 ? 1 : 0;
    constexpr int NQ = (GEOM == 0 && !BDY) ? 2 : 1;
    float4 q[2][NQ];
    auto fetch = [&](int r, float4 (&dst)[NQ]) {
        const int row = (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int i = 0; i < NQ; ++i) dst[i] = cb[row * 2 + Q0 + i];
    };
    fetch(0, q[0]);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cur = r & 1, nxt = cur ^ 1;
        if (r + 1 < 16) fetch(r + 1, q[nxt]);
        #ifndef SCASML_FACT_NOSCHED
        __builtin_amdgcn_sched_barrier(0);
#endif
        const float kap = __builtin_amdgcn_exp2f(lam[r]);
        if constexpr (GEOM == 0) {
            const float vsy = q[cur][0].x, vty = q[cur][0].y, c0 = q[cur][0].z;
            if constexpr (BDY) {                                    // boundary rows: c0 only
                const float F = c0 * kap;
                au += F;
                if constexpr (FORM == 0) at = fmaf(-(p.tx - vty), F, at);
                if constexpr (FORM != 1) ad = fmaf(-(p.sx - vsy), F, ad);
            } else {
                const float cL = q[cur][0].w, ct = q[cur][1].x, cS = q[cur][1].y;
                const float pp = p.tx - vty, ss = p.sx - vsy;       // a r_t, a S
                const float E = fmaf(ct, pp, fmaf(cS, ss, c0));
                au = fmaf(kap, E, au);
                if constexpr (FORM != 1) ad = fmaf(kap, fmaf(-ss, E, q[cur][1].w), ad);
                if constexpr (FORM == 0) {
                    at = fmaf(kap, fmaf(-pp, E, q[cur][1].z), at);
                    const float g0 = G[r];
                    al = fmaf(cL * kap, fmaf(g0, g0 - c.c4, -c.c10), al);
                }
            }
        } else if constexpr (GEOM == 1) {       // lap_y rows, coefficient cL (domain tiles only)
            const float vsy1 = q[cur][0].x, vay0 = q[cur][0].y, vwy1 = q[cur][0].z, cL = q[cur][0].w;
            const float v = cL * kap;
            if constexpr (FORM == 1) {
                au = fmaf(v, G[r], au);
            } else {
                const float w = v * G[r];
                au += w;
                ad = fmaf(-(p.sx - vsy1), w, fmaf(v, p.wxb - vwy1, ad));
                if constexpr (FORM == 0) at = fmaf(-(p.tx - vay0), w, at);
            }
        } else {                                // lap_x rows (full form only); boundary rows have ct = cS = 0: e0 = c0, cSw = 0
            const float ct = q[cur][0].x, cS = q[cur][0].y, e0 = q[cur][0].z, cSw = q[cur][0].w;
            if constexpr (BDY) {
                al = fmaf(kap * G[r], e0, al);
            } else {
                const float E2 = fmaf(ct, p.ax0, fmaf(cS, p.sx2, e0));
                const float t = fmaf(cS, p.wxa, -cSw);
                al = fmaf(kap, fmaf(G[r], E2, -t), al);
            }
        }
    }
}

// Stage block, in floats (scasml_gp_compat_pack_mfma): KS*256 planes | 256 Q fragment | 256 row constants (32 x 8) = (KS + 2) KiB
constexpr int kStageTail = 256 + 256;

template <int KS, int BPC, bool R16, int PLANES>
__global__ __launch_bounds__(256, BPC) void gp_eval_compat_mfma_kernel(const GpCompatArgs g) {
    static_assert(PLANES == 2 || !R16, "the as-coded form keeps both point planes");
    // three slots: stage s is read while s + 1 has landed or lands and s + 2 is issued into the slot stage s - 1 was read from, which
    // every wave left before the barrier that ended step s - 1
#ifndef SCASML_COMPAT_NSLOT
#define SCASML_COMPAT_NSLOT 3
#endif
    constexpr int WPB = 4, NSLOT = SCASML_COMPAT_NSLOT, AHEAD = 2;
    constexpr int STAGE = KS * 256 + kStageTail;       // floats per LDS slot
    constexpr int NCHUNK = STAGE / 256;
    constexpr int CLO = NCHUNK / WPB, CREM = NCHUNK % WPB;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int64_t blk = blockIdx.x;
    int64_t p0 = (blk * WPB + wv) * 32;
    const int n_tiles = g.n_pad / 32;
    const int nb0 = g.first_bdy_tile < n_tiles ? g.first_bdy_tile : n_tiles;

    // Site list (scasml_gp_eval_compat_site_list): the grid covers the LISTED sites only, in the list's order -- a sample-sharded rank launches no
    // workgroup over the sites of other ranks (85 % of the grid returned at once on each of 8 ranks), and the host lists the sites by falling cost,
    // so that the grid's last, partly filled round of workgroups is made of the cheapest ones.  rows_per_site is a multiple of 32 here: a wave is one site.
    int form_listed = -1;
    if (g.site_order) {
        const int64_t wps = g.rows_per_site >> 5;
        bool all_u = true, all_ud = true;
        for (int w = 0; w < WPB; ++w) {                     // the form is a property of the workgroup: from the sites of its four waves
            const int64_t vs = (blk * WPB + w) / wps;
            if (vs >= g.n_listed) continue;
            const int k = g.site_kinds ? g.site_kinds[g.site_order[vs]] : 0;
            if (!(k == 1 || k == 3)) all_u = false;
            if (!(k == 1 || k == 3 || k == 4)) all_ud = false;
        }
        form_listed = all_u ? 1 : (all_ud ? 2 : 0);
        const int64_t vw = blk * WPB + wv, vs = vw / wps;
        p0 = vs < g.n_listed ? (int64_t)g.site_order[vs] * g.rows_per_site + (vw - vs * wps) * 32 : g.n_inf;   // past the list: shadow rows, never stored
    }

    // Monte-Carlo sample sharding: a workgroup that lies wholly inside sites of other ranks has nothing to do
    if (!g.site_order && g.site_kinds && g.rows_per_site >= 32) {
        const int64_t b0 = blk * WPB * 32;
        if (b0 < g.n_inf) {
            const int64_t b1 = (b0 + WPB * 32 < g.n_inf ? b0 + WPB * 32 : g.n_inf) - 1;
            bool unowned = true;
            for (int64_t s = b0 / g.rows_per_site; s <= b1 / g.rows_per_site; ++s)
                if (g.site_kinds[s] != 2) unowned = false;
            if (unowned) return;
        }
    }
    // The form is a property of the WORKGROUP (all four waves walk the same stage sequence, because a stage is staged by all of
    // them): full unless every row of the workgroup lies in sites that need less.
    int form = 0;
    if (g.site_order) {
        form = form_listed;
    } else if (g.site_kinds && g.rows_per_site >= 32) {
        const int64_t b0 = blk * WPB * 32;
        const int64_t b1 = (b0 + WPB * 32 < g.n_inf ? b0 + WPB * 32 : g.n_inf) - 1;
        if (b0 < g.n_inf) {
            bool all_u = true, all_ud = true;
            for (int64_t s = b0 / g.rows_per_site; s <= b1 / g.rows_per_site; ++s) {
                const int k = g.site_kinds[s];
                if (k == 2) continue;
                if (!(k == 1 || k == 3)) all_u = false;
                if (!(k == 1 || k == 3 || k == 4)) all_ud = false;
            }
            form = all_u ? 1 : (all_ud ? 2 : 0);
        }
    }
    form = __builtin_amdgcn_readfirstlane(form);
    const int GD = form == 0 ? 3 : 2, GB = form == 0 ? 2 : 1;       // stages per domain / boundary tile
    const int n_stages = nb0 * GD + (n_tiles - nb0) * GB;

    const int wvs = __builtin_amdgcn_readfirstlane(wv);
    const bool extra = wvs < CREM;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    // stage s of this form -> its block of the model: (tile, geometry)
    auto stage_src = [&](int s) -> const float * {
        int tile, geom;
        if (s < nb0 * GD) {
            tile = s / GD;
            geom = s - tile * GD;
        } else {
            const int t = s - nb0 * GD;
            tile = nb0 + t / GB;
            geom = form == 0 ? 2 * (t - (t / GB) * GB) : 0;
        }
        return g.model + ((int64_t)tile * 3 + geom) * STAGE;
    };
    auto stage = [&](int s) {
        const uint32_t dst = lds_base + (uint32_t)((s % NSLOT) * STAGE) * 4u;
        const float *src = stage_src(s);
        auto chunk = [&](int c) { glds16_asm(src + c * 256, (uint32_t)lane * 16u, dst + (uint32_t)c * 1024u); };
#pragma unroll
        for (int i = 0; i < CLO; ++i) chunk(wvs + i * WPB);
        if (CREM && extra) chunk(wvs + CLO * WPB);
    };
    auto rendezvous = [&](bool newest_may_fly) {
        if (newest_may_fly) {
            if (CREM && extra) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CLO + 1) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CLO) : "memory");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    stage(0);
    if (n_stages > 1) stage(1);

    // ---- this wave's 32 points: fp16 planes (h, l) of 2 a^2 q x, the constants (1, 1, k1 a^2 |x|^2) in the last three columns
    const float qs = 0.5f * 1.44269504088896341f / g.a, k1 = -qs;
    const float hh = (float)g.d / (float)kHutch;
    const float ha2 = hh * g.a * g.a;
    s16x8 xb[PLANES][KS];
    float acc0 = 0.0f;          // PLANES == 1: the low part of the |x|^2 column, which the dropped plane would have carried
    CompatPoint pt;
    Frag qa[1], qb[1];          // B fragments of the Q products: components i_j + 1 (al, xs) and i_j (ys)
    {
        int64_t row = p0 + col;
        if (row >= g.n_inf) row = g.n_inf - 1;   // shadow rows, never stored
        const float *prow = g.points + row * g.kp;
        const int kbase = half * (8 * KS);
        const float4 *src = reinterpret_cast<const float4 *>(prow + kbase);
        float pn = 0.0f, ps = 0.0f;
        const float ptime = prow[g.d], px0 = prow[0];
        const float fold = 2.0f * g.a * g.a * qs;
        auto make_planes = [&](const float (&t)[8], Frag &fh, Frag &fl) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float h0 = (float)(_Float16)t[2 * c], h1 = (float)(_Float16)t[2 * c + 1];
                fh.u[c] = pack_h2(t[2 * c], t[2 * c + 1]);
                fl.u[c] = pack_h2(t[2 * c] - h0, t[2 * c + 1] - h1);
            }
        };
        float tlast[8];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 q0 = src[2 * s], q1 = src[2 * s + 1];
            const float e[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            float t[8];
#pragma unroll
            for (int c2 = 0; c2 < 8; ++c2) {
                pn = fmaf(e[c2], e[c2], pn);
                ps += e[c2];
                t[c2] = fold * e[c2];
            }
            if (s == KS - 1) {
#pragma unroll
                for (int c2 = 0; c2 < 8; ++c2) tlast[c2] = t[c2];
            } else {
                Frag fh, fl;
                make_planes(t, fh, fl);
                xb[0][s] = fh.v;
                if constexpr (PLANES == 2) xb[1][s] = fl.v;
            }
        }
        pn += __shfl_xor(pn, 32);
        ps += __shfl_xor(ps, 32);
        tlast[5] = half ? 1.0f : tlast[5];
        tlast[6] = half ? 1.0f : tlast[6];
        tlast[7] = half ? k1 * g.a * g.a * pn : tlast[7];
        {
            Frag fh, fl;
            make_planes(tlast, fh, fl);
            xb[0][KS - 1] = fh.v;
            if constexpr (PLANES == 2) xb[1][KS - 1] = fl.v;
            else acc0 = k1 * g.a * g.a * pn - (float)(_Float16)(k1 * g.a * g.a * pn);   // both halves: every row of this lane's column
        }
        pt.sx = g.a * (ps - ptime);           // the row sum includes t
        pt.tx = g.a * ptime;
        pt.ax0 = g.a * px0;
        pt.sx2 = pt.sx - pt.ax0 + pt.tx;
        // Hutchinson components of this point: x_{i_j+1} (geometries al, xs) and x_{i_j} (ys)
        float ua[kHutch], ub[kHutch], sa = 0.0f, sb = 0.0f, na = 0.0f, nb = 0.0f;
#pragma unroll
        for (int j = 0; j < kHutch; ++j) {
            const float va = prow[g.idx[j] + 1], vb = prow[g.idx[j]];
            sa += va;
            sb += vb;
            na = fmaf(va, va, na);
            nb = fmaf(vb, vb, nb);
            ua[j] = -2.0f * ha2 * va;
            ub[j] = -2.0f * ha2 * vb;
        }
        pt.wxa = 2.0f * ha2 * sa;
        pt.wxb = 2.0f * ha2 * sb;
        // Q fragment, K = 16: half 0 = (u_0..4 high parts, PX_h, PX_l, 1), half 1 = (u_0..4 low parts, 1, 0, 0)
        auto qfrag = [&](const float (&u)[kHutch], float px, Frag &f) {
            auto hi = [](float v) { return (float)(_Float16)v; };
            if (half == 0) {
                f.u[0] = pack_h2(u[0], u[1]);
                f.u[1] = pack_h2(u[2], u[3]);
                f.u[2] = pack_h2(u[4], px);
                f.u[3] = pack_h2(px - hi(px), 1.0f);
            } else {
                f.u[0] = pack_h2(u[0] - hi(u[0]), u[1] - hi(u[1]));
                f.u[1] = pack_h2(u[2] - hi(u[2]), u[3] - hi(u[3]));
                f.u[2] = pack_h2(u[4] - hi(u[4]), 1.0f);
                f.u[3] = 0u;
            }
        };
        const float c5ah = (float)kHutch * g.a * hh;
        qfrag(ua, fmaf(ha2, na, -c5ah), qa[0]);
        qfrag(ub, fmaf(ha2, nb, -c5ah), qb[0]);
    }
    CompatConsts cc;
    cc.a = g.a;
    cc.ad = g.a * (float)g.d;
    cc.c4 = 4.0f * g.a * hh;
    cc.c10 = 10.0f * g.a * g.a * hh * hh;
    float au = 0.0f, at = 0.0f, ad = 0.0f, al = 0.0f;
    f32x16 acc, accG;
    uint32_t region;
    asm volatile("s_mov_b32 %0, -1" : "=s"(region));   // basic-block boundaries the optimiser cannot remove (gp_eval_bf16.hip: one block spills)

    int s_now = 0;
    auto step = [&](auto geom_c, auto form_c, auto bdy_c) {
        constexpr int GEOM = decltype(geom_c)::value, FORM = decltype(form_c)::value;
        constexpr bool BDY = decltype(bdy_c)::value;
        constexpr bool NEEDG = GEOM != 0 || (FORM == 0 && !BDY);
        const int slot = s_now % NSLOT;
        if (s_now + AHEAD < n_stages) stage(s_now + AHEAD);
        const float *base = lds + slot * STAGE;
        if (region & 1) {
            compat_mfma_lam<KS, PLANES>(reinterpret_cast<const float4 *>(base), xb, acc, lane, acc0);
            if constexpr (NEEDG) {
                Frag aq;
                aq.f = reinterpret_cast<const float4 *>(base + KS * 256)[lane];
#pragma unroll
                for (int r = 0; r < 16; ++r) accG[r] = 0.0f;
                accG = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq.h, GEOM == 1 ? qb[0].h : qa[0].h, accG, 0, 0, 0);
            }
        }
        if (region & 2) {
            if constexpr (R16) compat_epilogue<GEOM, FORM, BDY, true>(base + (KS + 1) * 256, acc, accG, half, pt, cc, au, at, ad, al);
            else compat_epilogue_fact<GEOM, FORM, BDY>(base + (KS + 1) * 256, acc, accG, half, pt, cc, au, at, ad, al);
        }
        rendezvous(s_now + AHEAD < n_stages);
        ++s_now;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    auto sweep = [&](auto form_c) {
        constexpr int FORM = decltype(form_c)::value;
        for (int jt = 0; jt < nb0; ++jt) {
            step(I0{}, form_c, std::false_type{});
            step(I1{}, form_c, std::false_type{});
            if constexpr (FORM == 0) step(I2{}, form_c, std::false_type{});
        }
        for (int jt = nb0; jt < n_tiles; ++jt) {
            step(I0{}, form_c, std::true_type{});
            if constexpr (FORM == 0) step(I2{}, form_c, std::true_type{});
        }
    };
    rendezvous(n_stages > 1);   // stage 0 has landed (stage 1 may still be in flight)
    if (form == 2) sweep(I2{});
    else if (form == 1) sweep(I1{});
    else sweep(I0{});

    float u = au + __shfl_xor(au, 32);
    const float dt = at + __shfl_xor(at, 32);
    const float dv = ad + __shfl_xor(ad, 32);
    const float lp = al + __shfl_xor(al, 32);
    const int64_t row = p0 + col;
    if (half == 0 && row < g.n_inf) {
        if (g.round_out) u = (float)(_Float16)u;                               // predict(...).astype(float16), models/GP.py:671
        float eps = dt + g.mu * dv + 0.5f * g.sigma * g.sigma * lp + eq_f<float>(g.eq_id, u, g.sigma * dv, g.sigma, (float)g.d);   // :767-768
        if (g.round_out) eps = (float)(_Float16)eps;                           // :769
        g.out4[row] = make_float4(u, dv, eps, dt);
        if (g.lap) g.lap[row] = lp;
    }
}

// ---------------------------------------------------------------------------------------------------------------- pack
// One thread per collocation row: the three stage blocks of its tile (planes in MFMA A-fragment order, Q fragment, row constants).
__global__ void gp_compat_pack_mfma_kernel(int d, float a, const float *x_dom, int n_dom, const float *x_bdy, int n_bdy, const double *rv,
                                           CompatIdxF ix, float *model, int n_pad, int kp) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pad) return;
    const int N = n_dom + n_bdy, D = d + 1, KS = kp / 16;
    const float *src = j < n_dom ? x_dom + (int64_t)j * D : (j < N ? x_bdy + (int64_t)(j - n_dom) * D : nullptr);
    auto y = [&](int k) { return src ? src[k] : 0.0f; };
    const float qs = 0.5f * 1.44269504088896341f / a, k1 = -qs;
    const float hh = (float)d / (float)kHutch, ha2 = hh * a * a;
    float ny = 0.0f, sy = 0.0f;
    for (int k = 0; k < D; ++k) ny = fmaf(y(k), y(k), ny);
    for (int k = 0; k < d; ++k) sy += y(k);
    const float ay = k1 * a * a * ny;
    const _Float16 ayh = (_Float16)ay;
    const float ayl = ay - (float)ayh;
    float c0 = 0.0f, cL = 0.0f, ct = 0.0f, cS = 0.0f;
    if (j < n_dom) {
        c0 = (float)rv[j];
        cL = (float)rv[N + j];
        ct = (float)rv[N + n_dom + j];
        cS = (float)rv[N + 2 * n_dom + j];
    } else if (j < N) {
        c0 = (float)rv[j];
    }
    const int tile = j / 32, i = j % 32;
    const int stage_floats = KS * 256 + kStageTail;
    for (int g = 0; g < 3; ++g) {
        float *blk = model + ((int64_t)tile * 3 + g) * stage_floats;
        uint16_t *planes = reinterpret_cast<uint16_t *>(blk);
        // geometry g's collocation row: al y, ys y' = (y_1, .., t, y_0), xs y'' = (t, y_0, .., y_{d-1})
        auto yg = [&](int k) { return g == 0 ? y(k) : (g == 1 ? y((k + 1) % D) : y((k + D - 1) % D)); };
        for (int k = 0; k < kp; ++k) {
            const float v = k <= d ? yg(k) : (k == kp - 3 ? (float)ayh : (k == kp - 2 ? ayl : (k == kp - 1 ? 1.0f : 0.0f)));
            const int h = k / (kp / 2), kk = k % (kp / 2);
            const int64_t e = (((int64_t)(kk / 8)) * 64 + h * 32 + i) * 8 + (kk & 7);
            planes[e] = __builtin_bit_cast(unsigned short, (_Float16)v);
        }
        // Q fragment (row side): half 0 = (y_0..4, 1, 1, PY_h), half 1 = (y_0..4, PY_l, 0, 0); components i_j + 1 (al, ys) or i_j (xs)
        float yq[kHutch], nq = 0.0f, sq = 0.0f;
        for (int q = 0; q < kHutch; ++q) {
            yq[q] = y(ix.i[q] + (g == 2 ? 0 : 1));
            nq = fmaf(yq[q], yq[q], nq);
            sq += yq[q];
        }
        const float py = ha2 * nq;
        const _Float16 pyh = (_Float16)py;
        uint16_t *qf = reinterpret_cast<uint16_t *>(blk + KS * 256);
        auto put = [&](int h, int e, float v) { qf[(h * 32 + i) * 8 + e] = __builtin_bit_cast(unsigned short, (_Float16)v); };
        for (int q = 0; q < kHutch; ++q) {
            put(0, q, yq[q]);
            put(1, q, yq[q]);
        }
        put(0, 5, 1.0f);
        put(0, 6, 1.0f);
        put(0, 7, (float)pyh);
        put(1, 5, py - (float)pyh);
        put(1, 6, 0.0f);
        put(1, 7, 0.0f);
        float *rc = blk + (KS + 1) * 256 + i * 8;
        for (int q = 0; q < 8; ++q) rc[q] = 0.0f;
        if (g == 0) {
            rc[0] = a * sy;
            rc[1] = a * y(d);
            rc[2] = c0;
            rc[3] = cL;
            rc[4] = ct;
            rc[5] = cS;
            rc[6] = a * ct;
            rc[7] = a * (float)d * cS;
        } else if (g == 1) {
            rc[0] = a * (sy - y(0) + y(d));
            rc[1] = a * y(0);
            rc[2] = 2.0f * ha2 * sq;
            rc[3] = cL;
        } else {
            rc[0] = a * sy;
            rc[1] = a * y(d);
            rc[2] = 2.0f * ha2 * sq;
            rc[3] = c0;
            rc[4] = ct;
            rc[5] = cS;
            rc[6] = fmaf(-cS, rc[0], fmaf(-ct, rc[1], c0));
            rc[7] = cS * rc[2];
        }
    }
}

template <int KS, bool R16, int PLANES>
static int launch_compat(const GpCompatArgs &g, hipStream_t s) {
    // registers: 4 KS per point plane + ~72 (as-coded, reads not run ahead) / ~90 (geometry mode) for accumulators, Q fragments, row
    // constants and temporaries
    constexpr int REGS = 4 * PLANES * KS + (R16 ? 72 : 90);
    constexpr size_t lds_bytes = SCASML_COMPAT_NSLOT * (size_t)(KS * 256 + kStageTail) * sizeof(float);
    constexpr int BPC_REGS = REGS <= 128 ? 4 : (REGS <= 164 ? 3 : 2), BPC_LDS = (int)(160 * 1024 / lds_bytes);
    constexpr int BPC = BPC_REGS < BPC_LDS ? BPC_REGS : BPC_LDS;
    const int64_t waves = g.site_order ? (int64_t)g.n_listed * (g.rows_per_site >> 5) : (g.n_inf + 31) / 32;
    const int64_t blocks = (waves + 3) / 4;
    if (blocks == 0) return 0;
    if (blocks > 0x7FFFFFFF) return fail(SCASML_ERR_UNSUPPORTED, "gp_eval_compat_sites: too many points");
    static_assert(lds_bytes * BPC <= 160 * 1024, "LDS slots exceed 160 KiB");
    auto kern = gp_eval_compat_mfma_kernel<KS, BPC, R16, PLANES>;
    if (lds_bytes > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return fail(SCASML_ERR_HIP, "gp_eval_compat_sites: cannot reserve %zu bytes of LDS", lds_bytes);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds_bytes, s, g);
    return check_launch("gp_eval_compat_sites launch");
}

template <bool R16, int PLANES>
static int launch_compat_ks(const GpCompatArgs &g, hipStream_t s) {
    switch (g.kp / 16) {
#define SCASML_CASE(K) \
    case K: return launch_compat<K, R16, PLANES>(g, s);
        SCASML_CASE(1) SCASML_CASE(2) SCASML_CASE(3) SCASML_CASE(4) SCASML_CASE(5) SCASML_CASE(6) SCASML_CASE(7) SCASML_CASE(8)
        SCASML_CASE(9) SCASML_CASE(10) SCASML_CASE(11) SCASML_CASE(12) SCASML_CASE(13) SCASML_CASE(14) SCASML_CASE(15) SCASML_CASE(16)
#undef SCASML_CASE
    }
    return fail(SCASML_ERR_UNSUPPORTED, "gp_eval_compat_sites: kp=%d", g.kp);
}

static int check_idx5(const int32_t *idx_h, int d, int32_t (&out)[kHutch], const char *who) {
    if (!idx_h) return fail(SCASML_ERR_ARG, "%s: idx is null", who);
    for (int j = 0; j < kHutch; ++j) {
        if (idx_h[j] < 0 || idx_h[j] >= d) return fail(SCASML_ERR_ARG, "%s: idx[%d] = %d outside [0, d)", who, j, idx_h[j]);
        for (int q = 0; q < j; ++q)
            if (idx_h[q] == idx_h[j]) return fail(SCASML_ERR_ARG, "%s: idx has a repeated entry %d", who, idx_h[j]);
        out[j] = idx_h[j];
    }
    return 0;
}

}  // namespace scasml

using namespace scasml;

extern "C" int64_t scasml_gp_compat_model_floats(int32_t d, int32_t n_pad) {
    const int64_t kp = scasml_point_stride(d);
    return (int64_t)(n_pad / SCASML_GP_TILE) * 3 * ((kp / 16) * 256 + kStageTail);
}

extern "C" int scasml_gp_compat_pack_mfma(int32_t d, float a, const float *x_dom, int32_t n_dom, const float *x_bdy, int32_t n_bdy,
                                          const double *rv, const int32_t *idx_h, float *model_out, void *stream) {
    if (!x_dom || !rv || !model_out || (n_bdy > 0 && !x_bdy)) return fail(SCASML_ERR_ARG, "gp_compat_pack_mfma: null argument");
    if (d < kHutch || d > SCASML_MAX_DIM || n_dom < 1 || n_bdy < 0) return fail(SCASML_ERR_ARG, "gp_compat_pack_mfma: bad sizes (d >= %d needed)", kHutch);
    CompatIdxF ix;
    if (int rc = check_idx5(idx_h, d, ix.i, "gp_compat_pack_mfma")) return rc;
    const int n_pad = (n_dom + n_bdy + 31) / 32 * 32;
    const int kp = scasml_point_stride(d);
    hipLaunchKernelGGL(gp_compat_pack_mfma_kernel, dim3((n_pad + 63) / 64), dim3(64), 0, (hipStream_t)stream, d, a, x_dom, n_dom, x_bdy, n_bdy,
                       rv, ix, model_out, n_pad, kp);
    return check_launch("gp_compat_pack_mfma launch");
}

static int eval_compat_sites(int32_t d, float a, float sigma_eq, float mu_eq, int32_t eq_id, const float *model, int32_t n_dom,
                             int32_t n_bdy, const int32_t *idx_h, int32_t round16, float x_bound, const float *points, int64_t n_inf,
                             int64_t rows_per_site, const uint8_t *site_kinds, const int32_t *site_order, int32_t n_listed, float *out4, float *lap,
                             void *stream);

extern "C" int scasml_gp_eval_compat_sites(int32_t d, float a, float sigma_eq, float mu_eq, int32_t eq_id, const float *model, int32_t n_dom,
                                           int32_t n_bdy, const int32_t *idx_h, int32_t round16, float x_bound, const float *points, int64_t n_inf,
                                           int64_t rows_per_site, const uint8_t *site_kinds, float *out4, float *lap, void *stream) {
    return eval_compat_sites(d, a, sigma_eq, mu_eq, eq_id, model, n_dom, n_bdy, idx_h, round16, x_bound, points, n_inf, rows_per_site, site_kinds, nullptr,
                             0, out4, lap, stream);
}

extern "C" int scasml_gp_eval_compat_site_list(int32_t d, float a, float sigma_eq, float mu_eq, int32_t eq_id, const float *model, int32_t n_dom,
                                               int32_t n_bdy, const int32_t *idx_h, int32_t round16, float x_bound, const float *points, int64_t n_inf,
                                               int64_t rows_per_site, const uint8_t *site_kinds, const int32_t *site_order, int32_t n_listed,
                                               float *out4, float *lap, void *stream) {
    if (!site_order || n_listed < 0) return fail(SCASML_ERR_ARG, "gp_eval_compat_site_list: null or negative site list");
    if (rows_per_site < 32 || (rows_per_site & 31)) return fail(SCASML_ERR_ARG, "gp_eval_compat_site_list: rows_per_site must be a positive multiple of 32");
    if (n_inf % rows_per_site) return fail(SCASML_ERR_ARG, "gp_eval_compat_site_list: n_inf must be a whole number of sites");
    if (n_listed > n_inf / rows_per_site) return fail(SCASML_ERR_ARG, "gp_eval_compat_site_list: %d listed sites, the buffer holds %lld", n_listed, (long long)(n_inf / rows_per_site));
    if (n_listed == 0) return 0;
    return eval_compat_sites(d, a, sigma_eq, mu_eq, eq_id, model, n_dom, n_bdy, idx_h, round16, x_bound, points, n_inf, rows_per_site, site_kinds, site_order,
                             n_listed, out4, lap, stream);
}

static int eval_compat_sites(int32_t d, float a, float sigma_eq, float mu_eq, int32_t eq_id, const float *model, int32_t n_dom,
                             int32_t n_bdy, const int32_t *idx_h, int32_t round16, float x_bound, const float *points, int64_t n_inf,
                             int64_t rows_per_site, const uint8_t *site_kinds, const int32_t *site_order, int32_t n_listed, float *out4, float *lap,
                             void *stream) {
    if (n_inf == 0) return 0;
    if (!model || !points || !out4 || n_inf < 0) return fail(SCASML_ERR_ARG, "gp_eval_compat_sites: bad argument");
    if (d < kHutch || d > SCASML_MAX_DIM || n_dom < 1 || n_bdy < 0) return fail(SCASML_ERR_ARG, "gp_eval_compat_sites: bad sizes");
    if (site_kinds && rows_per_site < 1) return fail(SCASML_ERR_ARG, "gp_eval_compat_sites: rows_per_site must be positive");
    if (!eq_known(eq_id)) return fail(SCASML_ERR_UNSUPPORTED, "gp_eval_compat_sites: unknown equation id %d", eq_id);
    // the fp16 planes carry 0.72 a |x|^2 and 1.44 a x_k (gp_eval.hip, split = 22): refuse a stated bound that leaves the fp16 range
    const float xb = x_bound > 0.0f ? x_bound : 2.0f;
    if (0.7213f * a * xb * xb * (float)(d + 1) > 3.0e4f)
        return fail(SCASML_ERR_UNSUPPORTED, "gp_eval_compat_sites: a = %g with |x_k| <= %g is outside the fp16 range at d = %d; use scasml_gp_eval_compat",
                    (double)a, (double)xb, d);
    GpCompatArgs g;
    if (int rc = check_idx5(idx_h, d, g.idx, "gp_eval_compat_sites")) return rc;
    g.points = points;
    g.model = model;
    g.out4 = reinterpret_cast<float4 *>(out4);
    g.lap = lap;
    g.n_inf = n_inf;
    g.n_pad = (n_dom + n_bdy + 31) / 32 * 32;
    g.kp = scasml_point_stride(d);
    g.d = d;
    g.first_bdy_tile = (n_dom + SCASML_GP_TILE - 1) / SCASML_GP_TILE;
    g.a = a;
    g.sigma = sigma_eq;
    g.mu = mu_eq;
    g.eq_id = eq_id;
    g.round_out = (round16 & 2) ? 1 : 0;
    g.site_kinds = site_kinds;
    g.rows_per_site = rows_per_site;
    g.site_order = site_order;
    g.n_listed = n_listed;
    hipStream_t s = (hipStream_t)stream;
    if (round16 & 1) {
        if (round16 & 4) return fail(SCASML_ERR_ARG, "gp_eval_compat_sites: round16 bit 2 (one point plane) is the geometry mode's option, not the as-coded form's");
        return launch_compat_ks<true, 2>(g, s);
    }
    return (round16 & 4) ? launch_compat_ks<false, 1>(g, s) : launch_compat_ks<false, 2>(g, s);
}
