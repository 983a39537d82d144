// Pieces shared by the fp32-MFMA and the split-bf16 fused GP evaluation kernels: argument block,
// LDS stage view and the closed-form epilogue (derivation in gp_eval.hip).
#pragma once
#include "common.hpp"
#include "equations.hpp"

namespace scasml {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GpArgs {
    const float *points;       // n_inf x kp
    const float *colloc_frag;  // [n_tiles][NK4][64][4]
    const uint16_t *colloc_bf16;  // [n_tiles][3 planes][kp/16][64][8] truncated-bf16 planes
    const uint16_t *colloc_f16;   // [n_tiles][2 planes][kp/16][64][8] fp16 planes (h, l = fp16(v - h), unscaled)
    const float *coef;         // [n_pad][16]  FP32-kernel constants
    const float *coef2;        // [n_pad][16]  exponent-scaled constants of the 16-bit kernels (gp_epilogue_scaled)
    const float *coef3;        // [n_pad][8]   constants of the E-from-MFMA epilogue (gp_epilogue_em), scaled by 2^s
    const uint16_t *eplane;    // [n_tiles][1 KiB]: 64 x 4 fp16 A fragments of the linear part of E (K = 8), then 32 rows x 4 floats (gp_epilogue_em), scaled by 2^s
    const float *escale;       // escale[0] = 2^-s
    int32_t first_bdy_tile;    // collocation tiles from here on hold boundary (and padding) rows only: cL = ct = cS = 0
    float4 *out4;              // n_inf x (u, div, eps, dt)
    float *lap;                // n_inf or null
    int64_t n_inf;
    int32_t n_pad, kp, d;
    float a, sigma, mu;
    int32_t eq_id;
    int32_t colloc_is_f16;
    const uint8_t *site_u_only;   // per tree site: only u_hat needed (null = every row needs everything)
    int64_t rows_per_site;
    int32_t dbg;   // ablation switches (SCASML_GP_DBG, development only): 1 skip MFMA, 2 skip epilogue, 8 stage once
};

struct GpStageView {
    const float4 *y;    // A fragments of one collocation tile (layout depends on the kernel)
    const float *coef;  // [32 rows][16]
};

struct GpConsts {
    float a, a2, ad, kexp, dF;
};

// ---------------------------------------------------------------------------------------------------------------
// FP32 kernel (gp_eval.hip): C row = (r&3) + 8*(r>>2) + 4*half, column = lane & 31.  Per collocation row the
// coefficient tile holds 16 floats (gp_pack_kernel, `coef`):
//   0 a*sum y   1 a*t_y   2 c0   3 cL  |  4 ct   5 cS   6..11 unused here  |  12 |y|^2
constexpr int kCoefRow = 16;

template <int PT>
__device__ __forceinline__ void gp_epilogue_tile(const GpStageView &st, const f32x16 (&acc)[PT], const GpConsts &c, int half,
                                                 const float (&nx)[PT], const float (&sx)[PT], const float (&tx)[PT],
                                                 float (&au)[PT], float (&at)[PT], float (&ad)[PT], float (&al)[PT]) {
    const float4 *cb = reinterpret_cast<const float4 *>(__builtin_assume_aligned(st.coef + 4 * kCoefRow * half, 16));
    float4 q[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[0][i] = cb[i];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cur = r & 1, nxt = cur ^ 1;
        if (r + 1 < 16) {
            const int off = (((r + 1) & 3) + 8 * ((r + 1) >> 2)) * (kCoefRow / 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) q[nxt][i] = cb[off + i];
        }
        __builtin_amdgcn_sched_barrier(0);
        const float vsy = q[cur][0].x, vty = q[cur][0].y, vc0 = q[cur][0].z, vcL = q[cur][0].w;
        const float vct = q[cur][1].x, vcS = q[cur][1].y;
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const float pp = tx[p] - vty;                  // a * r_t
            const float ss = sx[p] - vsy;                  // a * S
            const float vny = q[cur][3].x;
            const float r2 = fmaf(-2.0f, acc[p][r], nx[p] + vny);
            const float kap = __builtin_amdgcn_exp2f(r2 * c.kexp);
            const float L = fmaf(-pp, pp, fmaf(c.a2, r2, -c.ad));  // a^2 (r2 - r_t^2) - a d
            const float E = fmaf(vcS, ss, fmaf(vct, pp, fmaf(vcL, L, vc0)));
            au[p] = fmaf(kap, E, au[p]);
            at[p] = fmaf(kap, fmaf(-pp, E, c.a * vct), at[p]);
            const float dv = fmaf(2.0f * vcL, ss, c.dF * vcS);
            ad[p] = fmaf(kap, fmaf(-ss, E, c.a * dv), ad[p]);
            const float lv = fmaf(vcL, fmaf(2.0f, L, c.ad), vcS * ss);
            al[p] = fmaf(kap, fmaf(L, E, -2.0f * c.a * lv), al[p]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 16-bit kernels (gp_eval_bf16.hip): everything is carried in the units of the exponent, so that the matrix
// product IS the exponent.  With q = log2(e) / (2a), k1 = -q:
//   the MFMA delivers  Lam = k1 a^2 |x - y|^2  (space and time; the operand planes hold 2 a^2 q x.y, k1 a^2 |y|^2
//   against a constant 1, and k1 a^2 |x|^2 in the last K column against a constant 1),  kappa = exp2(Lam);
//   pp = sqrt(q) a (t_x - t_y),  Lh = Lam + pp^2 = k1 (L + a d)  with L = a^2 |x - y|_space^2 - a d,  ss = a (S_x - S_y);
//   E = c0 + cL L + ct a r_t + cS ss = e0 + eL Lh + et pp + cS ss.
// Per collocation row `coef2` holds (gp_pack_kernel):
//   0 a*sum y   1 sqrt(q) a t_y   2 e0 = c0 - a d cL   3 eL = cL / k1  |  4 et = ct / sqrt(q)   5 cS   6 sqrt(q) a ct
//   7 2 a cL  |  8 a d cS   9 -4 a cL   10 -2 a k1 cS   11 2 a^2 d k1 cL  |  12 unused   13 e0T   14 eL   15 cS
// (12..15: the terminal-time form, E = e0T + eL Lam + cS (a S_x) for points with t = T exactly,
// e0T = e0 + eL pT^2 + et pT - cS a S_y with pT = sqrt(q) a (T - t_y): KIND 4)
// and the sums are  u = sum kappa E,  dt = (1/sqrt(q)) sum kappa (q6 - pp E),  div = sum kappa (-ss E + q7 ss + q8),
// lap = (1/k1) sum kappa (Lh (E + q9) + q10 ss + q11) - a d u   (the rescalings happen once per point, after the
// sweep).  17 VALU + 1 exp per (collocation, point) pair.  KIND selects what a wave needs:
//   0 all four sums;  1 u only (the root: 7 VALU + exp);  4 u only at t = T (terminal samples: 3 VALU + exp, one LDS read);
//   5 u and div only (Euler-Maruyama sites whose eps_PDE is not consumed: 10 VALU + exp);
//   2 / 3 the same on tiles of boundary rows (cL = ct = cS = 0, E = c0): 8 VALU + exp / 1 VALU + exp.
template <int KIND, bool PF>
__device__ __forceinline__ void gp_epilogue_scaled(const GpStageView &st, const f32x16 &acc, int half, float sx, float tx,
                                                   float &au, float &at, float &ad, float &al) {
    // C row = (r&3) + 8*(r>>2) + 4*half: one per-lane base (depends on the half-wave), compile-time row offsets
    const float4 *cb = reinterpret_cast<const float4 *>(__builtin_assume_aligned(st.coef + 4 * kCoefRow * half, 16));
    constexpr int NQ = (KIND == 0 || KIND == 5) ? 3 : (KIND == 1 ? 2 : 1);       // float4 reads per row
    constexpr int Q0 = KIND == 4 ? 3 : 0;                         // first float4 of the row this form reads
    // PF: rows ping-pong between two register sets (the reads of row r+1 are issued before row r is consumed);
    // one row per scheduling region keeps the VGPR budget flat
    float4 q[PF ? 2 : 1][NQ];
    if constexpr (PF) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) q[0][i] = cb[Q0 + i];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cur = PF ? (r & 1) : 0, nxt = cur ^ 1;
        if constexpr (PF) {
            if (r + 1 < 16) {
                const int off = (((r + 1) & 3) + 8 * ((r + 1) >> 2)) * (kCoefRow / 4);
#pragma unroll
                for (int i = 0; i < NQ; ++i) q[nxt][i] = cb[off + Q0 + i];
            }
        } else {
            const int off = ((r & 3) + 8 * (r >> 2)) * (kCoefRow / 4);
#pragma unroll
            for (int i = 0; i < NQ; ++i) q[0][i] = cb[off + Q0 + i];
        }
        __builtin_amdgcn_sched_barrier(0);
        const float lam = acc[r];
        const float kap = __builtin_amdgcn_exp2f(lam);
        const float vsy = q[cur][0].x, vty = q[cur][0].y, ve0 = q[cur][0].z, veL = q[cur][0].w;
        if constexpr (KIND == 3) {
            au = fmaf(kap, ve0, au);
        } else if constexpr (KIND == 4) {       // this form's float4 is (-, e0T, eL, cS): E = e0T + eL Lam + cS (a S_x)
            const float e0T = q[cur][0].y, eLT = q[cur][0].z, cST = q[cur][0].w;
            au = fmaf(kap, fmaf(cST, sx, fmaf(eLT, lam, e0T)), au);
        } else {
            const float pp = tx - vty;
            const float ss = sx - vsy;
            const float Lh = fmaf(pp, pp, lam);
            if constexpr (KIND == 2) {
                const float w = kap * ve0;
                au = fmaf(kap, ve0, au);        // the same rounding as KIND 3: u_hat must not depend on which sums a wave needs
                at = fmaf(-pp, w, at);
                ad = fmaf(-ss, w, ad);
                al = fmaf(Lh, w, al);
            } else {
                const float vet = q[cur][1].x, vcS = q[cur][1].y;
                const float E = fmaf(vcS, ss, fmaf(vet, pp, fmaf(veL, Lh, ve0)));
                au = fmaf(kap, E, au);
                if constexpr (KIND == 0) {
                    const float act = q[cur][1].z, c2 = q[cur][1].w;
                    const float c3 = q[cur][2].x, c4 = q[cur][2].y, c5 = q[cur][2].z, c6 = q[cur][2].w;
                    at = fmaf(kap, fmaf(-pp, E, act), at);
                    ad = fmaf(kap, fmaf(-ss, E, fmaf(c2, ss, c3)), ad);
                    al = fmaf(kap, fmaf(Lh, E + c4, fmaf(c5, ss, c6)), al);
                } else if constexpr (KIND == 5) {
                    const float c2 = q[cur][1].w, c3 = q[cur][2].x;
                    ad = fmaf(kap, fmaf(-ss, E, fmaf(c2, ss, c3)), ad);   // the same operations as KIND 0: div does not depend on the form
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// E-from-MFMA epilogue (fp16x2 mode with fp16-exact collocation points, gp_eval_bf16.hip EM).  The part of E that is bilinear in
// (point, collocation row),
//     E_lin = e0 + et pp + cS ss = [e0 - et vty - cS vsy] + et tx + cS sx           (pp = tx - vty, ss = sx - vsy),
// comes out of ONE extra MFMA per tile (32x32x8, fp16) instead of two vector instructions per pair: the collocation side carries
// S (cE_h, cE_l, et_h, et_h | et_l, cS_h, cS_h, cS_l) as fp16 (h = fp16(v), l = fp16(v - h); S = 2^s brings the largest entry to
// 2^12..2^13), the point side (1, 1, tx_h, tx_l | tx_h, sx_h, sx_l, sx_h): every product of two 11-bit significands is exact in the
// fp32 accumulator, the dropped l*l terms are 2^-22 of their product.  Everything downstream is carried times S (constants
// pre-scaled) and the four sums are multiplied by 2^-s once per point.  Tiles of terminal samples (KIND 4) and of boundary rows
// (KIND 2 / 3) are bound by the matrix pipe or trivial and keep their all-vector forms, with their own row constants.
// LDS slot behind the A planes: 512 B of E fragments (64 lanes x 4 halves) | 512 B `trow`: per row (S e0T, S eL, S cS, S e0) |
// 1 KiB `coef3`: per row  0 vsy  1 vty  2 S eL  3 S q6  |  4 S q7  5 S q8  6 S q10  7 S q11  (q9 = -2 q7).  KIND as in
// gp_epilogue_scaled:   0: 14 VALU + exp (16 before);  5 (u, div): 8 (10);  1 (u): 4 (7);  4, 2, 3: as before.
template <int KIND, bool PF>
__device__ __forceinline__ void gp_epilogue_em(const float *trow_lds, const f32x16 &acc, const f32x16 &accE, int half, float sx, float tx,
                                               float &au, float &at, float &ad, float &al) {
    const float4 *tb = reinterpret_cast<const float4 *>(__builtin_assume_aligned(trow_lds + 4 * 4 * half, 16));        // 1 float4 per row
    const float4 *cb = reinterpret_cast<const float4 *>(__builtin_assume_aligned(trow_lds + 128 + 4 * 8 * half, 16));  // 2 float4 per row
    constexpr bool USE_T = KIND == 2 || KIND == 3 || KIND == 4;                // reads the row's trow float4
    constexpr int NQ = (KIND == 0 || KIND == 5) ? 2 : ((KIND == 3 || KIND == 4) ? 0 : 1);   // coef3 float4 per row
    constexpr int NR = NQ + (USE_T ? 1 : 0);
    float4 q[PF ? 2 : 1][NR];
    auto fetch = [&](int r, float4 (&dst)[NR]) {
        const int row = (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int i = 0; i < NQ; ++i) dst[i] = cb[row * 2 + i];
        if constexpr (USE_T) dst[NQ] = tb[row];
    };
    if constexpr (PF) fetch(0, q[0]);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cur = PF ? (r & 1) : 0, nxt = cur ^ 1;
        if constexpr (PF) {
            if (r + 1 < 16) fetch(r + 1, q[nxt]);
        } else {
            fetch(r, q[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float lam = acc[r];
        const float kap = __builtin_amdgcn_exp2f(lam);
        if constexpr (KIND == 3) {
            au = fmaf(kap, q[cur][0].w, au);
        } else if constexpr (KIND == 4) {       // (S e0T, S eL, S cS, -): E = e0T + eL Lam + cS (a S_x)
            au = fmaf(kap, fmaf(q[cur][0].z, sx, fmaf(q[cur][0].y, lam, q[cur][0].x)), au);
        } else {
            const float vsy = q[cur][0].x, vty = q[cur][0].y;
            const float pp = tx - vty;
            const float Lh = fmaf(pp, pp, lam);
            if constexpr (KIND == 2) {
                const float ss = sx - vsy;
                const float ve0 = q[cur][1].w;
                const float w = kap * ve0;
                au = fmaf(kap, ve0, au);        // the same rounding as KIND 3
                at = fmaf(-pp, w, at);
                ad = fmaf(-ss, w, ad);
                al = fmaf(Lh, w, al);
            } else {
                const float E = fmaf(q[cur][0].z, Lh, accE[r]);
                au = fmaf(kap, E, au);
                if constexpr (KIND == 0 || KIND == 5) {
                    const float ss = sx - vsy;
                    const float c2 = q[cur][1].x, c3 = q[cur][1].y;
                    ad = fmaf(kap, fmaf(-ss, E, fmaf(c2, ss, c3)), ad);
                    if constexpr (KIND == 0) {
                        const float act = q[cur][0].w, c5 = q[cur][1].z, c6 = q[cur][1].w;
                        at = fmaf(kap, fmaf(-pp, E, act), at);
                        al = fmaf(kap, fmaf(Lh, fmaf(c2, -2.0f, E), fmaf(c5, ss, c6)), al);
                    }
                }
            }
        }
    }
}

// PDE residual of the surrogate at one point (models/GP.py:746-769): dt u + mu div u + sigma^2/2 Lap u + f(u, sigma div u);
// for Grad_Dependent_Nonlinear that is the reference's dt + (sigma^2 u - 1/d - sigma^2/2) div + sigma^2/2 Lap (:767-768)
__device__ __forceinline__ float gp_pde_residual(const GpArgs &g, float u, float dt, float dv, float lp) {
    return dt + g.mu * dv + 0.5f * g.sigma * g.sigma * lp + eq_f<float>(g.eq_id, u, g.sigma * dv, g.sigma, (float)g.d);
}

// Monte-Carlo sample sharding: true if every row of [b0, b0 + rows) lies in a tree site this rank does not own
// (site kind 2, scasml_plan_site_kinds).  Ownership alternates between neighbouring sites (units of one terminal
// draw), so every site the block touches is checked, not just its first and last row.
__device__ __forceinline__ bool gp_block_unowned(const GpArgs &g, int64_t b0, int64_t rows) {
    if (b0 >= g.n_inf) return false;
    const int64_t b1 = (b0 + rows < g.n_inf ? b0 + rows : g.n_inf) - 1;
    for (int64_t s = b0 / g.rows_per_site; s <= b1 / g.rows_per_site; ++s)
        if (g.site_u_only[s] != 2) return false;
    return true;
}

int launch_gp_eval_bf16(const GpArgs &g, int split, hipStream_t s);   // gp_eval_bf16.hip

}  // namespace scasml
