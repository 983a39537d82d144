// Pieces shared by the fp32-MFMA and the split-bf16 fused GP evaluation kernels: argument block,
// LDS stage view and the closed-form epilogue (derivation in gp_eval.hip).
#pragma once
#include "common.hpp"

namespace scasml {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GpArgs {
    const float *points;       // n_inf x kp
    const float *colloc_frag;  // [n_tiles][NK4][64][4]
    const uint16_t *colloc_bf16;  // [n_tiles][3 planes][kp/16][64][8] truncated-bf16 planes
    const uint16_t *colloc_f16;   // [n_tiles][2 planes][kp/16][64][8] fp16 planes (h, 2^11 * l)
    const float *coef;         // [n_pad][16]
    float4 *out4;              // n_inf x (u, div, eps, dt)
    float *lap;                // n_inf or null
    int64_t n_inf;
    int32_t n_pad, kp, d;
    float a, sigma;
    int32_t colloc_is_f16;
    const uint8_t *site_u_only;   // per tree site: only u_hat needed (null = every row needs everything)
    int64_t rows_per_site;
    int32_t dbg;   // ablation switches (SCASML_GP_DBG, development only): 1 skip MFMA, 2 skip epilogue, 4 no stagger, 8 stage once
};

struct GpStageView {
    const float4 *y;    // A fragments of one collocation tile (layout depends on the kernel)
    const float *coef;  // [32 rows][16]
};

struct GpConsts {
    float a, a2, ad, kexp, dF;
    float k1, k2;   // folded form: kappa = exp2(k1 * L0 + k2), L0 = a^2 r2 - a d
};

// C row = (r&3) + 8*(r>>2) + 4*half, column = lane & 31.  Coefficients of a collocation row are 8
// consecutive floats (|y|^2, a*sum y, a*t_y, c0, cL, ct, cS, 0): two broadcast ds_read_b128 per row.
// Per collocation row the coefficient tile holds 16 floats (gp_pack_kernel):
//   0 a*sum y   1 a*t_y   2 c0   3 cL  |  4 ct   5 cS   6 a*ct   7 2a*cL  |  8 a*d*cS   9 -4a*cL  10 -2a*cS
//   11 -2a^2*d*cL  |  12 |y|^2   13..15 0
// so the folded kernel consumes exactly three full ds_read_b128 per row (every component used: the
// compiler otherwise narrows the reads to ds_read2_b32 whose 8-bit offsets need an address add each).
constexpr int kCoefRow = 16;

// FOLD: the MFMA already delivered  a^2 (|y|^2 - 2 x.y)  (the collocation planes hold -2 a^2 y and an
// extra K column a^2 |y|^2 against a constant 1 in the point row), and nx[] holds a^2 |x|^2 - a d, so
// L0 = acc + nx = a^2 r2 - a d costs one add and kappa one fma + exp2; every row constant that would
// cost a multiply per element is precomputed.  19 VALU per (collocation, point) pair.
template <int PT, bool FOLD = false, bool PF = true, bool UONLY = false>
__device__ __forceinline__ void gp_epilogue_tile(const GpStageView &st, const f32x16 (&acc)[PT], const GpConsts &c, int half,
                                                 const float (&nx)[PT], const float (&sx)[PT], const float (&tx)[PT],
                                                 float (&au)[PT], float (&at)[PT], float (&ad)[PT], float (&al)[PT]) {
    // C row = (r&3) + 8*(r>>2) + 4*half: one per-lane base (depends on the half-wave), compile-time row offsets
    const float4 *cb = reinterpret_cast<const float4 *>(__builtin_assume_aligned(st.coef + 4 * kCoefRow * half, 16));
    constexpr int NQ = UONLY ? 2 : (FOLD ? 3 : 4);       // float4 reads per row (u_hat alone needs sy, ty, c0, cL, ct, cS)
    // PF: rows ping-pong between two register sets (the reads of row r+1 are issued before row r is
    // consumed); without PF one set is used and the other resident waves cover the LDS latency.  One
    // row (PT independent chains) per scheduling region keeps the VGPR budget flat.
    float4 q[PF ? 2 : 1][NQ];
    if constexpr (PF) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) q[0][i] = cb[i];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cur = PF ? (r & 1) : 0, nxt = cur ^ 1;
        if constexpr (PF) {
            if (r + 1 < 16) {
                const int off = (((r + 1) & 3) + 8 * ((r + 1) >> 2)) * (kCoefRow / 4);
#pragma unroll
                for (int i = 0; i < NQ; ++i) q[nxt][i] = cb[off + i];
            }
        } else {
            const int off = ((r & 3) + 8 * (r >> 2)) * (kCoefRow / 4);
#pragma unroll
            for (int i = 0; i < NQ; ++i) q[0][i] = cb[off + i];
        }
        __builtin_amdgcn_sched_barrier(0);
        const float vsy = q[cur][0].x, vty = q[cur][0].y, vc0 = q[cur][0].z, vcL = q[cur][0].w;
        const float vct = q[cur][1].x, vcS = q[cur][1].y;
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const float pp = tx[p] - vty;                  // a * r_t
            const float ss = sx[p] - vsy;                  // a * S
            if constexpr (UONLY) {                           // terminal-time points and the root: u_hat only, 10 VALU
                static_assert(FOLD, "u-only epilogue is written for the folded form");
                const float L0 = acc[p][r] + nx[p];
                const float kap = __builtin_amdgcn_exp2f(fmaf(L0, c.k1, c.k2));
                const float L = fmaf(-pp, pp, L0);
                au[p] = fmaf(kap, fmaf(vcS, ss, fmaf(vct, pp, fmaf(vcL, L, vc0))), au[p]);
            } else if constexpr (FOLD) {
                const float act = q[cur][1].z, c2 = q[cur][1].w;
                const float c3 = q[cur][2].x, c4 = q[cur][2].y, c5 = q[cur][2].z, c6 = q[cur][2].w;
                const float L0 = acc[p][r] + nx[p];        // a^2 r2 - a d
                const float kap = __builtin_amdgcn_exp2f(fmaf(L0, c.k1, c.k2));
                const float L = fmaf(-pp, pp, L0);
                const float E = fmaf(vcS, ss, fmaf(vct, pp, fmaf(vcL, L, vc0)));
                au[p] = fmaf(kap, E, au[p]);
                at[p] = fmaf(kap, fmaf(-pp, E, act), at[p]);
                ad[p] = fmaf(kap, fmaf(-ss, E, fmaf(c2, ss, c3)), ad[p]);
                al[p] = fmaf(kap, fmaf(L, E, fmaf(c4, L, fmaf(c5, ss, c6))), al[p]);
            } else {
                const float vny = q[cur][NQ - 1].x;
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
