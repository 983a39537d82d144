// Fused GP posterior evaluation with x.y on the 16-bit matrix cores (split arithmetic).
//
// Why: on gfx950 the fp32-input MFMA (v_mfma_f32_32x32x2_f32) runs at the fp32 VALU rate, 1/16 of the 16-bit MFMA, and in this kernel matrix
// time does not hide under vector time (the launch is the sum of the two: profiles/HISTORY.md 4.2, 4.4) -- so the matrix part is made short.
// Each fp32 operand is split by truncation into bf16 planes  v = hi + mid + lo  (exact to 2^-24 |v|); the products hi*hi, hi*mid, mid*hi, mid*mid, hi*lo,
// lo*hi carry every term down to 2^-24, i.e. x.y is as exact as the fp32 MFMA (SPLIT = 3, 6 MFMAs per
// K-step); SPLIT = 2 keeps hi/mid only (3 MFMAs, ~2^-17 per product).  Accumulation is fp32 in the MFMA.
//
// MODE 22 ("fp16x2"): two fp16 planes  v = h + l  (h = fp16(v), l = fp16(v - h): |l| <= 2^-12 |v| lands in the
// fp16 subnormal range for |v| < 1/4 and is then exact to 2^-25 absolute -- the MFMA honours fp16 subnormals, which
// tests/test_gpu_gp.py would show at once if it did not).  The products h*h and h*l + l*h are exact in the fp32
// accumulator and go into ONE accumulator (l*l ~ 2^-24 is dropped): 3 MFMAs per K-step instead of 6, product
// accuracy ~2^-22 instead of 2^-24, no combining instruction.  In this mode the scale factor sits on the
// point side (B = planes of 2 a^2 q x, constants 1 and 1 in columns kp-3 and kp-2) and the collocation planes
// hold y itself plus k1 a^2 |y|^2 as (h, l) in those two columns; when every collocation coordinate is
// exactly fp16 -- the reference's deepxde float16 arrays are -- plane l_y is zero, so the l_y * h_x MFMA
// and the staging of that plane are dropped (YEXACT): 2 MFMAs per K-step.
//
// In that case (fp16x2, YEXACT) the epilogue takes the bilinear part of E = e0 + eL Lh + et pp + cS ss from one extra
// v_mfma_f32_32x32x8_f16 per tile (EM below; gp_common.hpp, gp_epilogue_em): 10 % fewer vector instructions in a kernel whose
// vector ALUs are the busier pipe.
//
// Structure: workgroup of 8 waves, 32 points per wave held in VGPRs as 16-bit planes for the whole sweep; per
// collocation tile one LDS slot [planes*KS KiB of A fragments | 2 KiB constants] filled two tiles ahead by
// LDS-DMA; one barrier per tile.  The operands are scaled so that the product is the exponent of the kernel
// (gp_common.hpp, gp_epilogue_scaled); tiles of boundary rows run a shorter epilogue.  Every wave does MFMAs then
// epilogue of the same tile; the four waves of a SIMD belong to four unsynchronised workgroups.
#include <stdlib.h>
#include <type_traits>

#include "gp_common.hpp"
#include "gp_mfma16.hpp"

#ifndef SCASML_GP_EM
#define SCASML_GP_EM 1   // 0: development A/B against the all-vector epilogue
#endif

namespace scasml {

// KS = kp / 16 K-steps of the 32x32x16 MFMA; half-wave h covers k in [h*8*KS, (h+1)*8*KS)
template <int KS, int SPLIT, bool PF, bool F16, bool YEXACT>
__device__ __forceinline__ void gp_mfma_tile_bf16(const float4 *lds_a, const s16x8 (&xb)[SPLIT][KS], f32x16 &acc, int lane) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // A fragments ping-pong between two register sets selected by the (compile-time) parity of the
    // step: the ds_reads of step s+1 are issued before the MFMAs of step s, with no register copies.
    constexpr int NPL = YEXACT ? 1 : SPLIT;      // A planes actually staged and read
    Frag a[PF ? 2 : 1][SPLIT];
    if constexpr (PF) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) a[0][pl].f = lds_a[(pl * KS) * 64 + lane];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int cur = PF ? (s & 1) : 0, nxt = cur ^ 1;
        if constexpr (PF) {
            if (s + 1 < KS) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) a[nxt][pl].f = lds_a[(pl * KS + s + 1) * 64 + lane];
            }
        } else {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) a[0][pl].f = lds_a[(pl * KS + s) * 64 + lane];
        }
        if constexpr (F16) {   // plane 0 = h, plane 1 = l; small terms first
            Frag b0, b1;
            b0.v = xb[0][s];
            b1.v = xb[1][s];
            if constexpr (!YEXACT) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cur][1].h, b0.h, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cur][0].h, b1.h, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cur][0].h, b0.h, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            continue;
        }
        // small terms first; plane 0 = hi, 1 = mid, 2 = lo
        if (SPLIT == 3) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][2].v, xb[0][s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0].v, xb[SPLIT - 1][s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][1].v, xb[1][s], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][1].v, xb[0][s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0].v, xb[1][s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0].v, xb[0][s], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// WPB waves per workgroup (one workgroup per CU): 8 -> 2 waves/SIMD (<= 256 VGPRs, register prefetch
// of LDS operands), 12 -> 3 waves/SIMD (<= 168 VGPRs, no register prefetch: a single wave issues a VALU
// instruction only every ~7 cycles, so SIMD-level VALU throughput -- the epilogue -- scales with the
// number of resident waves until the matrix pipe becomes the limit).
// BPC = workgroups meant to be co-resident per CU (occupancy = WPB/4 * BPC waves per SIMD): with two
// 8-wave workgroups per CU one workgroup's point-row prologue and final stores overlap the other's sweep.
template <int KS, int SPLIT, int WPB, bool F16, int BPC, bool YEXACT>
__global__ __launch_bounds__(WPB * 64, WPB / 4 * BPC) void gp_eval_bf16_kernel(const GpArgs g) {
    constexpr int NPL = YEXACT ? 1 : SPLIT;          // A planes staged per tile
    constexpr bool PF = WPB * BPC <= 8 || (F16 && YEXACT && KS <= 7);
    // E-from-MFMA epilogue (gp_common.hpp, gp_epilogue_em): the slot's two coefficient KiB become one KiB of A fragments for the
    // linear part of E and one KiB of row constants (8 floats per row)
    constexpr bool EM = F16 && YEXACT && SCASML_GP_EM;
    static_assert(WPB == 4 || WPB == 8 || WPB == 12 || WPB == 16, "waves per workgroup");
    constexpr int STAGE = NPL * KS * 256 + 512;         // floats per LDS slot (A fragments + 32 rows x 16 coefficients, or E plane + 32 x 8)
    constexpr int NSLOT = (4 * STAGE * 4 * BPC <= 144 * 1024) ? 4 : 3;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // NSLOT slots
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    // (Walking the workgroups ACROSS the sites, so that matrix-bound terminal tiles and vector-bound Euler-Maruyama tiles share a
    // SIMD at all times, was measured: 7.94 / 7.96 ms against 7.94 / 7.85 in buffer order -- nothing.)
    const int64_t blk = blockIdx.x;
    const int64_t p0 = (blk * WPB + wv) * 32;
    const int n_tiles = g.n_pad / 32;

    // Monte-Carlo sample sharding: a workgroup whose rows all belong to sites this rank does not own has
    // nothing to do (block-uniform, so the barriers below stay consistent)
    if (g.site_u_only && g.rows_per_site >= 32 && gp_block_unowned(g, blk * WPB * 32, WPB * 32)) return;
    // stage one collocation tile: NCHUNK 1-KiB chunks (A fragments (plane, step), then the two coefficient KiB).
    // Wave w issues chunks w, w + WPB, ...: CLO of them, one more on the first NCHUNK % WPB waves (an LDS-DMA costs
    // its wave 60-185 issue cycles, MI355X_MICROARCH.md, so no padding copies).  The count is a wave-uniform
    // constant, so a counted s_waitcnt vmcnt(own count) means "everything but the newest tile has landed".
    constexpr int NCHUNK = NPL * KS + 2;
    constexpr int CLO = NCHUNK / WPB, CREM = NCHUNK % WPB;
    const int wvs = __builtin_amdgcn_readfirstlane(wv);              // the wave index as a scalar: staging addresses stay on the SALU
    const bool extra = wvs < CREM;                                   // this wave issues CLO + 1
    // low 32 bits of a flat pointer into the LDS aperture = the LDS byte address
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    auto stage = [&](int tile, int slot) {
        const uint32_t dst = lds_base + (uint32_t)(slot * STAGE) * 4u;
        const float *srcA = reinterpret_cast<const float *>(F16 ? g.colloc_f16 : g.colloc_bf16) + (int64_t)tile * (F16 ? 2 : 3) * KS * 256;
        const float *srcC = g.coef2 + (int64_t)tile * 512 - (int64_t)NPL * KS * 256;   // chunk c >= NPL*KS -> coef + (c - NPL*KS) KiB
        const float *srcE = reinterpret_cast<const float *>(g.eplane) + (int64_t)tile * 256 - (int64_t)NPL * KS * 256;        // EM: chunk NPL*KS
        const float *src3 = g.coef3 + (int64_t)tile * 256 - (int64_t)(NPL * KS + 1) * 256;                                  // EM: chunk NPL*KS + 1
        auto chunk = [&](int c) {
            const float *src = c < NPL * KS ? srcA : (EM ? (c == NPL * KS ? srcE : src3) : srcC);
            glds16_asm(src + c * 256, (uint32_t)lane * 16u, dst + (uint32_t)c * 1024u);
        };
#pragma unroll
        for (int i = 0; i < CLO; ++i) chunk(wvs + i * WPB);
        if (CREM && extra) chunk(wvs + CLO * WPB);
    };
    // NSLOT = 4: tiles are fetched TWO ahead and the per-tile rendezvous is a raw s_barrier behind a counted
    // vmcnt, so the newest tile's DMA stays in flight across the barrier (__syncthreads() would drain it:
    // vmcnt(0)); NSLOT = 3 (slots too big for four): one tile ahead, full drain.
    auto rendezvous = [&](bool newest_may_fly) {
        if (NSLOT == 4 && newest_may_fly) {
            if (CREM && extra) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CLO + 1) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CLO) : "memory");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // the raw barrier builtin has no memory semantics for the compiler (unlike __syncthreads()): the two
        // "memory"-clobbering statements around it keep every LDS read on its own side; this wave's own reads
        // of the slot about to be refilled have completed because their results were consumed above
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    stage(0, 0);
    if (NSLOT == 4 && n_tiles > 1) stage(1, 1);

    // ---- this wave's 32 points: fp32 row halves -> a*sum x, sqrt(q) a t, and the 16-bit planes ------
    // exponent scaling (gp_common.hpp): q = log2(e)/(2a), k1 = -q; the planes hold 2 a^2 q x (fp16 mode) or x (bf16
    // modes: the factor sits on the collocation side), the constants that meet k1 a^2 |y|^2, and k1 a^2 |x|^2 in
    // the last column (half 1, step KS-1, element 7) against the collocation side's constant 1
    const float qs = 0.5f * 1.44269504088896341f / g.a, k1 = -qs;
    s16x8 xb[SPLIT][KS];
    float sx, tx;
    {
        int64_t row = p0 + col;
        if (row >= g.n_inf) row = g.n_inf - 1;  // shadow rows, never stored
        const int kbase = half * (8 * KS);
        const float4 *src = reinterpret_cast<const float4 *>(g.points + row * g.kp + kbase);
        // No per-element selects: the point rows hold zeros beyond column d (scasml_hip.h), so |x|^2 (with t) and the
        // row sum need no masks, t is fetched on its own, and the three constants live in the LAST three columns
        // (half 1, step KS-1, elements 5..7: compile-time fragment positions), patched once after the loop.
        float pn = 0.0f, ps = 0.0f;
        const float pt = g.points[row * g.kp + g.d];
        const float fold = F16 ? 2.0f * g.a * g.a * qs : 1.0f;
        auto make_planes = [&](const float (&t)[8], Frag &fh, Frag &fm, Frag &fl) {
            uint32_t hb[8], mb[8], lb[8];
            if constexpr (!F16) {
#pragma unroll
                for (int c = 0; c < 8; ++c) split3(t[c], hb[c], mb[c], lb[c]);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {   // element 2c in the low half-word, 2c+1 in the high one
                if constexpr (F16) {
                    const float h0 = (float)(_Float16)t[2 * c], h1 = (float)(_Float16)t[2 * c + 1];
                    fh.u[c] = pack_h2(t[2 * c], t[2 * c + 1]);
                    fm.u[c] = pack_h2(t[2 * c] - h0, t[2 * c + 1] - h1);   // unscaled: fp16 subnormals carry it (see header)
                    fl.u[c] = 0;
                } else {
                    fh.u[c] = (hb[2 * c] >> 16) | hb[2 * c + 1];
                    fm.u[c] = (mb[2 * c] >> 16) | mb[2 * c + 1];
                    fl.u[c] = (lb[2 * c] >> 16) | lb[2 * c + 1];
                }
            }
        };
        float tlast[8];   // the operand values of the last step, kept to patch its last three columns once |x|^2 is known
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 q0 = src[2 * s], q1 = src[2 * s + 1];
            const float e[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            float t[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                pn = fmaf(e[c], e[c], pn);
                ps += e[c];
                t[c] = fold * e[c];
            }
            if (s == KS - 1) {
#pragma unroll
                for (int c = 0; c < 8; ++c) tlast[c] = t[c];
            } else {
                Frag fh, fm, fl;
                make_planes(t, fh, fm, fl);
                xb[0][s] = fh.v;
                xb[1][s] = fm.v;
                if (SPLIT == 3) xb[SPLIT - 1][s] = fl.v;
            }
        }
        pn += __shfl_xor(pn, 32);
        ps += __shfl_xor(ps, 32);
        // columns kp-3, kp-2 (fp16 mode: against the (h, l) parts of k1 a^2 |y|^2; bf16 modes: kp-2 against the whole
        // value) hold the constant 1, column kp-1 holds k1 a^2 |x|^2 against the collocation side's constant 1
        if (F16) tlast[5] = half ? 1.0f : tlast[5];
        tlast[6] = half ? 1.0f : tlast[6];
        tlast[7] = half ? k1 * g.a * g.a * pn : tlast[7];
        {
            Frag fh, fm, fl;
            make_planes(tlast, fh, fm, fl);
            xb[0][KS - 1] = fh.v;
            xb[1][KS - 1] = fm.v;
            if (SPLIT == 3) xb[SPLIT - 1][KS - 1] = fl.v;
        }
        sx = g.a * (ps - pt);                 // the row sum includes t
        tx = sqrtf(qs) * g.a * pt;
    }
    float au = 0.0f, at = 0.0f, ad = 0.0f, al = 0.0f;

    auto a_of = [&](int slot) { return reinterpret_cast<const float4 *>(lds + slot * STAGE); };
    auto view = [&](int slot) {
        const float *b = lds + slot * STAGE;
        return GpStageView{reinterpret_cast<const float4 *>(b), b + NPL * KS * 256};
    };
    // site-major buffers: all 32 rows of this wave belong to one tree site (rows_per_site a multiple of 32: the layout the
    // solvers use) or to two neighbouring ones.  form 1: only u_hat is consumed there (the root, terminal samples): the
    // epilogue drops the dt / div / Lap sums; form 2: additionally every row has t = T exactly (terminal samples, site kind
    // 3), which folds the time terms into the row constants.  Form 2 is taken only for single-site tiles, so that the
    // arithmetic a point sees never depends on how its batch was cut.
    int form = 0;
    if (g.site_u_only && g.rows_per_site >= 32) {
        const int64_t last = p0 + 31 < g.n_inf ? p0 + 31 : g.n_inf - 1;
        const int64_t s0 = p0 < g.n_inf ? p0 / g.rows_per_site : 0, s1 = last / g.rows_per_site;
        const int k0 = g.site_u_only[s0], k1s = g.site_u_only[s1];
        if ((k0 == 1 || k0 == 3) && (k1s == 1 || k1s == 3)) form = (k0 == 3 && s0 == s1 && g.rows_per_site % 32 == 0) ? 2 : 1;
        else if (k0 == 4 && s0 == s1 && g.rows_per_site % 32 == 0) form = 3;   // u_hat and div only
    }
    form = __builtin_amdgcn_readfirstlane(form);
    // EM: the point side of the E-plane product, (1, 1, tx_h, tx_l | tx_h, sx_h, sx_l, sx_h) as fp16 (gp_common.hpp, gp_epilogue_em)
    h16x4 xe;
    if constexpr (EM) {
        const float txh = (float)(_Float16)tx, sxh = (float)(_Float16)sx;
        union {
            h16x4 h;
            uint32_t u[2];
        } b;
        b.u[0] = half ? pack_h2(tx, sx) : pack_h2(1.0f, 1.0f);
        b.u[1] = half ? pack_h2(sx - sxh, sx) : pack_h2(tx, tx - txh);
        xe = b.h;
    }
    f32x16 acc, accE;
    uint32_t region;
    asm volatile("s_mov_b32 %0, -1" : "=s"(region));   // see tile(): basic-block boundaries, never false
    // the whole sweep is instantiated three times (full / u-only / u-only at t = T) and the wave-uniform choice is made
    // once, outside the tile loops: a branch inside them costs registers (the allocator then spills the point tile)
    auto sweep = [&](auto fm) {
        constexpr int FORM = decltype(fm)::value;
        constexpr bool UO = FORM == 1 || FORM == 2;
        constexpr int AHEAD = NSLOT == 4 ? 2 : 1;
        auto tile = [&](int jt, auto kind) {
            // (Measured and rejected here, profiles/r02_gp_eval_experiments.txt: s_setprio around either phase -- no effect;
            // a packed-f32 epilogue on two rows per instruction -- 45 % fewer VALU instructions, 6 % slower.)
            // Three basic blocks per tile -- staging, MFMAs, epilogue -- behind a scalar the optimiser cannot see through
            // (`region`, always all ones).  As one block the register allocator interleaves the point planes with the
            // epilogue's temporaries and spills 450-620 B per lane inside the loop (72 ms instead of 8, round 1);
            // __builtin_amdgcn_sched_barrier between the phases does not prevent that (tried: same spills), separate
            // blocks do.  tests/test_abi_and_host.py fails the build on scratch instructions inside these loops.
#ifdef SCASML_ABLATION     // development builds only (tools/build_variant.sh abl gp_eval_bf16.hip -DSCASML_ABLATION, and the same flag on gp_eval.hip): phases switched off by g.dbg, results are garbage
            if (jt + AHEAD < n_tiles && !(g.dbg & 8)) stage(jt + AHEAD, (jt + AHEAD) % NSLOT);
            if (!(g.dbg & 1)) gp_mfma_tile_bf16<KS, SPLIT, PF, F16, YEXACT>(a_of(jt % NSLOT), xb, acc, lane);
            if (!(g.dbg & 2)) gp_epilogue_scaled<decltype(kind)::value, PF>(view(jt % NSLOT), acc, half, sx, tx, au, at, ad, al);
#else
            if (jt + AHEAD < n_tiles) stage(jt + AHEAD, (jt + AHEAD) % NSLOT);
            constexpr int KIND = decltype(kind)::value;
            if (region & 1) {
                gp_mfma_tile_bf16<KS, SPLIT, PF, F16, YEXACT>(a_of(jt % NSLOT), xb, acc, lane);
                if constexpr (EM && (KIND == 0 || KIND == 1 || KIND == 5)) {
                    union {
                        float2 f;
                        h16x4 h;
                    } ae;
                    ae.f = reinterpret_cast<const float2 *>(lds + (jt % NSLOT) * STAGE + NPL * KS * 256)[lane];
#pragma unroll
                    for (int r = 0; r < 16; ++r) accE[r] = 0.0f;
                    accE = __builtin_amdgcn_mfma_f32_32x32x8f16(ae.h, xe, accE, 0, 0, 0);
                }
            }
            if (region & 2) {
                if constexpr (EM) gp_epilogue_em<KIND, PF>(lds + (jt % NSLOT) * STAGE + NPL * KS * 256 + 128, acc, accE, half, sx, tx, au, at, ad, al);
                else gp_epilogue_scaled<KIND, PF>(view(jt % NSLOT), acc, half, sx, tx, au, at, ad, al);
            }
#endif
            rendezvous(jt + AHEAD < n_tiles);
        };
        const int nb0 = g.first_bdy_tile < n_tiles ? g.first_bdy_tile : n_tiles;
        for (int jt = 0; jt < nb0; ++jt) tile(jt, std::integral_constant<int, FORM == 3 ? 5 : (FORM == 2 ? 4 : (UO ? 1 : 0))>{});
        for (int jt = nb0; jt < n_tiles; ++jt) tile(jt, std::integral_constant<int, UO ? 3 : 2>{});   // boundary rows only
    };
    rendezvous(n_tiles > 1);  // tile 0 has landed (tile 1 may still be in flight)
    if (form == 3) sweep(std::integral_constant<int, 3>{});
    else if (form == 2) sweep(std::integral_constant<int, 2>{});
    else if (form == 1) sweep(std::integral_constant<int, 1>{});
    else sweep(std::integral_constant<int, 0>{});

    // undo the exponent units (gp_common.hpp): dt = sum / sqrt(q), lap = sum / k1 - a d u
    const float s2 = g.sigma * g.sigma;
    if constexpr (EM) {   // the EM epilogue carries E times 2^s
        const float inv = g.escale[0];
        au *= inv;
        at *= inv;
        ad *= inv;
        al *= inv;
    }
    const float u = au + __shfl_xor(au, 32);
    const float dt = (at + __shfl_xor(at, 32)) / sqrtf(qs);
    const float dv = ad + __shfl_xor(ad, 32);
    const float lp = (al + __shfl_xor(al, 32)) / k1 - g.a * (float)g.d * u;
    const int64_t row = p0 + col;
    if (half == 0 && row < g.n_inf) {
        const float eps = gp_pde_residual(g, u, dt, dv, lp);   // models/GP.py:767-768
        g.out4[row] = make_float4(u, dv, eps, dt);
        if (g.lap) g.lap[row] = lp;
    }
}

template <int KS, int SPLIT, bool F16, bool YEXACT, int WPB, int BPC>
static int launch_cfg(const GpArgs &g, hipStream_t s) {
    const int64_t waves = (g.n_inf + 31) / 32;
    const int64_t blocks = (waves + WPB - 1) / WPB;
    if (blocks > 0x7FFFFFFF) return fail(SCASML_ERR_UNSUPPORTED, "gp_eval: too many points");
    constexpr size_t stage_bytes = ((YEXACT ? 1 : SPLIT) * KS * 256 + 512) * sizeof(float);
    constexpr size_t lds_bytes = ((4 * stage_bytes * BPC <= 144 * 1024) ? 4 : 3) * stage_bytes;
    static_assert(lds_bytes <= 160 * 1024, "LDS slots exceed 160 KiB");
    auto kern = gp_eval_bf16_kernel<KS, SPLIT, WPB, F16, BPC, YEXACT>;
    if (lds_bytes > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return fail(SCASML_ERR_HIP, "gp_eval: cannot reserve %zu bytes of LDS", lds_bytes);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WPB * 64), lds_bytes, s, g);
    return check_launch("gp_eval(bf16) launch");
}

template <int KS, int SPLIT, bool F16, bool YEXACT>
static int launch_one(const GpArgs &g, hipStream_t s) {
    // three waves per SIMD while the 16-bit planes of the point tile (SPLIT*4*KS VGPRs) leave room under 168
    constexpr int REGS = SPLIT * 4 * KS + (F16 ? 16 : 0);       // point-tile planes (+ head-room the fp16 prologue needs)
    constexpr int WPS = REGS <= 72 ? 4 : (REGS <= 96 ? 3 : 2);  // waves per SIMD the VGPR budget allows
    constexpr int BPC = WPS == 4 ? 2 : 1;
    constexpr int WPB = WPS * 4 / BPC;
    // four waves per SIMD as four 4-wave workgroups per CU when their 4 x 4 LDS slots fit (finer turnover: while one
    // workgroup reads its point rows the other three sweep; measured 9.19 vs 9.34 ms for 2 x 8 waves, 10.0 for 1 x 16)
    constexpr size_t stage_bytes = ((YEXACT ? 1 : SPLIT) * KS * 256 + 512) * sizeof(float);
    if constexpr (WPS == 4 && 16 * stage_bytes <= 144 * 1024) return launch_cfg<KS, SPLIT, F16, YEXACT, 4, 4>(g, s);
    return launch_cfg<KS, SPLIT, F16, YEXACT, WPB, BPC>(g, s);
}

template <int SPLIT, bool F16, bool YEXACT>
static int launch_split(const GpArgs &g, hipStream_t s) {
    switch (g.kp / 16) {
#define SCASML_CASE(K) \
    case K: return launch_one<K, SPLIT, F16, YEXACT>(g, s);
        SCASML_CASE(1) SCASML_CASE(2) SCASML_CASE(3) SCASML_CASE(4) SCASML_CASE(5) SCASML_CASE(6) SCASML_CASE(7) SCASML_CASE(8)
        SCASML_CASE(9) SCASML_CASE(10) SCASML_CASE(11) SCASML_CASE(12) SCASML_CASE(13) SCASML_CASE(14) SCASML_CASE(15) SCASML_CASE(16)
#undef SCASML_CASE
    }
    return fail(SCASML_ERR_UNSUPPORTED, "gp_eval: kp=%d", g.kp);
}

int launch_gp_eval_bf16(const GpArgs &g, int split, hipStream_t s) {
    if (split == 22) return g.colloc_is_f16 ? launch_split<2, true, true>(g, s) : launch_split<2, true, false>(g, s);
    return split == 3 ? launch_split<3, false, false>(g, s) : launch_split<2, false, false>(g, s);
}

}  // namespace scasml
