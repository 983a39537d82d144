// Fused GP posterior evaluation, fp16x2 split arithmetic, "collocation per lane" form (GP.eval_split = 22).
//
// What tools/ubench_valu_rate.hip and tools/ubench_valu_forms.hip measured on gfx950, and what this kernel does
// about it:
//  * an MFMA-only wave and a VALU-only wave sharing a SIMD take the SUM of their times (the wave that issues
//    MFMAs back to back holds the issue port), but MFMAs interleaved with plain f32 VALU in ONE wave overlap
//    fully -> every wave runs the epilogue of collocation subtile q with the MFMAs of subtile q+1 sprinkled
//    through it (two accumulator sets), instead of separate "MFMA phase" / "epilogue phase" waves;
//  * v_pk_fma_f32 does not overlap with MFMA at all, VALU with an SGPR source issues at half rate, a lone wave
//    issues one VALU per ~5 cycles and two or more per ~2.4 -> scalar f32 only, loop constants in VGPRs,
//    >= 2 waves per SIMD;
//  * a broadcast ds_read_b128 still costs 4 LDS cycles: with the 32x32 tiling (collocation = rows) the 12
//    per-collocation constants were re-read from LDS for every accumulator row (48 reads per tile and wave,
//    LDS as busy as the VALU).  Here the MFMA is v_mfma_f32_16x16x32_f16 with the POINTS as rows and the
//    collocation points as columns: lane (g, c) holds D[4g..4g+3][c], so the constants of collocation c sit in
//    that lane's VGPRs (3 conflict-free ds_read_b128 per subtile), the per-point quantities are 4 VGPRs each,
//    and the 16 column partial sums are reduced across lanes once, after the sweep.
//
// Arithmetic is unchanged from the 32x32 fp16x2 path (gp_eval_bf16.hip, MODE 22): v = h + 2^-11 l', point planes
// hold -2 a^2 x and the constants (1, 2^-11) that meet a^2 |y|^2 = (h, l') in two spare columns; h*h goes to one
// accumulator, h*l' + l'*h to a second one entering as 2^-11 * acc2; YEXACT (fp16 collocation points, the
// reference's deepxde arrays) drops the l'_y plane.  K is padded to a multiple of 32 with zeros.
#include <type_traits>

#include "gp_common.hpp"

#ifndef SCASML_GP_ABLATE
#define SCASML_GP_ABLATE 0
#endif

namespace scasml {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

union FragH {
    h16x8 h;
    uint32_t u[4];
    float4 f;
};

__device__ __forceinline__ uint32_t pack_h2f(float a, float b) {   // two fp16 (RNE) in one dword, a low
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    return (uint32_t)__builtin_bit_cast(unsigned short, ha) | ((uint32_t)__builtin_bit_cast(unsigned short, hb) << 16);
}

// one 16-byte-per-lane LDS-DMA from inline asm (see gp_eval_bf16.hip: the builtin form makes hipcc drain it at once)
__device__ __forceinline__ void glds16(const float *gsrc_lane, uint32_t lds_byte_addr_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc_lane), "s"(lds_byte_addr_uniform)
                 : "memory");
}

struct PairConsts {
    float k1, k2;   // kappa = exp2(k1 * L0 + k2), both kept in VGPRs
};

// one (point row, collocation column) pair; q = the column's 12 constants (layout: gp_pack_kernel, coef_t).
// `mid()` runs between the kappa/E half and the accumulation half: the caller issues an MFMA of the next
// subtile there, so that matrix instructions and VALU alternate in program order (back-to-back MFMAs would
// stall this wave's VALU behind the busy matrix pipe).
template <bool UO, class Mid>
__device__ __forceinline__ void pair_eval(float L0, float sx, float tx, const float4 (&q)[3], const PairConsts &c, float &au,
                                          float &at, float &ad, float &al, Mid mid) {
    const float vsy = q[0].x, vty = q[0].y, vc0 = q[0].z, vcL = q[0].w;
    const float vct = q[1].x, vcS = q[1].y;
    const float pp = tx - vty;                  // a * r_t
    const float ss = sx - vsy;                  // a * S
    const float kap = __builtin_amdgcn_exp2f(fmaf(L0, c.k1, c.k2));
    const float L = fmaf(-pp, pp, L0);
    const float E = fmaf(vcS, ss, fmaf(vct, pp, fmaf(vcL, L, vc0)));
    mid();
    au = fmaf(kap, E, au);
    if constexpr (!UO) {
        const float act = q[1].z, c2 = q[1].w;
        const float c3 = q[2].x, c4 = q[2].y, c5 = q[2].z, c6 = q[2].w;
        at = fmaf(kap, fmaf(-pp, E, act), at);
        ad = fmaf(kap, fmaf(-ss, E, fmaf(c2, ss, c3)), ad);
        al = fmaf(kap, fmaf(L, E, fmaf(c4, L, fmaf(c5, ss, c6))), al);
    }
}

// KS4 = K-steps of 32; PT = 16-point subtiles per wave; WPB waves per workgroup, BPC workgroups per CU
template <int KS4, int PT, int WPB, int BPC, bool YEXACT>
__global__ __launch_bounds__(WPB * 64, WPB / 4 * BPC) void gp_eval_f16_kernel(const GpArgs g) {
    constexpr int NPL = YEXACT ? 1 : 2;                  // collocation planes staged
    constexpr int NB = 2 * NPL * KS4;                    // 1-KiB B-fragment chunks per 32-collocation tile: [plane][sub][step]
    constexpr int NCHUNK = NB + 2;                       // + 2 KiB of per-collocation constants
    constexpr int STAGE = NCHUNK * 256;                  // floats per LDS slot
    constexpr int NSLOT = 4, AHEAD = 3;
    constexpr int CPW = (NCHUNK + WPB - 1) / WPB;
    constexpr int MPS = YEXACT ? 2 : 3;                  // MFMAs per (K-step, point subtile)
    constexpr int NM = KS4 * PT * MPS;                   // MFMAs per collocation subtile
    constexpr int NE = 4 * PT;                           // pair evaluations per lane and collocation subtile
    constexpr int NG = 2 * NE;                           // MFMA issue points per subtile (two per pair evaluation)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 15, grp = lane >> 4;
    const int64_t p0 = ((int64_t)blockIdx.x * WPB + wv) * (16 * PT);
    const int n_tiles = g.n_pad / 32;

    // Monte-Carlo sample sharding: a workgroup whose rows all belong to sites this rank does not own has nothing to do
    if (g.site_u_only && g.rows_per_site >= 16 * PT && gp_block_unowned(g, (int64_t)blockIdx.x * WPB * 16 * PT, WPB * 16 * PT)) return;
    // every wave issues exactly CPW LDS-DMAs per tile (surplus ones repeat the last chunk), so a counted
    // s_waitcnt vmcnt(CPW) means "everything but the newest tile has landed"
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    auto stage = [&](int tile, int slot) {
        const uint32_t dst = lds_base + (uint32_t)(slot * STAGE) * 4u;
        const float *srcB = reinterpret_cast<const float *>(g.colloc_f16t) + (int64_t)tile * (4 * KS4) * 256;   // 2 planes stored
        const float *srcC = g.coef_t + (int64_t)tile * 512 - (int64_t)NB * 256;
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            int c = wv + i * WPB;
            c = c < NCHUNK ? c : NCHUNK - 1;
            const float *src = c < NB ? srcB : srcC;
            glds16(src + c * 256 + lane * 4, (uint32_t)__builtin_amdgcn_readfirstlane((int)(dst + (uint32_t)c * 1024u)));
        }
    };
    // The tile loop body must stay ONE basic block (with a branch in it LLVM sinks the whole epilogue below the
    // branch, away from the MFMAs it is meant to overlap): a tile is staged on every iteration -- past the end the
    // last tile again, into a slot only the dropped look-ahead products read -- so the wait is always vmcnt(CPW).
    auto rendezvous = [&]() {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CPW) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    constexpr bool once = (SCASML_GP_ABLATE & 8) != 0;
    const int last_tile = n_tiles - 1;
    stage(0, 0);
    stage(1 < last_tile ? 1 : last_tile, 1);
    stage(2 < last_tile ? 2 : last_tile, 2);

    // ---- this wave's 16*PT points: A fragments (row = lane & 15, k = 32 s + 8 grp + c) of the two fp16 planes ----
    h16x8 xh[PT][KS4], xl[PT][KS4];
    float nx[PT][4], sx[PT][4], tx[PT][4];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        int64_t row = p0 + 16 * p + col;
        if (row >= g.n_inf) row = g.n_inf - 1;  // shadow rows, never stored
        const float *src = g.points + row * g.kp;
        float pn = 0.0f, ps = 0.0f, pt = 0.0f;
        const float fold = -2.0f * g.a * g.a;
#pragma unroll
        for (int s = 0; s < KS4; ++s) {
            const int k0 = 32 * s + 8 * grp;
            float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
            if (k0 < g.kp) {   // kp is a multiple of 16: an 8-chunk is wholly inside or outside the row
                q0 = *reinterpret_cast<const float4 *>(src + k0);
                q1 = *reinterpret_cast<const float4 *>(src + k0 + 4);
            }
            const float e[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            float t[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int k = k0 + c;
                pn = fmaf(e[c], e[c], pn);
                ps += k < g.d ? e[c] : 0.0f;
                pt += k == g.d ? e[c] : 0.0f;
                const float spare = k == g.d + 1 ? 1.0f : (k == g.d + 2 ? 0x1p-11f : 0.0f);
                t[c] = k <= g.d ? fold * e[c] : spare;
            }
            FragH fh, fl;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float h0 = (float)(_Float16)t[2 * c], h1 = (float)(_Float16)t[2 * c + 1];
                fh.u[c] = pack_h2f(t[2 * c], t[2 * c + 1]);
                fl.u[c] = pack_h2f((t[2 * c] - h0) * 2048.0f, (t[2 * c + 1] - h1) * 2048.0f);
            }
            xh[p][s] = fh.h;
            xl[p][s] = fl.h;
        }
        pn += __shfl_xor(pn, 16);
        ps += __shfl_xor(ps, 16);
        pt += __shfl_xor(pt, 16);
        pn += __shfl_xor(pn, 32);
        ps += __shfl_xor(ps, 32);
        pt += __shfl_xor(pt, 32);
        const float nrow = g.a * g.a * pn - g.a * (float)g.d;   // L0 = acc + nx
        const float srow = g.a * ps, trow = g.a * pt;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                               // this lane's accumulator rows are points 4 grp + i
            nx[p][i] = __shfl(nrow, 4 * grp + i);
            sx[p][i] = __shfl(srow, 4 * grp + i);
            tx[p][i] = __shfl(trow, 4 * grp + i);
        }
    }
    float au[PT][4], at[PT][4], ad[PT][4], al[PT][4];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) au[p][i] = at[p][i] = ad[p][i] = al[p][i] = 0.0f;
    PairConsts c;
    c.k1 = -0.5f * 1.44269504088896341f / g.a;    // exp(-a r2 / 2) = exp2(k1 * (a^2 r2 - a d) + k2)
    c.k2 = c.k1 * g.a * (float)g.d;
    asm volatile("" : "+v"(c.k1), "+v"(c.k2));   // VGPR operands: a VALU with an SGPR source issues at half rate

    bool uonly = false;
    if (g.site_u_only && g.rows_per_site >= 16 * PT) {
        const int64_t last = p0 + 16 * PT - 1 < g.n_inf ? p0 + 16 * PT - 1 : g.n_inf - 1;
        const int64_t s0 = p0 < g.n_inf ? p0 / g.rows_per_site : 0, s1 = last / g.rows_per_site;
        uonly = g.site_u_only[s0] == 1 && g.site_u_only[s1] == 1;
    }
    uonly = __builtin_amdgcn_readfirstlane((int)uonly) != 0;

    // LDS views of a slot: B fragments [plane][sub][step][64 lanes] float4, constants [3][32 columns] float4
    auto bfrag = [&](int slot, int pl, int sub, int s) {
        return reinterpret_cast<const float4 *>(lds + slot * STAGE)[((pl * 2 + sub) * KS4 + s) * 64 + lane];
    };
    auto cfrag = [&](int slot, int sub, int qi) {
        return reinterpret_cast<const float4 *>(lds + slot * STAGE + NB * 256)[qi * 32 + sub * 16 + col];
    };
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 accC[PT], accN[PT], accM[PT];   // current subtile (combined), next subtile: h*h and cross terms
    // ablation is a compile-time mask (-DSCASML_GP_ABLATE=1 no MFMA, 2 no epilogue, 8 stage only the first tiles): a
    // runtime test inside the unrolled half step would fence every pair evaluation into its own basic block
    constexpr bool do_mfma = !(SCASML_GP_ABLATE & 1), do_epi = !(SCASML_GP_ABLATE & 2);

    // MFMA number m of a collocation subtile whose B fragments are bh[] (and bl[]): order step-major, then
    // point subtile, then product -- consecutive MFMAs hit different accumulators
    auto mfma_one = [&](int m, const FragH (&bh)[KS4], const FragH (&bl)[KS4]) {
        const int s = m / (PT * MPS), p = (m / MPS) % PT, w = m % MPS;
        if (w == 0) accN[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[p][s], bh[s].h, s == 0 ? zero4 : accN[p], 0, 0, 0);
        if (w == 1) accM[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[p][s], bh[s].h, s == 0 ? zero4 : accM[p], 0, 0, 0);
        if (w == 2) accM[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[p][s], bl[s].h, accM[p], 0, 0, 0);
    };
    auto combine = [&]() {
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) accC[p][i] = fmaf(accM[p][i], 0x1p-11f, accN[p][i]);
    };
    auto load_b = [&](int slot, int sub, FragH (&bh)[KS4], FragH (&bl)[KS4]) {
#pragma unroll
        for (int s = 0; s < KS4; ++s) {
            bh[s].f = bfrag(slot, 0, sub, s);
            if constexpr (!YEXACT) bl[s].f = bfrag(slot, 1, sub, s);
        }
    };

    rendezvous();   // tiles 0 and 1 have landed (tile 2 may still be in flight)
    // Operand registers of the two half steps of a tile iteration.  Half step A (epilogue of subtile 0, products
    // of subtile 1) uses qA / bA, half step B (epilogue of subtile 1, products of the next tile's subtile 0) uses
    // qB / bB; each set is loaded from LDS during the OTHER half step, so no half step starts by waiting on LDS.
    FragH bAh[KS4], bAl[KS4], bBh[KS4], bBl[KS4];
    float4 qA[3], qB[3];
    {   // pipeline fill: the products of collocation subtile 0
        load_b(0, 0, bBh, bBl);
#pragma unroll
        for (int p = 0; p < PT; ++p) accN[p] = accM[p] = zero4;
        if (do_mfma) {
#pragma unroll
            for (int m = 0; m < NM; ++m) mfma_one(m, bBh, bBl);
        }
        combine();
        load_b(0, 1, bAh, bAl);
#pragma unroll
        for (int i = 0; i < 3; ++i) qA[i] = cfrag(0, 0, i);
    }

    // one half step: pair evaluations of the current subtile (constants q) interleaved with the MFMAs of the next
    // subtile (B fragments bh / bl); `prefetch()` issues the LDS reads of the following half step first
    auto half_step = [&](auto uo, const float4 (&q)[3], const FragH (&bh)[KS4], const FragH (&bl)[KS4], auto prefetch) {
        constexpr bool UO = decltype(uo)::value;
        prefetch();
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            auto issue = [&](int gi) {   // MFMAs of issue point gi, then pin the order
                if (do_mfma) {
#pragma unroll
                    for (int m = gi * NM / NG; m < (gi + 1) * NM / NG; ++m) mfma_one(m, bh, bl);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            const int p = e / 4, i = e % 4;
            issue(2 * e);
            if (do_epi) {
                pair_eval<UO>(accC[p][i] + nx[p][i], sx[p][i], tx[p][i], q, c, au[p][i], at[p][i], ad[p][i], al[p][i], [&] { issue(2 * e + 1); });
            } else {
                issue(2 * e + 1);
                au[p][i] += accC[p][i];   // ablation build: keep the products alive
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        combine();
    };
    auto sweep = [&](auto uo) {
        constexpr int NQ = decltype(uo)::value ? 2 : 3;
        for (int jt = 0; jt < n_tiles; ++jt) {
            const int ahead = jt + AHEAD < last_tile ? jt + AHEAD : last_tile;
            if (!once) stage(ahead, (jt + AHEAD) % NSLOT);
            const int sc = jt % NSLOT, sn = (jt + 1) % NSLOT;
            half_step(uo, qA, bAh, bAl, [&] {
#pragma unroll
                for (int i = 0; i < NQ; ++i) qB[i] = cfrag(sc, 1, i);
                load_b(sn, 0, bBh, bBl);   // on the last tile: a stale slot, its products are computed and dropped
            });
            half_step(uo, qB, bBh, bBl, [&] {
#pragma unroll
                for (int i = 0; i < NQ; ++i) qA[i] = cfrag(sn, 0, i);
                load_b(sn, 1, bAh, bAl);
            });
            rendezvous();
        }
    };
    if (uonly) sweep(std::true_type{});
    else sweep(std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation

    // ---- reduce the 16 column partial sums of every point row, lanes col = i store point 4 grp + i ----
    const float s2 = g.sigma * g.sigma;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        float u = 0.0f, dt = 0.0f, dv = 0.0f, lp = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float a0 = au[p][i], a1 = at[p][i], a2 = ad[p][i], a3 = al[p][i];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                a0 += __shfl_xor(a0, m);
                a1 += __shfl_xor(a1, m);
                a2 += __shfl_xor(a2, m);
                a3 += __shfl_xor(a3, m);
            }
            if (col == i) { u = a0; dt = a1; dv = a2; lp = a3; }
        }
        const int64_t row = p0 + 16 * p + 4 * grp + col;
        if (col < 4 && row < g.n_inf) {
            const float eps = dt + (s2 * u - 1.0f / (float)g.d - 0.5f * s2) * dv + 0.5f * s2 * lp;   // models/GP.py:767-768
            g.out4[row] = make_float4(u, dv, eps, dt);
            if (g.lap) g.lap[row] = lp;
        }
    }
}

template <int KS4, int PT, int WPB, int BPC, bool YEXACT>
static int launch_cfg(const GpArgs &g, hipStream_t s) {
    const int64_t waves = (g.n_inf + 16 * PT - 1) / (16 * PT);
    const int64_t blocks = (waves + WPB - 1) / WPB;
    if (blocks > 0x7FFFFFFF) return fail(SCASML_ERR_UNSUPPORTED, "gp_eval: too many points");
    constexpr size_t lds_bytes = (size_t)4 * (2 * (YEXACT ? 1 : 2) * KS4 + 2) * 1024;
    static_assert(lds_bytes * BPC <= 160 * 1024, "LDS slots exceed 160 KiB");
    auto kern = gp_eval_f16_kernel<KS4, PT, WPB, BPC, YEXACT>;
    if (lds_bytes > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return fail(SCASML_ERR_HIP, "gp_eval: cannot reserve %zu bytes of LDS", lds_bytes);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WPB * 64), lds_bytes, s, g);
    return check_launch("gp_eval(f16) launch");
}

template <int KS4, bool YEXACT>
static int launch_ks(const GpArgs &g, hipStream_t s) {
    // two 16-point subtiles per wave, two waves per SIMD as two 4-wave workgroups per CU (development switch
    // SCASML_GP_CFG=2: one 8-wave workgroup)
    static const int cfg = [] { const char *e = getenv("SCASML_GP_CFG"); return e ? atoi(e) : 0; }();
    if constexpr (KS4 <= 4 && YEXACT) {
        if (cfg == 2) return launch_cfg<KS4, 2, 8, 1, YEXACT>(g, s);
        return launch_cfg<KS4, 2, 4, 2, YEXACT>(g, s);
    } else {
        // long rows, or a second collocation plane: one point subtile per wave keeps everything in 256 VGPRs
        return launch_cfg<KS4, 1, 8, 1, YEXACT>(g, s);
    }
}

// float32 collocation points (second plane) with very long rows do not fit 256 VGPRs: the 32x32 kernel takes those
bool gp_eval_f16_supports(const GpArgs &g) { return g.colloc_is_f16 || (g.kp + 31) / 32 <= 5; }

int launch_gp_eval_f16(const GpArgs &g, hipStream_t s) {
    const int ks4 = (g.kp + 31) / 32;
#define SCASML_CASE(K) \
    case K: return g.colloc_is_f16 ? launch_ks<K, true>(g, s) : launch_ks<K, false>(g, s);
    switch (ks4) {
        SCASML_CASE(1) SCASML_CASE(2) SCASML_CASE(3) SCASML_CASE(4) SCASML_CASE(5) SCASML_CASE(6) SCASML_CASE(7) SCASML_CASE(8)
    }
#undef SCASML_CASE
    return fail(SCASML_ERR_UNSUPPORTED, "gp_eval: kp=%d", g.kp);
}

}  // namespace scasml
