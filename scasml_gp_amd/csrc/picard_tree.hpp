// Picard-tree walker and kernel template, shared by the translation units that instantiate it (picard_tree.hip: the Philox stream,
// picard_tree_jax.hip / picard_tree_jax_deep.hip: the reference's own stream, SCASML_RNG_JAX_STREAM) -- split so that the build compiles
// the instantiations in parallel (the JAX-stream kernels inline one Threefry + erf_inv per normal at every site of the unrolled tree).
#pragma once
#include <string.h>
#include "common.hpp"
#include "equations.hpp"
#include "philox_normal.hpp"

namespace scasml {

struct TreeArgs {
    scasml_plan plan;
    const float *x_t;
    float *points;
    const float4 *gpv;
    float *out_uz;
    float *out_uhat;
    int64_t B;
    int64_t Bs;   // site stride: row of (site, root) = site * Bs + root (>= B)
    int64_t ppr;  // points per root = plan.sites[n] + 1
    uint32_t k0, k1, stream, root0;
    int32_t rank, world;
    const uint8_t *owner;   // unit -> rank (scasml_plan_deal_units), or null: unit % world
    int32_t crn;  // SCASML_RNG_COMPAT_CRN: terminal draws keyed by the call's k = 0 position (reference key reuse, E-2/E-3)
    int32_t f16;  // SCASML_RNG_COMPAT_F16: the reference's solver-level float16 casts (g, f and every uz_solve return; E-5)
    const uint32_t *jk;   // SCASML_RNG_JAX_STREAM: key words [terminal k0 k1 | path sub-key 0 k0 k1 | sub-key 1 ...] (scasml_rng.jax_keys)
    int32_t d, G, logG, kp;
    float T, mu, sigma, clip;
};

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 b) {  // s*a + b
    return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}
__device__ __forceinline__ float4 add4(float4 a, float s) { return make_float4(a.x + s, a.y + s, a.z + s, a.w + s); }
__device__ __forceinline__ float4 f4_scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float clip1(float v, float c) { return v < -c ? -c : (v > c ? c : v); }  // keeps NaN (jnp.clip)
__device__ __forceinline__ float r16(float v) { return (float)(_Float16)v; }                          // .astype(jnp.float16), RNE
// Hardware reciprocal / square root / exp2 (<= 1 ulp) for the arithmetic that is NOT part of the bit-exact
// RNG specification: the IEEE-correct expansions cost ~10 VALU instructions each and these kernels are
// VALU-bound; the parity tolerance (1e-4 relative) is five orders of magnitude above the difference.
__device__ __forceinline__ float rcp_fast(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float sqrt_fast(float v) { return __builtin_amdgcn_sqrtf(v); }
__device__ __forceinline__ float exp_fast(float v) { return __builtin_amdgcn_exp2f(v * 1.44269504088896341f); }

// ACCUMULATE recovers the terminal normals as (X_T - x - drift) / vol.  X_T was rounded to binary32 when it was stored
// (|X_T| 2^-24 ~ 3e-8), so the recovered normal is off by ~3e-8 / vol and the z estimator, which divides by T - t, by
// ~g 1.2e-7 / (sqrt(T-t) (T-t) sqrt(mg)): 1e-4 g at T - t = 1.6e-3 but 0.03 g one fp16 ulp below T.  Below this
// volatility (sigma sqrt(T-t) < 1e-2, i.e. T - t < 1.6e-3 at sigma = 0.25) the normals are replayed instead.
constexpr float kReadbackMinVol = 1e-2f;
// Terminal samples whose rows ACCUMULATE requests ahead of the one it consumes.  Measured at the headline shape (same box,
// profiles/r02_accumulate_prefetch.txt): 1 -> 1.29 ms, 2 -> 1.30, 3 -> 1.65, 5 -> 1.66: beyond one the extra registers and
// moves cost more than the bytes in flight buy.
#ifndef SCASML_ACC_AHEAD
#define SCASML_ACC_AHEAD 1
#endif
constexpr int kAhead = SCASML_ACC_AHEAD;

// ---- SCASML_RNG_JAX_STREAM: the reference's own normals, jax.random.normal(key, shape, float16) under jax_threefry_partitionable,
// addressed by counter (oracle/jax_random.py is the NumPy statement; tests/test_gpu_jax_stream.py compares the two bit for bit):
// element with row-major index i of a draw under key (k0, k1) = low 16 bits of y0 ^ y1, (y0, y1) = Threefry-2x32-20(key, (i >> 32, i));
// bits >> 6 | 0x3C00 is a float16 in [1, 2); minus 1, times 2, plus nextafter(-1, 0), clamped below (each a float16 operation);
// sqrt(2) * erf_inv in float32 (XLA's ErfInv32) rounded to float16 before the float16 product.  One Threefry per normal: this is the
// parity mode, ~7x the integer work of the Philox stream.  The transform itself depends on ten bits only and is tabulated per workgroup.
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ uint32_t threefry_bits16(uint32_t k0, uint32_t k1, uint64_t index) {
    const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
    uint32_t x0 = (uint32_t)(index >> 32) + ks[0], x1 = (uint32_t)index + ks[1];
    const int rot[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x0 += x1;
            x1 = rotl32(x1, rot[i & 1][j]) ^ x0;
        }
        x0 += ks[(i + 1) % 3];
        x1 += ks[(i + 2) % 3] + (uint32_t)(i + 1);
    }
    return (x0 ^ x1) & 0xFFFFu;
}
// the float16 transform of the draw's TEN live bits (the float16 uniform has a 10-bit mantissa): a function of 1024 inputs, which the tree kernels
// tabulate once per workgroup (jax_table_to_lds) instead of evaluating log1p, a square root and a polynomial per normal
__device__ __forceinline__ float jax_normal_of_bits10(uint32_t b10) {
    const unsigned short hb = (unsigned short)(b10 | 0x3C00u);
    const _Float16 lo = (_Float16)-0.99951171875f;                      // nextafter(float16(-1), 0)
    _Float16 u = __builtin_bit_cast(_Float16, hb) - (_Float16)1.0f;
    u = u * (_Float16)2.0f + lo;                                        // (1 - lo) rounds to 2 in float16; both operations are exact or rounded once
    u = u < lo ? lo : u;
    const float x = (float)u;
    float w = -log1pf(-x * x);
    float p;
    if (w < 5.0f) {
        w -= 2.5f;
        p = 2.81022636e-08f;
        p = fmaf(p, w, 3.43273939e-07f);
        p = fmaf(p, w, -3.5233877e-06f);
        p = fmaf(p, w, -4.39150654e-06f);
        p = fmaf(p, w, 0.00021858087f);
        p = fmaf(p, w, -0.00125372503f);
        p = fmaf(p, w, -0.00417768164f);
        p = fmaf(p, w, 0.246640727f);
        p = fmaf(p, w, 1.50140941f);
    } else {
        w = sqrtf(w) - 3.0f;
        p = -0.000200214257f;
        p = fmaf(p, w, 0.000100950558f);
        p = fmaf(p, w, 0.00134934322f);
        p = fmaf(p, w, -0.00367342844f);
        p = fmaf(p, w, 0.00573950773f);
        p = fmaf(p, w, -0.0076224613f);
        p = fmaf(p, w, 0.00943887047f);
        p = fmaf(p, w, 1.00167406f);
        p = fmaf(p, w, 2.83297682f);
    }
    const _Float16 e = (_Float16)(p * x);
    return (float)((_Float16)1.4140625f * e);                           // float16(sqrt(2)) * float16(erf_inv): a float16 product
}
constexpr int kJaxTableRows = 1024;
__device__ __forceinline__ float *jax_table_lds() {
    __shared__ float t[kJaxTableRows];
    return t;
}
// every thread of the workgroup, before any return: the 1024 possible normals of jax.random.normal(float16), 4 KB of LDS
__device__ __forceinline__ void jax_table_to_lds() {
    float *t = jax_table_lds();
    for (int i = threadIdx.x; i < kJaxTableRows; i += blockDim.x) t[i] = jax_normal_of_bits10((uint32_t)i);
    __syncthreads();
}
__device__ __forceinline__ float jax_normal_f16(uint32_t k0, uint32_t k1, uint64_t index) {
    return jax_table_lds()[threefry_bits16(k0, k1, index) >> 6];
}
// jax.random.uniform(key, shape, float16): one of the 1024 values k / 1024 (0 included)
__device__ __forceinline__ float jax_uniform_f16(uint32_t k0, uint32_t k1, uint64_t index) {
    const unsigned short hb = (unsigned short)((threefry_bits16(k0, k1, index) >> 6) | 0x3C00u);
    return (float)(__builtin_bit_cast(_Float16, hb) - (_Float16)1.0f);
}

// ACCUMULATE's read-ahead of terminal rows (kAhead samples); empty in the other modes
template <bool ON>
struct PrefetchQueue {
    float4 XT[kAhead], gp[kAhead];
};
template <>
struct PrefetchQueue<false> {};

template <int VAR, int MODE, int EQ, bool JAX = false>
struct Walker {
    const TreeArgs &a;
    float4 mask;       // 1 on this lane's live spatial dims, 0 on padding
    float4 tmask;      // 1 on the component that holds t in a stored row (column d)
    uint32_t row_off4; // (local * kp + 4 * gl) / 4: this lane's float4 inside a site's block of B rows
    uint32_t gp_off;   // local
    bool row_lane;     // 4 * gl < kp
    uint32_t gl;       // lane index inside the root's group = Philox quad index
    uint32_t root;     // global root index (Philox counter word 2)
    int64_t local;     // this root's index inside the call's batch: row of site s = s * B + local
    int unit;          // running unit index of the ROOT call (sample sharding)

    __device__ __forceinline__ float group_sum(float v) const {
        for (int o = a.G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
    }
    __device__ __forceinline__ float dim_sum(float4 v) const {
        return group_sum(fmaf(mask.x, v.x, fmaf(mask.y, v.y, fmaf(mask.z, v.z, mask.w * v.w))));
    }
    __device__ __forceinline__ float4 normals(uint32_t site) const {
        return mul4(normal4(gl, site, root, a.stream, a.k0, a.k1), mask);
    }
    // sample m of a (batch, width, d) draw under key (k0, k1), for this root's row of the reference's flattened batch
    __device__ __forceinline__ float4 normals_jax(uint32_t kslot, uint64_t row, uint32_t width, uint32_t m) const {
        const uint32_t k0 = a.jk[2 * kslot], k1 = a.jk[2 * kslot + 1];
        const uint64_t first = (row * width + m) * (uint64_t)a.d + 4u * gl;
        float4 v;
        v.x = mask.x != 0.0f ? jax_normal_f16(k0, k1, first + 0) : 0.0f;
        v.y = mask.y != 0.0f ? jax_normal_f16(k0, k1, first + 1) : 0.0f;
        v.z = mask.z != 0.0f ? jax_normal_f16(k0, k1, first + 2) : 0.0f;
        v.w = mask.w != 0.0f ? jax_normal_f16(k0, k1, first + 3) : 0.0f;
        return v;
    }
    // path sub-keys one uz_solve call at level N consumes, its children's included (MLP.py:213-220, 231, 253)
    template <int N>
    __device__ __forceinline__ uint32_t nsplits() const {
        if constexpr (N <= 0) {
            return 0u;
        } else {
            return nsplits_from<N, 0>();
        }
    }
    template <int N, int L>
    __device__ __forceinline__ uint32_t nsplits_from() const {
        if constexpr (L >= N) {
            return 0u;
        } else {
            const uint32_t per = 1u + nsplits<L>() + nsplits<L - 1>();
            return (uint32_t)a.plan.term[N][L].q * per + nsplits_from<N, L + 1>();
        }
    }
    // Row addressing: row = site * B + local, so a site's rows start at a wave-uniform base (scalar 64-bit arithmetic) and
    // the lane contributes a fixed 32-bit element offset -- a 64-bit VGPR product per access costs three v_mad_u64_u32.
    __device__ __forceinline__ void emit_point(float4 X, float t, uint32_t site) const {
        if (!row_lane) return;                                    // lanes past the padded row
        float4 *base = reinterpret_cast<float4 *>(a.points + (int64_t)site * a.Bs * a.kp);
        base[row_off4] = fma4(t, tmask, mul4(X, mask));           // (X, t, zero pad)
    }
    __device__ __forceinline__ float4 gp_at(uint32_t site) const { return (a.gpv + (int64_t)site * a.Bs)[gp_off]; }
    __device__ __forceinline__ float4 load_point(uint32_t site) const {   // this lane's four dims of a stored tree point
        const float4 *base = reinterpret_cast<const float4 *>(a.points + (int64_t)site * a.Bs * a.kp);
        return base[row_off4];
    }
    __device__ __forceinline__ bool mine_at(int un) const {   // wave-uniform: scalar load
        return a.owner ? (int)a.owner[un] == a.rank : (un % a.world) == a.rank;
    }
    __device__ __forceinline__ bool owned(bool top) {
        if (!top || a.world == 1) return true;
        const bool mine = mine_at(unit);
        ++unit;
        return mine;
    }

    // Equation.g at time T (equations/equations.py:146-162, 248-261) through the registry; ScaSML.py:61-63 subtracts the surrogate
    __device__ __forceinline__ float g_terminal(float4 XT, float u_hat) const {
        float g = EqDef<EQ>::G(dim_sum(EqDef<EQ>::phi(XT)), a.T);
        if (a.f16) g = r16(g);                                    // terminal_constraint(...).astype(float16), equations.py:261
        if constexpr (MODE == SCASML_MODE_ACCUMULATE) {
            g -= u_hat;
            if (a.f16) g = r16(g);                                // float16 - float16 (ScaSML.py:62): a float16 value
        }
        return g;
    }
    // Equation.f (equations/equations.py:290-304, MLP.py:27-41); ScaSML.py:29-47: f(u_hat + u, sigma grad u_hat + z) - f(u_hat, sigma grad u_hat),
    // where f sees the gradient through its sum only, sum_i sigma d_i u_hat = sigma div u_hat
    __device__ __forceinline__ float f_eval(float uc, float4 zc, float4 gp) const {
        const float sz = dim_sum(zc);
        const float fd = (float)a.d;
        if constexpr (EqDef<EQ>::kGradSquare) {                   // f(u, sum z, |z|^2): one more sum over the dims (surrogate-free modes only)
            static_assert(MODE != SCASML_MODE_ACCUMULATE, "an f of |z|^2 needs the surrogate's full gradient per site: not in the fused evaluation");
            const float v = EqDef<EQ>::f2(uc, sz, dim_sum(mul4(zc, zc)), a.sigma, fd);
            return a.f16 ? r16(v) : v;
        } else if constexpr (MODE == SCASML_MODE_ACCUMULATE) {
            const float sg = a.sigma * gp.y;
            const float v1 = EqDef<EQ>::f(uc + gp.x, sg + sz, a.sigma, fd), v2 = EqDef<EQ>::f(gp.x, sg, a.sigma, fd);
            return a.f16 ? r16(r16(v1) - r16(v2)) : v1 - v2;      // generator(...).astype(float16) twice, then float16 - float16 (equations.py:304, ScaSML.py:45-47)
        } else {
            const float v = EqDef<EQ>::f(uc, sz, a.sigma, fd);
            return a.f16 ? r16(v) : v;
        }
    }

    // ---- one (n', l) term of the Picard sum ------------------------------------------------
    // jrow: this root's row in the reference's flattened batch of THIS call; jfirst: sub-key index of node (L, k = 0) of this call
    template <int N, int L, bool TOP>
    __device__ __forceinline__ void level(float4 x, float t, float tau, uint32_t base, uint32_t cbase, uint32_t &o, float &u, float4 &z,
                                          uint64_t jrow = 0, uint32_t jfirst = 0) {
        const scasml_term &tm = a.plan.term[N][L];
        const int q = tm.q, mc = tm.mc;
        const uint32_t s_l = (uint32_t)tm.sites_l, s_lm = (uint32_t)tm.sites_lm1;
        const float inv_mc = rcp_fast((float)mc);
        for (int m = 0; m < mc; ++m) {
            float4 X = x, W = f4(0.0f);
            // compat_crn: the children of every node k draw their terminal normals where the k = 0 children do
            // (MLP.py:167-168,178: one fixed key per uz_solve call, so calls of equal shape share their draws)
            const uint32_t c_plus = cbase + o + 1u, c_minus = c_plus + s_l;
            // sample sharding, quadrature paths: the draws of nodes other ranks own are replayed only up to the LAST node of this path that
            // this rank owns an addend of (every rank used to walk every path to its end: 35 steps per root at n = rho = 3, 5 % of a whole
            // step's draws on each of 8 ranks)
            int klast = q - 1;
            if constexpr (TOP && VAR == 0 && MODE != SCASML_MODE_ACCUMULATE) {
                if (a.world > 1) {
                    constexpr int per = L > 0 ? 2 : 1;
                    klast = -1;
                    for (int k = 0; k < q; ++k)
                        if (mine_at(unit + k * per) || (L > 0 && mine_at(unit + k * per + 1))) klast = k;
                }
            }
            for (int k = 0; k < q; ++k) {
                if constexpr (TOP && VAR == 0 && MODE != SCASML_MODE_ACCUMULATE) {
                    if (k > klast) {                             // nothing of this rank's further along the path
                        o += (uint32_t)(q - k) * (1u + s_l + s_lm);
                        unit += (q - k) * (L > 0 ? 2 : 1);
                        break;
                    }
                }
                const uint32_t site = base + o;
                o += 1;
                // Monte-Carlo sample sharding: the units dealt to ranks are the two ADDENDS a node (l, m, k) of the root call contributes to the root's
                // sums -- "+": w_k f(P_k, uz(l)) with the level-l subtree below the node (and, at l = 0, the surrogate's residual term), "-": the
                // level-(l-1) subtree's term (l > 0) -- not whole sample paths: the node's state is read back (ACCUMULATE), drawn directly (full
                // history) or replayed from the path's own cheap draws (below), and its surrogate values are evaluated by whoever owns either
                // addend.  At n = rho = 3 the largest unit shrinks from 264 sites to 58 and the dealt-load imbalance over 8 ranks from 3.19 to 1.00
                const bool mine = owned(TOP);                    // the "+" addend
                bool mine_minus = false;
                if constexpr (L > 0) mine_minus = owned(TOP);
                if constexpr (VAR == 1 || MODE == SCASML_MODE_ACCUMULATE) {
                    if (!mine && !mine_minus) {                  // nothing incremental in these forms
                        o += s_l + s_lm;
                        continue;
                    }
                }
                float tk, wk;
                float4 wvec;  // the vector multiplying y in the z estimator
                float dplus, dminus;
                if constexpr (VAR == 0 && MODE == SCASML_MODE_ACCUMULATE) {
                    // The pass that emitted the points already produced X_k, bit for bit; read it back instead
                    // of replaying Philox and the normal transform, and recover W_k = (X_k - x - mu (t_k - t)) / sigma (one
                    // rounding of a difference of O(1) numbers divided by sigma: ~1e-6 relative).
                    const float ck = tau * tm.cfrac[k];
                    X = load_point(site);
                    W = mul4(fma4(1.0f / a.sigma, add4(X, -a.mu * ck), f4_scale(x, -1.0f / a.sigma)), mask);
                    tk = t + ck;
                    wk = tau * tm.wfrac[k];
                    wvec = W;
                    dplus = rcp_fast(fmaf(tau, tm.dplus[k], 1e-6f));
                    dminus = rcp_fast(fmaf(tau, tm.cfrac[k], 1e-6f));
                } else if constexpr (VAR == 0) {                 // MLP.py:219-225
                    float4 xi;
                    if constexpr (JAX) xi = normals_jax(1u + jfirst + (uint32_t)k * (1u + nsplits<L>() + nsplits<L - 1>()), jrow, (uint32_t)mc, (uint32_t)m);
                    else xi = normals(site);
                    const float dk = tau * tm.dfrac[k];
                    const float sdk = sqrt_fast(dk);
                    W = fma4(sdk, xi, W);
                    X = fma4(a.sigma * sdk, xi, add4(X, a.mu * dk));
                    tk = fmaf(tau, tm.cfrac[k], t);
                    wk = tau * tm.wfrac[k];
                    wvec = W;
                    dplus = rcp_fast(fmaf(tau, tm.dplus[k], 1e-6f));   // MLP.py:249 (stale) / ScaSML.py:253
                    dminus = rcp_fast(fmaf(tau, tm.cfrac[k], 1e-6f));  // MLP.py:270
                } else {                                         // MLP_full_history.py:133-145
                    // JAX stream: the time and the normals of sample m of this call, and its terminal draws, all come from the ONE key
                    // split(PRNGKey(0), 1)[0] (MLP_full_history.py:92-93, 99, 133, 138), each at its own row-major index
                    float D;
                    if constexpr (JAX) D = jax_uniform_f16(a.jk[0], a.jk[1], jrow * (uint32_t)mc + (uint32_t)m) * tau;
                    else D = uniform_tau(site, root, a.stream, a.k0, a.k1) * tau;
                    const float sD = sqrt_fast(D);
                    // compat_crn: the level-0 draws ARE the terminal draws (MLP_full_history.py:92-93,99,138: one subkey).
                    // (ACCUMULATE replays these normals: reading the stored X back instead, as the terminal samples do, was
                    // measured slower -- 5.9 against 5.4 ms at n = 4, M = 3 -- the pass is bound by its reads, not its RNG.)
                    float4 xi;
                    if constexpr (JAX) xi = normals_jax(0u, jrow, (uint32_t)mc, (uint32_t)m);
                    else xi = normals(a.crn && L == 0 ? base + (uint32_t)m : site);
                    X = fma4(a.sigma * sD, xi, add4(x, a.mu * D));
                    tk = t + D;
                    wk = tau;
                    wvec = xi;
                    dplus = dminus = __builtin_amdgcn_rsqf(D + 1e-6f);     // :158-159
                }
                if constexpr (VAR == 0 && MODE != SCASML_MODE_ACCUMULATE) {
                    if (!mine && !mine_minus) {                  // the path has advanced (X, W); this node's terms belong to other ranks
                        o += s_l + s_lm;
                        continue;
                    }
                }
                if constexpr (MODE == SCASML_MODE_GENERATE) emit_point(X, tk, site);
                float4 gp = f4(0.0f);
                if constexpr (MODE == SCASML_MODE_ACCUMULATE) gp = gp_at(site);

                float uc;
                float4 zc;
                const uint32_t jkid = jfirst + (uint32_t)k * (1u + nsplits<L>() + nsplits<L - 1>()) + 1u;   // the children's first sub-key
                const uint64_t jkrow = jrow * (uint32_t)mc + (uint32_t)m;
                if (mine) {
                    uz<L, false>(X, tk, base + o, (a.crn && VAR == 0) ? c_plus : base + o, uc, zc, jkrow, jkid);
                    if constexpr (MODE != SCASML_MODE_GENERATE) {
                        const float y = f_eval(uc, zc, gp) * (wk * inv_mc);
                        u += y;                                  // MLP.py:248
                        z = fma4(y * dplus, wvec, z);            // MLP.py:249
                    }
                }
                o += s_l;
                if constexpr (L > 0) {
                    if (mine_minus) {
                        uz<L - 1, false>(X, tk, base + o, (a.crn && VAR == 0) ? c_minus : base + o, uc, zc, jkrow, jkid + nsplits<L>());
                        if constexpr (MODE != SCASML_MODE_GENERATE) {
                            const float y = f_eval(uc, zc, gp) * (wk * inv_mc);
                            u -= y;                              // MLP.py:269
                            z = fma4(-y * dminus, wvec, z);      // MLP.py:271
                        }
                    }
                    o += s_lm;
                } else if constexpr (MODE == SCASML_MODE_ACCUMULATE) {
                    const float e = gp.z * (wk * inv_mc);        // ScaSML.py:274-280 (l = 0: one addend, mine)
                    u += e;
                    z = fma4(e * dminus, wvec, z);
                }
            }
        }
        if constexpr (L + 1 < N) level<N, L + 1, TOP>(x, t, tau, base, cbase, o, u, z, jrow, jfirst + (uint32_t)q * (1u + nsplits<L>() + nsplits<L - 1>()));
    }

    // ---- uz_solve at compile-time level N -----------------------------------------------------
    template <int N, bool TOP>
    __device__ __forceinline__ void uz(float4 x, float t, uint32_t base, uint32_t cbase, float &u_out, float4 &z_out, uint64_t jrow = 0,
                                       uint32_t jsplit = 0) {
        if constexpr (N == 0) {                                  // MLP.py:205-207
            u_out = 0.0f;
            z_out = f4(0.0f);
        } else {
            // a child's time t + U (T - t) can round up to (or one ulp past) T in binary32: its horizon is then 0, not negative
            // (sqrt of a negative horizon would poison the whole root with NaN: seen once per ~1e7 full-history draws)
            const float tau = fmaxf(a.T - t, 0.0f);
            const int mg = a.plan.mg[N];
            const float drift = a.mu * tau, vol = a.sigma * sqrt_fast(tau);
            float su = 0.0f;
            float4 sz = f4(0.0f);
            // ACCUMULATE reads the stored X_T back: the rows of the next kAhead samples are requested before this one is
            // consumed (one dependent HBM round trip per sample otherwise; the pass is bound by the bytes it keeps in flight)
            const bool readback = MODE == SCASML_MODE_ACCUMULATE && vol >= kReadbackMinVol;
            // (the queue exists in ACCUMULATE only.  The 36 bytes of scratch rocprof reports for the level-3 / level-4 MLP and GENERATE kernels are
            // not these arrays but the stack slot of six spilled SGPRs -- v_writelane / v_readlane, no scratch instruction: DESIGN.md 4.1)
            PrefetchQueue<MODE == SCASML_MODE_ACCUMULATE> pq;
            if constexpr (MODE == SCASML_MODE_ACCUMULATE) {
#pragma unroll
                for (int p = 0; p < kAhead; ++p) {
                    pq.XT[p] = f4(0.0f);
                    pq.gp[p] = f4(0.0f);
                    if (!(TOP && a.world > 1) && p < mg) {
                        if (readback) pq.XT[p] = load_point(base + (uint32_t)p);
                        pq.gp[p] = gp_at(base + (uint32_t)p);
                    }
                }
            }
            for (int m = 0; m < mg; ++m) {                       // MLP.py:175-202
                const uint32_t site = base + (uint32_t)m;
                float4 nrm, XT, gpv = f4(0.0f);
                if constexpr (MODE == SCASML_MODE_ACCUMULATE) {
                    if (!(TOP && a.world > 1)) {                 // un-sharded: software-pipelined reads
                        XT = pq.XT[0];
                        gpv = pq.gp[0];
#pragma unroll
                        for (int p = 0; p + 1 < kAhead; ++p) {
                            pq.XT[p] = pq.XT[p + 1];
                            pq.gp[p] = pq.gp[p + 1];
                        }
                        if (m + kAhead < mg) {
                            if (readback) pq.XT[kAhead - 1] = load_point(site + kAhead);
                            pq.gp[kAhead - 1] = gp_at(site + kAhead);
                        }
                    } else {
                        if (!owned(TOP)) continue;
                        if (readback) XT = load_point(site);
                        gpv = gp_at(site);
                    }
                    // The emitting pass stored X_T bit for bit: recover the normals (one rounding of a difference
                    // of O(1) numbers: ~1e-6 relative) instead of replaying Philox and the normal transform, which would be most
                    // of this pass's VALU work.  Close to T they cannot be recovered accurately (kReadbackMinVol): replay.
                    if (__builtin_expect(readback, 1)) {
                        const float rv = rcp_fast(vol);
                        nrm = mul4(fma4(rv, add4(XT, -drift), f4_scale(x, -rv)), mask);
                    } else {
                        if constexpr (JAX) nrm = normals_jax(0u, jrow, (uint32_t)mg, (uint32_t)m);
                        else nrm = normals(cbase + (uint32_t)m);
                        XT = fma4(vol, nrm, add4(x, drift));
                    }
                } else {
                    if (!owned(TOP)) continue;
                    if constexpr (JAX) nrm = normals_jax(0u, jrow, (uint32_t)mg, (uint32_t)m);   // every call: split(PRNGKey(0), 1)[0] (MLP.py:167-168, 178)
                    else nrm = normals(cbase + (uint32_t)m);
                    XT = fma4(vol, nrm, add4(x, drift));
                }
                if constexpr (MODE == SCASML_MODE_GENERATE) {
                    emit_point(XT, a.T, site);
                } else {
                    const float g = g_terminal(XT, gpv.x);
                    su += g;
                    sz = fma4(g, nrm, sz);
                }
            }
            const float inv_mg = rcp_fast((float)mg);
            float u = su * inv_mg;
            const float zs = inv_mg * rcp_fast(VAR == 0 ? tau + 1e-6f : tau);   // MLP.py:201 / MLP_full_history.py:122
            // padding dims carry sz = 0: at T - t = 0 the full-history scale is 1/0 (MLP_full_history.py:122 has no epsilon) and
            // 0 * inf would put a NaN into the padding that dim_sum's 0 * NaN then spreads to the whole root
            float4 z = make_float4(mask.x != 0.0f ? sz.x * zs : 0.0f, mask.y != 0.0f ? sz.y * zs : 0.0f,
                                   mask.z != 0.0f ? sz.z * zs : 0.0f, mask.w != 0.0f ? sz.w * zs : 0.0f);
            uint32_t o = (uint32_t)mg;
            level<N, 0, TOP>(x, t, tau, base, cbase, o, u, z, jrow, jsplit);
            if (!(TOP && a.world > 1)) {                         // MLP.py:272-274
                u = clip1(u, a.clip);
                z = make_float4(clip1(z.x, a.clip), clip1(z.y, a.clip), clip1(z.z, a.clip), clip1(z.w, a.clip));
                // jnp.clip(...).astype(jnp.float16): MLP.py:274, ScaSML.py:284, MLP_full_history.py:180 -- ScaSML_full_history.py:199 does not cast
                if (a.f16 && !(VAR == 1 && MODE != SCASML_MODE_MLP)) {
                    u = r16(u);
                    z = make_float4(r16(z.x), r16(z.y), r16(z.z), r16(z.w));
                }
            }
            u_out = u;
            z_out = z;
        }
    }
};

// (An occupancy hint for ACCUMULATE was measured, profiles/r02_accumulate_prefetch.txt: 5 waves/SIMD 1.22 ms against 1.26, but the
// deeper levels then spill inside their loops; 6 and 8 are slower.  No hint.)
template <int VAR, int MODE, int N, int EQ, bool JAX = false>
__global__ __launch_bounds__(256) void picard_tree_kernel(const TreeArgs a) {
    if constexpr (JAX) jax_table_to_lds();   // every thread, before any return below: the reference's stream needs its own 4 KB table only
    else normal_table_to_lds();
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
    int64_t local;
    uint32_t gl;
    if constexpr (MODE == SCASML_MODE_GENERATE) {
        // GENERATE takes no sum over dims: lanes need not form power-of-two groups.  Flat (root, quad) mapping over the
        // kp / 4 float4 of a row, so no lane idles through the Philox and the normal transform work (at d = 100: 28 lanes per root
        // instead of 32, 25 of them drawing normals)
        const int64_t flat = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const int quads = a.kp >> 2;
        local = flat / quads;
        gl = (uint32_t)(flat - local * quads);
    } else {
        const int rpw = 64 >> a.logG;
        local = (int64_t)wave * rpw + (lane >> a.logG);
        gl = (uint32_t)(lane & (a.G - 1));
    }
    const bool valid = local < a.B;
    if (!valid) local = a.B - 1;  // idle lanes shadow the last root; their stores are masked

    Walker<VAR, MODE, EQ, JAX> w{a};
    w.gl = gl;
    w.root = a.root0 + (uint32_t)local;
    w.local = local;
    w.unit = 0;
    const int dim0 = 4 * (int)w.gl;
    w.mask = make_float4(dim0 + 0 < a.d ? 1.0f : 0.0f, dim0 + 1 < a.d ? 1.0f : 0.0f,
                         dim0 + 2 < a.d ? 1.0f : 0.0f, dim0 + 3 < a.d ? 1.0f : 0.0f);
    w.tmask = make_float4(dim0 + 0 == a.d ? 1.0f : 0.0f, dim0 + 1 == a.d ? 1.0f : 0.0f,
                          dim0 + 2 == a.d ? 1.0f : 0.0f, dim0 + 3 == a.d ? 1.0f : 0.0f);
    w.row_lane = dim0 < a.kp && (MODE != SCASML_MODE_GENERATE || valid);
    w.row_off4 = (uint32_t)((local * a.kp + (dim0 < a.kp ? dim0 : 0)) >> 2);   // a chunk's point buffer is < 2^32 floats per site block
    w.gp_off = (uint32_t)local;
    const float *row = a.x_t + local * (a.d + 1);
    float4 x;
    x.x = dim0 + 0 < a.d ? row[dim0 + 0] : 0.0f;
    x.y = dim0 + 1 < a.d ? row[dim0 + 1] : 0.0f;
    x.z = dim0 + 2 < a.d ? row[dim0 + 2] : 0.0f;
    x.w = dim0 + 3 < a.d ? row[dim0 + 3] : 0.0f;
    const float t = row[a.d];

    if constexpr (MODE == SCASML_MODE_GENERATE) {
        if (valid) w.emit_point(x, t, (uint32_t)(a.ppr - 1));   // the root itself, for ScaSML.py:303
    }
    float u;
    float4 z;
    w.template uz<N, true>(x, t, 0u, 0u, u, z, (uint64_t)a.root0 + (uint64_t)local, 0u);
    if constexpr (MODE != SCASML_MODE_GENERATE) {
        if (valid) {
            float *out = a.out_uz + local * (a.d + 1);
            if (w.gl == 0) {
                out[0] = u;
                if constexpr (MODE == SCASML_MODE_ACCUMULATE) {
                    if (a.out_uhat) a.out_uhat[local] = w.gp_at((uint32_t)(a.ppr - 1)).x;
                }
            }
            if (dim0 + 0 < a.d) out[1 + dim0 + 0] = z.x;
            if (dim0 + 1 < a.d) out[1 + dim0 + 1] = z.y;
            if (dim0 + 2 < a.d) out[1 + dim0 + 2] = z.z;
            if (dim0 + 3 < a.d) out[1 + dim0 + 3] = z.w;
        }
    }
}

// the reference's own random stream (compat_rng = "jax"): instantiated in picard_tree_jax.hip (n <= 3) and picard_tree_jax_deep.hip (n = 4, 5)
int launch_tree_jax(const TreeArgs &a, int variant, int mode, int eq_id, int n, dim3 grid, hipStream_t s);
int launch_tree_jax_deep(const TreeArgs &a, int variant, int mode, int eq_id, int n, dim3 grid, hipStream_t s);

}  // namespace scasml
