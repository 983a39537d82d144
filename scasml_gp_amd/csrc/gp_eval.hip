// Fused posterior evaluation of the PDE-constrained GP surrogate (inference side of models/GP.py).
//
// Replaces GP.predict (models/GP.py:653-671), GP.compute_gradient (:673-687; its spatial sum is
// produced here, the full gradient by gp_gradient_kernel) and
// GP_Grad_Dependent_Nonlinear.compute_PDE_loss (:746-769), which in the reference build three
// (n_inf x M) feature matrices by nested autodiff.  Here every feature is P(rho^2, S, r_t)*kappa
// (SURVEY.md Appendix C), so one pass over the N = N_dom + N_bdy collocation points suffices:
//
//   x.y          FP32 MFMA  v_mfma_f32_32x32x2_f32  (A = 32 collocation rows, B = 32 points)
//   epilogue     per (collocation i, point j):  r2 = |x|^2 + |y|^2 - 2 x.y,  kappa = exp(-a r2/2),
//                L = a^2 rho^2 - a d,  p = a r_t,  s = a S,  E = c0 + cL L + ct p + cS s, and
//                u   += kappa E
//                dt  += kappa (a ct - p E)
//                div += kappa (a (2 cL s + d cS) - s E)
//                lap += kappa (L E - 2a (cL (2L + a d) + cS s))
//   (these are the I / dt / div / lap rows of Appendix C contracted with right_vector; c0..cS are
//   the right_vector entries of the u, Lap, dt, div features of collocation point i).
//
// Layout: points are the MFMA N dimension (column = lane & 31), so the four running sums of a
// point live in the lane that owns its column and the Monte-Carlo-free reduction over collocation
// points is a plain register accumulation; only the two half-waves are combined at the end.
// The K loop is split between the half-waves (half h takes k in [h*kp/2, (h+1)*kp/2)), which makes
// every lane's operand a contiguous run of its row: 16-byte loads, no LDS staging; the point tile
// stays in VGPRs for the whole sweep and the collocation tile streams from L2 in fragment order.
#include <stdlib.h>

#include "gp_common.hpp"

namespace scasml {

// NK4 = kp / 8 float4 per lane per row; PT = point tiles (of 32) per wave.
//
// Workgroup = 8 waves (512 threads).  Per collocation tile the workgroup stages [NK4 KiB of A fragments | 2 KiB of
// constants] into one of three LDS slots with global_load_lds (16 B per lane, lane-linear image = the pre-packed
// fragment order), one tile ahead, and meets at ONE barrier per tile.  This FP32-input MFMA kernel is the
// arithmetic reference mode (GP.eval_split = 0); the production kernels are in gp_eval_bf16.hip.
template <int NK4>
__device__ __forceinline__ constexpr int gp_stage_floats() { return NK4 * 256 + 512; }

template <int NK4, int PT>
__device__ __forceinline__ void gp_mfma_tile(const GpStageView &st, const float4 (&xf)[PT][NK4], f32x16 (&acc)[PT], int lane) {
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;
    // one-deep register prefetch of the A fragment: the ds_read of step v+1 is issued before the 4*PT
    // MFMAs of step v (>= 256 cycles of cover); the scheduling barrier keeps the compiler from hoisting
    // all NK4 reads (4*NK4 VGPRs) to the top.
    float4 y = st.y[lane];
#pragma unroll
    for (int v = 0; v < NK4; ++v) {
        float4 yn = y;
        if (v + 1 < NK4) yn = st.y[(v + 1) * 64 + lane];
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(y.x, xf[p][v].x, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(y.y, xf[p][v].y, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(y.z, xf[p][v].z, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(y.w, xf[p][v].w, acc[p], 0, 0, 0);
        }
        y = yn;
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NK4, int PT>
__global__ __launch_bounds__(512, 2) void gp_eval_kernel(const GpArgs g) {
    constexpr int STAGE = gp_stage_floats<NK4>();       // floats per LDS slot
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 3 slots
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int64_t p0 = ((int64_t)blockIdx.x * 8 + wv) * (32 * PT);
    const int n_tiles = g.n_pad / 32;

    // ---- stage a collocation tile: wave w copies fragment rows v = w, w+8, ..; wave (NK4 % 8) the coefficients
    auto stage = [&](int tile, int slot) {
        float *dst = lds + slot * STAGE;
        const float *src = g.colloc_frag + (int64_t)tile * NK4 * 256;
        for (int v = wv; v < NK4; v += 8)
            __builtin_amdgcn_global_load_lds(src + v * 256 + lane * 4, dst + v * 256, 16, 0, 0);
        if (wv == (NK4 & 7))
            __builtin_amdgcn_global_load_lds(g.coef + (int64_t)tile * 512 + lane * 4, dst + NK4 * 256, 16, 0, 0);
        if (wv == ((NK4 + 1) & 7))
            __builtin_amdgcn_global_load_lds(g.coef + (int64_t)tile * 512 + 256 + lane * 4, dst + NK4 * 256 + 256, 16, 0, 0);
    };
    auto view = [&](int slot) {
        const float *b = lds + slot * STAGE;
        return GpStageView{reinterpret_cast<const float4 *>(b), b + NK4 * 256};
    };
    stage(0, 0);

    // ---- this wave's point tiles -> registers, plus |x|^2, a*sum x, a*t per point ------------
    float4 xf[PT][NK4];
    float nx[PT], sx[PT], tx[PT];
    const int kbase = half * (4 * NK4);
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        int64_t row = p0 + 32 * p + col;
        if (row >= g.n_inf) row = g.n_inf - 1;  // shadow rows, never stored
        const float4 *src = reinterpret_cast<const float4 *>(g.points + row * g.kp + kbase);
        float pn = 0.0f, ps = 0.0f, pt = 0.0f;
#pragma unroll
        for (int v = 0; v < NK4; ++v) {
            const float4 q = src[v];
            xf[p][v] = q;
            const float e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = kbase + 4 * v + c;
                pn = fmaf(e[c], e[c], pn);
                ps += k < g.d ? e[c] : 0.0f;
                pt += k == g.d ? e[c] : 0.0f;
            }
        }
        pn += __shfl_xor(pn, 32);
        ps += __shfl_xor(ps, 32);
        pt += __shfl_xor(pt, 32);
        nx[p] = pn;
        sx[p] = g.a * ps;
        tx[p] = g.a * pt;
    }
    float au[PT], at[PT], ad[PT], al[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) au[p] = at[p] = ad[p] = al[p] = 0.0f;
    GpConsts c;
    c.a = g.a;
    c.a2 = g.a * g.a;
    c.ad = g.a * (float)g.d;
    c.kexp = -0.5f * g.a * 1.44269504088896341f;  // exp(-a r2/2) = exp2(r2 * kexp)
    c.dF = (float)g.d;

    f32x16 acc[PT];
    __syncthreads();  // tile 0 has landed (the barrier drains the LDS-DMA: vmcnt(0))
    for (int jt = 0; jt < n_tiles; ++jt) {
        if (jt + 1 < n_tiles) stage(jt + 1, (jt + 1) % 3);
        gp_mfma_tile<NK4, PT>(view(jt % 3), xf, acc, lane);
        gp_epilogue_tile<PT>(view(jt % 3), acc, c, half, nx, sx, tx, au, at, ad, al);
        __syncthreads();
    }

    // ---- combine the two half-waves (rows 4h..4h+3 of each group) and store -------------------
    const float s2 = g.sigma * g.sigma;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const float u = au[p] + __shfl_xor(au[p], 32);
        const float dt = at[p] + __shfl_xor(at[p], 32);
        const float dv = ad[p] + __shfl_xor(ad[p], 32);
        const float lp = al[p] + __shfl_xor(al[p], 32);
        const int64_t row = p0 + 32 * p + col;
        if (half == 0 && row < g.n_inf) {
            // models/GP.py:767-768
            const float eps = gp_pde_residual(g, u, dt, dv, lp);   // models/GP.py:767-768
            g.out4[row] = make_float4(u, dv, eps, dt);
            if (g.lap) g.lap[row] = lp;
        }
    }
}

// Full gradient of the posterior mean (models/GP.py:673-687).  Not on the solver hot path
// (f of Grad_Dependent_Nonlinear needs only the spatial sum, which gp_eval_kernel returns);
// one wave per point, lanes stride the collocation points, d-vector partials in LDS-free form:
//   d/dx_i u = x_i * A1 - sum_j alpha_j y_ji + A2,  alpha_j = kappa_j a (2 a cL_j - E_j),
//   A1 = sum_j alpha_j,  A2 = sum_j kappa_j a cS_j;   d/dt u = sum_j kappa_j (a ct_j - p E_j).
__global__ __launch_bounds__(256) void gp_gradient_kernel(const float *points, const float *colloc, const float *coef,
                                                          float *grad, int64_t n_inf, int n_pad, int kp, int d, float a) {
    extern __shared__ float sh[];  // per wave: kp floats of the point + 64 alphas
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    const bool valid = row < n_inf;
    if (!valid) row = n_inf - 1;
    float *xs = sh + wv * (kp + 64);
    float *alpha = xs + kp;
    for (int k = lane; k < kp; k += 64) xs[k] = points[row * kp + k];
    __syncthreads();
    float nxv = 0.0f, sxv = 0.0f;
    for (int k = 0; k <= d; ++k) nxv = fmaf(xs[k], xs[k], nxv);
    for (int k = 0; k < d; ++k) sxv += xs[k];
    const float txa = a * xs[d], sxa = a * sxv;
    const float a2 = a * a, ad_ = a * (float)d;
    // each lane accumulates dims lane, lane+64, ... of  -sum_j alpha_j y_j
    float gacc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float A1 = 0.0f, A2 = 0.0f, gt = 0.0f;
    for (int j0 = 0; j0 < n_pad; j0 += 64) {
        const int j = j0 + lane;
        float al = 0.0f;
        if (j < n_pad) {
            const float *y = colloc + (int64_t)j * kp;
            float dot = 0.0f;
            for (int k = 0; k <= d; ++k) dot = fmaf(xs[k], y[k], dot);
            const float *cf = coef + (int64_t)j * kCoefRow;
            const float r2 = fmaf(-2.0f, dot, nxv + cf[12]);
            const float pp = txa - cf[1], ss = sxa - cf[0];
            const float kap = expf(-0.5f * a * r2);
            const float L = fmaf(-pp, pp, fmaf(a2, r2, -ad_));
            const float c0 = cf[2], cL = cf[3], ct = cf[4], cS = cf[5];
            const float E = fmaf(cS, ss, fmaf(ct, pp, fmaf(cL, L, c0)));
            al = kap * a * (2.0f * a * cL - E);
            A1 += al;
            A2 += kap * a * cS;
            gt += kap * fmaf(-pp, E, a * ct);
        }
        alpha[lane] = al;
        __syncthreads();
        const int jn = n_pad - j0 < 64 ? n_pad - j0 : 64;
        for (int jj = 0; jj < jn; ++jj) {
            const float aj = alpha[jj];
            const float *y = colloc + (int64_t)(j0 + jj) * kp;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = lane + 64 * c;
                if (k < d) gacc[c] = fmaf(-aj, y[k], gacc[c]);
            }
        }
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) {
        A1 += __shfl_xor(A1, o);
        A2 += __shfl_xor(A2, o);
        gt += __shfl_xor(gt, o);
    }
    if (!valid) return;
    float *out = grad + row * (d + 1);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k = lane + 64 * c;
        if (k < d) out[k] = fmaf(xs[k], A1, gacc[c]) + A2;
    }
    if (lane == 0) out[d] = gt;
}

// Build the device model from the training set and right_vector (models/GP.py:593-600).
__global__ void gp_pack_kernel(int d, float a, float T, const float *x_dom, int n_dom, const float *x_bdy, int n_bdy,
                               const double *rv, float *colloc, float *frag, uint16_t *bf, float *coef, int n_pad, int kp) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pad) return;
    const int N = n_dom + n_bdy;
    const float *src = j < n_dom ? x_dom + (int64_t)j * (d + 1) : (j < N ? x_bdy + (int64_t)(j - n_dom) * (d + 1) : nullptr);
    float ny = 0.0f, sy = 0.0f, ty = 0.0f;
    float ny_full = 0.0f;
    if (src)
        for (int k = 0; k <= d; ++k) ny_full = fmaf(src[k], src[k], ny_full);
    const int nk4 = kp / 8, tile = j / 32, i = j % 32;
    for (int k = 0; k < kp; ++k) {
        const float v = (src && k <= d) ? src[k] : 0.0f;
        colloc[(int64_t)j * kp + k] = v;
        // fragment order [tile][v][lane = half*32 + i][4], half h covers k in [h*kp/2, (h+1)*kp/2)
        const int h = k / (kp / 2), kk = k % (kp / 2);
        frag[(((int64_t)tile * nk4 + kk / 4) * 64 + h * 32 + i) * 4 + (kk & 3)] = v;
        // 16-bit operand planes of the exponent-scaled kernels (gp_common.hpp, gp_epilogue_scaled): with
        // q = log2(e)/(2a), k1 = -q the product of a collocation row and a point row is  Lam = k1 a^2 |x - y|^2.
        // MFMA 32x32x16 A-fragment order [tile][plane][step][lane = half*32 + i][8]: half h covers k in
        // [h*kp/2, (h+1)*kp/2), step = kk/8.  Column kp-1 is the constant 1 that meets k1 a^2 |x|^2 of the point row.
        const float qs = 0.5f * 1.44269504088896341f / a, k1 = -qs;
        const float ay = k1 * a * a * ny_full;
        // bf16 planes (truncation split, v = hi + mid + lo exactly up to 2^-24 |v|): the collocation side carries
        // the factor, 2 a^2 q y_k (k <= d), and k1 a^2 |y|^2 in column kp-2 against a constant 1 in the point row
        const float vf = k <= d ? 2.0f * a * a * qs * v : (k == kp - 2 ? ay : (k == kp - 1 ? 1.0f : 0.0f));
        const uint32_t hb = __float_as_uint(vf) & 0xFFFF0000u;
        const float r1 = vf - __uint_as_float(hb);
        const uint32_t mb = __float_as_uint(r1) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(mb);
        const uint32_t lb = __float_as_uint(r2) & 0xFFFF0000u;
        const int ks = kp / 16;
        const int64_t e = (((int64_t)(kk / 8)) * 64 + h * 32 + i) * 8 + (kk & 7);
        bf[((int64_t)tile * 3 + 0) * ks * 512 + e] = (uint16_t)(hb >> 16);
        bf[((int64_t)tile * 3 + 1) * ks * 512 + e] = (uint16_t)(mb >> 16);
        bf[((int64_t)tile * 3 + 2) * ks * 512 + e] = (uint16_t)(lb >> 16);
        // fp16 planes (h, l = fp16(v - h), unscaled), stored behind the bf16 planes.  Here the factor 2 a^2 q sits on
        // the POINT side (gp_eval_bf16.hip): the planes hold y itself, and k1 a^2 |y|^2 as h in column kp-3 and l in
        // column kp-2 of plane 0, met by the constants 1 and 1 in the point row (the kernel patches them in: the last
        // three columns are compile-time fragment positions; kp >= d + 4 keeps them clear of the data).  If y is exactly fp16 (float16
        // collocation points, as in the reference protocol) plane 1 is identically zero and is never read.
        uint16_t *hf = bf + (int64_t)3 * n_pad * kp;
        const _Float16 ayh = (_Float16)ay;
        const float vg = k <= d ? v : (k == kp - 3 ? (float)ayh : (k == kp - 2 ? ay - (float)ayh : (k == kp - 1 ? 1.0f : 0.0f)));
        const _Float16 fh = (_Float16)vg;
        const _Float16 fl = k <= d ? (_Float16)(vg - (float)fh) : (_Float16)0.0f;
        hf[((int64_t)tile * 2 + 0) * ks * 512 + e] = __builtin_bit_cast(unsigned short, fh);
        hf[((int64_t)tile * 2 + 1) * ks * 512 + e] = __builtin_bit_cast(unsigned short, fl);
        ny = fmaf(v, v, ny);
        if (k < d) sy += v;
        if (k == d) ty = v;
    }
    float c0 = 0.0f, cL = 0.0f, ct = 0.0f, cS = 0.0f;
    if (j < n_dom) {
        c0 = (float)rv[j];
        cL = (float)rv[n_dom + n_bdy + j];
        ct = (float)rv[2 * n_dom + n_bdy + j];
        cS = (float)rv[3 * n_dom + n_bdy + j];
    } else if (j < N) {
        c0 = (float)rv[j];
    }
    const float fd = (float)d;
    float *cf = coef + (int64_t)j * kCoefRow;   // layout: gp_common.hpp
    cf[0] = a * sy;
    cf[1] = a * ty;
    cf[2] = c0;
    cf[3] = cL;
    cf[4] = ct;
    cf[5] = cS;
    cf[6] = a * ct;
    cf[7] = 2.0f * a * cL;
    cf[8] = a * fd * cS;
    cf[9] = -4.0f * a * cL;
    cf[10] = -2.0f * a * cS;
    cf[11] = -2.0f * a * a * fd * cL;
    cf[12] = ny;
    cf[13] = cf[14] = cf[15] = 0.0f;
    // exponent-scaled constants of the 16-bit kernels (layout and derivation: gp_common.hpp, gp_epilogue_scaled)
    const float qs = 0.5f * 1.44269504088896341f / a, k1 = -qs, rq = sqrtf(qs);
    float *c2 = coef + ((int64_t)n_pad + j) * kCoefRow;
    c2[0] = a * sy;
    c2[1] = rq * a * ty;
    c2[2] = c0 - a * fd * cL;
    c2[3] = cL / k1;
    c2[4] = ct / rq;
    c2[5] = cS;
    c2[6] = rq * a * ct;
    c2[7] = 2.0f * a * cL;
    c2[8] = a * fd * cS;
    c2[9] = -4.0f * a * cL;
    c2[10] = -2.0f * a * k1 * cS;
    c2[11] = 2.0f * a * a * fd * k1 * cL;
    // terminal-time form (site kind 3): with t_x = T the scaled time difference pT = sqrt(q) a (T - t_y) is a constant of the
    // row, so E = e0 + eL (Lam + pT^2) + et pT + cS (a S_x - a S_y) = e0T + eL Lam + cS (a S_x) with the row's own terms folded
    // into e0T, and one ds_read_b128 carries the row
    const float pT = rq * a * (T - ty);
    c2[12] = 0.0f;
    c2[13] = c2[2] + c2[3] * pT * pT + c2[4] * pT - cS * (a * sy);
    c2[14] = c2[3];
    c2[15] = cS;
}

// ---- E-from-MFMA constants (gp_common.hpp, gp_epilogue_em): the four collocation-side entries of the linear part of E per row,
// from the exponent-scaled constants `c2` of the row and the terminal time
struct EmRow {
    float cE, et, cS;
};
__device__ __forceinline__ EmRow em_row(const float *c2) {
    EmRow r;
    r.cE = c2[2] - c2[4] * c2[1] - c2[5] * c2[0];   // e0 - et vty - cS vsy
    r.et = c2[4];
    r.cS = c2[5];
    return r;
}
// pass 1: largest magnitude over all rows -> bits of a non-negative float in *maxbits (zeroed by the caller)
__global__ void gp_em_max_kernel(const float *coef2, int n_pad, unsigned int *maxbits) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.0f;
    if (j < n_pad) {
        const EmRow r = em_row(coef2 + (int64_t)j * kCoefRow);
        m = fmaxf(fmaxf(fabsf(r.cE), fabsf(r.et)), fabsf(r.cS));
        if (!(m < 3.0e38f)) m = 3.0e38f;            // NaN / inf rows (an untrained model) must not poison the scale
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(maxbits, __float_as_uint(m));
}
// pass 2: scale 2^s (largest entry -> [2^12, 2^13)), fp16 (h, l) fragments of the E plane, scaled row constants
__global__ void gp_em_pack_kernel(const float *coef2, int n_pad, const unsigned int *maxbits, uint16_t *eplane, float *coef3, float *escale) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const float mx = __uint_as_float(*maxbits);
    int ex = 0;
    if (mx > 0.0f) frexpf(mx, &ex);                  // mx = f 2^ex, f in [0.5, 1)
    const float S = mx > 0.0f ? ldexpf(1.0f, 13 - ex) : 1.0f;
    if (j == 0) {
        escale[0] = 1.0f / S;
        escale[1] = S;
    }
    if (j >= n_pad) return;
    const float *c2 = coef2 + (int64_t)j * kCoefRow;
    const EmRow r = em_row(c2);
    auto hl = [&](float v, _Float16 &h, _Float16 &l) {
        const float sv = S * v;
        h = (_Float16)sv;
        l = (_Float16)(sv - (float)h);
    };
    _Float16 ch, cl, th, tl, sh, sl;
    hl(r.cE, ch, cl);
    hl(r.et, th, tl);
    hl(r.cS, sh, sl);
    const _Float16 k[8] = {ch, cl, th, th, tl, sh, sh, sl};
    const int tile = j / 32, i = j % 32;
    uint16_t *frag = eplane + (int64_t)tile * 512;                    // 1 KiB per tile: 512 B of fragments, 512 B of row constants
    for (int e = 0; e < 8; ++e) frag[((e / 4) * 32 + i) * 4 + (e & 3)] = __builtin_bit_cast(unsigned short, k[e]);
    float *trow = reinterpret_cast<float *>(frag + 256) + i * 4;
    trow[0] = S * c2[13];                            // e0T
    trow[1] = S * c2[3];                             // eL
    trow[2] = S * c2[5];                             // cS
    trow[3] = S * c2[2];                             // e0 (tiles of boundary rows)
    float *c3 = coef3 + (int64_t)j * 8;
    c3[0] = c2[0];
    c3[1] = c2[1];
    c3[2] = S * c2[3];
    c3[3] = S * c2[6];
    c3[4] = S * c2[7];
    c3[5] = S * c2[8];
    c3[6] = S * c2[10];
    c3[7] = S * c2[11];
}

template <int NK4>
static int launch_eval(const GpArgs &g, hipStream_t s) {
    constexpr int PT = NK4 <= 13 ? 2 : 1;
    const int64_t waves = (g.n_inf + 32 * PT - 1) / (32 * PT);
    const int64_t blocks = (waves + 7) / 8;
    if (blocks > 0x7FFFFFFF) return fail(SCASML_ERR_UNSUPPORTED, "gp_eval: too many points");
    constexpr size_t lds_bytes = 3 * (NK4 * 256 + 512) * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "LDS slots exceed 160 KiB");
    auto kern = gp_eval_kernel<NK4, PT>;
    if (lds_bytes > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return fail(SCASML_ERR_HIP, "gp_eval: cannot reserve %zu bytes of LDS", lds_bytes);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds_bytes, s, g);
    return check_launch("gp_eval launch");
}

static int check_model(const scasml_gp_model *m, const char *who) {
    if (!m || !m->colloc || !m->colloc_frag || !m->coef) return fail(SCASML_ERR_ARG, "%s: null model", who);
    if (m->d < 1 || m->d > SCASML_MAX_DIM) return fail(SCASML_ERR_UNSUPPORTED, "%s: d=%d outside 1..%d", who, m->d, SCASML_MAX_DIM);
    if (m->kp != scasml_point_stride(m->d)) return fail(SCASML_ERR_ARG, "%s: kp=%d does not match d=%d", who, m->kp, m->d);
    if (m->n_pad < 32 || m->n_pad % SCASML_GP_TILE || m->n_pad < m->n_dom + m->n_bdy)
        return fail(SCASML_ERR_ARG, "%s: n_pad=%d invalid for %d+%d points", who, m->n_pad, m->n_dom, m->n_bdy);
    return 0;
}

}  // namespace scasml

using namespace scasml;

// planes: 3 bf16 + 2 fp16 planes of n_pad x kp, then the E plane (16 halfwords per row);  coef: coef, coef2 (16 floats per row each),
// coef3 (8 per row), 16 floats of scale words
extern "C" int64_t scasml_gp_plane_halfwords(int32_t d, int32_t n_pad) {
    const int64_t kp = scasml_point_stride(d);
    return (int64_t)n_pad * 5 * kp + (int64_t)n_pad * 16;
}

extern "C" int64_t scasml_gp_coef_floats(int32_t n_pad) { return (int64_t)n_pad * (2 * kCoefRow + 8) + 16; }

extern "C" int scasml_gp_pack(int32_t d, float a, float T_terminal, const float *x_dom, int32_t n_dom, const float *x_bdy, int32_t n_bdy,
                              const double *rv, float *colloc_out, float *colloc_frag_out, uint16_t *colloc_bf16_out,
                              float *coef_out, void *stream) {
    if (!x_dom || !rv || !colloc_out || !colloc_frag_out || !colloc_bf16_out || !coef_out || (n_bdy > 0 && !x_bdy))
        return fail(SCASML_ERR_ARG, "gp_pack: null argument");
    if (d < 1 || d > SCASML_MAX_DIM || n_dom < 1 || n_bdy < 0) return fail(SCASML_ERR_ARG, "gp_pack: bad sizes");
    const int n_pad = (n_dom + n_bdy + 31) / 32 * 32;
    const int kp = scasml_point_stride(d);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gp_pack_kernel, dim3((n_pad + 63) / 64), dim3(64), 0, s, d, a, T_terminal, x_dom, n_dom, x_bdy,
                       n_bdy, rv, colloc_out, colloc_frag_out, colloc_bf16_out, coef_out, n_pad, kp);
    // E-from-MFMA constants: a global power-of-two scale needs the largest entry first
    float *coef2 = coef_out + (int64_t)n_pad * kCoefRow, *coef3 = coef_out + (int64_t)n_pad * 2 * kCoefRow;
    float *escale = coef3 + (int64_t)n_pad * 8;
    unsigned int *maxbits = reinterpret_cast<unsigned int *>(escale + 4);
    uint16_t *eplane = colloc_bf16_out + (int64_t)n_pad * 5 * kp;
    if (hipMemsetAsync(maxbits, 0, sizeof(unsigned int), s) != hipSuccess) return fail(SCASML_ERR_HIP, "gp_pack: memset failed");
    hipLaunchKernelGGL(gp_em_max_kernel, dim3((n_pad + 63) / 64), dim3(64), 0, s, coef2, n_pad, maxbits);
    hipLaunchKernelGGL(gp_em_pack_kernel, dim3((n_pad + 63) / 64), dim3(64), 0, s, coef2, n_pad, maxbits, eplane, coef3, escale);
    return check_launch("gp_pack launch");
}

static int gp_eval_impl(const scasml_gp_model *m, const float *points, int64_t n_inf, float *out4, float *lap,
                        int64_t rows_per_site, const uint8_t *site_u_only, void *stream) {
    if (n_inf == 0) return 0;
    if (int rc = check_model(m, "gp_eval")) return rc;
    if (n_inf < 0 || !points || !out4) return fail(SCASML_ERR_ARG, "gp_eval: bad argument");
    GpArgs g;
    g.points = points;
    g.colloc_frag = m->colloc_frag;
    g.colloc_bf16 = m->colloc_bf16;
    g.colloc_f16 = m->colloc_bf16 ? m->colloc_bf16 + (int64_t)3 * m->n_pad * m->kp : nullptr;
    g.colloc_is_f16 = m->colloc_is_f16;
    g.site_u_only = site_u_only;
    g.rows_per_site = rows_per_site;
    g.coef = m->coef;
    g.coef2 = m->coef + (int64_t)m->n_pad * kCoefRow;
    g.coef3 = m->coef + (int64_t)m->n_pad * 2 * kCoefRow;
    g.escale = g.coef3 + (int64_t)m->n_pad * 8;
    g.eplane = m->colloc_bf16 ? m->colloc_bf16 + (int64_t)5 * m->n_pad * m->kp : nullptr;
    g.first_bdy_tile = (m->n_dom + SCASML_GP_TILE - 1) / SCASML_GP_TILE;
    g.out4 = reinterpret_cast<float4 *>(out4);
    g.lap = lap;
    g.n_inf = n_inf;
    g.n_pad = m->n_pad;
    g.kp = m->kp;
    g.d = m->d;
    g.a = m->a;
    g.sigma = m->sigma_eq;
    g.mu = m->mu_eq;
    g.eq_id = m->eq_id;
    if (!eq_known(m->eq_id)) return fail(SCASML_ERR_UNSUPPORTED, "gp_eval: unknown equation id %d", m->eq_id);
#ifdef SCASML_ABLATION   // development builds only: phases of the 16-bit kernels switched off through the environment
    const char *dbg = getenv("SCASML_GP_DBG");
    g.dbg = dbg ? atoi(dbg) : 0;
#else
    g.dbg = 0;
#endif
    hipStream_t s = (hipStream_t)stream;
    if (m->split == 2 || m->split == 3 || m->split == 22) {
        if (!m->colloc_bf16) return fail(SCASML_ERR_ARG, "gp_eval: split=%d needs colloc_bf16", m->split);
        // fp16 planes carry k1 a^2 |x|^2 = -0.72 a |x|^2 itself and 2 a^2 q x_k = 1.44 a x_k: the caller states a bound on the
        // coordinates (scasml_gp_model.x_bound, 0 = the default 2) and the mode is refused where that bound would leave the
        // fp16 range -- rows outside the bound are the caller's breach of the precondition (scasml_hip.h)
        const float xb = m->x_bound > 0.0f ? m->x_bound : 2.0f;
        if (m->split == 22 && 0.7213f * m->a * xb * xb * (float)(m->d + 1) > 3.0e4f)
            return fail(SCASML_ERR_UNSUPPORTED, "gp_eval: a = 1/sigma^2 = %g with |x_k| <= %g is outside the range of the fp16x2 mode at d = %d; use split = 3",
                        (double)m->a, (double)xb, m->d);
        return launch_gp_eval_bf16(g, m->split, s);
    }
    if (m->split != 0) return fail(SCASML_ERR_ARG, "gp_eval: split must be 0, 2, 3 or 22");
#define SCASML_EVAL_CASE(NK) \
    case NK: return launch_eval<NK>(g, s);
    switch (m->kp / 8) {
         SCASML_EVAL_CASE(2)  SCASML_EVAL_CASE(4)  SCASML_EVAL_CASE(6)
         SCASML_EVAL_CASE(8)  SCASML_EVAL_CASE(10)  SCASML_EVAL_CASE(12)
         SCASML_EVAL_CASE(14)  SCASML_EVAL_CASE(16)  SCASML_EVAL_CASE(18)
         SCASML_EVAL_CASE(20)  SCASML_EVAL_CASE(22)  SCASML_EVAL_CASE(24)
         SCASML_EVAL_CASE(26)  SCASML_EVAL_CASE(28)  SCASML_EVAL_CASE(30)
         SCASML_EVAL_CASE(32)
    }
#undef SCASML_EVAL_CASE
    return fail(SCASML_ERR_UNSUPPORTED, "gp_eval: kp=%d", m->kp);
}

extern "C" int scasml_gp_eval(const scasml_gp_model *m, const float *points, int64_t n_inf, float *out4, float *lap,
                              void *stream) {
    return gp_eval_impl(m, points, n_inf, out4, lap, 0, nullptr, stream);
}

extern "C" int scasml_gp_eval_sites(const scasml_gp_model *m, const float *points, int64_t n_inf, int64_t rows_per_site,
                                    const uint8_t *site_u_only, float *out4, void *stream) {
    if (site_u_only && rows_per_site < 1) return fail(SCASML_ERR_ARG, "gp_eval_sites: rows_per_site must be positive");
    return gp_eval_impl(m, points, n_inf, out4, nullptr, rows_per_site, site_u_only, stream);
}

extern "C" int scasml_gp_gradient(const scasml_gp_model *m, const float *points, int64_t n_inf, float *grad, void *stream) {
    if (n_inf == 0) return 0;
    if (int rc = check_model(m, "gp_gradient")) return rc;
    if (n_inf < 0 || !points || !grad) return fail(SCASML_ERR_ARG, "gp_gradient: bad argument");
    const int64_t blocks = (n_inf + 3) / 4;
    if (blocks > 0x7FFFFFFF) return fail(SCASML_ERR_UNSUPPORTED, "gp_gradient: too many points");
    const size_t lds = 4 * (m->kp + 64) * sizeof(float);
    hipLaunchKernelGGL(gp_gradient_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, points, m->colloc,
                       m->coef, grad, n_inf, m->n_pad, m->kp, m->d, m->a);
    return check_launch("gp_gradient launch");
}
