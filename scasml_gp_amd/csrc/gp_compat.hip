// The reference's surrogate AS CODED (GP.compat = "reference"): the 5-index Hutchinson "Laplacian" on a cyclically
// shifted argument (models/GP.py:28-39, 87-105, 119-127, 141-179) and the float16 rounding of every kernel entry
// (:43 and the .astype(jnp.float16) that closes every derivative kernel).  SURVEY.md Appendix E-5/E-6; derivation and
// the CPU statement: oracle/gp_compat.py.  This is the parity mode for the reference's own experiment protocol
// (1000 + 200 collocation points, 1200 evaluation points, n = rho = 2: 6.8e7 kernel pairs per solve) -- the default
// surrogate with the exact operators runs on the MFMA kernels of gp_eval_bf16.hip.  Arithmetic is float64 throughout so
// that the float16 rounding of an entry is decided exactly as the float64 NumPy statement decides it.
//
// Geometries (v' = (v_2, ..., v_d, t, v_1), the shift of models/GP.py:91-93; time is the last column):
//   al: r = x - y      everything without a Laplacian, and lap_x lap_y (components i+1)
//   ys: r = x - y'     lap_y of kappa, dt_x kappa, div_x kappa
//   xs: r = x' - y     lap_x of kappa, dt_y kappa, div_y kappa
// idx[5] indexes the SHIFTED vector (0 <= i < d): component i of x', i.e. original coordinate i+1.
#include "common.hpp"
#include "equations.hpp"

namespace scasml {

constexpr int kMC = 5;   // models/GP.py:30

struct CompatIdx {
    int32_t i[kMC];
};

__device__ __forceinline__ double round16(double v, int on) {
    // float64 -> float16 (RNE, one rounding: v_cvt_f16_f32 of a float64 would round twice) -> float64
    if (!on) return v;
    const double av = fabs(v);
    if (!(av == av)) return v;
    if (av >= 65520.0) return v < 0 ? -INFINITY : INFINITY;
    if (av < 5.9604644775390625e-8 * 0.5) return v < 0 ? -0.0 : 0.0;     // below half the smallest subnormal
    int e;
    frexp(av, &e);                                     // av = m 2^e, m in [0.5, 1)
    int ulp_exp = e - 11;                              // 11 significant bits
    if (ulp_exp < -24) ulp_exp = -24;                  // subnormal spacing 2^-24
    const double q = ldexp(av, -ulp_exp);              // integer part carries the kept bits
    const double r = rint(q);                          // RNE (default rounding mode)
    const double o = ldexp(r, ulp_exp);
    return v < 0 ? -o : o;
}

// The three geometries of one (x, y) pair.  x, y: pointers to d+1 coordinates with strides sx, sy (elements).
struct PairGeom {
    double kap[3];   // al, ys, xs
    double S[3];
    double rD[3];
    double ri[3][kMC];
};

template <class FX, class FY>
__device__ __forceinline__ void pair_geometry(int d, double a, const CompatIdx &ix, FX x, FY y, PairGeom &g) {
    const int D = d + 1;
    double r2[3] = {0.0, 0.0, 0.0}, S[3] = {0.0, 0.0, 0.0};
    // one pass over k: x_k, x'_k = x_{k+1 mod D}, y_k, y'_k
    double xk = x(0), yk = y(0);
    const double x0 = xk, y0 = yk;
    for (int k = 0; k < D; ++k) {
        const double xn = k + 1 < D ? x(k + 1) : x0, yn = k + 1 < D ? y(k + 1) : y0;
        const double ral = xk - yk, rys = xk - yn, rxs = xn - yk;
        r2[0] = fma(ral, ral, r2[0]);
        r2[1] = fma(rys, rys, r2[1]);
        r2[2] = fma(rxs, rxs, r2[2]);
        if (k < d) {
            S[0] += ral;
            S[1] += rys;
            S[2] += rxs;
        } else {
            g.rD[0] = ral;
            g.rD[1] = rys;
            g.rD[2] = rxs;
        }
        xk = xn;
        yk = yn;
    }
#pragma unroll
    for (int j = 0; j < kMC; ++j) {
        const int i = ix.i[j];                                  // 0 <= i < d, so i + 1 <= d
        g.ri[0][j] = x(i + 1) - y(i + 1);                       // al: shifted component i of (x - y)'
        g.ri[1][j] = x(i) - y(i + 1);                           // ys: x_i - y'_i
        g.ri[2][j] = x(i + 1) - y(i);                           // xs: x'_i - y_i
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        g.kap[q] = exp(-0.5 * a * r2[q]);
        g.S[q] = S[q];
    }
}

// The 16 operator pairs (opx, opy), ops: 0 = I, 1 = Lap, 2 = dt, 3 = div, each rounded to float16 if asked.
__device__ __forceinline__ void compat_blocks(int d, double a, const PairGeom &g, int r16, double (&P)[4][4]) {
    const double h = (double)d / kMC, a2 = a * a, a3 = a2 * a;
    const double kal = g.kap[0], S = g.S[0], rt = g.rD[0];
    double sg[3], mix[3], dbl = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        double s = 0.0, m = 0.0;
#pragma unroll
        for (int j = 0; j < kMC; ++j) {
            const double r = g.ri[q][j];
            s += a2 * r * r - a;
            m += 2.0 * a2 * r + a2 * g.S[q] - a3 * g.S[q] * r * r;
            if (q == 0) dbl += 2.0 * a2 - 4.0 * a3 * r * r;
        }
        sg[q] = s;
        mix[q] = m;
    }
    P[0][0] = kal;
    P[0][2] = a * rt * kal;
    P[0][3] = a * S * kal;
    P[2][0] = -a * rt * kal;
    P[2][2] = (a - a2 * rt * rt) * kal;
    P[2][3] = -a2 * rt * S * kal;
    P[3][0] = -a * S * kal;
    P[3][2] = -a2 * rt * S * kal;
    P[3][3] = (a * d - a2 * S * S) * kal;
    P[0][1] = h * sg[1] * g.kap[1];                                  // lap_y kappa            [ys]
    P[2][1] = -a * g.rD[1] * h * sg[1] * g.kap[1];                   // dt_x lap_y             [ys]
    P[3][1] = h * mix[1] * g.kap[1];                                 // div_x lap_y            [ys]
    P[1][0] = h * sg[2] * g.kap[2];                                  // lap_x kappa            [xs]
    P[1][2] = a * g.rD[2] * h * sg[2] * g.kap[2];                    // lap_x dt_y             [xs]
    P[1][3] = -h * mix[2] * g.kap[2];                                // lap_x div_y            [xs]
    P[1][1] = h * h * (sg[0] * sg[0] + dbl) * kal;                   // lap_x lap_y            [al]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) P[i][j] = round16(P[i][j], r16);
}

// ---- the reference's FLOAT16 ARITHMETIC on float16 rows (round16 bit 2; GP(f16_graph=True); oracle: OracleGPCompat(f16_graph=2)) ------------
// On float16 rows (the collocation points, the harness's test points) kappa = exp(-sum((x - y)**2) / (2 sigma**2)) is float16 arithmetic
// throughout -- self.sigma is weakly typed (models/GP.py:25, 41-43) -- and the derivative kernels are reverse-mode autodiff THROUGH it (:55-85;
// the dt / div second-order blocks reverse over reverse, :107-139).  This restates that op sequence for the nine Laplacian-free operator pairs:
//   r_k = f16(x_k - y_k); S = f16(sum_f32 f16(r_k^2)); q = f16(-S inv16); kappa16 = f16(exp(q)); t1 = f16(kappa16 inv16); m_k = 2 r_k
//   d/dx_k kappa = f16(-t1 m_k);  div_x = f16(sum_f32 over k < d);  d/dy = -d/dx
//   second order, h = dt_x or div_x of kappa, w = m_d or f16(sum_f32 m_k):  gS = f16(f16(f16(w inv16) kappa16) inv16),
//   grad_y h [k] = -f16(-2 t1 [k differentiated directly] + f16(gS m_k));  dt_y picks k = d, div_y is the float16 sum over k < d.
// inv16 = f16(1 / f16(2 sigma^2)): the division by the constant reaches the device as a multiplication by its folded reciprocal (XLA's
// algebraic simplifier); the reference's logged GP errors decide for this reading (profiles/HISTORY.md section 2, tests/studies/f16_graph_study.py).
// The Hutchinson blocks keep one rounding per entry (compat_blocks).  Float32 accumulations run in index order.
__device__ __forceinline__ _Float16 hmul(_Float16 a, _Float16 b) { return (_Float16)((float)a * (float)b); }   // exact product, one rounding

template <class FX, class FY>
__device__ __forceinline__ void f16_graph_blocks(int d, double a, FX x, FY y, double (&P)[4][4]) {
    const _Float16 c16 = (_Float16)(2.0 / a);
    const _Float16 inv16 = (_Float16)(1.0f / (float)c16);
    float Sf = 0.0f;
    for (int k = 0; k <= d; ++k) {
        const _Float16 r = (_Float16)x(k) - (_Float16)y(k);
        Sf += (float)(r * r);
    }
    const _Float16 S = (_Float16)Sf;
    const _Float16 q = hmul(-S, inv16);
    const _Float16 kap = (_Float16)(float)exp((double)q);
    const _Float16 t1 = hmul(kap, inv16);
    const _Float16 two_t1 = (_Float16)2.0f * t1;
    float sum_m = 0.0f, sum_dx = 0.0f;
    for (int k = 0; k < d; ++k) {
        const _Float16 m = (_Float16)2.0f * ((_Float16)x(k) - (_Float16)y(k));
        sum_m += (float)m;
        sum_dx += (float)hmul(-t1, m);
    }
    const _Float16 m_d = (_Float16)2.0f * ((_Float16)x(d) - (_Float16)y(d));
    const _Float16 dx_d = hmul(-t1, m_d), div_x = (_Float16)sum_dx;
    auto to_S = [&](_Float16 w) { return hmul(hmul(hmul(w, inv16), kap), inv16); };
    const _Float16 gS_dt = to_S(m_d), gS_div = to_S((_Float16)sum_m);
    float s_dtdiv = 0.0f, s_divdiv = 0.0f;
    for (int k = 0; k < d; ++k) {
        const _Float16 m = (_Float16)2.0f * ((_Float16)x(k) - (_Float16)y(k));
        s_dtdiv += (float)(-hmul(gS_dt, m));
        s_divdiv += (float)(_Float16)((float)two_t1 - (float)hmul(gS_div, m));
    }
    P[0][0] = (double)kap;
    P[2][0] = (double)dx_d;
    P[0][2] = -(double)dx_d;
    P[3][0] = (double)div_x;
    P[0][3] = -(double)div_x;
    P[2][2] = (double)(_Float16)((float)two_t1 - (float)hmul(gS_dt, m_d));
    P[2][3] = (double)(_Float16)s_dtdiv;
    P[3][2] = -(double)hmul(gS_div, m_d);
    P[3][3] = (double)(_Float16)s_divdiv;
}
// (round16 bit 3, exploratory: lap_y kappa and lap_x kappa through the same float16 sequence -- the mean over the five drawn indices of the Hessian
// diagonal of kappa in the SHIFTED argument, H_i = f16(f16(gS_i m_i) - 2 t1), gS_i from w = m_i; jnp.mean accumulates in float32 and rounds once,
// the product with the weakly typed d is float16.  The logs do not decide for it: profiles/HISTORY.md section 2.)
template <class FX, class FY>
__device__ __forceinline__ double f16_graph_hutchinson(int d, double a, const CompatIdx &ix, FX x, FY y) {   // x, y: the geometry's rows, already shifted
    const _Float16 c16 = (_Float16)(2.0 / a);
    const _Float16 inv16 = (_Float16)(1.0f / (float)c16);
    float Sf = 0.0f;
    for (int k = 0; k <= d; ++k) {
        const _Float16 r = (_Float16)x(k) - (_Float16)y(k);
        Sf += (float)(r * r);
    }
    const _Float16 kap = (_Float16)(float)exp((double)hmul(-(_Float16)Sf, inv16));
    const _Float16 t1 = hmul(kap, inv16), two_t1 = (_Float16)2.0f * t1;
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < kMC; ++j) {
        const int i = ix.i[j];
        const _Float16 m = (_Float16)2.0f * ((_Float16)x(i) - (_Float16)y(i));
        const _Float16 gS = hmul(hmul(hmul(m, inv16), kap), inv16);
        acc += (float)(_Float16)((float)hmul(gS, m) - (float)two_t1);
    }
#ifdef SCASML_F16_HUTCH_FUSED      // development: mean * d without the intermediate float16 rounding
    return (double)(_Float16)(acc / (float)kMC * (float)d);
#else
    const _Float16 mean = (_Float16)(acc / (float)kMC);
    return (double)(_Float16)((float)mean * (float)d);
#endif
}

template <class FX>
__device__ __forceinline__ bool row_is_f16(int d, FX x) {
    bool ok = true;
    for (int k = 0; k <= d; ++k) ok = ok && (double)(float)(_Float16)x(k) == (double)x(k);
    return ok;
}

// ---------------------------------------------------------------------------------- Gram (models/GP.py:182-258)
__global__ void gp_gram_compat_kernel(int d, double a, const float *x_dom, int n_dom, const float *x_bdy, int n_bdy,
                                      CompatIdx ix, int r16, double *K) {
    const int N = n_dom + n_bdy;
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || j >= N) return;
    const int64_t M = 4 * (int64_t)n_dom + n_bdy;
    const float *xi = i < n_dom ? x_dom + (int64_t)i * (d + 1) : x_bdy + (int64_t)(i - n_dom) * (d + 1);
    const float *yj = j < n_dom ? x_dom + (int64_t)j * (d + 1) : x_bdy + (int64_t)(j - n_dom) * (d + 1);
    PairGeom g;
    pair_geometry(d, a, ix, [&](int k) { return (double)xi[k]; }, [&](int k) { return (double)yj[k]; }, g);
    double P[4][4];
    compat_blocks(d, a, g, r16 & 1, P);
    if (r16 & 4) f16_graph_blocks(d, a, [&](int k) { return xi[k]; }, [&](int k) { return yj[k]; }, P);   // the caller vouches for float16 rows
    if (r16 & 8) {
        const int D = d + 1;
        P[0][1] = f16_graph_hutchinson(d, a, ix, [&](int k) { return xi[k]; }, [&](int k) { return yj[(k + 1) % D]; });   // ys: r = x - y'
        P[1][0] = f16_graph_hutchinson(d, a, ix, [&](int k) { return xi[(k + 1) % D]; }, [&](int k) { return yj[k]; });   // xs: r = x' - y
    }
    const int nops_i = i < n_dom ? 4 : 1, nops_j = j < n_dom ? 4 : 1;
    for (int ox = 0; ox < nops_i; ++ox) {
        const int64_t row = ox == 0 ? i : (int64_t)n_dom + n_bdy + (int64_t)(ox - 1) * n_dom + i;
        for (int oy = 0; oy < nops_j; ++oy) {
            const int64_t col = oy == 0 ? j : (int64_t)n_dom + n_bdy + (int64_t)(oy - 1) * n_dom + j;
            K[row * M + col] = P[ox][oy];
        }
    }
}

// Rows of the same matrix for the block-row distributed fit (scasml_gp_gram_compat_rows; BASELINE configs[4]): the feature rows of ONE
// operator `ox` at the points [i_lo, i_lo + n_i) against every collocation point j, columns < ncols only (the distributed factor stores
// the lower triangle).  Same per-pair arithmetic as gp_gram_compat_kernel (pair_geometry + compat_blocks): the rows are bit-identical.
__global__ void gp_gram_compat_rows_kernel(int d, double a, const float *x_dom, int n_dom, const float *x_bdy, int n_bdy, CompatIdx ix, int r16,
                                           int ox, int i_lo, int n_i, int64_t ncols, double *out, int64_t ld) {
    const int N = n_dom + n_bdy;
    const int ii = blockIdx.y * blockDim.y + threadIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (ii >= n_i || j >= N || (int64_t)j >= ncols) return;       // column j of the u block is this pair's smallest column
    const int i = i_lo + ii;
    const float *xi = i < n_dom ? x_dom + (int64_t)i * (d + 1) : x_bdy + (int64_t)(i - n_dom) * (d + 1);
    const float *yj = j < n_dom ? x_dom + (int64_t)j * (d + 1) : x_bdy + (int64_t)(j - n_dom) * (d + 1);
    PairGeom g;
    pair_geometry(d, a, ix, [&](int k) { return (double)xi[k]; }, [&](int k) { return (double)yj[k]; }, g);
    double P[4][4];
    compat_blocks(d, a, g, r16 & 1, P);
    if (r16 & 4) f16_graph_blocks(d, a, [&](int k) { return xi[k]; }, [&](int k) { return yj[k]; }, P);
    if (r16 & 8) {
        const int D = d + 1;
        P[0][1] = f16_graph_hutchinson(d, a, ix, [&](int k) { return xi[k]; }, [&](int k) { return yj[(k + 1) % D]; });
        P[1][0] = f16_graph_hutchinson(d, a, ix, [&](int k) { return xi[(k + 1) % D]; }, [&](int k) { return yj[k]; });
    }
    const int nops_j = j < n_dom ? 4 : 1;
    for (int oy = 0; oy < nops_j; ++oy) {
        const int64_t col = oy == 0 ? j : (int64_t)N + (int64_t)(oy - 1) * n_dom + j;
        if (col < ncols) out[(int64_t)ii * ld + col] = P[ox][oy];
    }
}

// ---------------------------------------------------------------------------------- cross-kernel rows (models/GP.py:271-411, 630-651)
// The feature rows of ONE operator `ox` (0 = I, 1 = Lap, 2 = dt, 3 = div, applied in x) at ARBITRARY points x_inf against every collocation
// point: kernel_x_t_phi (ox = 0), laplacian_x_t_ / dt_x_t_ / div_x_t_kernel_x_t_phi -- the matrices the reference materialises for predict and
// compute_PDE_loss (the hot path contracts them on the fly: gp_eval_compat_kernel, gp_eval_compat_mfma.hip).  Columns in the Gram's order
// [u(dom), u(bdy), Lap(dom), dt(dom), div(dom)].  surrogate 0: the reference's code (pair_geometry + compat_blocks, as the Gram rows); 1: the
// operators it documents (exact Laplacian, no rounding).
__device__ __forceinline__ void exact_blocks(int d, double a, double kap, double rho2, double S, double rt, double (&P)[4][4]) {
    const double a2 = a * a, lap = a2 * rho2 - a * d;
    P[0][0] = kap;
    P[0][1] = P[1][0] = lap * kap;
    P[0][2] = a * rt * kap;
    P[2][0] = -a * rt * kap;
    P[0][3] = a * S * kap;
    P[3][0] = -a * S * kap;
    P[2][2] = (a - a2 * rt * rt) * kap;
    P[2][3] = P[3][2] = -a2 * rt * S * kap;
    P[3][3] = (a * d - a2 * S * S) * kap;
    P[2][1] = -a * rt * lap * kap;
    P[1][2] = a * rt * lap * kap;
    P[3][1] = -(a * S * lap - 2.0 * a2 * S) * kap;
    P[1][3] = (a * S * lap - 2.0 * a2 * S) * kap;
    P[1][1] = (a2 * a2 * rho2 * rho2 - (2.0 * d + 4.0) * a2 * a * rho2 + ((double)d * d + 2.0 * d) * a2) * kap;
}

__global__ void gp_cross_rows_kernel(int d, double a, const float *x_dom, int n_dom, const float *x_bdy, int n_bdy, CompatIdx ix, int r16, int surrogate,
                                     int ox, const float *x_inf, int64_t n_inf, int64_t ld_inf, double *out, int64_t ld) {
    const int N = n_dom + n_bdy;
    const int64_t i = (int64_t)blockIdx.y * blockDim.y + threadIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inf || j >= N) return;
    const float *xi = x_inf + i * ld_inf;
    const float *yj = j < n_dom ? x_dom + (int64_t)j * (d + 1) : x_bdy + (int64_t)(j - n_dom) * (d + 1);
    double P[4][4];
    if (surrogate == 1) {
        double r2 = 0.0, S = 0.0;
        for (int k = 0; k < d; ++k) {
            const double r = (double)xi[k] - (double)yj[k];
            r2 = fma(r, r, r2);
            S += r;
        }
        const double rt = (double)xi[d] - (double)yj[d];
        exact_blocks(d, a, exp(-0.5 * a * fma(rt, rt, r2)), r2, S, rt, P);
    } else {
        PairGeom g;
        pair_geometry(d, a, ix, [&](int k) { return (double)xi[k]; }, [&](int k) { return (double)yj[k]; }, g);
        compat_blocks(d, a, g, r16 & 1, P);
        if ((r16 & 4) && row_is_f16(d, [&](int k) { return (double)xi[k]; })) {      // float16 rows on float16 collocation points (the caller vouches for those)
            f16_graph_blocks(d, a, [&](int k) { return xi[k]; }, [&](int k) { return yj[k]; }, P);
            if (r16 & 8) {
                const int D = d + 1;
                P[0][1] = f16_graph_hutchinson(d, a, ix, [&](int k) { return xi[k]; }, [&](int k) { return yj[(k + 1) % D]; });
                P[1][0] = f16_graph_hutchinson(d, a, ix, [&](int k) { return xi[(k + 1) % D]; }, [&](int k) { return yj[k]; });
            }
        }
    }
    const int nops_j = j < n_dom ? 4 : 1;
    for (int oy = 0; oy < nops_j; ++oy) {
        const int64_t col = oy == 0 ? j : (int64_t)N + (int64_t)(oy - 1) * n_dom + j;
        out[i * ld + col] = P[ox][oy];
    }
}

// dx_t_kernel_x_t_phi (models/GP.py:296-324): the gradient in x of the FIRST feature row -- out[i][col][k] = d/dx_k of (kappa, lap_y kappa,
// dt_y kappa, div_y kappa)(x_i, y_j), k <= d with the time derivative last; autodiff passes through the entries' float16 casts, so these are the
// derivatives of the un-rounded entries, each rounded once (.astype(float16), :324).  As coded, with r = x - y, r1 = x - y' and
// sg1 = sum_j (a^2 r1_{i_j}^2 - a):  d kappa = -a r_k kappa0;  d(dt_y) = (a [k = d] - a^2 r_t r_k) kappa0;  d(div_y) = (a [k < d] - a^2 S r_k) kappa0;
// d(lap_y) = h kappa1 r1_k (2 a^2 [k in idx] - a sg1).  Documented: d(lap_y) = (2 a^2 r_k [k < d] - a r_k (a^2 rho^2 - a d)) kappa0.
__global__ void gp_cross_grad_rows_kernel(int d, double a, const float *x_dom, int n_dom, const float *x_bdy, int n_bdy, CompatIdx ix, int r16,
                                          int surrogate, const float *x_inf, int64_t n_inf, int64_t ld_inf, double *out, int64_t M) {
    const int N = n_dom + n_bdy, D = d + 1;
    const int64_t i = (int64_t)blockIdx.y * blockDim.y + threadIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inf || j >= N) return;
    const float *xi = x_inf + i * ld_inf;
    const float *yj = j < n_dom ? x_dom + (int64_t)j * D : x_bdy + (int64_t)(j - n_dom) * D;
    double r2 = 0.0, S = 0.0, r2s = 0.0;
    for (int k = 0; k < D; ++k) {
        const double r = (double)xi[k] - (double)yj[k], r1 = (double)xi[k] - (double)yj[(k + 1) % D];
        r2 = fma(r, r, r2);
        r2s = fma(r1, r1, r2s);
        if (k < d) S += r;
    }
    const double rt = (double)xi[d] - (double)yj[d];
    const double kap0 = exp(-0.5 * a * r2), kap1 = exp(-0.5 * a * r2s), a2 = a * a, h = (double)d / kMC;
    double sg1 = 0.0;
    for (int q = 0; q < kMC; ++q) {
        const double r1 = (double)xi[ix.i[q]] - (double)yj[ix.i[q] + 1];
        sg1 += a2 * r1 * r1 - a;
    }
    const double lap = a2 * (r2 - rt * rt) - a * d;
    const int on = surrogate == 1 ? 0 : (r16 & 1);
    double *o = out + (i * M + j) * D;
    for (int k = 0; k < D; ++k) o[k] = round16(-a * ((double)xi[k] - (double)yj[k]) * kap0, on);
    if (j >= n_dom) return;
    double *oL = out + (i * M + N + j) * D, *ot = out + (i * M + N + n_dom + j) * D, *oS = out + (i * M + N + 2 * (int64_t)n_dom + j) * D;
    for (int k = 0; k < D; ++k) {
        const double r = (double)xi[k] - (double)yj[k];
        ot[k] = round16(((k == d ? a : 0.0) - a2 * rt * r) * kap0, on);
        oS[k] = round16(((k < d ? a : 0.0) - a2 * S * r) * kap0, on);
        if (surrogate == 1) {
            oL[k] = ((k < d ? 2.0 * a2 * r : 0.0) - a * r * lap) * kap0;
        } else {
            bool in_idx = false;
            for (int q = 0; q < kMC; ++q) in_idx |= ix.i[q] == k;
            const double r1 = (double)xi[k] - (double)yj[(k + 1) % D];
            oL[k] = round16(h * kap1 * r1 * ((in_idx ? 2.0 * a2 : 0.0) - a * sg1), on);
        }
    }
}

// diagonal of K + nugget I rounded to float16 (kernel_phi_phi_perturb.astype(float16), models/GP.py:268): the entries of K are
// float16 values already, so only the diagonal moves
__global__ void round16_diag_kernel(double *A, int64_t M, int64_t lda, double nugget) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) A[i * lda + i] = round16(A[i * lda + i] + nugget, 1);
}

__global__ void round16_vec_kernel(double *v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = round16(v[i], 1);
}

// ---------------------------------------------------------------------------------- evaluation
// One wavefront per evaluation point; lanes stride the collocation points (transposed float64 copy, so a lane's reads
// of coordinate k are contiguous across lanes); the point's coordinates sit in LDS.  Outputs as scasml_gp_eval:
// (u_hat, div_x u_hat, eps_PDE, dt u_hat) and optionally the "Laplacian".
// Feature rows (models/GP.py:326-411, 630-651): with (c0, cL, ct, cS) the right_vector entries of the u / Lap / dt / div
// features of a domain point and c0 that of a boundary point,
//   L^x u_hat = sum_j c0 P[x][0] + cL P[x][1] + ct P[x][2] + cS P[x][3],   x in {I, dt, div, lap}.
__global__ __launch_bounds__(256) void gp_eval_compat_kernel(int d, double a, double sigma, double mu, int eq_id, const double *colloc_t, int n_dom,
                                                             int n_bdy, int64_t ldc, const double *rv, CompatIdx ix, int r16,
                                                             const float *points, int64_t n_inf, int kp, float4 *out4,
                                                             float *lap_out) {
    extern __shared__ double xs_all[];   // 4 waves x (d + 1)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int D = d + 1;
    int64_t row = (int64_t)blockIdx.x * 4 + wv;
    const bool valid = row < n_inf;
    if (!valid) row = n_inf - 1;
    double *xs = xs_all + wv * D;
    for (int k = lane; k < D; k += 64) xs[k] = (double)points[row * kp + k];
    __syncthreads();
    const int N = n_dom + n_bdy;
    // round16 bit 2: on a float16 evaluation row (the collocation points are float16: the caller vouches) the Laplacian-free entries follow the
    // reference's float16 op sequence (f16_graph_blocks)
    const bool graph = (r16 & 4) && row_is_f16(d, [&](int k) { return xs[k]; });
    double acc[4] = {0.0, 0.0, 0.0, 0.0};   // I, lap, dt, div applied in x
    for (int j = lane; j < N; j += 64) {
        PairGeom g;
        pair_geometry(d, a, ix, [&](int k) { return xs[k]; }, [&](int k) { return colloc_t[(int64_t)k * ldc + j]; }, g);
        double P[4][4];
        compat_blocks(d, a, g, r16 & 1, P);
        if (graph) f16_graph_blocks(d, a, [&](int k) { return xs[k]; }, [&](int k) { return colloc_t[(int64_t)k * ldc + j]; }, P);
        if (graph && (r16 & 8)) {
            P[0][1] = f16_graph_hutchinson(d, a, ix, [&](int k) { return xs[k]; }, [&](int k) { return colloc_t[(int64_t)((k + 1) % D) * ldc + j]; });
            P[1][0] = f16_graph_hutchinson(d, a, ix, [&](int k) { return xs[(k + 1) % D]; }, [&](int k) { return colloc_t[(int64_t)k * ldc + j]; });
        }
        const double c0 = rv[j];
        double cL = 0.0, ct = 0.0, cS = 0.0;
        if (j < n_dom) {
            cL = rv[(int64_t)N + j];
            ct = rv[(int64_t)N + n_dom + j];
            cS = rv[(int64_t)N + 2 * n_dom + j];
        }
#pragma unroll
        for (int ox = 0; ox < 4; ++ox) acc[ox] += c0 * P[ox][0] + cL * P[ox][1] + ct * P[ox][2] + cS * P[ox][3];
    }
#pragma unroll
    for (int ox = 0; ox < 4; ++ox)
        for (int o = 32; o > 0; o >>= 1) acc[ox] += __shfl_xor(acc[ox], o);
    if (valid && lane == 0) {
        const double s2 = sigma * sigma;
        const double u = round16(acc[0], r16 & 2), lp = acc[1], dt = acc[2], dv = acc[3];                              // predict(...).astype(float16), models/GP.py:671
        const double eps = round16(dt + mu * dv + 0.5 * s2 * lp + eq_f<double>(eq_id, u, sigma * dv, sigma, (double)d), r16 & 2);   // :767-769
        out4[row] = make_float4((float)u, (float)dv, (float)eps, (float)dt);
        if (lap_out) lap_out[row] = (float)lp;
    }
}

// ---------------------------------------------------------------------------------- gradient (models/GP.py:673-687)
// compute_gradient differentiates  u_hat(x) = dot(kernel_x_t_phi_single(x), right_vector)  by autodiff: through the float16 casts
// (identity for the derivative) and through the shifted Hutchinson feature lap_y kappa(x, y') = h sg1 kappa1.  With r = x - y
// (aligned), r1 = x - y' (y shifted), E0 = c0 + ct a r_t + cS a S:
//   d u_hat / d x_i = sum_j  -a r_i kappa0 E0 + a kappa0 (ct [i = d] + cS [i < d])  +  cL h kappa1 r1_i (2 a^2 [i in idx] - a sg1).
// One wavefront per point: lanes stride the collocation points for the pair scalars, then the coordinates for the d+1 sums.
__global__ __launch_bounds__(256) void gp_gradient_compat_kernel(int d, double a, const double *colloc_t, int n_dom, int n_bdy, int64_t ldc,
                                                                 const double *rv, CompatIdx ix, int r16, const float *points, int64_t n_inf,
                                                                 int kp, float *grad) {
    extern __shared__ double sh_all[];   // per wave: (d + 1) coordinates + 3 x 64 pair scalars
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int D = d + 1, N = n_dom + n_bdy;
    int64_t row = (int64_t)blockIdx.x * 4 + wv;
    const bool valid = row < n_inf;
    if (!valid) row = n_inf - 1;
    double *xs = sh_all + wv * (D + 192);
    double *pa = xs + D, *pd = pa + 64, *pe = pd + 64;
    for (int k = lane; k < D; k += 64) xs[k] = (double)points[row * kp + k];
    __syncthreads();
    const double h = (double)d / kMC, a2 = a * a;
    double g[4] = {0.0, 0.0, 0.0, 0.0};          // coordinates lane, lane + 64, ...
    double sA = 0.0, sB = 0.0, sC = 0.0, sD = 0.0, sE = 0.0;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        double al = 0.0, dl = 0.0, el = 0.0;
        if (j < N) {
            double r2 = 0.0, r2s = 0.0, S = 0.0;
            for (int k = 0; k < D; ++k) {
                const double yk = colloc_t[(int64_t)k * ldc + j], yn = colloc_t[(int64_t)(k + 1 < D ? k + 1 : 0) * ldc + j];
                const double r = xs[k] - yk, rs = xs[k] - yn;
                r2 = fma(r, r, r2);
                r2s = fma(rs, rs, r2s);
                if (k < d) S += r;
            }
            const double rt = xs[d] - colloc_t[(int64_t)d * ldc + j];
            const double kap0 = exp(-0.5 * a * r2), kap1 = exp(-0.5 * a * r2s);
            const double c0 = rv[j];
            double cL = 0.0, ct = 0.0, cS = 0.0;
            if (j < n_dom) {
                cL = rv[(int64_t)N + j];
                ct = rv[(int64_t)N + n_dom + j];
                cS = rv[(int64_t)N + 2 * n_dom + j];
            }
            double sg1 = 0.0;
#pragma unroll
            for (int q = 0; q < kMC; ++q) {
                const double r = xs[ix.i[q]] - colloc_t[(int64_t)(ix.i[q] + 1) * ldc + j];
                sg1 += a2 * r * r - a;
            }
            const double E0 = c0 + ct * a * rt + cS * a * S;
            al = -a * kap0 * E0;
            dl = -a * h * cL * kap1 * sg1;
            el = 2.0 * a2 * h * cL * kap1;
            sA += al;
            sB += a * kap0 * ct;
            sC += a * kap0 * cS;
            sD += dl;
            sE += el;
        }
        pa[lane] = al;
        pd[lane] = dl;
        pe[lane] = el;
        __syncthreads();
        const int jn = N - j0 < 64 ? N - j0 : 64;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = lane + 64 * c;
            if (i < D) {
                bool in_idx = false;
#pragma unroll
                for (int q = 0; q < kMC; ++q) in_idx |= ix.i[q] == i;
                const double *yi = colloc_t + (int64_t)i * ldc + j0, *ys = colloc_t + (int64_t)(i + 1 < D ? i + 1 : 0) * ldc + j0;
                double acc = 0.0;
                for (int jj = 0; jj < jn; ++jj) acc -= pa[jj] * yi[jj] + (pd[jj] + (in_idx ? pe[jj] : 0.0)) * ys[jj];
                g[c] += acc;
            }
        }
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) {
        sA += __shfl_xor(sA, o);
        sB += __shfl_xor(sB, o);
        sC += __shfl_xor(sC, o);
        sD += __shfl_xor(sD, o);
        sE += __shfl_xor(sE, o);
    }
    if (!valid) return;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int i = lane + 64 * c;
        if (i < D) {
            bool in_idx = false;
#pragma unroll
            for (int q = 0; q < kMC; ++q) in_idx |= ix.i[q] == i;
            const double v = g[c] + xs[i] * (sA + sD + (in_idx ? sE : 0.0)) + (i < d ? sC : sB);
            grad[row * D + i] = (float)round16(v, r16);
        }
    }
}

__global__ void transpose_colloc_kernel(const float *x_dom, int n_dom, const float *x_bdy, int n_bdy, int d, int64_t ldc, double *out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    const int N = n_dom + n_bdy;
    if (j >= N) return;
    const float *src = j < n_dom ? x_dom + (int64_t)j * (d + 1) : x_bdy + (int64_t)(j - n_dom) * (d + 1);
    out[(int64_t)k * ldc + j] = (double)src[k];
}

static int check_idx(const int32_t *idx_h, int d, CompatIdx &ix, const char *who) {
    if (!idx_h) return fail(SCASML_ERR_ARG, "%s: idx is null", who);
    for (int j = 0; j < kMC; ++j) {
        if (idx_h[j] < 0 || idx_h[j] >= d) return fail(SCASML_ERR_ARG, "%s: idx[%d] = %d outside [0, d)", who, j, idx_h[j]);
        for (int q = 0; q < j; ++q)
            if (idx_h[q] == idx_h[j]) return fail(SCASML_ERR_ARG, "%s: idx has a repeated entry %d", who, idx_h[j]);
        ix.i[j] = idx_h[j];
    }
    return 0;
}

}  // namespace scasml

using namespace scasml;

extern "C" int scasml_gp_gram_compat(int32_t d, double a, const float *x_dom, int32_t n_dom, const float *x_bdy, int32_t n_bdy,
                                     const int32_t *idx_h, int32_t round16, double *K, void *stream) {
    if (!x_dom || !K || (n_bdy > 0 && !x_bdy)) return fail(SCASML_ERR_ARG, "gp_gram_compat: null argument");
    if (d < kMC || n_dom < 1 || n_bdy < 0) return fail(SCASML_ERR_ARG, "gp_gram_compat: bad sizes (d >= %d needed)", kMC);
    CompatIdx ix;
    if (int rc = check_idx(idx_h, d, ix, "gp_gram_compat")) return rc;
    const int N = n_dom + n_bdy;
    hipLaunchKernelGGL(gp_gram_compat_kernel, dim3((N + 15) / 16, (N + 15) / 16), dim3(16, 16), 0, (hipStream_t)stream, d, a,
                       x_dom, n_dom, x_bdy, n_bdy, ix, round16, K);
    return check_launch("gp_gram_compat launch");
}

extern "C" int scasml_gp_gram_compat_rows(int32_t d, double a, const float *x_dom, int32_t n_dom, const float *x_bdy, int32_t n_bdy,
                                          const int32_t *idx_h, int32_t round16, int64_t row0, int32_t nrows, int64_t ncols, double *out, int64_t ld,
                                          void *stream) {
    if (!x_dom || !out || (n_bdy > 0 && !x_bdy)) return fail(SCASML_ERR_ARG, "gp_gram_compat_rows: null argument");
    const int64_t M = 4 * (int64_t)n_dom + n_bdy;
    if (d < kMC || n_dom < 1 || n_bdy < 0 || row0 < 0 || nrows < 0 || row0 + nrows > M || ncols < 0 || ncols > M || ld < ncols)
        return fail(SCASML_ERR_ARG, "gp_gram_compat_rows: bad sizes (d >= %d needed)", kMC);
    CompatIdx ix;
    if (int rc = check_idx(idx_h, d, ix, "gp_gram_compat_rows")) return rc;
    if (nrows == 0 || ncols == 0) return 0;
    const int N = n_dom + n_bdy;
    int64_t r = row0;
    const int64_t r_end = row0 + nrows;
    while (r < r_end) {                         // one launch per operator segment of the row range (as scasml_gp_gram_rows)
        int ox, i_lo;
        int64_t seg_end;
        if (r < N) {
            ox = 0;
            i_lo = (int)r;
            seg_end = N;
        } else {
            const int64_t q = r - N;
            ox = 1 + (int)(q / n_dom);
            i_lo = (int)(q % n_dom);
            seg_end = (int64_t)N + (int64_t)ox * n_dom;
        }
        const int64_t stop = seg_end < r_end ? seg_end : r_end;
        const int n_i = (int)(stop - r);
        const unsigned gy = (unsigned)((n_i + 15) / 16);
        if (gy > 65535) return fail(SCASML_ERR_UNSUPPORTED, "gp_gram_compat_rows: too many rows per call");
        hipLaunchKernelGGL(gp_gram_compat_rows_kernel, dim3((N + 15) / 16, gy), dim3(16, 16), 0, (hipStream_t)stream, d, a, x_dom, n_dom, x_bdy, n_bdy,
                           ix, round16, ox, i_lo, n_i, ncols, out + (r - row0) * ld, ld);
        r = stop;
    }
    return check_launch("gp_gram_compat_rows launch");
}

extern "C" int scasml_gp_cross_rows(int32_t d, double a, const float *x_dom, int32_t n_dom, const float *x_bdy, int32_t n_bdy, const int32_t *idx_h,
                                    int32_t round16, int32_t surrogate, int32_t op, const float *x_inf, int64_t n_inf, int64_t ld_inf, double *out,
                                    int64_t ld, void *stream) {
    if (!x_dom || !out || !x_inf || (n_bdy > 0 && !x_bdy)) return fail(SCASML_ERR_ARG, "gp_cross_rows: null argument");
    const int64_t M = 4 * (int64_t)n_dom + n_bdy;
    if (d < 1 || d > SCASML_MAX_DIM || n_dom < 1 || n_bdy < 0 || n_inf < 0 || ld_inf < d + 1) return fail(SCASML_ERR_ARG, "gp_cross_rows: bad sizes");
    if (surrogate != 0 && surrogate != 1) return fail(SCASML_ERR_ARG, "gp_cross_rows: surrogate must be 0 (as coded) or 1 (documented operators)");
    if (op < 0 || op > 4) return fail(SCASML_ERR_ARG, "gp_cross_rows: op must be 0 (I), 1 (Lap), 2 (dt), 3 (div) or 4 (gradient of the I row)");
    if (op < 4 && ld < M) return fail(SCASML_ERR_ARG, "gp_cross_rows: ld %lld < M = %lld", (long long)ld, (long long)M);
    CompatIdx ix = {{0, 0, 0, 0, 0}};
    if (surrogate == 0) {
        if (d < kMC) return fail(SCASML_ERR_ARG, "gp_cross_rows: the as-coded surrogate needs d >= %d", kMC);
        if (int rc = check_idx(idx_h, d, ix, "gp_cross_rows")) return rc;
    }
    if (n_inf == 0) return 0;
    const int N = n_dom + n_bdy;
    const int64_t gy = (n_inf + 15) / 16;
    if (gy > 65535) return fail(SCASML_ERR_UNSUPPORTED, "gp_cross_rows: at most %d rows per call", 65535 * 16);
    if (op == 4)
        hipLaunchKernelGGL(gp_cross_grad_rows_kernel, dim3((N + 15) / 16, (unsigned)gy), dim3(16, 16), 0, (hipStream_t)stream, d, a, x_dom, n_dom, x_bdy,
                           n_bdy, ix, round16, surrogate, x_inf, n_inf, ld_inf, out, M);
    else
        hipLaunchKernelGGL(gp_cross_rows_kernel, dim3((N + 15) / 16, (unsigned)gy), dim3(16, 16), 0, (hipStream_t)stream, d, a, x_dom, n_dom, x_bdy, n_bdy,
                           ix, round16, surrogate, op, x_inf, n_inf, ld_inf, out, ld);
    return check_launch("gp_cross_rows launch");
}

extern "C" int scasml_round16_diag(double *A, int64_t M, int64_t lda, double nugget, void *stream) {
    if (!A || M < 1 || lda < M) return fail(SCASML_ERR_ARG, "round16_diag: bad argument");
    hipLaunchKernelGGL(round16_diag_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A, M, lda, nugget);
    return check_launch("round16_diag launch");
}

extern "C" int scasml_round16(double *v, int64_t n, void *stream) {
    if (!v || n < 0) return fail(SCASML_ERR_ARG, "round16: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(round16_vec_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, n);
    return check_launch("round16 launch");
}

extern "C" int scasml_gp_compat_pack(int32_t d, const float *x_dom, int32_t n_dom, const float *x_bdy, int32_t n_bdy,
                                     double *colloc_t, int64_t ldc, void *stream) {
    if (!x_dom || !colloc_t || (n_bdy > 0 && !x_bdy) || d < 1 || n_dom < 1 || n_bdy < 0 || ldc < n_dom + n_bdy)
        return fail(SCASML_ERR_ARG, "gp_compat_pack: bad argument");
    const int N = n_dom + n_bdy;
    hipLaunchKernelGGL(transpose_colloc_kernel, dim3((N + 255) / 256, d + 1), dim3(256), 0, (hipStream_t)stream, x_dom, n_dom,
                       x_bdy, n_bdy, d, ldc, colloc_t);
    return check_launch("gp_compat_pack launch");
}

extern "C" int scasml_gp_eval_compat(int32_t d, double a, double sigma_eq, double mu_eq, int32_t eq_id, const double *colloc_t, int32_t n_dom, int32_t n_bdy,
                                     int64_t ldc, const double *rv, const int32_t *idx_h, int32_t round16, const float *points,
                                     int64_t n_inf, int32_t kp, float *out4, float *lap, void *stream) {
    if (n_inf == 0) return 0;
    if (!colloc_t || !rv || !points || !out4 || n_inf < 0) return fail(SCASML_ERR_ARG, "gp_eval_compat: bad argument");
    if (d < kMC || d > SCASML_MAX_DIM || n_dom < 1 || n_bdy < 0 || ldc < n_dom + n_bdy || kp < d + 1)
        return fail(SCASML_ERR_ARG, "gp_eval_compat: bad sizes");
    if (!eq_known(eq_id)) return fail(SCASML_ERR_UNSUPPORTED, "gp_eval_compat: unknown equation id %d", eq_id);
    CompatIdx ix;
    if (int rc = check_idx(idx_h, d, ix, "gp_eval_compat")) return rc;
    const int64_t blocks = (n_inf + 3) / 4;
    if (blocks > 0x7FFFFFFF) return fail(SCASML_ERR_UNSUPPORTED, "gp_eval_compat: too many points");
    const size_t lds = 4 * (size_t)(d + 1) * sizeof(double);
    hipLaunchKernelGGL(gp_eval_compat_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, d, a, sigma_eq, mu_eq, eq_id, colloc_t,
                       n_dom, n_bdy, ldc, rv, ix, round16, points, n_inf, kp, reinterpret_cast<float4 *>(out4), lap);
    return check_launch("gp_eval_compat launch");
}

extern "C" int scasml_gp_gradient_compat(int32_t d, double a, const double *colloc_t, int32_t n_dom, int32_t n_bdy, int64_t ldc, const double *rv,
                                         const int32_t *idx_h, int32_t round16, const float *points, int64_t n_inf, int32_t kp, float *grad,
                                         void *stream) {
    if (n_inf == 0) return 0;
    if (!colloc_t || !rv || !points || !grad || n_inf < 0) return fail(SCASML_ERR_ARG, "gp_gradient_compat: bad argument");
    if (d < kMC || d > SCASML_MAX_DIM || n_dom < 1 || n_bdy < 0 || ldc < n_dom + n_bdy || kp < d + 1)
        return fail(SCASML_ERR_ARG, "gp_gradient_compat: bad sizes");
    CompatIdx ix;
    if (int rc = check_idx(idx_h, d, ix, "gp_gradient_compat")) return rc;
    const int64_t blocks = (n_inf + 3) / 4;
    if (blocks > 0x7FFFFFFF) return fail(SCASML_ERR_UNSUPPORTED, "gp_gradient_compat: too many points");
    const size_t lds = 4 * (size_t)(d + 1 + 192) * sizeof(double);
    hipLaunchKernelGGL(gp_gradient_compat_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, d, a, colloc_t, n_dom, n_bdy, ldc, rv,
                       ix, round16, points, n_inf, kp, grad);
    return check_launch("gp_gradient_compat launch");
}
