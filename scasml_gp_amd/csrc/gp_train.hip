// Training side of the PDE-constrained GP (models/GP.py:182-268, 430-444, 487-604): the
// 25-block derivative-feature Gram matrix in closed form, a blocked right-looking Cholesky of
// K + nugget*I (trailing updates on the FP64 matrix cores, v_mfma_f64_16x16x4_f64) (the reference factors by SVD, models/GP.py:260-267; for symmetric PSD K the
// product L L^T is the same matrix) and blocked triangular solves.  Everything here is float64,
// as the reference's x64 SVD is.  Matrix order M must be a multiple of 32; the host pads with an
// identity block (chol([[A,0],[0,I]]) = [[L,0],[0,I]]).
//
// Block order of K (models/GP.py:251-258): [u(dom), u(bdy), Lap(dom), dt(dom), div(dom)].
#include <stdlib.h>

#include "common.hpp"
#include "equations.hpp"
#include "f64_tile_dma.hpp"

namespace scasml {

constexpr int NB = 32;

// ---------------------------------------------------------------------------------- Cholesky
// (1) factor the NB x NB diagonal block in LDS.  One barrier per column: the thread that owns the NEXT pivot finishes it (its own
// update, the square root and the reciprocal) inside the current column's step, so a column needs no separate pivot and scaling
// phases (three barriers per column before; M = 4224: 5.4 -> 5.0 ms; the block is one link of the factorisation's serial chain).
__global__ __launch_bounds__(NB *NB) void chol_diag_kernel(double *A, int64_t M, int64_t k0, int32_t *info) {
    __shared__ double T[NB][NB + 1], Lk[NB][NB + 1];
    __shared__ double sd[2], pinv[2];   // sqrt(pivot) and its reciprocal, ping-pong over columns
    const int r = threadIdx.y, c = threadIdx.x;
    T[r][c] = A[(k0 + r) * M + (k0 + c)];
    __syncthreads();
    if (r == 0 && c == 0) {
        const double v = T[0][0];
        if (!(v > 0.0)) {
            if (*info == 0) *info = (int32_t)(k0 + 1);
            sd[0] = pinv[0] = nan("");
        } else {
            sd[0] = sqrt(v);
            pinv[0] = 1.0 / sd[0];
        }
    }
    __syncthreads();
    for (int j = 0; j < NB; ++j) {
        const double is = pinv[j & 1];
        if (c == j) {
            Lk[r][j] = r == j ? sd[j & 1] : (r > j ? T[r][j] * is : 0.0);
        } else if (c > j && r >= c) {
            const double v = fma(-(T[r][j] * is), T[c][j] * is, T[r][c]);
            T[r][c] = v;
            if (r == j + 1 && c == j + 1) {
                if (!(v > 0.0)) {
                    if (*info == 0) *info = (int32_t)(k0 + j + 2);
                    sd[(j + 1) & 1] = pinv[(j + 1) & 1] = nan("");
                } else {
                    const double sq = sqrt(v);
                    sd[(j + 1) & 1] = sq;
                    pinv[(j + 1) & 1] = 1.0 / sq;
                }
            }
        }
        __syncthreads();
    }
    A[(k0 + r) * M + (k0 + c)] = Lk[r][c];
}

// (2) panel: rows below the diagonal block, X * L_kk^T = A_panel  (one thread per row)
__global__ __launch_bounds__(256) void chol_panel_kernel(double *A, int64_t M, int64_t k0) {
    __shared__ double Lk[NB][NB + 1];
    for (int idx = threadIdx.x; idx < NB * NB; idx += blockDim.x) Lk[idx / NB][idx % NB] = A[(k0 + idx / NB) * M + k0 + idx % NB];
    __syncthreads();
    const int64_t r = k0 + NB + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    double x[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) x[j] = A[r * M + k0 + j];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        double v = x[j];
#pragma unroll
        for (int p = 0; p < j; ++p) v = fma(-x[p], Lk[j][p], v);
        x[j] = v / Lk[j][j];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) A[r * M + k0 + j] = x[j];
}

// ---- C(TB x TB) -= A(TB x K) * B(K x TB) on the FP64 matrix cores ------------------------------------
// v_mfma_f64_16x16x4_f64: lane l supplies A[row = l&15][k = l>>4] and B[k = l>>4][col = l&15]; the four
// results of a lane are C[row = (l>>4) + 4*i][col = l&15], i = 0..3 (cdna_hip_programming.md section 3: the
// f64 C/D map differs from the f32 one).  Four waves per workgroup, each owning a quadrant of NT x NT MFMA tiles:
// NT = 2 -> 64 x 64 output tile (the one in use: the K = 256 updates of large matrices are L2-bandwidth-bound at
// 6.4 flop/B with it, 32 TFLOP/s; NT = 4 -> 128 x 128 doubles the flops per operand byte but needs 438 VGPRs, one wave
// per SIMD, and measured 23 TFLOP/s).  Operands come from LDS panels
// [TB][LDP] (A rows / B^T rows), double-buffered: the global loads of chunk k+1 are in flight while chunk k feeds
// the matrix cores.
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int TB = 64;           // output tile edge of the NT = 2 kernels (grid geometry of the callers)
constexpr int LDP = NB + 2;      // padded leading dimension: 2 (row LDP + k) mod 64 is distinct over a ds_read_b64 group (16 rows x 2 k)
constexpr int64_t kOuterRows = 8 * NB;   // outer block of the two-level factorisation / substitutions
constexpr int64_t kOuterBigRows = 16 * NB;   // outer panel of scasml_cholesky on matrices large enough for the look-ahead

template <int NT>
struct TileAcc {
    f64x4 v[NT][NT];
};
// WS = waves per tile side: a workgroup is WS x WS waves (64 WS^2 threads), each owning NT x NT MFMA tiles of 16 x 16, i.e. an
// output tile of (16 NT WS)^2.  (NT, WS) = (2, 2): 64 x 64, 4 waves, 8 flop per operand byte; (2, 4): 128 x 128, 16 waves --
// the same registers per wave, half the operand traffic per flop and four waves per SIMD to hide it (139 KB of LDS: one
// workgroup per CU).  The large trailing updates use (2, 4), everything small (2, 2).
template <int NT, int WS = 2>
constexpr size_t tile_lds_bytes() { return (size_t)2 * 2 * (16 * NT * WS) * LDP * sizeof(double); }   // 2 operands x 2 buffers

template <int NT>
__device__ __forceinline__ void mfma_tile_zero(TileAcc<NT> &t) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) t.v[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
}

// t += As (TB x NB) * Bt (TB x NB)^T
template <int NT, int WS = 2>
__device__ __forceinline__ void mfma_tile_accumulate(const double (*As)[LDP], const double (*Bt)[LDP], TileAcc<NT> &t) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wr = (wv / WS) * (16 * NT), wc = (wv % WS) * (16 * NT);   // this wave's origin inside the tile
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int k0 = 0; k0 < NB; k0 += 4) {
        double a[NT], b[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) a[i] = As[wr + 16 * i + l15][k0 + l4];
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = Bt[wc + 16 * j + l15][k0 + l4];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) t.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], t.v[i][j], 0, 0, 0);
    }
}

// Output-tile addressing: ONE 32-bit byte offset per thread -- its place (row wr + l4, column wc + l15) inside the tile; tile rows < 128 and
// ldc < 2^21, so below 2^32 bytes -- against a scalar base per (i, j, e) fragment element: the accesses take the SGPR-base form and no 64-bit
// address lives in a VGPR (sixteen of them, hoisted out of the tile loop, spilled in the streamed kernel).
template <int NT, int WS = 2>
__device__ __forceinline__ uint32_t tile_thread_offset(int64_t ldc) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t wr = (wv / WS) * (16 * NT), wc = (wv % WS) * (16 * NT);
    return ((wr + (lane >> 4)) * (uint32_t)ldc + wc + (lane & 15)) * 8u;
}
__device__ __forceinline__ double *tile_element(double *C, int64_t ldc, int i, int j, int e, uint32_t toff) {
    return reinterpret_cast<double *>(reinterpret_cast<char *>(C + (int64_t)(16 * i + 4 * e) * ldc + 16 * j) + toff);
}

// C -= t (rows x cols valid)
template <int NT, int WS = 2>
__device__ __forceinline__ void mfma_tile_subtract(const TileAcc<NT> &t, double *C, int64_t ldc, int64_t rows, int64_t cols) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wr = (wv / WS) * (16 * NT), wc = (wv % WS) * (16 * NT);
    const int l15 = lane & 15, l4 = lane >> 4;
    const uint32_t toff = tile_thread_offset<NT, WS>(ldc);
    constexpr int TBX = 16 * NT * WS;
    const int nr = rows < TBX ? (int)rows : TBX, nc = cols < TBX ? (int)cols : TBX;   // 32-bit tests: sixteen 64-bit row indices per lane spilled
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = wr + 16 * i + l4 + 4 * e, c = wc + 16 * j + l15;
                if (r < nr && c < nc) *tile_element(C, ldc, i, j, e, toff) -= t.v[i][j][e];
            }
}

// The same in two halves for a tile that lies wholly inside the matrix (no bounds tests, so the 4 NT^2 loads of a lane are issued
// back to back instead of one round trip at a time behind a branch each): the loads are issued before the LAST chunk of the K loop
// and fly under its matrix work, the tail is a subtraction and a store.  Measured at M = 35 008: the guarded read-modify-write
// tail was 88 ms of a 391 ms factorisation (13 us per 128 x 128 tile against 34 us of matrix work).
template <int NT, int WS = 2>
__device__ __forceinline__ void mfma_tile_load_full(TileAcc<NT> &c, double *C, int64_t ldc) {
    const uint32_t toff = tile_thread_offset<NT, WS>(ldc);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) c.v[i][j][e] = *tile_element(C, ldc, i, j, e, toff);
}
template <int NT, int WS = 2>
__device__ __forceinline__ void mfma_tile_store_diff_full(const TileAcc<NT> &c, const TileAcc<NT> &t, double *C, int64_t ldc) {
    const uint32_t toff = tile_thread_offset<NT, WS>(ldc);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) *tile_element(C, ldc, i, j, e, toff) = c.v[i][j][e] - t.v[i][j][e];
}

// The K loop shared by the Cholesky and triangular-solve updates: `fetch(kk, ra, rb)` loads this thread's elements
// of the two operand chunks of columns [kk, kk + NB) into registers (element e of a thread is panel entry
// idx = tid + 256 e, row idx / NB, column idx % NB).
// Which chunk entry (tile row rr, chunk column cc) is a thread's element e: consecutive lanes walk the direction that is contiguous
// in memory -- chunk columns for a row-major panel, tile rows for an operand stored k-major (KMAJOR: entry (rr, cc) at P[cc * ld + rr];
// with the row-major mapping there every lane of a load touched a cache line of its own).
template <int TBX, int THREADS, bool KMAJOR>
__device__ __forceinline__ void chunk_entry(int e, int &rr, int &cc) {
    const int idx = threadIdx.x + e * THREADS;
    if (KMAJOR) {
        rr = idx % TBX;
        cc = idx / TBX;
    } else {
        rr = idx / NB;
        cc = idx % NB;
    }
}

template <int NT, int WS = 2, bool KMAJOR_A = false, bool KMAJOR_B = false, class Fetch, class PreTail>
__device__ __forceinline__ void mfma_tile_k_loop(double *smem, int64_t k_begin, int64_t k_end, Fetch fetch, PreTail pre_tail, TileAcc<NT> &t) {
    constexpr int TBX = 16 * NT * WS, THREADS = 64 * WS * WS, PER = TBX * NB / THREADS;
    double (*Pa)[TBX][LDP] = reinterpret_cast<double (*)[TBX][LDP]>(smem);
    double (*Pb)[TBX][LDP] = reinterpret_cast<double (*)[TBX][LDP]>(smem + 2 * TBX * LDP);
    double ra[PER], rb[PER];
    auto park = [&](int buf) {
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            int rr, cc;
            chunk_entry<TBX, THREADS, KMAJOR_A>(e, rr, cc);
            Pa[buf][rr][cc] = ra[e];
            chunk_entry<TBX, THREADS, KMAJOR_B>(e, rr, cc);
            Pb[buf][rr][cc] = rb[e];
        }
    };
    fetch(k_begin, ra, rb);
    park(0);
    __syncthreads();
    int cur = 0;
    for (int64_t kk = k_begin; kk + NB < k_end; kk += NB) {
        fetch(kk + NB, ra, rb);
        mfma_tile_accumulate<NT, WS>(Pa[cur], Pb[cur], t);
        park(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    pre_tail();   // the caller's loads of the output tile: in flight under the last chunk
    mfma_tile_accumulate<NT, WS>(Pa[cur], Pb[cur], t);
}

// ---------------------------------------------------------------------------------- Gram on the FP64 matrix cores
// K(phi, phi) (models/GP.py:182-258): the only O(N^2 d) part of a pair's 16 entries is x_i . y_j, so the pair geometry of a
// 64 x 64 tile of collocation pairs is ONE tile of the GEMM X Y^T (v_mfma_f64_16x16x4_f64, K = d + 1 streamed through LDS as
// in the Cholesky update) and everything else is the epilogue of SURVEY.md K5:  |x - y|^2 = |x|^2 + |y|^2 - 2 x.y,
// r_t = t_x - t_y, S = sum x - sum y (per-point statistics, computed once per tile into LDS), kappa = exp(-a r^2 / 2) and the 16
// operator polynomials of Appendix C, written straight into their blocks.  float64 throughout: the reference factors this
// matrix in x64 (models/GP.py:258) and the norm expansion loses ~1e-15 of r^2.
__global__ __launch_bounds__(256) void gp_gram_mfma_kernel(int d, double a, const float *x_dom, int n_dom, const float *x_bdy, int n_bdy,
                                                           double *K) {
    constexpr int NT = 2, TBX = 32 * NT, PER = TBX * NB / 256;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double stat[2 * TBX][4];          // (|x|^2 with t, sum of the spatial coordinates, t) of the tile's 64 + 64 points
    const int N = n_dom + n_bdy;
    const int64_t M = 4 * (int64_t)n_dom + n_bdy;
    const int i0 = blockIdx.y * TBX, j0 = blockIdx.x * TBX;
    auto row_of = [&](int p) { return p < n_dom ? x_dom + (int64_t)p * (d + 1) : x_bdy + (int64_t)(p - n_dom) * (d + 1); };
    {   // two threads per point, even / odd coordinates, combined with one shuffle
        const int q = threadIdx.x >> 1, par = threadIdx.x & 1;
        const int p = q < TBX ? i0 + q : j0 + q - TBX;
        double n = 0.0, sx = 0.0, tt = 0.0;
        if (p < N) {
            const float *x = row_of(p);
            for (int k = par; k < d; k += 2) {
                const double v = (double)x[k];
                n = fma(v, v, n);
                sx += v;
            }
            tt = (double)x[d];
        }
        n += __shfl_xor(n, 1);
        sx += __shfl_xor(sx, 1);
        if (par == 0) {
            stat[q][0] = fma(tt, tt, n);
            stat[q][1] = sx;
            stat[q][2] = tt;
        }
    }   // visible to everyone after the barriers of the K loop below
    TileAcc<NT> t;
    mfma_tile_zero(t);
    const int64_t kend = ((int64_t)d + 1 + NB - 1) / NB * NB;
    mfma_tile_k_loop<NT>(smem, 0, kend, [&](int64_t kk, double (&ra)[PER], double (&rb)[PER]) {
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int idx = threadIdx.x + e * 256, rr = idx / NB, cc = idx % NB;
            const int k = (int)kk + cc;
            ra[e] = (i0 + rr < N && k <= d) ? (double)row_of(i0 + rr)[k] : 0.0;
            rb[e] = (j0 + rr < N && k <= d) ? (double)row_of(j0 + rr)[k] : 0.0;
        }
    }, [] {}, t);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wr = (wv >> 1) * (16 * NT), wc = (wv & 1) * (16 * NT);
    const int l15 = lane & 15, l4 = lane >> 4;
    const double fd = (double)d;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        const int j = j0 + wc + 16 * jt + l15;
        if (j >= N) continue;
        const double nj = stat[TBX + j - j0][0], sj = stat[TBX + j - j0][1], tj = stat[TBX + j - j0][2];
        const int nops_j = j < n_dom ? 4 : 1;
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + wr + 16 * it + l4 + 4 * e;
                if (i >= N) continue;
                const double ni = stat[i - i0][0], si = stat[i - i0][1], ti = stat[i - i0][2];
                double r2 = fma(-2.0, t.v[it][jt][e], ni + nj);
                r2 = r2 < 0.0 ? 0.0 : r2;
                const double rt = ti - tj, S = si - sj;
                double rho2 = fma(-rt, rt, r2);
                rho2 = rho2 < 0.0 ? 0.0 : rho2;
                const double kap = exp(-0.5 * a * r2);
                const double lap = a * a * rho2 - a * fd;
                const int nops_i = i < n_dom ? 4 : 1;
                // operator polynomials P(opx, opy); ops: 0 = I, 1 = Lap, 2 = dt, 3 = div (the table of gp_gram_rows_kernel)
                const double aS = a * S, art = a * rt, mix = aS * lap - 2.0 * a * aS;
                const double P[4][4] = {
                    {1.0, lap, art, aS},
                    {lap, a * a * a * a * rho2 * rho2 - (2.0 * fd + 4.0) * a * a * a * rho2 + (fd * fd + 2.0 * fd) * a * a, art * lap, mix},
                    {-art, -art * lap, a - art * art, -art * aS},
                    {-aS, -mix, -art * aS, a * fd - aS * aS}};
                for (int ox = 0; ox < nops_i; ++ox) {
                    const int64_t row = ox == 0 ? i : (int64_t)N + (int64_t)(ox - 1) * n_dom + i;
                    for (int oy = 0; oy < nops_j; ++oy) {
                        const int64_t col = oy == 0 ? j : (int64_t)N + (int64_t)(oy - 1) * n_dom + j;
                        K[row * M + col] = P[ox][oy] * kap;
                    }
                }
            }
    }
}

// The same tile for a RANGE OF FEATURE ROWS (block-row distributed fits, scasml_gp_gram_rows): the rows of one operator `ox` at the
// points [i_lo, i_lo + n_i), against every collocation point j; each pair writes its 1 or 4 entries of that row (columns < ncols
// only -- the distributed factor stores the lower triangle).  out points at the first of those rows; tiles whose smallest column
// (the u column of their first point) is already >= ncols have nothing to write.
__global__ __launch_bounds__(256) void gp_gram_rows_mfma_kernel(int d, double a, const float *x_dom, int n_dom, const float *x_bdy, int n_bdy,
                                                                int ox, int i_lo, int n_i, int64_t ncols, double *out, int64_t ld) {
    constexpr int NT = 2, TBX = 32 * NT, PER = TBX * NB / 256;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double stat[2 * TBX][4];
    const int N = n_dom + n_bdy;
    const int i0 = i_lo + blockIdx.y * TBX, j0 = blockIdx.x * TBX, i_end = i_lo + n_i;
    if ((int64_t)j0 >= ncols) return;             // block-uniform: every column of this tile lies beyond the stored part of these rows
    auto row_of = [&](int p) { return p < n_dom ? x_dom + (int64_t)p * (d + 1) : x_bdy + (int64_t)(p - n_dom) * (d + 1); };
    {
        const int q = threadIdx.x >> 1, par = threadIdx.x & 1;
        const int p = q < TBX ? i0 + q : j0 + q - TBX;
        const bool live = q < TBX ? p < i_end : p < N;
        double n = 0.0, sx = 0.0, tt = 0.0;
        if (live) {
            const float *x = row_of(p);
            for (int k = par; k < d; k += 2) {
                const double v = (double)x[k];
                n = fma(v, v, n);
                sx += v;
            }
            tt = (double)x[d];
        }
        n += __shfl_xor(n, 1);
        sx += __shfl_xor(sx, 1);
        if (par == 0) {
            stat[q][0] = fma(tt, tt, n);
            stat[q][1] = sx;
            stat[q][2] = tt;
        }
    }
    TileAcc<NT> t;
    mfma_tile_zero(t);
    const int64_t kend = ((int64_t)d + 1 + NB - 1) / NB * NB;
    mfma_tile_k_loop<NT>(smem, 0, kend, [&](int64_t kk, double (&ra)[PER], double (&rb)[PER]) {
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int idx = threadIdx.x + e * 256, rr = idx / NB, cc = idx % NB;
            const int k = (int)kk + cc;
            ra[e] = (i0 + rr < i_end && k <= d) ? (double)row_of(i0 + rr)[k] : 0.0;
            rb[e] = (j0 + rr < N && k <= d) ? (double)row_of(j0 + rr)[k] : 0.0;
        }
    }, [] {}, t);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wr = (wv >> 1) * (16 * NT), wc = (wv & 1) * (16 * NT);
    const int l15 = lane & 15, l4 = lane >> 4;
    const double fd = (double)d;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        const int j = j0 + wc + 16 * jt + l15;
        if (j >= N) continue;
        const double nj = stat[TBX + j - j0][0], sj = stat[TBX + j - j0][1], tj = stat[TBX + j - j0][2];
        const int nops_j = j < n_dom ? 4 : 1;
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + wr + 16 * it + l4 + 4 * e;
                if (i >= i_end) continue;
                const double ni = stat[i - i0][0], si = stat[i - i0][1], ti = stat[i - i0][2];
                double r2 = fma(-2.0, t.v[it][jt][e], ni + nj);
                r2 = r2 < 0.0 ? 0.0 : r2;
                const double rt = ti - tj, S = si - sj;
                double rho2 = fma(-rt, rt, r2);
                rho2 = rho2 < 0.0 ? 0.0 : rho2;
                const double kap = exp(-0.5 * a * r2);
                const double lap = a * a * rho2 - a * fd;
                const double aS = a * S, art = a * rt, mix = aS * lap - 2.0 * a * aS;
                // the operator table of gp_gram_mfma_kernel; this launch writes its row `ox`
                const double P[4][4] = {
                    {1.0, lap, art, aS},
                    {lap, a * a * a * a * rho2 * rho2 - (2.0 * fd + 4.0) * a * a * a * rho2 + (fd * fd + 2.0 * fd) * a * a, art * lap, mix},
                    {-art, -art * lap, a - art * art, -art * aS},
                    {-aS, -mix, -art * aS, a * fd - aS * aS}};
                double *orow = out + (int64_t)(i - i_lo) * ld;
                for (int oy = 0; oy < nops_j; ++oy) {
                    const int64_t col = oy == 0 ? j : (int64_t)N + (int64_t)(oy - 1) * n_dom + j;
                    if (col < ncols) orow[col] = P[ox][oy] * kap;
                }
            }
    }
}

// (3) trailing update with a panel of K columns [J, J + K), lower triangle only:
//   C[r][c] -= sum_k A[r][J + k] * A[c][J + k]   for rows r >= R0, columns R0 <= c < col_end,
// one tile per workgroup, the panel streamed through LDS NB columns at a time while the tile stays in the
// accumulators: C is read and written ONCE per K columns.  The factorisation below uses it twice: inside an outer
// panel (K = NB, columns of that panel only) and for the rest of the matrix once per outer panel (K = kOuter) --
// with K = NB everywhere the factorisation moves M^3 / (3 NB) * 16 bytes through HBM (7 TB at M = 35 000).
template <int NT, int WS>
__global__ __launch_bounds__(64 * WS * WS) void chol_update_k_kernel(double *A, int64_t M, int64_t J, int64_t K, int64_t R0, int64_t col_end) {
    constexpr int TBX = 16 * NT * WS, THREADS = 64 * WS * WS, PER = TBX * NB / THREADS;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int64_t ti = blockIdx.y, tj = blockIdx.x;
    if (WS == 4) {   // 1-D grid in super-tile order (host_common.hpp)
        const int64_t nt = (M - R0 + TBX - 1) / TBX;
        if (!super_tile_of_block(blockIdx.x, nt, nt, true, ti, tj)) return;
    }
    if (tj > ti) return;
    const int64_t r0 = R0 + ti * TBX, c0 = R0 + tj * TBX;
    if (r0 >= M || c0 >= col_end) return;
    const bool interior = r0 + TBX <= M && c0 + TBX <= col_end;   // block-uniform
    double *Ct = A + r0 * M + c0;
    TileAcc<NT> t, cin;
    mfma_tile_zero(t);
    mfma_tile_k_loop<NT, WS>(smem, J, J + K, [&](int64_t kk, double (&ra)[PER], double (&rb)[PER]) {
        if (interior) {
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                const int idx = threadIdx.x + e * THREADS, rr = idx / NB, cc = idx % NB;
                ra[e] = A[(r0 + rr) * M + kk + cc];
                rb[e] = A[(c0 + rr) * M + kk + cc];
            }
        } else {
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                const int idx = threadIdx.x + e * THREADS, rr = idx / NB, cc = idx % NB;
                ra[e] = r0 + rr < M ? A[(r0 + rr) * M + kk + cc] : 0.0;
                rb[e] = c0 + rr < M ? A[(c0 + rr) * M + kk + cc] : 0.0;
            }
        }
    }, [&] { if (interior) mfma_tile_load_full<NT, WS>(cin, Ct, M); }, t);
    if (interior) mfma_tile_store_diff_full<NT, WS>(cin, t, Ct, M);
    else mfma_tile_subtract<NT, WS>(t, Ct, M, M - r0, col_end - c0);
}

// The large trailing updates: the same 128 x 128 tile with its operand panels staged by LDS-DMA, three stages ahead (f64_tile_dma.hpp);
// results bit-identical to chol_update_k_kernel<2, 4> (same summation order), which stays for matrices whose base is not 16-byte aligned.
__global__ __launch_bounds__(kDmaThreads) void chol_update_dma_kernel(double *A, int64_t M, int64_t J, int64_t K, int64_t R0, int64_t col_end) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int64_t nt = (M - R0 + kDmaTile - 1) / kDmaTile;
    int64_t ti, tj;
    if (!super_tile_of_block(blockIdx.x, nt, nt, true, ti, tj) || tj > ti) return;
    const int64_t r0 = R0 + ti * kDmaTile, c0 = R0 + tj * kDmaTile;
    if (r0 >= M || c0 >= col_end) return;
    const int ra = (int)(M - r0 < kDmaTile ? M - r0 : kDmaTile), rb = (int)(col_end - c0 < kDmaTile ? col_end - c0 : kDmaTile);
    f64_tile_dma(smem, A + r0 * M + J, M, ra, A + c0 * M + J, M, rb, K, A + r0 * M + c0, M);
}

// ---------------------------------------------------------------------------------- TRSM
// diagonal-block solve: one thread per right-hand-side column
template <int TRANS>
__global__ __launch_bounds__(256) void trsm_diag_kernel(const double *L, int64_t M, double *B, int64_t nrhs, int64_t k0,
                                                        int64_t col_limit) {
    __shared__ double Lk[NB][NB + 1];
    for (int idx = threadIdx.x; idx < NB * NB; idx += blockDim.x) Lk[idx / NB][idx % NB] = L[(k0 + idx / NB) * M + k0 + idx % NB];
    __syncthreads();
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= col_limit) return;   // col_limit <= nrhs: columns beyond it are known zero / not needed
    double x[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) x[j] = B[(k0 + j) * nrhs + col];
    if (TRANS == 0) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            double v = x[j];
#pragma unroll
            for (int p = 0; p < j; ++p) v = fma(-Lk[j][p], x[p], v);
            x[j] = v / Lk[j][j];
        }
    } else {
#pragma unroll
        for (int j = NB - 1; j >= 0; --j) {
            double v = x[j];
#pragma unroll
            for (int p = j + 1; p < NB; ++p) v = fma(-Lk[p][j], x[p], v);
            x[j] = v / Lk[j][j];
        }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) B[(k0 + j) * nrhs + col] = x[j];
}

// off-diagonal update with K freshly solved rows X[J : J + K, :], 64x64 tile per workgroup, the K dimension streamed
// through LDS NB at a time while the tile stays in the accumulators (B is read and written once per K rows):
//   TRANS == 0:  B[r, :] -= L[r, J:J+K] * X[J:J+K, :]          for rows rbase <= r < rend  (rows below the solved ones)
//   TRANS == 1:  B[r, :] -= L[J:J+K, r]^T * X[J:J+K, :]        for rows rbase <= r < rend  (rows above them)
// tri != 0: only tiles on or below the block diagonal (c0 < r0 + TB) are updated -- the lower triangle of a symmetric result
template <int TRANS, int NT, int WS>
__global__ __launch_bounds__(64 * WS * WS) void trsm_update_kernel(const double *L, int64_t M, double *B, int64_t nrhs, int64_t J, int64_t K,
                                                          int64_t rbase, int64_t rend, int tri, int64_t ntc) {
    constexpr int TBX = 16 * NT * WS, THREADS = 64 * WS * WS, PER = TBX * NB / THREADS;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int64_t ti = blockIdx.y, tj = blockIdx.x;
    if (WS == 4 && !super_tile_of_block(blockIdx.x, (rend - rbase + TBX - 1) / TBX, ntc, false, ti, tj)) return;   // 1-D grid, super-tile order
    const int64_t r0 = rbase + ti * TBX, c0 = tj * TBX;
    if (r0 >= rend || c0 >= nrhs || (tri && c0 >= r0 + TBX)) return;   // block-uniform
    const bool interior = r0 + TBX <= rend && c0 + TBX <= nrhs;   // block-uniform
    double *Ct = B + r0 * nrhs + c0;
    TileAcc<NT> t, cin;
    mfma_tile_zero(t);
    // the right-hand side is always read k-major (Xt[col][k] = B[k][col]); L is k-major in the transposed solve
    mfma_tile_k_loop<NT, WS, TRANS == 1, true>(smem, J, J + K, [&](int64_t kk, double (&rl)[PER], double (&rx)[PER]) {
        if (interior) {
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                int rr, cc;
                chunk_entry<TBX, THREADS, TRANS == 1>(e, rr, cc);
                rl[e] = TRANS == 0 ? L[(r0 + rr) * M + kk + cc] : L[(kk + cc) * M + r0 + rr];
                chunk_entry<TBX, THREADS, true>(e, rr, cc);
                rx[e] = B[(kk + cc) * nrhs + c0 + rr];
            }
        } else {
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                int rr, cc;
                chunk_entry<TBX, THREADS, TRANS == 1>(e, rr, cc);
                rl[e] = r0 + rr < rend ? (TRANS == 0 ? L[(r0 + rr) * M + kk + cc] : L[(kk + cc) * M + r0 + rr]) : 0.0;
                chunk_entry<TBX, THREADS, true>(e, rr, cc);
                rx[e] = c0 + rr < nrhs ? B[(kk + cc) * nrhs + c0 + rr] : 0.0;
            }
        }
    }, [&] { if (interior) mfma_tile_load_full<NT, WS>(cin, Ct, nrhs); }, t);
    if (interior) mfma_tile_store_diff_full<NT, WS>(cin, t, Ct, nrhs);
    else mfma_tile_subtract<NT, WS>(t, Ct, nrhs, rend - r0, nrhs - c0);
}

// ---------------------------------------------------------------------------------- Newton system
// Unknowns sol = [z1, z3, z5] (values of u, Lap u, div u at the N domain points); feature vector
// b(sol) = [z1, g, z3, F(sol), z5] with F = -s2 z1 z5 + (1/d + s2/2) z5 - (s2/2) z3 (models/GP.py:430-444, 705-719).
struct EqArgs {
    int eq_id;
    double d, sigma, mu;
};

__global__ void gp_newton_b_kernel(EqArgs q, const double *sol, const double *bdy_g, int N, int Nb, double *b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int M = 4 * N + Nb;
    if (i >= M) return;
    double v;
    if (i < N) v = sol[i];
    else if (i < N + Nb) v = bdy_g[i - N];
    else if (i < 2 * N + Nb) v = sol[N + (i - N - Nb)];
    else if (i < 3 * N + Nb) {
        const int k = i - 2 * N - Nb;
        v = eq_F(q.eq_id, sol[k], sol[N + k], sol[2 * N + k], q.mu, q.sigma, q.d).F;
    } else v = sol[2 * N + (i - 3 * N - Nb)];
    b[i] = v;
}

// y = A x for a dense row-major M x M float64 matrix: one wavefront per row
__global__ __launch_bounds__(256) void gemv_kernel(const double *A, int64_t M, int64_t lda, const double *x, double *y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    double acc = 0.0;
    for (int64_t k = lane; k < M; k += 64) acc = fma(A[row * lda + k], x[k], acc);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) y[row] = acc;
}

// Gradient and full Hessian of J(sol) = b^T A b (A = K_p^-1 symmetric), written into a zero-padded
// (ldh x ldh) buffer whose padding is the identity (ready for scasml_cholesky):
//   dF/dz1 = -s2 z5, dF/dz3 = -s2/2, dF/dz5 = -s2 z1 + (1/d + s2/2)  (cf. models/GP.py:722-743)
//   grad_i = 2 (Ab[r_i] + dF_i Ab[r4_i]),
//   H_ij   = 2 (A[r_i,r_j] + dF_i A[r4_i,r_j] + A[r_i,r4_j] dF_j + dF_i A[r4_i,r4_j] dF_j)
//            + 2 (-s2) Ab[r4_i] on the (z1_i, z5_i) / (z5_i, z1_i) pairs          (second derivative of F).
__global__ void gp_newton_system_kernel(EqArgs q, const double *A, int64_t lda, int N, int Nb, const double *sol,
                                        const double *Ab, double *grad, double *H, int64_t ldh, int gauss_newton) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = (int64_t)blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= ldh || j >= ldh) return;
    const int n3 = 3 * N;
    if (i >= n3 || j >= n3) {
        H[i * ldh + j] = i == j ? 1.0 : 0.0;
        return;
    }
    const int bi = (int)(i / N), ii = (int)(i % N), bj = (int)(j / N), jj = (int)(j % N);
    const int64_t off[3] = {0, (int64_t)N + Nb, (int64_t)3 * N + Nb};
    const int64_t ri = off[bi] + ii, rj = off[bj] + jj, r4i = 2 * (int64_t)N + Nb + ii, r4j = 2 * (int64_t)N + Nb + jj;
    const FOp Fi = eq_F(q.eq_id, sol[ii], sol[N + ii], sol[2 * N + ii], q.mu, q.sigma, q.d);
    const FOp Fj = eq_F(q.eq_id, sol[jj], sol[N + jj], sol[2 * N + jj], q.mu, q.sigma, q.d);
    const double dFi = bi == 0 ? Fi.d1 : (bi == 1 ? Fi.d3 : Fi.d5);
    const double dFj = bj == 0 ? Fj.d1 : (bj == 1 ? Fj.d3 : Fj.d5);
    double h = 2.0 * (A[ri * lda + rj] + dFi * A[r4i * lda + rj] + A[ri * lda + r4j] * dFj + dFi * A[r4i * lda + r4j] * dFj);
    if (!gauss_newton && ii == jj && bi != 1 && bj != 1)     // the Hessian of F_i in (z1_i, z5_i), weighted by 2 (K_p^-1 b)_{F_i}
        h += 2.0 * (bi == 0 && bj == 0 ? Fi.F11 : (bi == 2 && bj == 2 ? Fi.F55 : Fi.F15)) * Ab[r4i];
    H[i * ldh + j] = h;
    if (j == 0) grad[i] = 2.0 * (Ab[ri] + dFi * Ab[r4i]);
}

// J v and J^T w for the matrix-free Newton iteration (J = d b / d sol; b = [z1, g, z3, F, z5])
__global__ void gp_newton_jv_kernel(EqArgs q, const double *sol, const double *v, int N, int Nb, double *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int M = 4 * N + Nb;
    if (i >= M) return;
    double r;
    if (i < N) r = v[i];
    else if (i < N + Nb) r = 0.0;
    else if (i < 2 * N + Nb) r = v[N + (i - N - Nb)];
    else if (i < 3 * N + Nb) {
        const int k = i - 2 * N - Nb;
        const FOp F = eq_F(q.eq_id, sol[k], sol[N + k], sol[2 * N + k], q.mu, q.sigma, q.d);
        r = F.d1 * v[k] + F.d3 * v[N + k] + F.d5 * v[2 * N + k];
    } else r = v[2 * N + (i - 3 * N - Nb)];
    out[i] = r;
}

__global__ void gp_newton_jtv_kernel(EqArgs q, const double *sol, const double *w, const double *Ab, const double *v,
                                     double scale, int N, int Nb, double *out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const FOp F = eq_F(q.eq_id, sol[k], sol[N + k], sol[2 * N + k], q.mu, q.sigma, q.d);
    const double w4 = w[2 * N + Nb + k];
    double o1 = w[k] + F.d1 * w4;
    double o3 = w[N + Nb + k] + F.d3 * w4;
    double o5 = w[3 * N + Nb + k] + F.d5 * w4;
    if (Ab && v) {
        const double ab = Ab[2 * N + Nb + k];
        o1 += ab * (F.F11 * v[k] + F.F15 * v[2 * N + k]);
        o5 += ab * (F.F15 * v[k] + F.F55 * v[2 * N + k]);
    }
    out[k] = scale * o1;
    out[N + k] = scale * o3;
    out[2 * N + k] = scale * o5;
}

__global__ void add_diag_kernel(double *A, int64_t M, double v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) A[i * M + i] += v;
}

// whole-matrix passes: grid.y strides over the rows (a grid dimension other than x holds at most 65535 workgroups; M = 70 016 has more rows)
constexpr unsigned kRowGrid = 32768;
static unsigned row_grid(int64_t M) { return (unsigned)(M < (int64_t)kRowGrid ? M : (int64_t)kRowGrid); }

__global__ void set_identity_kernel(double *A, int64_t M) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= M) return;
    for (int64_t r = blockIdx.y; r < M; r += gridDim.y) A[r * M + c] = r == c ? 1.0 : 0.0;
}

__global__ void mirror_lower_kernel(double *A, int64_t M) {   // A[r][c] = A[c][r] for c > r
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= M) return;
    for (int64_t r = blockIdx.y; r < c; r += gridDim.y) A[r * M + c] = A[c * M + r];
}

__global__ void zero_upper_kernel(double *A, int64_t M) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= M) return;
    for (int64_t r = blockIdx.y; r < c; r += gridDim.y) A[r * M + c] = 0.0;
}

}  // namespace scasml

using namespace scasml;

// The large substitution updates on the LDS-DMA tile (f64_tile_dma.hpp): the right-hand sides (and, in the transposed solve, the factor) are
// k-major operands.  Results bit-identical to trsm_update_kernel<TRANS, 2, 4> (same summation order), which stays as the fallback.
template <int TRANS>
__global__ __launch_bounds__(kDmaThreads) void trsm_update_dma_kernel(const double *L, int64_t M, double *B, int64_t nrhs, int64_t J, int64_t K, int64_t rbase,
                                                                      int64_t rend, int tri, int64_t ntc) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int64_t ti, tj;
    if (!super_tile_of_block(blockIdx.x, (rend - rbase + kDmaTile - 1) / kDmaTile, ntc, false, ti, tj)) return;
    const int64_t r0 = rbase + ti * kDmaTile, c0 = tj * kDmaTile;
    if (r0 >= rend || c0 >= nrhs || (tri && c0 >= r0 + kDmaTile)) return;   // block-uniform
    const int ra = (int)(rend - r0 < kDmaTile ? rend - r0 : kDmaTile), rb = (int)(nrhs - c0 < kDmaTile ? nrhs - c0 : kDmaTile);
    const double *At = TRANS == 0 ? L + r0 * M + J : L + J * M + r0;
    f64_tile_dma<kDmaNB, kDmaStages, TRANS == 1, true>(smem, At, M, ra, B + J * nrhs + c0, nrhs, rb, K, B + r0 * nrhs + c0, nrhs);
}

// ---- launch helpers ------------------------------------------------------------------------------------------
template <class Kern>
static bool reserve_lds(Kern kern, size_t bytes) {
    return bytes <= 64 * 1024 ||
           hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

// tiles at least this many on both sides: the 128 x 128 tile (one workgroup per CU) still fills the chip
constexpr int64_t kBigTileRows = 4096;
// matrices at least this large factor with a one-panel lookahead on a second stream (scasml_cholesky)
constexpr int64_t kLookaheadRows = 8192;

template <int NT, int WS>
static void launch_chol_update_ws(double *A, int64_t M, int64_t J, int64_t K, int64_t R0, int64_t col_end, hipStream_t s) {
    constexpr int TBX = 16 * NT * WS;
    const int64_t nti = (M - R0 + TBX - 1) / TBX, ntj = (col_end - R0 + TBX - 1) / TBX;
    if (nti <= 0 || ntj <= 0) return;
    if (WS == 4 && K >= kDmaStages * kDmaNB && (uintptr_t)A % 16 == 0 && M < (1 << 21) && !getenv("SCASML_F64_TILE_REGISTER_STAGED")) {   // (development knob: A/B runs)
        if (!reserve_lds(chol_update_dma_kernel, kDmaLdsBytes)) return;   // reported by check_launch through hipGetLastError
        hipLaunchKernelGGL(chol_update_dma_kernel, dim3(super_tile_grid(nti, nti, true)), dim3(kDmaThreads), kDmaLdsBytes, s, A, M, J, K, R0, col_end);
        return;
    }
    auto kern = chol_update_k_kernel<NT, WS>;
    constexpr size_t lds = tile_lds_bytes<NT, WS>();
    if (!reserve_lds(kern, lds)) return;
    // WS == 4: the kernel enumerates the lower triangle of nti x nti tiles itself (columns beyond col_end return at once)
    const dim3 grid = WS == 4 ? dim3(super_tile_grid(nti, nti, true)) : dim3((unsigned)ntj, (unsigned)nti);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WS * WS), lds, s, A, M, J, K, R0, col_end);
}

template <int NT>
static void launch_chol_update(double *A, int64_t M, int64_t J, int64_t K, int64_t R0, int64_t col_end, hipStream_t s) {
    if (K >= 8 * NB && M - R0 >= kBigTileRows && col_end - R0 >= kBigTileRows) launch_chol_update_ws<NT, 4>(A, M, J, K, R0, col_end, s);
    else launch_chol_update_ws<NT, 2>(A, M, J, K, R0, col_end, s);
}

// rows [rbase, rend) x columns [0, ncols) of B
template <int TRANS, int NT, int WS>
static void launch_trsm_update_ws(const double *L, int64_t M, double *B, int64_t nrhs, int64_t J, int64_t K, int64_t rbase, int64_t rend,
                                  int64_t ncols, int tri, hipStream_t s) {
    constexpr int TBX = 16 * NT * WS;
    const int64_t ntr = (rend - rbase + TBX - 1) / TBX, ntc = (ncols + TBX - 1) / TBX;
    if (ntr <= 0 || ntc <= 0) return;
    if (WS == 4 && K >= kDmaStages * kDmaNB && ((uintptr_t)L | (uintptr_t)B) % 16 == 0 && M % 2 == 0 && nrhs % 2 == 0 && (rend - rbase) % 2 == 0 &&
        M < (1 << 21) && nrhs < (1 << 21) && !getenv("SCASML_F64_TILE_REGISTER_STAGED")) {
        if (!reserve_lds(trsm_update_dma_kernel<TRANS>, kDmaLdsBytes)) return;
        hipLaunchKernelGGL(trsm_update_dma_kernel<TRANS>, dim3(super_tile_grid(ntr, ntc, false)), dim3(kDmaThreads), kDmaLdsBytes, s, L, M, B, nrhs, J, K, rbase,
                           rend, tri, ntc);
        return;
    }
    auto kern = trsm_update_kernel<TRANS, NT, WS>;
    constexpr size_t lds = tile_lds_bytes<NT, WS>();
    if (!reserve_lds(kern, lds)) return;
    const dim3 grid = WS == 4 ? dim3(super_tile_grid(ntr, ntc, false)) : dim3((unsigned)ntc, (unsigned)ntr);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WS * WS), lds, s, L, M, B, nrhs, J, K, rbase, rend, tri, ntc);
}

template <int TRANS, int NT>
static void launch_trsm_update(const double *L, int64_t M, double *B, int64_t nrhs, int64_t J, int64_t K, int64_t rbase, int64_t rend,
                               int64_t ncols, int tri, hipStream_t s) {
    if (K >= 8 * NB && rend - rbase >= kBigTileRows && ncols >= kBigTileRows) launch_trsm_update_ws<TRANS, NT, 4>(L, M, B, nrhs, J, K, rbase, rend, ncols, tri, s);
    else launch_trsm_update_ws<TRANS, NT, 2>(L, M, B, nrhs, J, K, rbase, rend, ncols, tri, s);
}

template <int TRANS>
static void trsm_update(const double *L, int64_t M, double *B, int64_t nrhs, int64_t J, int64_t K, int64_t rbase, int64_t rend,
                        int64_t ncols, int tri, hipStream_t s) {
    launch_trsm_update<TRANS, 2>(L, M, B, nrhs, J, K, rbase, rend, ncols, tri, s);
}

// One-panel lookahead for the blocked factorisation and substitutions below.  A panel (kOuterRows columns or rows) is a chain of
// 2-3 x 8 short dependent kernels (0.37 ms in scasml_cholesky at M = 35 000: 50 ms over the factorisation with the chip idle).
// The chain of panel p+1 needs only that panel's own block updated, so it runs on a second stream while the caller's stream applies
// panel p to everything beyond panel p+1:
//   side:  chain(p) . [factored] . wait(applied p-1) . apply p to the block of panel p+1 . chain(p+1) ...
//   main:  wait(factored p) . apply p to everything beyond panel p+1 . [applied p] ...
// The two applications of a panel write disjoint blocks; the one on the side stream waits for the main stream because both
// subtract from panel p+1's block.  The stream and the two events live for one call (nothing survives it; re-entrant as before).
struct Lookahead {
    hipStream_t main = nullptr, side = nullptr;
    hipEvent_t factored = nullptr, applied = nullptr;
    bool open(hipStream_t s) {
        main = s;
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);   // (least, greatest): the short kernels of the chain go first
        if (hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi) != hipSuccess ||
            hipEventCreateWithFlags(&factored, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&applied, hipEventDisableTiming) != hipSuccess) {
            close_handles();
            return false;
        }
        hipEventRecord(applied, main);   // everything queued on the caller's stream so far
        hipStreamWaitEvent(side, applied, 0);
        return true;
    }
    void chain_done() { hipEventRecord(factored, side); }
    void side_waits_for_main() { hipStreamWaitEvent(side, applied, 0); }
    void main_waits_for_chain() { hipStreamWaitEvent(main, factored, 0); }
    void main_applied() { hipEventRecord(applied, main); }
    void join() {   // both streams' work before anything later on the caller's stream; also lets the side stream see it
        hipEventRecord(factored, side);
        hipStreamWaitEvent(main, factored, 0);
        hipEventRecord(applied, main);
        hipStreamWaitEvent(side, applied, 0);
    }
    void close() {
        hipEventRecord(factored, side);
        hipStreamWaitEvent(main, factored, 0);
        close_handles();   // released when the recorded work completes
    }
    void close_handles() {
        if (factored) hipEventDestroy(factored);
        if (applied) hipEventDestroy(applied);
        if (side) hipStreamDestroy(side);
        factored = applied = nullptr;
        side = nullptr;
    }
};

// Outer panel (factorisation) / row group (inverse) of a matrix of order M: kOuterRows below the look-ahead size, kOuterBigRows above it,
// twice that from kOuterHugeFrom rows on (measured, scasml_cholesky: M = 35 008 277 / 279 ms at 512 / 1024 columns, M = 70 016 2007 / 1942 ms).
constexpr int64_t kOuterHugeFrom = 49152;
static int64_t outer_rows(int64_t M) {
    static const int64_t forced = [] { const char *e = getenv("SCASML_CHOL_OUTER"); const int64_t v = e ? atoll(e) : 0; return v >= kOuterRows && v % kOuterRows == 0 ? v : 0; }();   // development
    if (M < kLookaheadRows) return kOuterRows;
    if (forced) return forced;
    return M >= kOuterHugeFrom ? 2 * kOuterBigRows : kOuterBigRows;
}

extern "C" int scasml_cholesky(double *A, int64_t M, double nugget, int32_t *info_dev, void *stream) {
    if (!A || !info_dev || M < 1) return fail(SCASML_ERR_ARG, "cholesky: bad argument");
    if (M % NB) return fail(SCASML_ERR_UNSUPPORTED, "cholesky: M=%lld is not a multiple of %d (pad with an identity block)", (long long)M, NB);
    if (M > 65535 * (int64_t)NB) return fail(SCASML_ERR_UNSUPPORTED, "cholesky: M too large for this build");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(info_dev, 0, sizeof(int32_t), s) != hipSuccess) return fail(SCASML_ERR_HIP, "cholesky: memset failed");
    if (nugget != 0.0) hipLaunchKernelGGL(add_diag_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, A, M, nugget);
    // two-level right-looking factorisation: columns are eliminated NB at a time inside an outer panel of kOuter
    // columns (updates confined to that panel's columns), then the whole trailing matrix is updated once with K = kOuter
    // Large matrices (the look-ahead path) take outer panels of kOuterBigRows columns, factored as sub-panels of kOuterRows with one K = kOuterRows
    // update between them: the trailing update's K doubles, i.e. the 256 KB read-modify-write of every 128 x 128 tile of C is paid half as often
    // (tools/gemm_bench.py, C of 16 384^2: 49 TFLOP/s at K = 256, 59 at K = 512, 61.5 at K = 1024).
    const int64_t kOuter = outer_rows(M);
    auto factor_panel = [&](int64_t J, int64_t jend, hipStream_t q) {   // columns [J, jend) final, all rows
        for (int64_t Js = J; Js < jend; Js += kOuterRows) {
            const int64_t Je = Js + kOuterRows < jend ? Js + kOuterRows : jend;
            for (int64_t k0 = Js; k0 < Je; k0 += NB) {
                // the diagonal block and the panel stay two launches: refactoring the block inside every panel workgroup (one launch fewer per
                // step) measured SLOWER, 6.1 against 4.9 ms at M = 4224 (profiles/r05_cholesky_fused_diag_panel.txt)
                hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(NB, NB), 0, q, A, M, k0, info_dev);
                const int64_t rest = M - k0 - NB;
                if (rest <= 0) break;
                hipLaunchKernelGGL(chol_panel_kernel, dim3((unsigned)((rest + 255) / 256)), dim3(256), 0, q, A, M, k0);
                if (k0 + NB < Je) launch_chol_update<2>(A, M, k0, NB, k0 + NB, Je, q);
            }
            if (Je < jend) launch_chol_update<2>(A, M, Js, Je - Js, Je, jend, q);   // sub-panel -> the rest of this outer panel's columns
        }
    };
    if (M < kLookaheadRows) {
        for (int64_t J = 0; J < M; J += kOuter) {
            const int64_t jend = J + kOuter < M ? J + kOuter : M;
            factor_panel(J, jend, s);
            if (jend < M) launch_chol_update<2>(A, M, J, jend - J, jend, M, s);
        }
    } else {
        Lookahead la;
        if (!la.open(s)) return fail(SCASML_ERR_HIP, "cholesky: cannot create the lookahead stream");
        for (int64_t J = 0; J < M; J += kOuter) {
            const int64_t jend = J + kOuter < M ? J + kOuter : M;
            const int64_t next_end = jend + kOuter < M ? jend + kOuter : M;
            factor_panel(J, jend, la.side);
            la.chain_done();
            if (jend < M) {
                la.side_waits_for_main();                                            // panel J-1 applied to panel J+1's columns
                launch_chol_update<2>(A, M, J, jend - J, jend, next_end, la.side);   // panel J -> columns of panel J+1 (rows >= jend)
                if (next_end < M) {
                    la.main_waits_for_chain();
                    launch_chol_update<2>(A, M, J, jend - J, next_end, M, s);        // panel J -> everything right of panel J+1
                    la.main_applied();
                }
            }
        }
        la.close();
    }
    hipLaunchKernelGGL(zero_upper_kernel, dim3((unsigned)((M + 255) / 256), row_grid(M)), dim3(256), 0, s, A, M);
    return check_launch("cholesky launch");
}

extern "C" int scasml_trsm_lower(const double *L, int64_t M, double *Bmat, int64_t nrhs, int trans, void *stream) {
    if (!L || !Bmat || M < 1 || nrhs < 0) return fail(SCASML_ERR_ARG, "trsm: bad argument");
    if (nrhs == 0) return 0;
    if (M % NB) return fail(SCASML_ERR_UNSUPPORTED, "trsm: M=%lld is not a multiple of %d", (long long)M, NB);
    if (M > 65535 * (int64_t)NB) return fail(SCASML_ERR_UNSUPPORTED, "trsm: M too large for this build");
    hipStream_t s = (hipStream_t)stream;
    const unsigned cb = (unsigned)((nrhs + 255) / 256), ct = (unsigned)((nrhs + TB - 1) / TB);
    // two-level blocked substitution (see scasml_cholesky): NB rows at a time inside a group of kOuterRows rows, then
    // one update of all remaining rows per group
    auto tiles = [](int64_t n) { return (unsigned)((n + TB - 1) / TB); };
    if (trans == 0) {
        for (int64_t J = 0; J < M; J += kOuterRows) {
            const int64_t jend = J + kOuterRows < M ? J + kOuterRows : M;
            for (int64_t k0 = J; k0 < jend; k0 += NB) {
                hipLaunchKernelGGL(trsm_diag_kernel<0>, dim3(cb), dim3(256), 0, s, L, M, Bmat, nrhs, k0, nrhs);
                if (k0 + NB < jend)
                    trsm_update<0>(L, M, Bmat, nrhs, k0, NB, k0 + NB, jend, nrhs, 0, s);
            }
            if (jend < M)
                trsm_update<0>(L, M, Bmat, nrhs, J, jend - J, jend, M, nrhs, 0, s);
        }
    } else {
        for (int64_t jend = M; jend > 0; jend -= kOuterRows) {
            const int64_t J = jend > kOuterRows ? jend - kOuterRows : 0;
            for (int64_t k0 = jend - NB; k0 >= J; k0 -= NB) {
                hipLaunchKernelGGL(trsm_diag_kernel<1>, dim3(cb), dim3(256), 0, s, L, M, Bmat, nrhs, k0, nrhs);
                if (k0 > J)
                    trsm_update<1>(L, M, Bmat, nrhs, k0, NB, J, k0, nrhs, 0, s);
            }
            if (J > 0)
                trsm_update<1>(L, M, Bmat, nrhs, J, jend - J, 0, J, nrhs, 0, s);
        }
    }
    return check_launch("trsm launch");
}

extern "C" int scasml_gp_newton_b(int32_t eq_id, int32_t d, double sigma, double mu, const double *sol, const double *bdy_g, int32_t n_dom,
                                  int32_t n_bdy, double *b, void *stream) {
    if (!eq_known(eq_id)) return fail(SCASML_ERR_UNSUPPORTED, "gp_newton_b: unknown equation id %d", eq_id);
    if (!sol || !b || n_dom < 1 || n_bdy < 0 || (n_bdy > 0 && !bdy_g)) return fail(SCASML_ERR_ARG, "gp_newton_b: bad argument");
    const int M = 4 * n_dom + n_bdy;
    hipLaunchKernelGGL(gp_newton_b_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, EqArgs{eq_id, (double)d, sigma, mu}, sol, bdy_g, n_dom, n_bdy, b);
    return check_launch("gp_newton_b launch");
}

extern "C" int scasml_gp_newton_jv(int32_t eq_id, int32_t d, double sigma, double mu, const double *sol, const double *v, int32_t n_dom, int32_t n_bdy,
                                   double *out, void *stream) {
    if (!eq_known(eq_id)) return fail(SCASML_ERR_UNSUPPORTED, "gp_newton_jv: unknown equation id %d", eq_id);
    if (!sol || !v || !out || n_dom < 1 || n_bdy < 0) return fail(SCASML_ERR_ARG, "gp_newton_jv: bad argument");
    const int M = 4 * n_dom + n_bdy;
    hipLaunchKernelGGL(gp_newton_jv_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, EqArgs{eq_id, (double)d, sigma, mu}, sol, v, n_dom, n_bdy, out);
    return check_launch("gp_newton_jv launch");
}

extern "C" int scasml_gp_newton_jtv(int32_t eq_id, int32_t d, double sigma, double mu, const double *sol, const double *w, const double *Ab,
                                    const double *v, double scale, int32_t n_dom, int32_t n_bdy, double *out, void *stream) {
    if (!eq_known(eq_id)) return fail(SCASML_ERR_UNSUPPORTED, "gp_newton_jtv: unknown equation id %d", eq_id);
    if (!sol || !w || !out || n_dom < 1 || n_bdy < 0) return fail(SCASML_ERR_ARG, "gp_newton_jtv: bad argument");
    hipLaunchKernelGGL(gp_newton_jtv_kernel, dim3((n_dom + 255) / 256), dim3(256), 0, (hipStream_t)stream, EqArgs{eq_id, (double)d, sigma, mu}, sol, w, Ab, v,
                       scale, n_dom, n_bdy, out);
    return check_launch("gp_newton_jtv launch");
}

extern "C" int scasml_gemv(const double *A, int64_t M, int64_t lda, const double *x, double *y, void *stream) {
    if (!A || !x || !y || M < 1 || lda < M) return fail(SCASML_ERR_ARG, "gemv: bad argument");
    hipLaunchKernelGGL(gemv_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, A, M, lda, x, y);
    return check_launch("gemv launch");
}

extern "C" int scasml_gp_newton_system(int32_t eq_id, int32_t d, double sigma, double mu, const double *A, int64_t lda, int32_t n_dom,
                                       int32_t n_bdy, const double *sol, const double *Ab, double *grad, double *H, int64_t ldh,
                                       int gauss_newton, void *stream) {
    if (!eq_known(eq_id)) return fail(SCASML_ERR_UNSUPPORTED, "gp_newton_system: unknown equation id %d", eq_id);
    if (!A || !sol || !Ab || !grad || !H || n_dom < 1 || ldh < 3 * (int64_t)n_dom || lda < 4 * (int64_t)n_dom + n_bdy)
        return fail(SCASML_ERR_ARG, "gp_newton_system: bad argument");
    const unsigned gx = (unsigned)((ldh + 15) / 16);
    if (gx > 65535) return fail(SCASML_ERR_UNSUPPORTED, "gp_newton_system: system too large for this build");
    hipLaunchKernelGGL(gp_newton_system_kernel, dim3(gx, gx), dim3(16, 16), 0, (hipStream_t)stream, EqArgs{eq_id, (double)d, sigma, mu}, A, lda, n_dom,
                       n_bdy, sol, Ab, grad, H, ldh, gauss_newton);
    return check_launch("gp_newton_system launch");
}

// K_p^-1 = L^-T L^-1 from the Cholesky factor, exploiting structure: X = L^-1 is lower triangular (forward
// substitution on the identity touches only columns left of the current block) and the result is symmetric
// (the backward substitution is carried out for the lower triangle only, then mirrored): about a third of
// the tile updates of two general triangular solves.
extern "C" int scasml_cholesky_inverse(const double *L, int64_t M, double *A, void *stream) {
    if (!L || !A || M < 1) return fail(SCASML_ERR_ARG, "cholesky_inverse: bad argument");
    if (M % NB) return fail(SCASML_ERR_UNSUPPORTED, "cholesky_inverse: M=%lld is not a multiple of %d", (long long)M, NB);
    if (M > 65535 * (int64_t)NB) return fail(SCASML_ERR_UNSUPPORTED, "cholesky_inverse: M too large for this build");
    hipStream_t s = (hipStream_t)stream;
    auto tiles = [](int64_t n) { return (unsigned)((n + TB - 1) / TB); };
    hipLaunchKernelGGL(set_identity_kernel, dim3((unsigned)((M + 255) / 256), row_grid(M)), dim3(256), 0, s, A, M);
    auto forward_chain = [&](int64_t J, int64_t jend, hipStream_t q) {    // rows [J, jend) of X = L^-1 final
        for (int64_t k0 = J; k0 < jend; k0 += NB) {
            const int64_t lim = k0 + NB;
            hipLaunchKernelGGL(trsm_diag_kernel<0>, dim3((unsigned)((lim + 255) / 256)), dim3(256), 0, q, L, M, A, M, k0, lim);
            if (k0 + NB < jend)
                trsm_update<0>(L, M, A, M, k0, NB, k0 + NB, jend, lim, 0, q);
        }
    };
    auto backward_chain = [&](int64_t J, int64_t jend, hipStream_t q) {   // rows [J, jend) of Z = L^-T X final
        for (int64_t k0 = jend - NB; k0 >= J; k0 -= NB) {
            const int64_t lim = k0 + NB;
            hipLaunchKernelGGL(trsm_diag_kernel<1>, dim3((unsigned)((lim + 255) / 256)), dim3(256), 0, q, L, M, A, M, k0, lim);
            if (k0 > J)
                trsm_update<1>(L, M, A, M, k0, NB, J, k0, k0, 1, q);
        }
    };
    if (M < kLookaheadRows) {
        for (int64_t J = 0; J < M; J += kOuterRows) {            // X = L^-1: row r is nonzero in columns <= r
            const int64_t jend = J + kOuterRows < M ? J + kOuterRows : M;
            forward_chain(J, jend, s);
            if (jend < M)
                trsm_update<0>(L, M, A, M, J, jend - J, jend, M, jend, 0, s);
        }
        for (int64_t jend = M; jend > 0; jend -= kOuterRows) {   // Z = L^-T X, lower triangle only
            const int64_t J = jend > kOuterRows ? jend - kOuterRows : 0;
            backward_chain(J, jend, s);
            if (J > 0)
                trsm_update<1>(L, M, A, M, J, jend - J, 0, J, J, 1, s);
        }
    } else {   // the same with the chains on the lookahead stream (struct Lookahead): groups of rows instead of panels of columns
        // groups of outer_rows(M) rows, solved as sub-groups of kOuterRows with one K = kOuterRows update between them (scasml_cholesky's
        // outer panels): the updates of everything beyond a group run at twice or four times the K, C read and written that much less often
        const int64_t G = outer_rows(M);
        auto forward_group = [&](int64_t J, int64_t jend, hipStream_t q) {
            for (int64_t Js = J; Js < jend; Js += kOuterRows) {
                const int64_t Je = Js + kOuterRows < jend ? Js + kOuterRows : jend;
                forward_chain(Js, Je, q);
                if (Je < jend) trsm_update<0>(L, M, A, M, Js, Je - Js, Je, jend, Je, 0, q);   // sub-group -> the group's later rows
            }
        };
        auto backward_group = [&](int64_t J, int64_t jend, hipStream_t q) {
            for (int64_t Je = jend; Je > J; Je -= kOuterRows) {
                const int64_t Js = Je - J > kOuterRows ? Je - kOuterRows : J;
                backward_chain(Js, Je, q);
                if (Js > J) trsm_update<1>(L, M, A, M, Js, Je - Js, J, Js, Js, 1, q);         // sub-group -> the group's earlier rows
            }
        };
        Lookahead la;
        if (!la.open(s)) return fail(SCASML_ERR_HIP, "cholesky_inverse: cannot create the lookahead stream");
        for (int64_t J = 0; J < M; J += G) {
            const int64_t jend = J + G < M ? J + G : M;
            const int64_t next_end = jend + G < M ? jend + G : M;
            forward_group(J, jend, la.side);
            la.chain_done();
            if (jend < M) {
                la.side_waits_for_main();
                trsm_update<0>(L, M, A, M, J, jend - J, jend, next_end, jend, 0, la.side);   // group J -> rows of group J+1
                if (next_end < M) {
                    la.main_waits_for_chain();
                    trsm_update<0>(L, M, A, M, J, jend - J, next_end, M, jend, 0, s);        // group J -> all rows below group J+1
                    la.main_applied();
                }
            }
        }
        la.join();
        for (int64_t jend = M; jend > 0; jend -= G) {
            const int64_t J = jend > G ? jend - G : 0;
            const int64_t prev = J > G ? J - G : 0;
            backward_group(J, jend, la.side);
            la.chain_done();
            if (J > 0) {
                la.side_waits_for_main();
                trsm_update<1>(L, M, A, M, J, jend - J, prev, J, J, 1, la.side);             // group J -> rows of the group above
                if (prev > 0) {
                    la.main_waits_for_chain();
                    trsm_update<1>(L, M, A, M, J, jend - J, 0, prev, J, 1, s);               // group J -> all rows above that
                    la.main_applied();
                }
            }
        }
        la.close();
    }
    hipLaunchKernelGGL(mirror_lower_kernel, dim3((unsigned)((M + 255) / 256), row_grid(M)), dim3(256), 0, s, A, M);
    return check_launch("cholesky_inverse launch");
}

extern "C" int scasml_gp_gram(int32_t d, double a, const float *x_dom, int32_t n_dom, const float *x_bdy, int32_t n_bdy,
                              double *K, void *stream) {
    if (!x_dom || !K || (n_bdy > 0 && !x_bdy)) return fail(SCASML_ERR_ARG, "gp_gram: null argument");
    if (d < 1 || n_dom < 1 || n_bdy < 0) return fail(SCASML_ERR_ARG, "gp_gram: bad sizes");
    const int N = n_dom + n_bdy;
    const unsigned gt = (unsigned)((N + 63) / 64);
    if (gt > 65535) return fail(SCASML_ERR_UNSUPPORTED, "gp_gram: too many collocation points for one launch");
    if (!reserve_lds(gp_gram_mfma_kernel, tile_lds_bytes<2>())) return fail(SCASML_ERR_HIP, "gp_gram: cannot reserve LDS");
    hipLaunchKernelGGL(gp_gram_mfma_kernel, dim3(gt, gt), dim3(256), tile_lds_bytes<2>(), (hipStream_t)stream, d, a, x_dom, n_dom, x_bdy, n_bdy, K);
    return check_launch("gp_gram launch");
}

// Feature rows [row0, row0 + nrows) x columns [0, ncols) of K(phi, phi) for the block-row distributed factorisation: the row range is
// cut at the operator boundaries of the feature order [u(dom), u(bdy), Lap(dom), dt(dom), div(dom)] and every piece -- one operator,
// a contiguous range of points -- is one launch of the FP64-MFMA pair tile.
extern "C" int scasml_gp_gram_rows(int32_t d, double a, const float *x_dom, int32_t n_dom, const float *x_bdy, int32_t n_bdy,
                                   int64_t row0, int32_t nrows, int64_t ncols, double *out, int64_t ld, void *stream) {
    if (!x_dom || !out || (n_bdy > 0 && !x_bdy)) return fail(SCASML_ERR_ARG, "gp_gram_rows: null argument");
    const int64_t M = 4 * (int64_t)n_dom + n_bdy;
    if (d < 1 || n_dom < 1 || n_bdy < 0 || row0 < 0 || nrows < 0 || row0 + nrows > M || ncols < 0 || ncols > M || ld < ncols)
        return fail(SCASML_ERR_ARG, "gp_gram_rows: bad sizes");
    if (nrows == 0 || ncols == 0) return 0;
    const int N = n_dom + n_bdy;
    if (!reserve_lds(gp_gram_rows_mfma_kernel, tile_lds_bytes<2>())) return fail(SCASML_ERR_HIP, "gp_gram_rows: cannot reserve LDS");
    const unsigned gx = (unsigned)((N + 63) / 64);
    int64_t r = row0;
    const int64_t r_end = row0 + nrows;
    while (r < r_end) {
        int ox, i_lo;
        int64_t seg_end;                          // first row past this operator's block
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
        const unsigned gy = (unsigned)((n_i + 63) / 64);
        if (gy > 65535) return fail(SCASML_ERR_UNSUPPORTED, "gp_gram_rows: too many rows per call");
        hipLaunchKernelGGL(gp_gram_rows_mfma_kernel, dim3(gx, gy), dim3(256), tile_lds_bytes<2>(), (hipStream_t)stream, d, a, x_dom, n_dom, x_bdy,
                           n_bdy, ox, i_lo, n_i, ncols, out + (r - row0) * ld, ld);
        r = stop;
    }
    return check_launch("gp_gram_rows launch");
}
