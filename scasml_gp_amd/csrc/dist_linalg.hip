// Building blocks of the block-row DISTRIBUTED Gram / Cholesky / triangular solves (scasml_gp_amd/dist_gp.py; BASELINE
// configs[4]: 1e5 collocation points -> M = 350 000 features, 980 GB float64 -- only the 8 GPUs of a node together hold it).
// Replaces models/GP.py:182-268 (kernel_phi_phi + factor) and the solve of :599 at sizes one GPU cannot hold.
//
// Storage: K (and then L) is cut into block rows of kDistBlock = 256 feature rows; block row i is owned by rank i % world
// (1-D block-cyclic) and stored as a dense row-major panel of 256 x (i + 1) * 256 doubles -- the lower triangle only.
// Everything here works on such panels through (pointer, leading dimension) pairs, so the same kernels serve any layout:
//   (scasml_gp_gram_rows, the Gram rows of a block row, is the FP64-MFMA pair tile of gp_train.hip)
//   scasml_gemm_nt_sub     C -= A B^T              (FP64 MFMA, v_mfma_f64_16x16x4_f64, 64 x 64 tile per workgroup)
//   scasml_trsm_right_lt   X <- X L^-T             (the panel solve of the right-looking factorisation)
//   scasml_gemv_sub        y -= A x  or  y -= A^T x  (the block steps of the distributed substitutions)
#include <stdlib.h>

#include "common.hpp"
#include "f64_tile_dma.hpp"

namespace scasml {

constexpr int kNB = 32;

// ---------------------------------------------------------------------------------- C -= A B^T
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int kLDP = kNB + 2;   // padded LDS leading dimension (gp_train.hip)

// (32 WS)^2 tile per workgroup of WS x WS waves, each wave 2 x 2 MFMA tiles of 16 x 16 (WS = 2: 64 x 64, 4 waves; WS = 4:
// 128 x 128, 16 waves -- half the operand traffic per flop, used when both extents are large; gp_train.hip has the same pair);
// the K dimension is streamed through LDS 32 at a time, double-buffered (global loads of chunk k+1 in flight while chunk k
// feeds the matrix cores).  A: rows x K (lda), B: cols x K (ldb), C: rows x cols (ldc).  K must be a multiple of 32.
// Triangular skipping for block-cyclic row panels: local row block lb (of SCASML_DIST_BLOCK rows) is global block row
// tri_row0 + lb * tri_stride, tile column block cb is global block column tri_col0 + cb; tiles strictly above the block
// diagonal are not touched (tri_stride = 0: no skipping).
struct TriMap {
    int64_t row0, stride, col0;
};
constexpr size_t gemm_lds_bytes(int ws) { return (size_t)2 * 2 * (32 * ws) * kLDP * sizeof(double); }   // 2 operands x 2 buffers

template <int WS>
__global__ __launch_bounds__(64 * WS * WS) void gemm_nt_sub_kernel(double *C, int64_t ldc, int64_t rows, int64_t cols, const double *A,
                                                                   int64_t lda, const double *B, int64_t ldb, int64_t K, TriMap tri) {
    constexpr int TBX = 32 * WS, THREADS = 64 * WS * WS, PER = TBX * kNB / THREADS;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double (*Pa)[TBX][kLDP] = reinterpret_cast<double (*)[TBX][kLDP]>(smem);
    double (*Pb)[TBX][kLDP] = reinterpret_cast<double (*)[TBX][kLDP]>(smem + 2 * TBX * kLDP);
    int64_t ti = blockIdx.y, tj = blockIdx.x;
    if (WS == 4 && !super_tile_of_block(blockIdx.x, (rows + TBX - 1) / TBX, (cols + TBX - 1) / TBX, false, ti, tj)) return;   // 1-D grid, super-tile order (common.hpp)
    const int64_t r0 = ti * TBX, c0 = tj * TBX;
    if (tri.stride > 0 && tri.col0 + c0 / SCASML_DIST_BLOCK > tri.row0 + (r0 / SCASML_DIST_BLOCK) * tri.stride) return;   // block-uniform
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wr = (wv / WS) * 32, wc = (wv % WS) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const bool interior = r0 + TBX <= rows && c0 + TBX <= cols;   // block-uniform: no bounds tests, loads issued back to back
    double ra[PER], rb[PER];
    auto fetch = [&](int64_t kk) {
        if (interior) {
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                const int idx = threadIdx.x + e * THREADS, rr = idx / kNB, cc = idx % kNB;
                ra[e] = A[(r0 + rr) * lda + kk + cc];
                rb[e] = B[(c0 + rr) * ldb + kk + cc];
            }
        } else {
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                const int idx = threadIdx.x + e * THREADS, rr = idx / kNB, cc = idx % kNB;
                ra[e] = r0 + rr < rows ? A[(r0 + rr) * lda + kk + cc] : 0.0;
                rb[e] = c0 + rr < cols ? B[(c0 + rr) * ldb + kk + cc] : 0.0;
            }
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int idx = threadIdx.x + e * THREADS, rr = idx / kNB, cc = idx % kNB;
            Pa[buf][rr][cc] = ra[e];
            Pb[buf][rr][cc] = rb[e];
        }
    };
    auto accumulate = [&](int cur) {
#pragma unroll
        for (int k0 = 0; k0 < kNB; k0 += 4) {
            double av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = Pa[cur][wr + 16 * i + l15][k0 + l4];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = Pb[cur][wc + 16 * j + l15][k0 + l4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };
    fetch(0);
    park(0);
    __syncthreads();
    int cur = 0;
    for (int64_t kk = 0; kk + kNB < K; kk += kNB) {
        fetch(kk + kNB);
        accumulate(cur);
        park(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // the output tile's loads fly under the last chunk's matrix work (gp_train.hip, mfma_tile_load_full: the guarded
    // read-modify-write tail cost a quarter of the factorisation)
    double *Ct = C + (r0 + wr + l4) * ldc + c0 + wc + l15;
    f64x4 cin[2][2];
    if (interior) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) cin[i][j][e] = Ct[(int64_t)(16 * i + 4 * e) * ldc + 16 * j];
    }
    accumulate(cur);
    if (interior) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) Ct[(int64_t)(16 * i + 4 * e) * ldc + 16 * j] = cin[i][j][e] - acc[i][j][e];
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t r = r0 + wr + 16 * i + l4 + 4 * e, c = c0 + wc + 16 * j + l15;
                if (r < rows && c < cols) C[r * ldc + c] -= acc[i][j][e];
            }
}

// The 128 x 128 tile with LDS-DMA operand staging (f64_tile_dma.hpp): the large trailing updates of the distributed factorisation.
// Same results as gemm_nt_sub_kernel<4>, bit for bit (same summation order); that kernel stays for operands the DMA cannot take
// (odd leading dimensions, bases off 16 bytes).
__global__ __launch_bounds__(kDmaThreads) void gemm_nt_sub_dma_kernel(double *C, int64_t ldc, int64_t rows, int64_t cols, const double *A, int64_t lda,
                                                                      const double *B, int64_t ldb, int64_t K, TriMap tri) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int64_t ti, tj;
    if (!super_tile_of_block(blockIdx.x, (rows + kDmaTile - 1) / kDmaTile, (cols + kDmaTile - 1) / kDmaTile, false, ti, tj)) return;
    const int64_t r0 = ti * kDmaTile, c0 = tj * kDmaTile;
    if (tri.stride > 0 && tri.col0 + c0 / SCASML_DIST_BLOCK > tri.row0 + (r0 / SCASML_DIST_BLOCK) * tri.stride) return;   // block-uniform
    const int ra = (int)(rows - r0 < kDmaTile ? rows - r0 : kDmaTile), rb = (int)(cols - c0 < kDmaTile ? cols - c0 : kDmaTile);
    f64_tile_dma(smem, A + r0 * lda, lda, ra, B + c0 * ldb, ldb, rb, K, C + r0 * ldc + c0, ldc);
}

// ---------------------------------------------------------------------------------- X <- X L^-T (32 columns at a time)
// one thread per row of X: x L_kk^T = rhs with the 32 x 32 lower-triangular L_kk at (L, ldl)
__global__ __launch_bounds__(256) void trsm_right_lt32_kernel(const double *L, int64_t ldl, double *X, int64_t ldx, int64_t rows) {
    __shared__ double Lk[kNB][kNB + 1];
    for (int idx = threadIdx.x; idx < kNB * kNB; idx += blockDim.x) Lk[idx / kNB][idx % kNB] = L[(int64_t)(idx / kNB) * ldl + idx % kNB];
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    double x[kNB];
#pragma unroll
    for (int j = 0; j < kNB; ++j) x[j] = X[r * ldx + j];
#pragma unroll
    for (int j = 0; j < kNB; ++j) {
        double v = x[j];
#pragma unroll
        for (int p = 0; p < j; ++p) v = fma(-x[p], Lk[j][p], v);
        x[j] = v / Lk[j][j];
    }
#pragma unroll
    for (int j = 0; j < kNB; ++j) X[r * ldx + j] = x[j];
}

// ---------------------------------------------------------------------------------- y -= A x, y -= A^T x
// tri.stride > 0 (block-row panels of a lower-triangular factor, TriMap above with col0 unused): row block lb is global block row
// tri.row0 + lb * tri.stride and holds nothing beyond its own diagonal block -- the sweep stops there instead of reading the zeros
__global__ __launch_bounds__(256) void gemv_sub_kernel(const double *A, int64_t lda, int64_t rows, int64_t cols, const double *x, double *y, TriMap tri) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (tri.stride > 0) {
        const int64_t lim = (tri.row0 + (row / SCASML_DIST_BLOCK) * tri.stride + 1) * SCASML_DIST_BLOCK;
        cols = lim < cols ? lim : cols;
    }
    double acc = 0.0;
    for (int64_t k = lane; k < cols; k += 64) acc = fma(A[row * lda + k], x[k], acc);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) y[row] -= acc;
}

// y[c] -= sum_r A[r][c] x[r]: a workgroup takes 64 columns (one per lane: a row's 512 bytes are one coalesced read) and its four waves interleaved
// row slices, four independent accumulators each -- sixteen rows in flight per workgroup (one thread per column walking its rows alone was latency
// bound: 0.4 ms for a 1024-row sweep, 55 of the 61 ms of a distributed solve at M = 70 001) -- combined through LDS in a FIXED association.
// Rows beyond `rows_per_block` are split over blockIdx.y and those partial sums meet in atomics (scasml_gemv_sub beyond 1024 rows).
__device__ __forceinline__ double gemv_t_columns(const double *A, int64_t lda, int64_t rb, int64_t re, int64_t c, bool live, const double *x) {
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (live) {
        int64_t r = rb + wv;
        for (; r + 12 < re; r += 16) {
            a0 = fma(A[r * lda + c], x[r], a0);
            a1 = fma(A[(r + 4) * lda + c], x[r + 4], a1);
            a2 = fma(A[(r + 8) * lda + c], x[r + 8], a2);
            a3 = fma(A[(r + 12) * lda + c], x[r + 12], a3);
        }
        for (; r < re; r += 4) a0 = fma(A[r * lda + c], x[r], a0);
    }
    part[wv][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    return (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

__global__ __launch_bounds__(256) void gemv_t_sub_kernel(const double *A, int64_t lda, int64_t rows, int64_t cols, const double *x, double *y,
                                                         int64_t rows_per_block) {
    const int64_t c = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int64_t rb = (int64_t)blockIdx.y * rows_per_block;
    const int64_t re = rb + rows_per_block < rows ? rb + rows_per_block : rows;
    const double acc = gemv_t_columns(A, lda, rb, re, c, c < cols, x);
    if (threadIdx.x >= 64 || c >= cols) return;
    if (gridDim.y == 1) y[c] -= acc;     // single writer: deterministic (the distributed solves sweep at most 1024 rows per launch)
    else atomicAdd(&y[c], -acc);
}

// the same product with the row groups' partial sums written to a scratch buffer and added in FIXED order: bitwise reproducible between runs
// (the atomic path's last bits depend on the order the row groups retire in)
__global__ __launch_bounds__(256) void gemv_t_partial_kernel(const double *A, int64_t lda, int64_t rows, int64_t cols, const double *x, double *partial,
                                                             int64_t rows_per_block, TriMap tri) {
    const int64_t c = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    int64_t rb = (int64_t)blockIdx.y * rows_per_block;
    const int64_t re = rb + rows_per_block < rows ? rb + rows_per_block : rows;
    if (tri.stride > 0) {   // the 64 columns of this workgroup lie in ONE block column: the rows above its first block row hold zeros there (block-uniform)
        const int64_t cb = (int64_t)blockIdx.x * 64 / SCASML_DIST_BLOCK;
        const int64_t lb = cb <= tri.row0 ? 0 : (cb - tri.row0 + tri.stride - 1) / tri.stride;
        rb = lb * SCASML_DIST_BLOCK > rb ? lb * SCASML_DIST_BLOCK : rb;
    }
    const double acc = gemv_t_columns(A, lda, rb, re, c, c < cols, x);
    if (threadIdx.x < 64 && c < cols) partial[(int64_t)blockIdx.y * cols + c] = acc;
}
__global__ __launch_bounds__(256) void gemv_t_reduce_kernel(const double *partial, int64_t groups, int64_t cols, double *y) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    double acc = 0.0;
    for (int64_t g = 0; g < groups; ++g) acc += partial[g * cols + c];
    y[c] -= acc;
}

}  // namespace scasml

using namespace scasml;

extern "C" int scasml_gemm_nt_sub(double *C, int64_t ldc, int64_t rows, int64_t cols, const double *A, int64_t lda, const double *B,
                                  int64_t ldb, int64_t K, int64_t tri_row0, int64_t tri_stride, int64_t tri_col0, void *stream) {
    if (!C || !A || !B || rows < 0 || cols < 0 || K < 0 || ldc < cols || lda < K || ldb < K) return fail(SCASML_ERR_ARG, "gemm_nt_sub: bad argument");
    if (K % kNB) return fail(SCASML_ERR_UNSUPPORTED, "gemm_nt_sub: K=%lld is not a multiple of %d", (long long)K, kNB);
    if (rows == 0 || cols == 0 || K == 0) return 0;
    const TriMap tri{tri_row0, tri_stride, tri_col0};
    const bool big = rows >= 4096 && cols >= 4096 && K >= 256;
    const int tb = big ? 128 : 64;
    const int64_t gx = (cols + tb - 1) / tb, gy = (rows + tb - 1) / tb;
    if (gy > 65535) return fail(SCASML_ERR_UNSUPPORTED, "gemm_nt_sub: too many rows for one launch");
    const bool dma = big && K >= kDmaStages * kDmaNB && lda % 2 == 0 && ldb % 2 == 0 && ((uintptr_t)A | (uintptr_t)B) % 16 == 0 && lda < (1 << 21) && ldb < (1 << 21) &&
                     ldc < (1 << 21) && !getenv("SCASML_F64_TILE_REGISTER_STAGED");   // (development: the register-staged tile for A/B runs)
    if (dma) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_nt_sub_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDmaLdsBytes) != hipSuccess)
            return fail(SCASML_ERR_HIP, "gemm_nt_sub: cannot reserve %zu bytes of LDS", kDmaLdsBytes);
        hipLaunchKernelGGL(gemm_nt_sub_dma_kernel, dim3(super_tile_grid(gy, gx, false)), dim3(kDmaThreads), kDmaLdsBytes, (hipStream_t)stream, C, ldc, rows, cols, A,
                           lda, B, ldb, K, tri);
    } else if (big) {
        constexpr size_t lds = gemm_lds_bytes(4);
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_nt_sub_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return fail(SCASML_ERR_HIP, "gemm_nt_sub: cannot reserve %zu bytes of LDS", lds);
        hipLaunchKernelGGL(gemm_nt_sub_kernel<4>, dim3(super_tile_grid(gy, gx, false)), dim3(1024), lds, (hipStream_t)stream, C, ldc, rows, cols, A, lda, B, ldb, K, tri);
    } else {
        constexpr size_t lds = gemm_lds_bytes(2);
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_nt_sub_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return fail(SCASML_ERR_HIP, "gemm_nt_sub: cannot reserve %zu bytes of LDS", lds);
        hipLaunchKernelGGL(gemm_nt_sub_kernel<2>, dim3((unsigned)gx, (unsigned)gy), dim3(256), lds, (hipStream_t)stream, C, ldc, rows, cols, A, lda, B, ldb, K, tri);
    }
    return check_launch("gemm_nt_sub launch");
}

extern "C" int scasml_trsm_right_lt(const double *L, int64_t ldl, int64_t nb, double *X, int64_t ldx, int64_t rows, void *stream) {
    if (!L || !X || nb < 1 || rows < 0 || ldl < nb || ldx < nb) return fail(SCASML_ERR_ARG, "trsm_right_lt: bad argument");
    if (nb % kNB) return fail(SCASML_ERR_UNSUPPORTED, "trsm_right_lt: nb=%lld is not a multiple of %d", (long long)nb, kNB);
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const unsigned gb = (unsigned)((rows + 255) / 256);
    for (int64_t j = 0; j < nb; j += kNB) {
        // X[:, j:j+32] <- X[:, j:j+32] L_jj^-T, then X[:, j+32:] -= X[:, j:j+32] L[j+32:, j:j+32]^T
        hipLaunchKernelGGL(trsm_right_lt32_kernel, dim3(gb), dim3(256), 0, s, L + j * ldl + j, ldl, X + j, ldx, rows);
        const int64_t rest = nb - j - kNB;
        if (rest > 0) {
            const int64_t gy = (rows + 63) / 64;
            if (gy > 65535) return fail(SCASML_ERR_UNSUPPORTED, "trsm_right_lt: too many rows for one launch");
            constexpr size_t lds = gemm_lds_bytes(2);
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_nt_sub_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return fail(SCASML_ERR_HIP, "trsm_right_lt: cannot reserve LDS");
            hipLaunchKernelGGL(gemm_nt_sub_kernel<2>, dim3((unsigned)((rest + 63) / 64), (unsigned)gy), dim3(256), lds, s, X + j + kNB, ldx, rows, rest,
                               X + j, ldx, L + (j + kNB) * ldl + j, ldl, (int64_t)kNB, TriMap{0, 0, 0});
        }
    }
    return check_launch("trsm_right_lt launch");
}

extern "C" int64_t scasml_gemv_t_ordered_scratch(int64_t rows, int64_t cols) {
    if (rows < 0 || cols < 0) return -1;
    const int64_t rpb = rows <= 64 * 64 ? 64 : (rows + 63) / 64;      // at most 64 row groups
    const int64_t groups = rows ? (rows + rpb - 1) / rpb : 0;
    return groups * cols;
}

static int gemv_t_sub_ordered(const double *A, int64_t lda, int64_t rows, int64_t cols, const double *x, double *y, double *scratch, int64_t scratch_elems,
                              TriMap tri, void *stream);
extern "C" int scasml_gemv_t_sub_ordered(const double *A, int64_t lda, int64_t rows, int64_t cols, const double *x, double *y, double *scratch,
                                         int64_t scratch_elems, void *stream) {
    return gemv_t_sub_ordered(A, lda, rows, cols, x, y, scratch, scratch_elems, TriMap{0, 0, 0}, stream);
}
extern "C" int scasml_gemv_t_sub_ordered_tri(const double *A, int64_t lda, int64_t rows, int64_t cols, const double *x, double *y, double *scratch,
                                             int64_t scratch_elems, int64_t tri_row0, int64_t tri_stride, void *stream) {
    if (tri_row0 < 0 || tri_stride < 1) return fail(SCASML_ERR_ARG, "gemv_t_sub_ordered_tri: bad block map");
    return gemv_t_sub_ordered(A, lda, rows, cols, x, y, scratch, scratch_elems, TriMap{tri_row0, tri_stride, 0}, stream);
}
static int gemv_t_sub_ordered(const double *A, int64_t lda, int64_t rows, int64_t cols, const double *x, double *y, double *scratch, int64_t scratch_elems,
                              TriMap tri, void *stream) {
    if (!A || !x || !y || rows < 0 || cols < 0 || lda < cols) return fail(SCASML_ERR_ARG, "gemv_t_sub_ordered: bad argument");
    if (rows == 0 || cols == 0) return 0;
    const int64_t need = scasml_gemv_t_ordered_scratch(rows, cols);
    if (!scratch || scratch_elems < need) return fail(SCASML_ERR_ARG, "gemv_t_sub_ordered: scratch holds %lld doubles, %lld needed", (long long)scratch_elems, (long long)need);
    const int64_t rpb = rows <= 64 * 64 ? 64 : (rows + 63) / 64;
    const int64_t groups = (rows + rpb - 1) / rpb;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gemv_t_partial_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)groups), dim3(256), 0, s, A, lda, rows, cols, x, scratch, rpb, tri);
    hipLaunchKernelGGL(gemv_t_reduce_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, s, scratch, groups, cols, y);
    return check_launch("gemv_t_sub_ordered launch");
}

extern "C" int scasml_gemv_sub_tri(const double *A, int64_t lda, int64_t rows, int64_t cols, const double *x, double *y, int64_t tri_row0, int64_t tri_stride,
                                   void *stream) {
    if (!A || !x || !y || rows < 0 || cols < 0 || lda < cols || tri_row0 < 0 || tri_stride < 1) return fail(SCASML_ERR_ARG, "gemv_sub_tri: bad argument");
    if (rows == 0 || cols == 0) return 0;
    hipLaunchKernelGGL(gemv_sub_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, A, lda, rows, cols, x, y, TriMap{tri_row0, tri_stride, 0});
    return check_launch("gemv_sub_tri launch");
}

extern "C" int scasml_gemv_sub(const double *A, int64_t lda, int64_t rows, int64_t cols, const double *x, double *y, int trans, void *stream) {
    if (!A || !x || !y || rows < 0 || cols < 0 || lda < cols) return fail(SCASML_ERR_ARG, "gemv_sub: bad argument");
    if (rows == 0 || cols == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (!trans) {
        hipLaunchKernelGGL(gemv_sub_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, A, lda, rows, cols, x, y, TriMap{0, 0, 0});
    } else {
        const int64_t rpb = rows <= 1024 ? rows : 64;     // up to 1024 rows (a group of four block rows of the distributed substitutions): one writer per column
        const int64_t gy = (rows + rpb - 1) / rpb;
        if (gy > 65535) return fail(SCASML_ERR_UNSUPPORTED, "gemv_sub: too many rows for one launch");
        hipLaunchKernelGGL(gemv_t_sub_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)gy), dim3(256), 0, s, A, lda, rows, cols, x, y, rpb);
    }
    return check_launch("gemv_sub launch");
}
