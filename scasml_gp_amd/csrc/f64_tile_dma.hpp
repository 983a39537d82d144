// The 128 x 128 FP64 update tile  C -= A B^T  (v_mfma_f64_16x16x4_f64) with its operands staged by LDS-DMA: the trailing update of
// scasml_cholesky (gp_train.hip) and scasml_gemm_nt_sub (dist_linalg.hip), i.e. the factor of models/GP.py:260-267 at sizes where the update
// is all of the run time.  Both operands are row-major panels (row = output row / output column, K contiguous).
//
// 1024 threads = 16 waves as 4 x 4, each wave 2 x 2 MFMA tiles of 16 x 16.  K streams through LDS 16 columns at a time in FOUR stages:
// a stage is [A rows 0..127][B rows 0..127], a row 16 doubles = 8 granules of 16 bytes, unpadded, because global_load_lds_dwordx4 writes a
// wave's 64 x 16 bytes contiguously (8 rows per instruction; wave w stages rows 8 w .. 8 w + 7 of both operands: two instructions per
// stage and no staging registers, no ds_write).  Granule g of row r sits in slot g ^ ((r >> 1) & 7): the 16 rows x 2 k of one ds_read_b64
// pass then fall on 64 distinct banks (unswizzled, rows 128 bytes apart alternate between two bank halves: 8-way conflicts).
// Three stages fly while one feeds the matrix cores -- the register-staged double buffer (32 columns per stage, 139 KB) had one chunk of
// look-ahead, and under the read-modify-write traffic of the other CUs' tiles its loads did not arrive in time: the same tile with the
// same summation order (results bit-identical) measured 41.6 / 46.7 / 49.1 TFLOP/s at K = 256 / 512 / 1024 against 49.4 / 55.7 / 60.6
// here (tools/ubench_f64_tile.hip, C of 16 384^2, profiles/r06_f64_tile_phases.txt).  One barrier per stage; completion is counted by hand
// (counted vmcnt, then the barrier), the DMA being issued from inline asm (gp_mfma16.hpp: hipcc would drain it at once).
// The output tile is read under the last two stages' matrix work and written as C - acc.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scasml {

typedef double dma_f64x4 __attribute__((ext_vector_type(4)));
constexpr int kDmaTile = 128, kDmaNB = 16, kDmaStages = 4, kDmaThreads = 1024;
constexpr int kDmaOpBytes = kDmaTile * kDmaNB * 8, kDmaStageBytes = 2 * kDmaOpBytes;
constexpr size_t kDmaLdsBytes = (size_t)kDmaStages * kDmaStageBytes;   // 128 KB: one workgroup per CU

__device__ __forceinline__ void f64_glds16(const void *gsrc_uniform, uint32_t lane_byte_offset, uint32_t lds_byte_addr_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_byte_offset), "s"(gsrc_uniform), "s"(lds_byte_addr_uniform)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void f64_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void f64_rendezvous() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// One tile.  At, Bt: first valid row of the tile's operand panels at K column 0 (16-byte aligned, lda / ldb even); rows_a / rows_b (1..128):
// valid rows -- rows beyond them re-read the last valid row, their products land in output rows / columns that are never stored.
// K: a multiple of 16, >= 64.  Ct: the tile's corner in C.  All arguments are workgroup-uniform.
__device__ __forceinline__ void f64_tile_dma(double *smem, const double *At, int64_t lda, int rows_a, const double *Bt, int64_t ldb, int rows_b, int64_t K,
                                             double *Ct, int64_t ldc) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wvs = __builtin_amdgcn_readfirstlane(wv);
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    const int wr = (wv >> 2) * 32, wc = (wv & 3) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    // DMA: this lane's source inside a chunk of either operand (32-bit byte offsets: rows < 128, leading dimensions < 2^21)
    const uint32_t drow = 8 * wvs + (lane >> 3), dgran = (lane & 7) ^ ((drow >> 1) & 7);
    const uint32_t ra = drow < (uint32_t)rows_a ? drow : (uint32_t)rows_a - 1u, rb = drow < (uint32_t)rows_b ? drow : (uint32_t)rows_b - 1u;
    const uint32_t offa = ra * ((uint32_t)lda * 8u) + dgran * 16u, offb = rb * ((uint32_t)ldb * 8u) + dgran * 16u;
    auto stage = [&](int64_t c) {   // chunk c -> slot c % kDmaStages
        const uint32_t dst = lds_base + (uint32_t)(c % kDmaStages) * kDmaStageBytes + (uint32_t)wvs * 1024u;
        f64_glds16(At + c * kDmaNB, offa, dst);
        f64_glds16(Bt + c * kDmaNB, offb, dst + kDmaOpBytes);
    };
    // fragment reads: byte offsets inside a stage; the K step k0 enters by XOR (k0 * 8 flips granule bits 1..2 of the slot)
    uint32_t fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t r_a = wr + 16 * i + l15, r_b = wc + 16 * i + l15;
        fa[i] = r_a * 128u + ((((uint32_t)l4 >> 1) ^ ((r_a >> 1) & 7)) * 16u) + ((uint32_t)l4 & 1) * 8u;
        fb[i] = kDmaOpBytes + r_b * 128u + ((((uint32_t)l4 >> 1) ^ ((r_b >> 1) & 7)) * 16u) + ((uint32_t)l4 & 1) * 8u;
    }
    dma_f64x4 acc[2][2], cin[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (dma_f64x4){0.0, 0.0, 0.0, 0.0};
    const char *sm = reinterpret_cast<const char *>(smem);
    auto compute = [&](int64_t c) {
        const char *st = sm + (c % kDmaStages) * kDmaStageBytes;
#pragma unroll
        for (int k0 = 0; k0 < kDmaNB; k0 += 4) {
            double av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const double *>(st + (fa[i] ^ (uint32_t)(k0 * 8)));
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const double *>(st + (fb[j] ^ (uint32_t)(k0 * 8)));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };
    const bool interior = rows_a == kDmaTile && rows_b == kDmaTile;   // block-uniform
    // this thread's place in the tile (row wr + l4, column wc + l15) as ONE 32-bit byte offset against scalar bases per fragment element
    const uint32_t toff = ((uint32_t)(wr + l4) * (uint32_t)ldc + (uint32_t)(wc + l15)) * 8u;
    auto elem = [&](int i, int j, int e) {
        return reinterpret_cast<double *>(reinterpret_cast<char *>(Ct + (int64_t)(16 * i + 4 * e) * ldc + 16 * j) + toff);
    };
    const int64_t nc = K / kDmaNB;
    stage(0);
    stage(1);
    stage(2);
    for (int64_t c = 0; c + 2 < nc; ++c) {
        f64_wait_vm<4>();   // this wave's share of chunk c has landed (chunks c + 1, c + 2 may fly) ...
        f64_rendezvous();   // ... and everyone's; every wave is past chunk c - 1, whose slot chunk c + 3 takes
        if (c + 3 < nc) stage(c + 3);
        compute(c);
    }
    f64_wait_vm<2>();
    f64_rendezvous();
    if (interior) {   // the output tile's loads fly under the last two chunks
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) cin[i][j][e] = *elem(i, j, e);
    }
    compute(nc - 2);
    if (interior) f64_wait_vm<16>();
    else f64_wait_vm<0>();
    f64_rendezvous();
    compute(nc - 1);
    if (interior) {
        f64_wait_vm<0>();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) *elem(i, j, e) = cin[i][j][e] - acc[i][j][e];
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = wr + 16 * i + l4 + 4 * e, c = wc + 16 * j + l15;
                if (r < rows_a && c < rows_b) *elem(i, j, e) -= acc[i][j][e];
            }
}

}  // namespace scasml
