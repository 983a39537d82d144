// The 128 x 128 FP64 update tile  C -= A B^T  (v_mfma_f64_16x16x4_f64) with its operands staged by LDS-DMA: the trailing update of
// scasml_cholesky (gp_train.hip) and scasml_gemm_nt_sub (dist_linalg.hip), i.e. the factor of models/GP.py:260-267 at sizes where the update
// is all of the run time.  Both operands are row-major panels (row = output row / output column, K contiguous).
//
// 1024 threads = 16 waves as 4 x 4, each wave 2 x 2 MFMA tiles of 16 x 16.  K streams through LDS NB columns at a time in STAGES stages:
// a stage is [A rows 0..127][B rows 0..127], a row NB doubles = NB / 2 granules of 16 bytes, unpadded, because global_load_lds_dwordx4 writes a
// wave's 64 x 16 bytes contiguously (1 KB = 4 or 8 rows per instruction; wave w stages rows 8 w .. 8 w + 7 of both operands; no staging
// registers, no ds_write).  Granule g of row r sits in slot g ^ swz(r), so that the 16 rows x 2 k of one ds_read_b64 pass fall on 64 distinct
// banks (unswizzled, 256-byte rows all start on bank 0: 16-way conflicts).  One barrier per stage; completion is counted by hand (counted
// vmcnt, then the barrier), the DMA being issued from inline asm (gp_mfma16.hpp: hipcc would drain it at once).  The output tile is read under
// the last stage's (with deeper look-ahead: the last two stages') matrix work and written as C - acc.
// Measured (tools/ubench_f64_tile.hip, C of 16 384^2, operands random; profiles/r06_f64_update_tile.txt), TFLOP/s at K = 256 / 512 / 1024, all
// three forms bit-identical (same summation order): a stand-alone copy of the register-staged double buffer 41.6 / 46.7 / 49.1; (16 columns,
// 4 stages: three stages of look-ahead) 48.9 / 55.5 / 61.5 -- its K loop ran at 87 % of the matrix pipe's rate, the rest being the barrier and
// first-read bubble of every 16-column stage; (32 columns, 2 stages: one stage of look-ahead, half the barriers) 51.2 / 60.2 / 66.9: in use.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scasml {

typedef double dma_f64x4 __attribute__((ext_vector_type(4)));
constexpr int kDmaTile = 128, kDmaThreads = 1024;
// NB columns of K per stage, STAGES stages: (16, 4) = three stages of look-ahead, a barrier per 16 columns; (32, 2) = one stage of look-ahead, a
// barrier per 32 columns.  Both 128 KB: one workgroup per CU.
template <int NB, int STAGES>
struct DmaShape {
    static constexpr int kRowBytes = NB * 8, kGranules = NB / 2, kOpBytes = kDmaTile * kRowBytes, kStageBytes = 2 * kOpBytes;
    static constexpr int kRowsPerInstr = 1024 / kRowBytes, kInstrPerOperand = 8 / kRowsPerInstr, kInstrPerStage = 2 * kInstrPerOperand;
    static constexpr int kRowsPerBankRow = 256 / kRowBytes;   // rows that share one pass over the 64 banks
    static constexpr size_t kLdsBytes = (size_t)STAGES * kStageBytes;
    static_assert(NB == 16 || NB == 32, "16 or 32 columns per stage");
    static_assert(kLdsBytes <= 160 * 1024, "LDS");
    __host__ __device__ static constexpr uint32_t swz(uint32_t row) { return (row / kRowsPerBankRow) & (kGranules - 1); }
};
constexpr int kDmaNB = 32, kDmaStages = 2;                    // the shape in use: +5 / +8.5 / +9 % over (16, 4) at K = 256 / 512 / 1024 (profiles/r06_f64_update_tile.txt)
constexpr size_t kDmaLdsBytes = DmaShape<kDmaNB, kDmaStages>::kLdsBytes;

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
// valid rows -- rows beyond them re-read the last valid row (pair of rows), their products land in output rows / columns that are never stored.
// KMA / KMB: that operand is stored K-MAJOR, element (row, k) at P[k * ld + row] (the right-hand sides and the transposed factor of the
// triangular-solve updates): its stage is [k][row] in LDS -- the DMA's 16 bytes are two adjacent ROWS of one k, one instruction per k --
// with the granule (row pair) index XORed by (k & 1) << 3, so that the two k of a ds_read_b64 pass use different halves of the banks;
// rows_a / rows_b must then be even.
// K: a multiple of NB, >= STAGES * NB.  Ct: the tile's corner in C.  All arguments are workgroup-uniform.
template <int NB = kDmaNB, int STAGES = kDmaStages, bool KMA = false, bool KMB = false>
__device__ __forceinline__ void f64_tile_dma(double *smem, const double *At, int64_t lda, int rows_a, const double *Bt, int64_t ldb, int rows_b, int64_t K,
                                             double *Ct, int64_t ldc) {
    using S = DmaShape<NB, STAGES>;
    constexpr int D = STAGES - 1;                             // stages in flight ahead of the one being consumed
    constexpr int I = S::kInstrPerStage;                      // DMA instructions per wave and stage
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wvs = __builtin_amdgcn_readfirstlane(wv);
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    const int wr = (wv >> 2) * 32, wc = (wv & 3) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    // DMA, row-major operand: wave w stages rows 8 w .. 8 w + 7, kRowsPerInstr rows per instruction; k-major operand: wave w stages the
    // k-rows w * NB / 16 .. (one instruction = the 128 rows of one k).  32-bit byte offsets: rows < 128, k < NB, leading dimensions < 2^21.
    constexpr int Q = S::kInstrPerOperand;
    uint32_t offa[Q], offb[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        auto row_major = [&](int rows, int64_t ld) {
            const uint32_t drow = 8 * wvs + q * S::kRowsPerInstr + lane / S::kGranules, dgran = (lane % S::kGranules) ^ S::swz(drow);
            const uint32_t r = drow < (uint32_t)rows ? drow : (uint32_t)rows - 1u;
            return r * ((uint32_t)ld * 8u) + dgran * 16u;
        };
        auto k_major = [&](int rows, int64_t ld) {
            const uint32_t kq = Q * wvs + q, g = (uint32_t)lane ^ ((kq & 1u) << 3), last = (uint32_t)rows / 2u - 1u;
            return kq * ((uint32_t)ld * 8u) + (g < last ? g : last) * 16u;
        };
        offa[q] = KMA ? k_major(rows_a, lda) : row_major(rows_a, lda);
        offb[q] = KMB ? k_major(rows_b, ldb) : row_major(rows_b, ldb);
    }
    auto stage = [&](int64_t c) {   // chunk c -> slot c % STAGES
        const uint32_t dst = lds_base + (uint32_t)(c % STAGES) * S::kStageBytes + (uint32_t)wvs * (Q * 1024);   // either layout: Q KB per wave and operand
        const double *pa = KMA ? At + c * NB * lda : At + c * NB, *pb = KMB ? Bt + c * NB * ldb : Bt + c * NB;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            f64_glds16(pa, offa[q], dst + q * 1024);
            f64_glds16(pb, offb[q], dst + S::kOpBytes + q * 1024);
        }
    };
    // fragment reads: byte offsets inside a stage.  Row-major: the K step k0 enters by XOR (k0 * 8 flips the upper bits of the granule's slot);
    // k-major: by addition (k0 KB further)
    uint32_t fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t r_a = wr + 16 * i + l15, r_b = wc + 16 * i + l15;
        auto frag = [&](bool km, uint32_t r) {
            return km ? (uint32_t)l4 * 1024u + (((r >> 1) ^ (((uint32_t)l4 & 1u) << 3)) * 16u) + (r & 1u) * 8u
                      : r * S::kRowBytes + ((((uint32_t)l4 >> 1) ^ S::swz(r)) * 16u) + ((uint32_t)l4 & 1) * 8u;
        };
        fa[i] = frag(KMA, r_a);
        fb[i] = S::kOpBytes + frag(KMB, r_b);
    }
    dma_f64x4 acc[2][2], cin[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (dma_f64x4){0.0, 0.0, 0.0, 0.0};
    const char *sm = reinterpret_cast<const char *>(smem);
    auto compute = [&](int64_t c) {
        const char *st = sm + (c % STAGES) * S::kStageBytes;
#pragma unroll
        for (int k0 = 0; k0 < NB; k0 += 4) {
            double av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const double *>(st + (KMA ? fa[i] + (uint32_t)(k0 * 1024) : fa[i] ^ (uint32_t)(k0 * 8)));
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const double *>(st + (KMB ? fb[j] + (uint32_t)(k0 * 1024) : fb[j] ^ (uint32_t)(k0 * 8)));
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
    auto load_c = [&] {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) cin[i][j][e] = *elem(i, j, e);
    };
    const int64_t nc = K / NB;
    constexpr int kCLead = D >= 2 ? 1 : 0;   // the output tile's loads are issued this many chunks before the last one (they fly under the rest)
    for (int c = 0; c < D; ++c) stage(c);
    // main part: every chunk but the last 1 + kCLead; chunk c's wait lets the younger chunks already issued (at most D - 1) fly
    for (int64_t c = 0; c + 1 + kCLead < nc; ++c) {
        if (nc - 1 - c >= D - 1) f64_wait_vm<(D - 1) * I>();
        else f64_wait_vm<(D >= 2 ? I : 0)>();                 // (D = 3 only: one younger chunk left)
        f64_rendezvous();   // everyone's share of chunk c has landed; every wave is past chunk c - 1, whose slot chunk c + D takes
        if (c + D < nc) stage(c + D);
        compute(c);
    }
    if (kCLead) {           // chunk nc - 2: chunk nc - 1 may fly; then the output tile's loads, under the last two chunks' matrix work
        f64_wait_vm<(D >= 2 ? I : 0)>();
        f64_rendezvous();
        if (interior) load_c();
        compute(nc - 2);
        if (interior) f64_wait_vm<16>();
        else f64_wait_vm<0>();
        f64_rendezvous();
    } else {                // one stage of look-ahead: the loads fly under the last chunk alone
        f64_wait_vm<0>();
        f64_rendezvous();
        if (interior) load_c();
    }
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
