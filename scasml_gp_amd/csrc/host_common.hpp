// Host-only helpers of libscasml_hip.so that need no HIP header: the thread-local error string and fail().
// plan_host.cpp (plain C++, also built with -fsanitize=address,undefined by tests/test_host_sanitizers.py) includes only this.
#pragma once
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "scasml_hip.h"

namespace scasml {

char *error_buffer();  // thread-local, 512 bytes (plan_host.cpp)

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// host + device where the HIP compiler sees it (the kernels' tile order), plain host code in plan_host.cpp (scasml_tile_order: the CPU tests)
#if defined(__HIPCC__)
#define SCASML_HD __host__ __device__
#else
#define SCASML_HD
#endif

// Tile order of the 128 x 128 kernels (WS = 4: one workgroup per CU).  Workgroups b and b + 8 of a launch are observed to share an XCD and
// its 4 MB L2 (MI355X_MICROARCH.md, Workgroup dispatch: round-robin placement; used for speed only, any placement gives the same result),
// and an XCD runs 32 of them at a time.  With a row-major 2-D grid the 32 tiles an XCD holds at one moment lie in ONE tile row, eight
// columns apart: one shared row panel and 32 column panels of their own -- the update read 3.1 x the bytes it wrote
// (profiles/r06_cholesky_counters_70k.txt: 274 KB of operands per tile against 256 KB of C traffic), and at 15.8 flop per fabric byte the
// FP64 matrix cores cannot be fed.  The 1-D grid here deals an XCD SUPER-TILES of 8 x 4 tiles: 8 row panels + 4 column panels (3 MB at
// K = 256, streamed 32 columns at a time by workgroups that start together) feed 32 tiles.  Triangular launches (tj <= ti, nti == ntj)
// enumerate the super-tiles of the lower triangle only: super-row sr holds super-columns 0 .. 2 sr + 1.
constexpr int kSuperRows = 8, kSuperCols = 4, kSuperTiles = kSuperRows * kSuperCols, kXcds = 8;
SCASML_HD inline int64_t super_tile_count(int64_t nti, int64_t ntj, bool tri) {
    const int64_t nsr = (nti + kSuperRows - 1) / kSuperRows, nsc = (ntj + kSuperCols - 1) / kSuperCols;
    return tri ? nsr * nsr + nsr : nsr * nsc;
}
SCASML_HD inline unsigned super_tile_grid(int64_t nti, int64_t ntj, bool tri) {
    const int64_t n = super_tile_count(nti, ntj, tri);
    return (unsigned)((n + kXcds - 1) / kXcds * kXcds * kSuperTiles);
}
// false: this workgroup has no tile (beyond the last super-tile; the caller still tests its tile against the matrix)
SCASML_HD inline bool super_tile_of_block(int64_t b, int64_t nti, int64_t ntj, bool tri, int64_t &ti, int64_t &tj) {
    const int64_t x = b % kXcds, q = b / kXcds;
    const int64_t S = (q / kSuperTiles) * kXcds + x, w = q % kSuperTiles;
    const int64_t nsr = (nti + kSuperRows - 1) / kSuperRows, nsc = (ntj + kSuperCols - 1) / kSuperCols;
    int64_t sr, sc;
    if (tri) {
        if (S >= nsr * nsr + nsr) return false;
        sr = (int64_t)((sqrt(4.0 * (double)S + 1.0) - 1.0) * 0.5);   // largest sr with sr^2 + sr <= S
        while (sr * sr + sr > S) --sr;
        while ((sr + 1) * (sr + 1) + (sr + 1) <= S) ++sr;
        sc = S - (sr * sr + sr);
    } else {
        if (S >= nsr * nsc) return false;
        sr = S / nsc;
        sc = S % nsc;
    }
    ti = sr * kSuperRows + w / kSuperCols;
    tj = sc * kSuperCols + w % kSuperCols;
    return ti < nti && tj < ntj;
}

}  // namespace scasml
