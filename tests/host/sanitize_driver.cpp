// Driver for the address/undefined-behaviour sanitizer build of the library's host-only translation unit
// (scasml_gp_amd/csrc/plan_host.cpp; SURVEY.md section 5 "race detection / sanitizers").  Built and run by
// tests/test_host_sanitizers.py with  g++ -fsanitize=address,undefined -fno-sanitize-recover=all :
// every entry point is driven with exactly-sized heap buffers, so a write past an end, a read of freed memory or signed
// overflow in the planning arithmetic ends the process with a report instead of passing silently.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "scasml_hip.h"

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, scasml_last_error()); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

// a schedule of the shape tables.build_plan makes: level n' has mg terminal samples and, per sub-level l < n', mc paths of q nodes
static scasml_plan make_plan(int variant, int n, int mg_base, int q, int mc_base) {
    scasml_plan p;
    memset(&p, 0, sizeof(p));
    p.variant = variant;
    p.n = n;
    p.mg[0] = 1;
    for (int np = 1; np <= n; ++np) {
        p.mg[np] = p.mg[np - 1] * mg_base;
        int64_t sites = p.mg[np];
        for (int l = 0; l < np; ++l) {
            scasml_term &t = p.term[np][l];
            t.q = variant ? 1 : q;
            t.mc = mc_base + (np - l - 1);
            t.sites_l = p.sites[l];
            t.sites_lm1 = l > 0 ? p.sites[l - 1] : 0;
            sites += (int64_t)t.mc * t.q * (1 + t.sites_l + t.sites_lm1);
        }
        p.sites[np] = (int32_t)sites;
    }
    return p;
}

int main() {
    CHECK(scasml_abi_version() == SCASML_ABI_VERSION);
    CHECK(scasml_sizeof(3) == sizeof(scasml_plan) && scasml_sizeof(99) == 0);
    CHECK(scasml_point_stride(100) == 112 && scasml_point_stride(252) == 256);

    // ---- the normal table: exact capacity accepted, one row short refused and untouched, null refused
    const int32_t rows = scasml_normal_table_rows();
    CHECK(rows == SCASML_NORMAL_TABLE_ROWS);
    {
        std::vector<float> exact((size_t)rows * 4);
        CHECK(scasml_normal_table(exact.data(), rows) == 0);
        CHECK(exact[0] != 0.0f);
        std::vector<float> small((size_t)(rows - 1) * 4, 7.0f);
        CHECK(scasml_normal_table(small.data(), rows - 1) == SCASML_ERR_ARG);
        for (float v : small) CHECK(v == 7.0f);
        CHECK(scasml_normal_table(nullptr, rows) == SCASML_ERR_ARG);
        CHECK(scasml_normal_table(exact.data(), -5) == SCASML_ERR_ARG);
    }

    // ---- site kinds and unit dealing over every level and both variants, exactly-sized buffers
    for (int variant = 0; variant < 2; ++variant)
        for (int n = 0; n <= SCASML_MAX_LEVEL; ++n) {
            const scasml_plan p = make_plan(variant, n, variant ? 3 : 2, 3, 2);
            const int64_t ppr = scasml_points_per_root(&p);
            CHECK(ppr == (int64_t)p.sites[n] + 1);
            std::vector<uint8_t> kinds((size_t)ppr);
            CHECK(scasml_plan_site_kinds(&p, 0, 1, nullptr, kinds.data()) == 0);
            CHECK(kinds[(size_t)ppr - 1] == 1);
            if (n == 0) continue;
            int units = p.mg[n];
            for (int l = 0; l < n; ++l) units += p.term[n][l].mc * p.term[n][l].q * (l > 0 ? 2 : 1);   // terminal samples + the addends of the nodes (m, k)
            for (int world : {1, 2, 3, 8, 255}) {
                std::vector<uint8_t> owner((size_t)units);
                std::vector<double> load((size_t)world);
                {
                    const double measured[4] = {1.0, 0.62, 0.50, 0.04};      // the as-coded surrogate's site costs (replay charged on the quadrature paths)
                    CHECK(scasml_plan_deal_units(&p, world, (world & 1) ? measured : nullptr, owner.data(), units, load.data()) == units);
                }
                for (uint8_t o : owner) CHECK(o < world);
                CHECK(scasml_plan_deal_units(&p, world, nullptr, owner.data(), units - 1, load.data()) == SCASML_ERR_ARG);   // capacity one short: refused
                for (int rank = 0; rank < world && rank < 4; ++rank) {
                    CHECK(scasml_plan_site_kinds(&p, rank, world, owner.data(), kinds.data()) == 0);
                    CHECK(scasml_plan_site_kinds(&p, rank, world, nullptr, kinds.data()) == 0);
                }
            }
            CHECK(scasml_plan_deal_units(&p, 0, nullptr, kinds.data(), units, nullptr) == SCASML_ERR_ARG);
            CHECK(scasml_plan_deal_units(&p, 256, nullptr, kinds.data(), units, nullptr) == SCASML_ERR_ARG);
            CHECK(scasml_plan_site_kinds(&p, 2, 2, nullptr, kinds.data()) == SCASML_ERR_ARG);
        }
    // an inconsistent plan (sites that do not match the terms) is refused before anything past the buffer is written
    {
        scasml_plan p = make_plan(0, 2, 2, 3, 2);
        std::vector<uint8_t> kinds((size_t)p.sites[2] + 1);
        p.sites[2] += 1;
        std::vector<uint8_t> kinds2((size_t)p.sites[2] + 1);
        CHECK(scasml_plan_site_kinds(&p, 0, 1, nullptr, kinds2.data()) == SCASML_ERR_ARG);
        p.n = SCASML_MAX_LEVEL + 1;
        CHECK(scasml_points_per_root(&p) == -1);
        CHECK(scasml_plan_site_kinds(&p, 0, 1, nullptr, kinds.data()) == SCASML_ERR_ARG);
        CHECK(scasml_plan_deal_units(nullptr, 1, nullptr, kinds.data(), 1, nullptr) == SCASML_ERR_ARG);
    }
    CHECK(strlen(scasml_last_error()) > 0);
    printf("host sanitizer driver ok\n");
    return 0;
}
