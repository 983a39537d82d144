// Host-only entry points of libscasml_hip.so -- ABI bookkeeping, the static schedule's helpers (site kinds, dealing of the
// Monte-Carlo units) and the normal transform's table as data.  Plain C++: no HIP header, no device code, so the same file is
// built a second time with -fsanitize=address,undefined and driven on the CPU (tests/test_host_sanitizers.py).
#include <string.h>

#include <new>
#include <vector>

#include "host_common.hpp"

namespace scasml {
char *error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace scasml

using namespace scasml;

extern "C" int scasml_abi_version(void) { return SCASML_ABI_VERSION; }
extern "C" const char *scasml_last_error(void) { return scasml::error_buffer(); }
extern "C" size_t scasml_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(scasml_problem);
        case 1: return sizeof(scasml_rng);
        case 2: return sizeof(scasml_term);
        case 3: return sizeof(scasml_plan);
        case 4: return sizeof(scasml_gp_model);
    }
    return 0;
}

extern "C" int64_t scasml_points_per_root(const scasml_plan *plan_h) {
    if (!plan_h || plan_h->n < 0 || plan_h->n > SCASML_MAX_LEVEL) return -1;
    return (int64_t)plan_h->sites[plan_h->n] + 1;
}

// Site kinds in the kernels' enumeration order (terminal samples first, then per level / path / node the
// Euler-Maruyama site followed by its child subtrees): 3 = terminal sample (only u_hat consumed, at t = T), 1 = only u_hat consumed (the root row),
// 4 = Euler-Maruyama site of a level l > 0 term (u_hat and div u_hat consumed, eps_PDE not), 0 = everything consumed.
static void site_kinds_rec(const scasml_plan *p, int n, uint8_t *&out) {
    if (n == 0) return;
    for (int m = 0; m < p->mg[n]; ++m) *out++ = 3;   // terminal samples: u_hat only, at t = T
    for (int l = 0; l < n; ++l) {
        const scasml_term &t = p->term[n][l];
        for (int m = 0; m < t.mc; ++m)
            for (int k = 0; k < t.q; ++k) {
                *out++ = l > 0 ? 4 : 0;   // eps_PDE enters the sum only in the level-0 term (ScaSML.py:274-280)
                site_kinds_rec(p, l, out);
                if (l > 0) site_kinds_rec(p, l - 1, out);
            }
    }
}

// cost of the subtree of a level-n call, in the units of scasml_plan_deal_units (Euler-Maruyama site 1, terminal site 0.6)
static double subtree_cost(const scasml_plan *p, int n) {
    if (n == 0) return 0.0;
    double c = 0.6 * p->mg[n];
    for (int l = 0; l < n; ++l) {
        const scasml_term &t = p->term[n][l];
        c += (double)t.mc * t.q * (1.0 + subtree_cost(p, l) + (l > 0 ? subtree_cost(p, l - 1) : 0.0));
    }
    return c;
}

extern "C" int32_t scasml_plan_deal_units(const scasml_plan *plan_h, int32_t world, uint8_t *owner_h, int32_t capacity, double *load_h) {
    if (!plan_h || !owner_h || plan_h->n < 1 || plan_h->n > SCASML_MAX_LEVEL) return fail(SCASML_ERR_ARG, "plan_deal_units: bad argument");
    if (world < 1 || world > 255) return fail(SCASML_ERR_ARG, "plan_deal_units: world must be 1..255");
    const int n = plan_h->n;
    int64_t units = plan_h->mg[n];
    // units past the terminal samples: per node (l, m, k) its "+" addend (the node and the level-l subtree) and, for l > 0, its "-" addend (the
    // level-(l-1) subtree) -- enumerated node by node, "+" first, exactly as Walker::level() asks owned()
    for (int l = 0; l < n; ++l) units += (int64_t)plan_h->term[n][l].mc * plan_h->term[n][l].q * (l > 0 ? 2 : 1);
    if (units > capacity) return fail(SCASML_ERR_ARG, "plan_deal_units: %lld units exceed the capacity %d", (long long)units, capacity);
    // unit costs in enumeration order; the levels come in blocks of equal cost, the most expensive level last
    std::vector<double> cost, load;
    std::vector<char> done;
    try {
        cost.assign((size_t)units, 0.0);
        load.assign((size_t)world, 0.0);
        done.assign((size_t)units, 0);
    } catch (const std::bad_alloc &) {
        return fail(SCASML_ERR_ARG, "plan_deal_units: out of host memory for %lld units", (long long)units);
    }
    int64_t u = 0;
    for (int m = 0; m < plan_h->mg[n]; ++m) cost[u++] = 0.6;
    for (int l = 0; l < n; ++l) {
        const scasml_term &t = plan_h->term[n][l];
        const double cp = 1.0 + subtree_cost(plan_h, l), cm = l > 0 ? subtree_cost(plan_h, l - 1) : 0.0;
        for (int m = 0; m < t.mc * t.q; ++m) {
            cost[u++] = cp;
            if (l > 0) cost[u++] = cm;
        }
    }
    for (int64_t k = 0; k < units; ++k) {           // longest processing time first (ties: lower unit index, lower rank)
        int64_t best = -1;
        for (int64_t i = 0; i < units; ++i)
            if (!done[i] && (best < 0 || cost[i] > cost[best])) best = i;
        int r = 0;
        for (int j = 1; j < world; ++j)
            if (load[j] < load[r]) r = j;
        done[best] = 1;
        owner_h[best] = (uint8_t)r;
        load[r] += cost[best];
    }
    if (load_h)
        for (int j = 0; j < world; ++j) load_h[j] = load[j];
    return (int32_t)units;
}

extern "C" int scasml_plan_site_kinds(const scasml_plan *plan_h, int32_t rank, int32_t world, const uint8_t *unit_owner_h, uint8_t *kinds_h) {
    if (!plan_h || !kinds_h || plan_h->n < 0 || plan_h->n > SCASML_MAX_LEVEL) return fail(SCASML_ERR_ARG, "plan_site_kinds: bad argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(SCASML_ERR_ARG, "plan_site_kinds: bad rank/world");
    uint8_t *w = kinds_h;
    site_kinds_rec(plan_h, plan_h->n, w);
    if (w - kinds_h != plan_h->sites[plan_h->n]) return fail(SCASML_ERR_ARG, "plan_site_kinds: plan.sites is inconsistent with its terms");
    *w = 1;   // the root row (every rank evaluates it)
    if (world > 1 && plan_h->n > 0) {   // units of the ROOT call, dealt exactly as Walker::owned() does
        const int n = plan_h->n;
        int unit = 0;
        int64_t o = 0;
        auto mine = [&](int un) { return unit_owner_h ? (int)unit_owner_h[un] == rank : un % world == rank; };
        for (int m = 0; m < plan_h->mg[n]; ++m, ++unit, ++o)
            if (!mine(unit)) kinds_h[o] = 2;
        for (int l = 0; l < n; ++l) {
            const scasml_term &t = plan_h->term[n][l];
            for (int m = 0; m < t.mc * t.q; ++m) {                     // node by node: its own site, the "+" subtree, the "-" subtree (l > 0)
                const bool plus = mine(unit++);
                const bool minus = l > 0 ? mine(unit++) : false;
                if (!plus && !minus) kinds_h[o] = 2;                     // the node's point is evaluated by whoever owns either addend
                o += 1;
                if (!plus)
                    for (int64_t k = 0; k < t.sites_l; ++k) kinds_h[o + k] = 2;
                o += t.sites_l;
                if (!minus)
                    for (int64_t k = 0; k < t.sites_lm1; ++k) kinds_h[o + k] = 2;
                o += t.sites_lm1;
            }
        }
    }
    return 0;
}

extern "C" int32_t scasml_point_stride(int32_t d) { return (d + 4 + 15) / 16 * 16; }


// The 768 x 4 coefficients of the table-driven inverse normal CDF (philox_normal.hpp includes the same file for the kernels).
static const float kNormalTableHost[][4] = {
#include "normal_table.inc"
};
constexpr int32_t kNormalTableHostRows = (int32_t)(sizeof(kNormalTableHost) / sizeof(kNormalTableHost[0]));
static_assert(kNormalTableHostRows == SCASML_NORMAL_TABLE_ROWS, "normal_table.inc and SCASML_NORMAL_TABLE_ROWS disagree");

extern "C" int32_t scasml_normal_table_rows(void) { return kNormalTableHostRows; }

extern "C" int scasml_normal_table(float *table_h, int32_t capacity_rows) {
    if (!table_h) return fail(SCASML_ERR_ARG, "normal_table: null argument");
    if (capacity_rows < kNormalTableHostRows)
        return fail(SCASML_ERR_ARG, "normal_table: the table has %d rows of 4 floats, the buffer holds %d", kNormalTableHostRows, capacity_rows);
    memcpy(table_h, kNormalTableHost, sizeof(kNormalTableHost));
    return 0;
}
