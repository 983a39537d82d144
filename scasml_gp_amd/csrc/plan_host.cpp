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

// What a site costs, relative to an Euler-Maruyama site of a level-0 term whose surrogate values are all consumed (kind 0): {kind 0, kind 4
// (u_hat and div only), kind 3 (terminal: u_hat only), one replayed path step (quadrature paths: the draws of a node another rank owns)}.
// Defaults: the figures of ABI <= 6 (any Euler-Maruyama site 1, terminal site 0.6, replay free).  Measured at the headline shape with the as-coded
// surrogate (profiles/r06_sample_sharding_rank_times.txt): GP evaluation 37.1 / 21.8 / 17.0 us per site of 16384 roots + 3.4 us of GENERATE and
// ACCUMULATE per site -> {1, 0.62, 0.50, 0.04}; the host passes the weights of the surrogate in use (solvers/_picard.py site_cost).
struct SiteCost {
    double full, udiv, term, replay;
};
static SiteCost site_cost_of(const double *w) {
    SiteCost c{1.0, 1.0, 0.6, 0.0};
    if (w) c = SiteCost{w[0], w[1], w[2], w[3]};
    return c;
}

// cost of the subtree of a level-n call (below the root call every site has one owner: no replay, no shared node points)
static double subtree_cost(const scasml_plan *p, int n, const SiteCost &w) {
    if (n == 0) return 0.0;
    double c = w.term * p->mg[n];
    for (int l = 0; l < n; ++l) {
        const scasml_term &t = p->term[n][l];
        c += (double)t.mc * t.q * ((l > 0 ? w.udiv : w.full) + subtree_cost(p, l, w) + (l > 0 ? subtree_cost(p, l - 1, w) : 0.0));
    }
    return c;
}

extern "C" int32_t scasml_plan_deal_units(const scasml_plan *plan_h, int32_t world, const double *site_cost_h, uint8_t *owner_h, int32_t capacity,
                                          double *load_h) {
    if (!plan_h || !owner_h || plan_h->n < 1 || plan_h->n > SCASML_MAX_LEVEL) return fail(SCASML_ERR_ARG, "plan_deal_units: bad argument");
    if (world < 1 || world > 255) return fail(SCASML_ERR_ARG, "plan_deal_units: world must be 1..255");
    const SiteCost w = site_cost_of(site_cost_h);
    if (!(w.full > 0.0) || !(w.udiv > 0.0) || !(w.term > 0.0) || !(w.replay >= 0.0)) return fail(SCASML_ERR_ARG, "plan_deal_units: site costs must be positive (replay >= 0)");
    const int n = plan_h->n;
    int64_t units = plan_h->mg[n];
    // units past the terminal samples: per node (l, m, k) its "+" addend (the node and the level-l subtree) and, for l > 0, its "-" addend (the
    // level-(l-1) subtree) -- enumerated node by node, "+" first, exactly as Walker::level() asks owned()
    for (int l = 0; l < n; ++l) units += (int64_t)plan_h->term[n][l].mc * plan_h->term[n][l].q * (l > 0 ? 2 : 1);
    if (units > capacity) return fail(SCASML_ERR_ARG, "plan_deal_units: %lld units exceed the capacity %d", (long long)units, capacity);
    // unit costs in enumeration order; the levels come in blocks of equal cost, the most expensive level last.  Two costs depend on WHO gets a unit:
    // a "-" addend on another rank than its node's "+" addend makes that rank evaluate the node's point too (scasml_plan_site_kinds keeps it for
    // either owner), and on the quadrature paths a rank replays the draws of a path up to the LAST node it owns an addend of (Walker::level stops
    // there; the steps of nodes it owns are in the node's cost already)
    const bool replays = plan_h->variant == 0 && w.replay > 0.0 && world > 1;
    int64_t paths = 0;
    for (int l = 0; l < n; ++l) paths += plan_h->term[n][l].mc;
    std::vector<double> cost, load, node;
    std::vector<int32_t> path, knode, last, own;
    std::vector<char> done;
    try {
        cost.assign((size_t)units, 0.0);
        node.assign((size_t)units, 0.0);     // > 0 on "-" addends: what the node's own point costs whoever does not hold the "+" addend already
        path.assign((size_t)units, -1);      // sample path of the unit's node (terminal samples: none) and the node's position on it
        knode.assign((size_t)units, 0);
        load.assign((size_t)world, 0.0);
        done.assign((size_t)units, 0);
        last.assign((size_t)(world * paths), -1);   // per (rank, path): last node the rank owns an addend of, and how many nodes it owns
        own.assign((size_t)(world * paths), 0);
    } catch (const std::bad_alloc &) {
        return fail(SCASML_ERR_ARG, "plan_deal_units: out of host memory for %lld units", (long long)units);
    }
    int64_t u = 0;
    int32_t pid = 0;
    for (int m = 0; m < plan_h->mg[n]; ++m) cost[u++] = w.term;
    for (int l = 0; l < n; ++l) {
        const scasml_term &t = plan_h->term[n][l];
        const double nw = l > 0 ? w.udiv : w.full;
        const double cp = nw + subtree_cost(plan_h, l, w), cm = l > 0 ? subtree_cost(plan_h, l - 1, w) : 0.0;
        for (int m = 0; m < t.mc; ++m, ++pid)
            for (int k = 0; k < t.q; ++k) {
                path[u] = pid, knode[u] = k;
                cost[u++] = cp;
                if (l > 0) {
                    path[u] = pid, knode[u] = k;
                    node[u] = nw;
                    cost[u++] = cm;
                }
            }
    }
    // what rank r pays on top of the unit's own cost
    auto extra = [&](int64_t i, int r, bool commit) {
        double e = 0.0;
        const bool with_plus = node[i] > 0.0 && owner_h[i - 1] == r;      // a "-" addend next to its "+" partner (dealt before it: it costs more)
        if (node[i] > 0.0 && !with_plus) e += node[i];
        if (replays && path[i] >= 0) {
            int32_t &la = last[(size_t)r * paths + path[i]], &ow = own[(size_t)r * paths + path[i]];
            const int32_t la2 = knode[i] > la ? knode[i] : la, ow2 = ow + (with_plus ? 0 : 1);
            e += w.replay * ((la2 + 1 - ow2) - (la + 1 - ow));
            if (commit) la = la2, ow = ow2;
        }
        return e;
    };
    for (int64_t k = 0; k < units; ++k) {           // longest processing time first (ties: lower unit index)
        int64_t best = -1;
        for (int64_t i = 0; i < units; ++i)
            if (!done[i] && (best < 0 || cost[i] > cost[best])) best = i;
        // the rank that leaves the largest load lowest; among those the one that pays least for the unit, then the one that ends lowest (a unit
        // that costs its partner's rank nothing is not moved to a rank where it costs a second node point, however idle that rank is)
        double top = 0.0;
        for (int j = 0; j < world; ++j) top = load[j] > top ? load[j] : top;
        int r = 0;
        double span_r = 0.0, pay_r = 0.0, end_r = 0.0;
        for (int j = 0; j < world; ++j) {
            const double pay = cost[best] + extra(best, j, false), e = load[j] + pay, span = e > top ? e : top;
            const bool better = j == 0 || span < span_r || (span == span_r && (pay < pay_r || (pay == pay_r && e < end_r)));
            if (better) r = j, span_r = span, pay_r = pay, end_r = e;
        }
        done[best] = 1;
        owner_h[best] = (uint8_t)r;
        load[r] += cost[best] + extra(best, r, true);
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

// The tile order of the 128 x 128 FP64 update kernels (host_common.hpp), for inspection and the CPU tests
extern "C" int64_t scasml_tile_order_blocks(int64_t nti, int64_t ntj, int32_t tri) {
    if (nti < 1 || ntj < 1 || (tri && nti != ntj)) return fail(SCASML_ERR_ARG, "tile_order_blocks: bad argument");
    return (int64_t)super_tile_grid(nti, ntj, tri != 0);
}
extern "C" int scasml_tile_order(int64_t block, int64_t nti, int64_t ntj, int32_t tri, int64_t *ti_h, int64_t *tj_h) {
    if (!ti_h || !tj_h || block < 0 || nti < 1 || ntj < 1 || (tri && nti != ntj)) return fail(SCASML_ERR_ARG, "tile_order: bad argument");
    int64_t ti = -1, tj = -1;
    const bool live = super_tile_of_block(block, nti, ntj, tri != 0, ti, tj);
    *ti_h = live ? ti : -1;
    *tj_h = live ? tj : -1;
    return live ? 1 : 0;
}


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
