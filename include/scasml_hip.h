/*
 * scasml_hip.h -- C ABI of libscasml_hip.so, the MI355X (gfx950) hot path of SCaSML_GP.
 *
 * The reference (Francis-Fan-create/SCaSML_GP) is pure Python/JAX and has no FFI boundary;
 * its boundary is the Python class surface.  This header is the
 * C ABI the build puts UNDER that surface: every entry point names the reference routine
 * it replaces.  Conventions:
 *   - all pointers are DEVICE pointers unless the name ends in _h (host);
 *   - row-major, float32 unless stated, time in the LAST column of a point row
 *     (solvers/MLP.py:161-162);
 *   - the caller owns every buffer; nothing is allocated, freed or synchronised inside;
 *   - `stream` is a hipStream_t (NULL = default stream); calls are asynchronous on it;
 *   - return 0 on success, <0 on error (message via scasml_last_error(), thread-local);
 *     no entry point throws or aborts;
 *   - re-entrant: no global mutable state besides the thread-local error string.
 */
#ifndef SCASML_HIP_H
#define SCASML_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCASML_ABI_VERSION 7
#define SCASML_MAX_LEVEL 5   /* Picard level n <= 5 (kernels are instantiated per level)      */
#define SCASML_MAX_Q 6       /* quadrature nodes per rule <= 6 (rho <= 5, solvers/MLP.py:132)  */
#define SCASML_MAX_DIM 252   /* spatial dimension d <= 252 (one 4-dim quad per lane, +t, +3 spare columns) */
#define SCASML_GP_TILE 32     /* collocation points per MFMA tile; n_pad is a multiple of it    */
#define SCASML_NORMAL_TABLE_ROWS 768   /* rows (of 4 floats) of the inverse-normal-CDF table: 24 octaves x 32 segments */

enum { SCASML_ERR_ARG = -1, SCASML_ERR_UNSUPPORTED = -2, SCASML_ERR_HIP = -3 };

/* Equation registry (equations/equations.py:15-230 is the reference's plug-in base).  The kernels cover the family
 *   u_t + mu sum_i d_i u + sigma^2/2 Lap u + f(u, sum_i z_i) = 0,  z = sigma grad u,  u(T, x) = g(x),  mu and sigma constant,
 * one device functor set per id (csrc/equations.hpp):
 *   0  equations/equations.py:232-417 `Grad_Dependent_Nonlinear`: f = sigma*u*sum(z), g = 1 - 1/(1+exp(t+sum x)),
 *      mu = -1/d - sigma^2/2, sigma = 0.25
 *   1  `Cubic_Reaction_Diffusion` (no reference counterpart): f = -u (1-u) (1 + (sigma^2 d/2)(1-2u)), mu = 0, same g;
 *      exact solution 1 - 1/(1+exp(t+sum x))
 * and, ABI 7, one step wider for the solvers without a surrogate -- f(u, sum_i z_i, sum_i z_i^2), one more xor-shuffle sum per evaluation of f:
 *   2  `Quadratic_Gradient_Reaction_Diffusion`: f = -u (1-u) (1 + (sigma^2 d/2)(1-2u)) + (|z|^2 - sigma^2 d (u (1-u))^2), mu = 0, same g and
 *      exact solution (the |z|^2 term vanishes on the travelling wave).  scasml_picard_tree in SCASML_MODE_MLP only (MLP, MLP_full_history,
 *      Philox stream): ScaSML's defect f(u_hat + u, sigma grad u_hat + z) - f(u_hat, sigma grad u_hat) would need the surrogate's FULL gradient at
 *      every tree site, where the fused evaluation delivers div u_hat; the GP kernels refuse this id. */
enum { SCASML_EQ_GRAD_DEPENDENT_NONLINEAR = 0, SCASML_EQ_CUBIC_REACTION_DIFFUSION = 1, SCASML_EQ_QUADRATIC_GRADIENT_REACTION_DIFFUSION = 2 };

typedef struct {
    int32_t d;        /* spatial dimension (n_input - 1)                                    */
    int32_t eq_id;    /* SCASML_EQ_*                                                        */
    float T;          /* terminal time (equations.py:356)                                   */
    float mu;         /* drift    (equations.py:263-276)                                    */
    float sigma;      /* diffusion (equations.py:278-288)                                   */
    float clip;       /* norm_estimation (MLP.py:273-274) or uncertainty (ScaSML.py:282-284)*/
} scasml_problem;

/* Philox4x32-10 stream: counter = (quad, site, root0 + local root, stream), key = seed. */
typedef struct {
    uint64_t seed;
    uint32_t stream;  /* advanced by the host once per solver call (replaces MLP.py:220 key splitting) */
    uint32_t root0;   /* global index of the first root in this call (root sharding)       */
    int32_t rank;     /* Monte-Carlo sample sharding of the ROOT call: this rank's units (see unit_owner) */
    int32_t world;    /* 1 = no sample sharding; >1 => outputs are un-clipped partial sums  */
    uint32_t flags;   /* SCASML_RNG_*                                                       */
    uint32_t reserved;
    const uint8_t *unit_owner;  /* DEVICE bytes, one per unit of the root call (terminal samples, then the addends of the nodes (m, k) of the
                                   sample paths of level 0, 1, ...: scasml_plan_deal_units): the rank that owns it, from scasml_plan_deal_units.  NULL: unit % world. */
    const uint32_t *jax_keys;   /* SCASML_RNG_JAX_STREAM: DEVICE words [k0 k1] x (1 + S): the key every call's terminal draws use,
                                   split(PRNGKey(0), 1)[0] (solvers/MLP.py:167-168), then the S sub-keys this solve takes from the solver's
                                   stateful key (MLP.py:220) in the reference's call order (quadrature solvers; the full-history solvers
                                   draw everything from the first key: S = 0).  NULL otherwise. */
} scasml_rng;

/* scasml_rng.flags.  COMPAT_CRN reproduces the reference's key reuse as pure counter
 * keying: every uz_solve call draws its terminal normals from the same fixed key (solvers/MLP.py:167-168,178), so
 * the child calls of the q quadrature nodes of one sample path -- calls of equal shape -- share their terminal
 * draws: a call's terminal sample m is keyed by the site it would have at node k = 0 of every ancestor path.  In
 * the full-history solvers the level-0 normals equal the terminal ones (solvers/MLP_full_history.py:92-93,99,138).
 * Default (0): independent draws everywhere.
 * COMPAT_F16 applies the reference's solver-level float16 casts: Equation.g and Equation.f return
 * float16 (equations/equations.py:261, 304), ScaSML.g / ScaSML.f subtract float16 from float16 (solvers/ScaSML.py:45-47, 62), and every
 * uz_solve returns clip(...).astype(float16) (solvers/MLP.py:274, ScaSML.py:284, MLP_full_history.py:180; ScaSML_full_history.py:199
 * does not cast), so a child's (u, z) is a float16 value before the parent's f sees it.  Sample-sharded partial sums (world > 1) are
 * left unrounded: the cast follows the clip, which follows the all-reduce (scasml_clip_round16).
 * JAX_STREAM (every level 1..SCASML_MAX_LEVEL; with sample sharding every rank passes the same key words) replaces the Philox stream by the REFERENCE's own normals (and, full history, uniform times) --
 * jax.random.normal(key, shape, float16) under jax_threefry_partitionable, each element addressed by the row-major index it has in the
 * reference's batch-vectorised draw (one Threefry-2x32 per normal) -- under the keys in scasml_rng.jax_keys; seed / stream are ignored.
 * With it a solve on the reference's test set lands on the numbers its runs logged (tests/test_gpu_jax_stream.py). */
enum { SCASML_RNG_COMPAT_CRN = 1, SCASML_RNG_COMPAT_F16 = 2, SCASML_RNG_JAX_STREAM = 4 };

/* One (level n', sub-level l) term of the Picard sum: MLP.py:210-271 / MLP_full_history.py:131-177. */
typedef struct {
    int32_t q;                    /* quadrature nodes (1 for full history)                  */
    int32_t mc;                   /* sample paths MC_f                                      */
    int32_t sites_l;              /* RNG sites of one child subtree of level l              */
    int32_t sites_lm1;            /* ... of level l-1 (0 when l == 0)                       */
    float dfrac[SCASML_MAX_Q];    /* (c[k]-c[k-1])/T : time step k as a fraction of T-t     */
    float cfrac[SCASML_MAX_Q];    /* c[k]/T : node time; also delta_t of the "-" term       */
    float wfrac[SCASML_MAX_Q];    /* w[k]/T : quadrature weight                             */
    float dplus[SCASML_MAX_Q];    /* delta_t fraction of the "+" term (MLP.py:249 stale value) */
} scasml_term;

/* Static schedule of the recursion, built on the host from the tables of
 * MLP.approx_parameters (solvers/MLP.py:111-139).  Passed by value to the kernels. */
typedef struct {
    int32_t variant;                                   /* 0 = quadrature, 1 = full history  */
    int32_t n;                                         /* level of the root call            */
    int32_t mg[SCASML_MAX_LEVEL + 1];                  /* terminal samples of a level-n' call */
    int32_t sites[SCASML_MAX_LEVEL + 1];               /* RNG sites of a level-n' subtree   */
    scasml_term term[SCASML_MAX_LEVEL + 1][SCASML_MAX_LEVEL];   /* [n'][l], l < n'          */
} scasml_plan;

enum {
    SCASML_MODE_MLP = 0,        /* MLP / MLP_full_history: no surrogate                      */
    SCASML_MODE_GENERATE = 1,   /* ScaSML pass 1: emit every tree point for the GP          */
    SCASML_MODE_ACCUMULATE = 2  /* ScaSML pass 2: Picard sums on the defect, GP values given */
};

int scasml_abi_version(void);
const char *scasml_last_error(void);
/* sizeof() of the ABI structs as compiled, for binding self-checks:
 * 0 scasml_problem, 1 scasml_rng, 2 scasml_term, 3 scasml_plan, 4 scasml_gp_model. */
size_t scasml_sizeof(int which);

/* Sites of the point buffer / GP-value buffer per root: sites[n] + 1 (the root itself last). */
int64_t scasml_points_per_root(const scasml_plan *plan_h);
/* Padded row length (floats) of a point row: round_up(d + 4, 16): X, t, three spare columns the GP
 * evaluation uses for folded constants (d+1, d+2 and the last one), zero pad to a whole 16-bit MFMA K-step. */
int32_t scasml_point_stride(int32_t d);

/*
 * The Picard tree: replaces MLP.uz_solve (solvers/MLP.py:141-274), ScaSML.uz_solve
 * (solvers/ScaSML.py:149-284) and the two *_full_history.uz_solve
 * (solvers/MLP_full_history.py:64-180, solvers/ScaSML_full_history.py:75-199) -- on-device
 * Philox normals, Euler-Maruyama stepping, the recursive quadrature, f and g
 * (equations/equations.py:248-304), clipping.
 *   x_t      : B x (d+1) evaluation points.
 *   site_stride : rows between consecutive tree sites in `points` / `gp_vals` (>= B; 0 means B).  A multiple of 32
 *              makes every 32-row wavefront tile of scasml_gp_eval_sites a single site, which is what lets it pick the
 *              cheapest epilogue per site (and makes u_hat independent of how a batch is chunked); rows B .. site_stride-1
 *              of a site are padding that is neither written nor read here.
 *   points   : MODE_GENERATE out: points_per_root x site_stride x point_stride, SITE-major (row = site*site_stride + root).
 *              Row content (X, t, zero pad).
 *              MODE_ACCUMULATE in: the same buffer (Euler-Maruyama states are read back, not recomputed).
 *   gp_vals  : MODE_ACCUMULATE in: points_per_root x site_stride x 4 = (u_hat, div_x u_hat, eps_PDE, dt u_hat), same row
 *              order, from scasml_gp_eval_sites on `points`.
 *   out_uz   : B x (1+d): (u, z) clipped [MLP, ACCUMULATE]; un-clipped partial sums if world > 1.
 *   out_uhat : B: u_hat at the root [ACCUMULATE] (ScaSML.py:303), may be NULL.
 */
int scasml_picard_tree(const scasml_problem *prob_h, const scasml_plan *plan_h, int mode,
                       const float *x_t, int64_t B, int64_t site_stride, scasml_rng rng,
                       float *points, const float *gp_vals,
                       float *out_uz, float *out_uhat, void *stream);

/* Clip all-reduced partial sums in place (world > 1): MLP.py:272-274 / ScaSML.py:281-284. */
int scasml_clip(float *uz, int64_t count, float clip, void *stream);
/* The same followed, when round16 != 0, by the root call's .astype(float16) (MLP.py:274, ScaSML.py:284, MLP_full_history.py:180 --
 * ScaSML_full_history.py:199 does not cast): what a sample-sharded solve under SCASML_RNG_COMPAT_F16 still owes after its all-reduce. */
int scasml_clip_round16(float *uz, int64_t count, float clip, int32_t round16, void *stream);

/* Raw RNG access for parity tests: normals of `site` for roots root0..root0+B-1 -> B x d. */
int scasml_debug_normals(scasml_rng rng, uint32_t site, int32_t d, int64_t B, float *out, void *stream);
/* Elements index0 .. index0 + count - 1 of jax.random.normal((key0, key1), shape, float16) under jax_threefry_partitionable, as float32
 * values (SCASML_RNG_JAX_STREAM; the NumPy statement is oracle/jax_random.py: bit-identical, tests/test_gpu_jax_stream.py). */
int scasml_debug_jax_normals(uint32_t key0, uint32_t key1, uint64_t index0, int64_t count, float *out, void *stream);
/* The normal transform on its whole input domain, for exhaustive parity tests: a normal is a function of the top 24 bits of
 * its Philox word;  out[i] = N(word = (k0 + i) << 8)  for i < n, k0 + n <= 2^24 (device pointer). */
int scasml_debug_transform(uint32_t k0, int64_t n, float *out, void *stream);
/* The binary32 coefficients of the table-driven inverse normal CDF (csrc/normal_table.inc) -> HOST memory: the specification
 * of the normal transform as data, so that a second implementation can be checked against it.  The table has
 * scasml_normal_table_rows() (= SCASML_NORMAL_TABLE_ROWS) rows of 4 floats; the caller states how many rows table_h holds and a
 * buffer that is too small is refused (ABI 2 copied the library's table into whatever it was given). */
int32_t scasml_normal_table_rows(void);
int scasml_normal_table(float *table_h, int32_t capacity_rows);

/* ------------------------------------------------------------------ Gaussian process */

/* Trained surrogate, device resident (models/GP.py:185-192, 590-600). */
typedef struct {
    int32_t d;
    int32_t n_dom, n_bdy;        /* N_Omega, N_dOmega                                        */
    int32_t n_pad;               /* (n_dom + n_bdy) rounded up to 32                         */
    int32_t kp;                  /* point stride = round_up(d+4, 16)                         */
    int32_t split;               /* x.y arithmetic: 0 = fp32 MFMA; 3 / 2 = bf16 MFMA on 3 (fp32-exact) / 2 bf16 planes;
                                    22 = fp16 MFMA on two fp16 planes (22-bit products, 3 MFMAs per K-step) */
    float a;                     /* 1/sigma_k^2, sigma_k = 0.25*sqrt(d) (models/GP.py:25)    */
    float sigma_eq;              /* equation sigma (models/GP.py:748)                        */
    float mu_eq;                 /* equation mu: the PDE residual is dt + mu div + sigma^2/2 Lap + f(u, sigma div)  */
    int32_t eq_id;               /* SCASML_EQ_*: which f closes the residual (models/GP.py:767-768 for id 0)      */
    const float *colloc;         /* n_pad x kp   collocation points, domain first, zero pad  */
    const float *colloc_frag;    /* the same, in fp32 MFMA A-fragment order [tile][kp/8][64][4] */
    const uint16_t *colloc_bf16; /* scasml_gp_plane_halfwords(): 3 truncated-bf16 planes [tile][plane][kp/16][64][8], then 2 fp16
                                    planes in the same order */
    int32_t colloc_is_f16;       /* 1 if every collocation coordinate is exactly representable in fp16 (the reference's
                                    deepxde float16 points are): split = 22 then needs 2 instead of 3 MFMAs per K-step */
    const float *coef;           /* scasml_gp_coef_floats(): n_pad x 16 per-row constants (a*sum y, a*t_y, c0, cL, ct, cS, ..., |y|^2) for
                                    the FP32 kernel, then n_pad x 16 in the exponent-scaled form of the 16-bit kernels */
    float x_bound;               /* PRECONDITION of split = 22: every coordinate of every evaluation row satisfies |x_k| <= x_bound
                                    (0 = 2.0, which covers the unit cube plus any path of this equation: 0.5 + |mu| T + 5.8 sigma sqrt(T)
                                    = 1.6).  The fp16 planes carry 0.72 a |x|^2; the call is refused if that could leave the fp16 range
                                    under the stated bound.  Rows beyond the bound overflow to inf / NaN: use split = 3 for them. */
    int32_t reserved;
} scasml_gp_model;

/* Build `coef` and the padded `colloc` from points and right_vector (models/GP.py:599-600):
 * c0 = rv[u(X)] (domain and boundary rows), cL = rv[Lap], ct = rv[dt], cS = rv[div] (zero on
 * boundary rows).  x_dom: n_dom x (d+1), x_bdy: n_bdy x (d+1), rv: 4*n_dom + n_bdy (float64).
 * T_terminal: the equation's terminal time.  A third of a ScaSML step's evaluation points are terminal samples at exactly
 * t = T (ScaSML.py:61); for them the time difference to a collocation point is a per-row constant, which the pack folds into
 * a 4-float form of the constants (site kind 3 of scasml_plan_site_kinds). */
int scasml_gp_pack(int32_t d, float a, float T_terminal, const float *x_dom, int32_t n_dom, const float *x_bdy, int32_t n_bdy,
                   const double *rv, float *colloc_out, float *colloc_frag_out,
                   uint16_t *colloc_bf16_out /* scasml_gp_plane_halfwords(d, n_pad) */,
                   float *coef_out /* scasml_gp_coef_floats(n_pad) */, void *stream);
/* Sizes of the two packed buffers above (n_pad = n_dom + n_bdy rounded up to SCASML_GP_TILE): the 16-bit operand planes
 * in MFMA fragment order (3 bf16 + 2 fp16 planes) and the per-collocation constants. */
int64_t scasml_gp_plane_halfwords(int32_t d, int32_t n_pad);
int64_t scasml_gp_coef_floats(int32_t n_pad);

/*
 * Fused posterior evaluation: replaces GP.predict (models/GP.py:653-671), the spatial-sum of
 * GP.compute_gradient (:673-687) and GP_Grad_Dependent_Nonlinear.compute_PDE_loss (:746-769)
 * without materialising any (n_inf x M) feature matrix.  x.y on the matrix cores -- bf16 MFMA over
 * 3 (fp32-exact) or 2 bf16 planes of each operand, or FP32 MFMA (gp_h->split) -- and the closed-form
 * derivative features in the epilogue.
 *   points : n_inf x kp rows (X, t, zero pad); columns d+1 .. kp-1 MUST be zero (row sums are taken unmasked and the
 *            last three columns carry folded constants inside the kernel)
 *   out4   : n_inf x 4 = (u_hat, div_x u_hat, eps_PDE, dt u_hat)
 *   lap    : n_inf Laplacian of u_hat, may be NULL
 */
int scasml_gp_eval(const scasml_gp_model *gp_h, const float *points, int64_t n_inf,
                   float *out4, float *lap, void *stream);

/* Same, for the site-major point buffer of scasml_picard_tree: rows [s*rows_per_site, (s+1)*rows_per_site) are
 * tree site s; site_u_only[s] (device bytes, from scasml_plan_site_kinds): 0 = every output needed; 1 = only u_hat is
 * consumed (the root, ScaSML.py:303) -- its rows get u_hat only (div, eps, dt = 0), which halves the epilogue work;
 * 3 = only u_hat AND every row of the site has t = T_terminal of scasml_gp_pack exactly (terminal samples, ScaSML.py:61):
 * a quarter of the full epilogue, used when rows_per_site is a multiple of 32 (otherwise treated as 1); 4 = u_hat and
 * div_x u_hat are consumed (eps, dt = 0); 2 = skip the site. */
int scasml_gp_eval_sites(const scasml_gp_model *gp_h, const float *points, int64_t n_inf, int64_t rows_per_site,
                         const uint8_t *site_u_only, float *out4, void *stream);

/* Host helper: fill kinds_h[0 .. points_per_root) with 3 for terminal samples (u_hat only, at t = T), 1 for the trailing
 * root row (u_hat only, at the root's own time), 0 for the Euler-Maruyama sites of level-0 terms (u_hat, div, eps_PDE needed),
 * 4 for those of level l > 0 terms (u_hat and div only: eps_PDE enters the sum in the level-0 term alone, ScaSML.py:274-280) and
 * 2 for sites of root-call units this rank does not own under Monte-Carlo sample sharding (unit_owner_h[unit] !=
 * rank, or unit % world != rank when unit_owner_h is NULL -- the same rule as scasml_rng): scasml_picard_tree
 * neither writes nor reads those rows and scasml_gp_eval_sites skips workgroups that lie entirely inside them.
 * world = 1: no site is skipped. */
int scasml_plan_site_kinds(const scasml_plan *plan_h, int32_t rank, int32_t world, const uint8_t *unit_owner_h, uint8_t *kinds_h);

/* Monte-Carlo sample sharding: the shardable units of the ROOT call are its mg[n] terminal
 * samples and, for every level l < n and every node (m, k) of its mc sample paths (q nodes each), the ADDENDS that node
 * contributes to the root's sums: "+" = w_k f(P_k, uz(l)) with the level-l subtree below the node (at l = 0 also the
 * surrogate's residual term) and, for l > 0, "-" = the level-(l-1) subtree's term -- enumerated node by node, "+" first.
 * A node's state X_k, W_k is read back (ACCUMULATE), drawn directly (full history) or replayed from the path's own cheap
 * draws (solvers/MLP.py:215-225), and its surrogate values are evaluated by whoever owns either addend, so neither a
 * path nor a node need stay on one rank.  Costs are unequal (at n = rho = 3: 27 terminal samples, 20 addends of 1 site,
 * 9 of 10, 6 of 79 and 6 of 9 sites), so they are dealt by cost -- longest processing time first onto the least loaded rank.
 * site_cost_h (host, may be NULL): {Euler-Maruyama site of a level-0 term (surrogate kind 0), of a level l > 0 term (kind 4: u_hat and
 * div only), terminal site (kind 3: u_hat only), one replayed path step}, relative; NULL = {1, 1, 0.6, 0} (ABI <= 6).  ABI 7: the weights
 * are the caller's (the as-coded surrogate at the headline shape measures {1, 0.62, 0.50, 0.04}: a rank that drew nine u_hat-and-div sites
 * for free was 7 % slower than its dealt load said); a "-" addend stays with its node's "+" addend unless the least loaded rank ends lower
 * even after evaluating the node's point a second time (the zero-cost "-" addends of level-1 nodes therefore never move a node point to a
 * second rank); load_h includes the second evaluations and, for the quadrature solvers, the steps a rank replays up to the last node of a
 * path it owns an addend of.  Dealt-load max / mean with the measured weights: 1.00 / 1.00 / 1.01 at 2 / 4 / 8 ranks (whole paths as
 * units, as up to ABI 5: 1.00, 1.60, 3.19); full history n = 4, M = 3: 1.00, 1.00, 1.2 (whole samples: 1.43).
 * Returns the number of units (<0 on error); fills owner_h[0 .. units) (needs capacity >= units, world <= 255) and, if not NULL,
 * load_h[0 .. world) with the cost dealt to every rank.  Philox is keyed by tree site, so the sum over ranks does not depend on the dealing. */
int32_t scasml_plan_deal_units(const scasml_plan *plan_h, int32_t world, const double *site_cost_h, uint8_t *owner_h, int32_t capacity, double *load_h);

/* Host only: the workgroup -> tile map of the 128 x 128 FP64 update kernels (the trailing update of scasml_cholesky, the substitution
 * updates of scasml_cholesky_inverse / scasml_trsm_*, scasml_gemm_nt_sub; models/GP.py:260-267, 599 are what they replace).  Workgroups
 * b and b + 8 share an XCD's L2, 32 at a time: the 1-D grid deals every XCD super-tiles of 8 x 4 tiles (8 row panels + 4 column panels
 * feed 32 tiles).  tri != 0 (needs nti == ntj): the super-tiles of the lower triangle only.  scasml_tile_order_blocks: workgroups to
 * launch (< 0 on error); scasml_tile_order: 1 and the tile of workgroup `block` in ti_h / tj_h (tiles above the diagonal of a triangular
 * launch are returned and skipped by the kernel), 0 (and -1, -1) when that workgroup has none, < 0 on error. */
int64_t scasml_tile_order_blocks(int64_t nti, int64_t ntj, int32_t tri);
int scasml_tile_order(int64_t block, int64_t nti, int64_t ntj, int32_t tri, int64_t *ti_h, int64_t *tj_h);

/* Full gradient of the posterior mean, n_inf x (d+1), time last: GP.compute_gradient (:673-687). */
int scasml_gp_gradient(const scasml_gp_model *gp_h, const float *points, int64_t n_inf,
                       float *grad, void *stream);

/* 25-block feature Gram K(phi,phi), M x M float64, M = 4*n_dom + n_bdy, block order
 * [u(dom), u(bdy), Lap(dom), dt(dom), div(dom)]: GP.kernel_phi_phi (models/GP.py:182-258). */
int scasml_gp_gram(int32_t d, double a, const float *x_dom, int32_t n_dom, const float *x_bdy, int32_t n_bdy,
                   double *K, void *stream);

/* In-place lower Cholesky of the SPD matrix A + nugget*I (float64, row-major, M x M); the strict
 * upper triangle is zeroed.  Replaces the SVD factor U*sqrt(S+nugget), models/GP.py:260-267.
 * info_dev: one int32 on the device, set to the 1-based index of a non-positive pivot, else 0. */
int scasml_cholesky(double *A, int64_t M, double nugget, int32_t *info_dev, void *stream);

/* Solve L X = B (trans=0) or L^T X = B (trans=1) in place; L lower M x M, B M x nrhs row-major
 * float64: the solves of models/GP.py:439, 533, 599. */
int scasml_trsm_lower(const double *L, int64_t M, double *Bmat, int64_t nrhs, int trans, void *stream);

/* A = (L L^T)^-1 from the lower Cholesky factor L (M x M float64, M a multiple of 32; A is overwritten, any
 * content): the K_p^-1 that GPsolver's loss, gradient and Hessian need (models/GP.py:439, 599).  Uses the
 * triangular structure of L^-1 and the symmetry of the result (about a third of two general solves). */
int scasml_cholesky_inverse(const double *L, int64_t M, double *A, void *stream);

/* Newton iteration on J(sol) = b(sol)^T K_p^-1 b(sol), sol = [z1, z3, z5] (3*n_dom), float64: the pieces of
 * GP.GPsolver's loop (models/GP.py:510-588) that the reference obtains by autodiff of loss_function (:430-444).
 * eq_id / sigma / mu select and parametrise F = -mu z5 - sigma^2/2 z3 - f(z1, sigma z5) (csrc/equations.hpp).
 *   scasml_gp_newton_b       b = [z1, g, z3, F(sol), z5]  (4*n_dom + n_bdy), F = time_der_rep (:705-719)
 *   scasml_gemv              y = A x, A row-major M x M with leading dimension lda (A b, and right_vector = A z, :599)
 *   scasml_gp_newton_system  grad (3*n_dom) and the full Hessian, written into an ldh x ldh buffer whose padding
 *                            beyond 3*n_dom is the identity (ldh a multiple of 32: ready for scasml_cholesky);
 *                            gauss_newton != 0 omits the second-derivative term of F (always positive semidefinite) */
int scasml_gp_newton_b(int32_t eq_id, int32_t d, double sigma, double mu, const double *sol, const double *bdy_g, int32_t n_dom,
                       int32_t n_bdy, double *b, void *stream);
int scasml_gemv(const double *A, int64_t M, int64_t lda, const double *x, double *y, void *stream);
int scasml_gp_newton_system(int32_t eq_id, int32_t d, double sigma, double mu, const double *A, int64_t lda, int32_t n_dom,
                            int32_t n_bdy, const double *sol, const double *Ab, double *grad, double *H, int64_t ldh,
                            int gauss_newton, void *stream);
/* The same Newton iteration without K_p^-1 in memory (distributed fits: products with K_p^-1 are two triangular solves):
 *   scasml_gp_newton_jv    out (4*n_dom + n_bdy) = J v,   J = d b / d sol at `sol`, v of length 3*n_dom
 *   scasml_gp_newton_jtv   out (3*n_dom) = scale * (J^T w  [+ the second-derivative term of F: with Ab = K_p^-1 b and a
 *                          direction v (both may be NULL = omitted): -sigma^2 Ab[F_i] v5_i on z1_i, -sigma^2 Ab[F_i] v1_i on z5_i])
 * so that  grad = jtv(Ab, scale 2)  and  H v = jtv(K_p^-1 jv(v), Ab, v, scale 2). */
int scasml_gp_newton_jv(int32_t eq_id, int32_t d, double sigma, double mu, const double *sol, const double *v, int32_t n_dom, int32_t n_bdy,
                        double *out, void *stream);
int scasml_gp_newton_jtv(int32_t eq_id, int32_t d, double sigma, double mu, const double *sol, const double *w, const double *Ab,
                         const double *v, double scale, int32_t n_dom, int32_t n_bdy, double *out, void *stream);

/* ------------------------------------------------------------------ reference-compat surrogate
 * The reference's surrogate AS CODED differs from the operators it documents:
 * laplacian_op (models/GP.py:28-39) is a 5-index Hutchinson subsample applied to a cyclically shifted argument
 * (:87-105, 119-127, 141-179: t_x = x_t[0] although time is the last column), and every kernel entry is rounded to
 * float16 (:43, 55-179).  These entry points compute exactly that, in float64, for GP(compat="reference"):
 *   idx_h   : the five Hutchinson indices, HOST int32[5], distinct, 0 <= i < d.  The reference draws them with
 *             random.choice(PRNGKey(0), d, (5,), replace=False) -- JAX threefry, not reproducible without JAX -- so they are
 *             an argument.  Index i differentiates along component i of the SHIFTED vector (original coordinate i+1).
 *   round16 : bit 2 (Gram, Gram rows and scasml_gp_eval_compat; the caller vouches that the collocation rows are float16 values) evaluates the nine
 *             Laplacian-free operator pairs on float16 rows through the reference's float16 op sequence -- kappa in float16 arithmetic, its
 *             derivative kernels reverse-mode autodiff through it (models/GP.py:41-85, 107-139) -- instead of one rounding per entry;
 *             bit 0 rounds every kernel entry to float16 (RNE) before it is stored / used; bit 1 (evaluation only) also returns
 *             u_hat and eps_PDE as float16 values, as predict / compute_PDE_loss do (.astype(float16), models/GP.py:671, 769;
 *             eps_PDE is then formed from the rounded u_hat).
 * scasml_gp_gram_compat   K(phi, phi), same block order as scasml_gp_gram              (models/GP.py:182-258)
 * scasml_round16_diag     A[i][i] = float16(A[i][i] + nugget): K_p as right_vector sees it   (:267-268, 599)
 * scasml_round16          v = float16(v) elementwise (time_der_rep(...).astype(float16), :719)
 * scasml_gp_compat_pack   colloc_t[k][j] (float64, leading dimension ldc >= n_dom + n_bdy) = coordinate k of collocation point j
 * scasml_gp_eval_compat   out4 = (u_hat, div_x u_hat, eps_PDE, dt u_hat) and optionally the "Laplacian" of u_hat, as
 *                         scasml_gp_eval, from kernel_x_t_phi_single and the {dt, div, laplacian}_x_t_kernel_x_t_phi rows
 *                         (:326-411, 630-651, 746-769); points: n_inf rows of stride kp floats (X, t, ...). */
int scasml_gp_gram_compat(int32_t d, double a, const float *x_dom, int32_t n_dom, const float *x_bdy, int32_t n_bdy,
                          const int32_t *idx_h, int32_t round16, double *K, void *stream);
int scasml_round16_diag(double *A, int64_t M, int64_t lda, double nugget, void *stream);
int scasml_round16(double *v, int64_t n, void *stream);
int scasml_gp_compat_pack(int32_t d, const float *x_dom, int32_t n_dom, const float *x_bdy, int32_t n_bdy,
                          double *colloc_t, int64_t ldc, void *stream);
int scasml_gp_eval_compat(int32_t d, double a, double sigma_eq, double mu_eq, int32_t eq_id, const double *colloc_t, int32_t n_dom, int32_t n_bdy,
                          int64_t ldc, const double *rv, const int32_t *idx_h, int32_t round16, const float *points,
                          int64_t n_inf, int32_t kp, float *out4, float *lap, void *stream);
/* Full gradient of the as-coded posterior mean, n_inf x (d+1), time last: GP.compute_gradient (:673-687) differentiates
 * dot(kernel_x_t_phi_single(x), right_vector) by autodiff, i.e. THROUGH the float16 casts of the entries (a cast is the identity
 * for the derivative), so the gradient is that of the unrounded shifted-Hutchinson features; round16 != 0 rounds the result
 * (.astype(float16), :687). */
int scasml_gp_gradient_compat(int32_t d, double a, const double *colloc_t, int32_t n_dom, int32_t n_bdy, int64_t ldc, const double *rv,
                              const int32_t *idx_h, int32_t round16, const float *points, int64_t n_inf, int32_t kp, float *grad, void *stream);

/* The same surrogate on the matrix cores -- the form the solvers use (csrc/gp_eval_compat_mfma.hip).  The three pair geometries
 * (aligned, y shifted, x shifted) are one x.y product against three cyclic shifts of the collocation row, the 5-component
 * Hutchinson sums one extra K = 16 product per geometry; entries are rounded with v_cvt_pk_f16_f32 from float32 values, so a
 * rounding decision can differ from scasml_gp_eval_compat's float64 one where the value lies within ~2^-20 of a float16 midpoint.
 * PRECONDITIONS: every collocation coordinate is exactly representable in float16 (the reference's deepxde float16 points are;
 * the caller checks) and |x_k| <= x_bound (0 = 2.0) for every evaluation row, as for scasml_gp_model.split = 22.
 *   scasml_gp_compat_model_floats  size of the packed model (floats) for n_pad = (n_dom + n_bdy) rounded up to SCASML_GP_TILE
 *   scasml_gp_compat_pack_mfma     per (tile, geometry): fp16 A fragments of the shifted rows, the Hutchinson fragment, 8 row constants
 *   scasml_gp_eval_compat_sites    out4 / lap as scasml_gp_eval_compat; rows_per_site / site_kinds as scasml_gp_eval_sites
 *                                  (site_kinds may be NULL: every row gets everything).  A 128-row workgroup runs the geometries
 *                                  its sites need: all three where eps_PDE is consumed, two where only u_hat (and div) are.
 *                                  round16 here: bit 0 = every entry rounded to float16 (the reference's code, models/GP.py:43, 55-179);
 *                                  bit 1 = u_hat and eps_PDE leave as float16 values (:671, 769); with bit 0 OFF the kernel runs the
 *                                  GEOMETRY mode -- the same sixteen entries in the same three geometries, not rounded, so that the four
 *                                  sums factor per geometry (GP(compat="reference-geometry"); 1.5x faster, relative L2 of the solvers
 *                                  within 3e-5 of the as-coded mode on the reference's experiments) -- and bit 2 then lets the evaluation
 *                                  point enter x.y as ONE float16 plane (kappa moves by <= 3.5e-5 relative; half the MFMAs).  3 = the
 *                                  reference's code, 6 = the geometry mode as GP uses it; bit 2 with bit 0 is refused. */
int64_t scasml_gp_compat_model_floats(int32_t d, int32_t n_pad);
int scasml_gp_compat_pack_mfma(int32_t d, float a, const float *x_dom, int32_t n_dom, const float *x_bdy, int32_t n_bdy,
                               const double *rv, const int32_t *idx_h, float *model_out, void *stream);
int scasml_gp_eval_compat_sites(int32_t d, float a, float sigma_eq, float mu_eq, int32_t eq_id, const float *model, int32_t n_dom,
                                int32_t n_bdy, const int32_t *idx_h, int32_t round16, float x_bound, const float *points, int64_t n_inf,
                                int64_t rows_per_site, const uint8_t *site_kinds, float *out4, float *lap, void *stream);
/* ABI 7.  The same evaluation over a LIST of sites: site_order[0 .. n_listed) (DEVICE int32, distinct site indices < n_inf / rows_per_site) are
 * the sites to evaluate, in launch order; rows_per_site must be a multiple of 32 (a 32-row wavefront tile is then one site) and n_inf a whole
 * number of sites.  The grid covers the listed sites only: a Monte-Carlo sample-sharded rank launches nothing over the sites of other ranks
 * (scasml_gp_eval_compat_sites launches a workgroup per 128 rows of the whole buffer and returns at once from 85 % of them on each of 8 ranks),
 * and a list in falling cost order (site kind 0, then 4, then 3 and 1: solvers/_picard.py site_order) leaves the cheapest workgroups to the
 * grid's last, partly filled round.  Every row's values are those of scasml_gp_eval_compat_sites bit for bit, whatever the order
 * (tests/test_gpu_compat_mfma.py).  Rows of sites that are not listed are left untouched. */
int scasml_gp_eval_compat_site_list(int32_t d, float a, float sigma_eq, float mu_eq, int32_t eq_id, const float *model, int32_t n_dom,
                                    int32_t n_bdy, const int32_t *idx_h, int32_t round16, float x_bound, const float *points, int64_t n_inf,
                                    int64_t rows_per_site, const uint8_t *site_kinds, const int32_t *site_order, int32_t n_listed,
                                    float *out4, float *lap, void *stream);

/* ABI 7.  Cross-kernel feature rows K(x, phi): GP.kernel_x_t_phi (models/GP.py:271-294; op 0), laplacian_x_t_kernel_x_t_phi (:326-354; op 1),
 * dt_x_t_kernel_x_t_phi (:356-383; op 2), div_x_t_kernel_x_t_phi (:385-411; op 3), dx_t_kernel_x_t_phi (:296-324; op 4) and, with a single
 * row, kernel_x_t_phi_single (:630-651) -- the matrices the reference materialises for predict / compute_gradient / compute_PDE_loss; the hot
 * path contracts them on the fly (scasml_gp_eval*), these entry points exist for callers of the reference's class surface.
 *   x_inf  : n_inf rows of d+1 coordinates (time last), row stride ld_inf floats (the point buffers of this library qualify)
 *   op 0-3 : out[i * ld + col] = (L^op_x L^oy_y kappa)(x_i, y_j), columns in the Gram's order [u(dom), u(bdy), Lap(dom), dt(dom), div(dom)], ld >= M
 *   op 4   : out[(i * M + col) * (d+1) + k] = d/dx_k of the op-0 entry (ld ignored); n_inf * M * (d+1) doubles
 *   surrogate 0: the reference's code (5-index Hutchinson on the shifted argument; round16 as scasml_gp_gram_compat: bit 0 = every entry a
 *   float16 value, bit 2 = the float16 op sequence on float16 rows); 1: the operators it documents (idx_h, round16 ignored).
 * The as-coded rows are the per-pair arithmetic of scasml_gp_gram_compat: at the collocation points themselves they ARE the Gram's rows. */
int scasml_gp_cross_rows(int32_t d, double a, const float *x_dom, int32_t n_dom, const float *x_bdy, int32_t n_bdy, const int32_t *idx_h,
                         int32_t round16, int32_t surrogate, int32_t op, const float *x_inf, int64_t n_inf, int64_t ld_inf, double *out,
                         int64_t ld, void *stream);

/* ------------------------------------------------------------------ block-row distributed Gram / Cholesky / solves
 * For collocation sets whose K(phi, phi) does not fit one GPU (BASELINE configs[4]: 1e5 points, M = 350 000, 980 GB float64)
 * the matrix is cut into block rows of SCASML_DIST_BLOCK feature rows, block row i owned by rank i % world and stored as a
 * row-major panel of its lower-triangle columns; scasml_gp_amd/dist_gp.py sequences these kernels and the RCCL broadcasts /
 * all-gathers between them.  Replaces models/GP.py:182-268 and the solve of :599 at that size.  All float64.
 *   scasml_gp_gram_rows    rows [row0, row0 + nrows) x columns [0, ncols) of K(phi, phi) into out (leading dimension ld),
 *                          block order as scasml_gp_gram
 *   scasml_gemm_nt_sub     C (rows x cols, ldc) -= A (rows x K, lda) * B (cols x K, ldb)^T on the FP64 matrix cores; K % 32 == 0.
 *                          tri_stride > 0: C is a stack of block rows of a lower-triangular matrix -- local row block lb is global
 *                          block row tri_row0 + lb * tri_stride, column block cb is global block column tri_col0 + cb -- and the
 *                          64 x 64 tiles strictly above the block diagonal are skipped
 *   scasml_trsm_right_lt   X (rows x nb, ldx) <- X * L^-T, L lower triangular nb x nb (ldl), nb % 32 == 0: the panel solve
 *   scasml_gemv_sub        trans == 0: y (rows) -= A (rows x cols, lda) x (cols);  trans != 0: y (cols) -= A^T x (rows): one writer per column up
 *                          to 1024 rows, beyond that 64-row groups combined with atomics (the last bits depend on their order)
 *   scasml_gemv_t_sub_ordered   (ABI 7) y (cols) -= A^T x (rows) for any number of rows, bitwise reproducible: at most 64 row groups write their
 *                          partial sums to `scratch` (scasml_gemv_t_ordered_scratch(rows, cols) doubles, the caller's) and one pass adds them in
 *                          fixed order -- the K_p v sweep of the Newton-CG fit (scasml_gp_amd/dist_gp.py matvec) */
#define SCASML_DIST_BLOCK 256
int scasml_gp_gram_rows(int32_t d, double a, const float *x_dom, int32_t n_dom, const float *x_bdy, int32_t n_bdy,
                        int64_t row0, int32_t nrows, int64_t ncols, double *out, int64_t ld, void *stream);
/* The same rows of the as-coded Gram (scasml_gp_gram_compat; models/GP.py:182-258 with the shifted Hutchinson blocks and float16 entries):
 * bit-identical to the rows of the full matrix, so a block-row distributed fit builds the reference's estimator too. */
int scasml_gp_gram_compat_rows(int32_t d, double a, const float *x_dom, int32_t n_dom, const float *x_bdy, int32_t n_bdy, const int32_t *idx_h,
                               int32_t round16, int64_t row0, int32_t nrows, int64_t ncols, double *out, int64_t ld, void *stream);
int scasml_gemm_nt_sub(double *C, int64_t ldc, int64_t rows, int64_t cols, const double *A, int64_t lda, const double *B,
                       int64_t ldb, int64_t K, int64_t tri_row0, int64_t tri_stride, int64_t tri_col0, void *stream);
int scasml_trsm_right_lt(const double *L, int64_t ldl, int64_t nb, double *X, int64_t ldx, int64_t rows, void *stream);
int scasml_gemv_sub(const double *A, int64_t lda, int64_t rows, int64_t cols, const double *x, double *y, int trans, void *stream);
int64_t scasml_gemv_t_ordered_scratch(int64_t rows, int64_t cols);
int scasml_gemv_t_sub_ordered(const double *A, int64_t lda, int64_t rows, int64_t cols, const double *x, double *y, double *scratch,
                              int64_t scratch_elems, void *stream);
/* The same two sweeps over the stacked block rows of a LOWER-TRIANGULAR factor (DistCholesky's panel: local row block lb is global block row
 * tri_row0 + lb * tri_stride, blocks of SCASML_DIST_BLOCK rows; nothing is stored beyond a row's own diagonal block): the row sweep stops at the
 * diagonal block, the transposed sweep starts at the first block row that reaches the column -- half the bytes of the full-width sweeps
 * (K_p v = L (L^T v), models/GP.py:430-444 as used by the matrix-free Newton iteration).  The skipped entries are zeros: the same sums, added in a
 * fixed order (bitwise reproducible between runs). */
int scasml_gemv_sub_tri(const double *A, int64_t lda, int64_t rows, int64_t cols, const double *x, double *y, int64_t tri_row0, int64_t tri_stride, void *stream);
int scasml_gemv_t_sub_ordered_tri(const double *A, int64_t lda, int64_t rows, int64_t cols, const double *x, double *y, double *scratch,
                                  int64_t scratch_elems, int64_t tri_row0, int64_t tri_stride, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SCASML_HIP_H */
