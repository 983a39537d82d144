// Picard-tree kernel: the whole multilevel-Picard recursion of one evaluation point, walked
// depth-first inside one group of lanes of a 64-wide wavefront.
//
// Replaces MLP.uz_solve (solvers/MLP.py:141-274), ScaSML.uz_solve (solvers/ScaSML.py:149-284),
// MLP_full_history.uz_solve (solvers/MLP_full_history.py:64-180) and
// ScaSML_full_history.uz_solve (solvers/ScaSML_full_history.py:75-199).
//
// Mapping (DESIGN.md "picard_tree"): a root point owns G = pow2 >= round_up(d+4,16)/4 lanes; lane
// `gl` of the group holds spatial dims 4gl..4gl+3 of every d-vector (x, X, W, z) in one float4,
// so one Philox4x32 block per lane per path-step yields exactly that lane's four normals.
// 64/G roots share a wavefront and walk the SAME static tree in lock step -- the tree depends
// only on the tables (MLP.py:111-139), never on data, so there is no divergence.  The
// recursion is unrolled at compile time (level is a template parameter): every frame lives in
// VGPRs, sum_i over the dims is a log2(G)-step xor-shuffle, sum over Monte-Carlo samples is a
// register accumulation, and nothing but the root row is read from or written to HBM in
// MODE_MLP.  For ScaSML the same walk runs twice around the batched GP evaluation:
// MODE_GENERATE emits every tree point (coalesced float4 rows), MODE_ACCUMULATE reads those states
// back (recovering the normals from them; it replays Philox only where that would be inaccurate, see
// kReadbackMinVol, and for the full-history draws) and consumes (u_hat, div u_hat, eps_PDE) per point.

#include "picard_tree.hpp"

namespace scasml {

template <int VAR, int MODE, int EQ>
static int launch_level(const TreeArgs &a, int n, dim3 grid, hipStream_t s) {
    switch (n) {
        case 1: hipLaunchKernelGGL((picard_tree_kernel<VAR, MODE, 1, EQ>), grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((picard_tree_kernel<VAR, MODE, 2, EQ>), grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((picard_tree_kernel<VAR, MODE, 3, EQ>), grid, dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((picard_tree_kernel<VAR, MODE, 4, EQ>), grid, dim3(256), 0, s, a); break;
        case 5: hipLaunchKernelGGL((picard_tree_kernel<VAR, MODE, 5, EQ>), grid, dim3(256), 0, s, a); break;
        default: return fail(SCASML_ERR_UNSUPPORTED, "picard_tree: level n=%d outside 1..%d", n, SCASML_MAX_LEVEL);
    }
    return check_launch("picard_tree launch");
}

template <int VAR, int EQ>
static int launch_mode(const TreeArgs &a, int mode, int n, dim3 grid, hipStream_t s) {
    if (a.jk) {   // the reference's own random stream: instantiated in translation units of their own (picard_tree.hpp)
        if (mode != SCASML_MODE_MLP && mode != SCASML_MODE_GENERATE && mode != SCASML_MODE_ACCUMULATE) return fail(SCASML_ERR_ARG, "picard_tree: unknown mode %d", mode);
        if (n < 1 || n > SCASML_MAX_LEVEL) return fail(SCASML_ERR_UNSUPPORTED, "picard_tree: level n=%d outside 1..%d", n, SCASML_MAX_LEVEL);
        return n <= 3 ? launch_tree_jax(a, VAR, mode, EQ, n, grid, s) : launch_tree_jax_deep(a, VAR, mode, EQ, n, grid, s);
    }
    switch (mode) {
        case SCASML_MODE_MLP: return launch_level<VAR, SCASML_MODE_MLP, EQ>(a, n, grid, s);
        // GENERATE evaluates neither f nor g: one instantiation (equation 0) serves every equation
        case SCASML_MODE_GENERATE: return launch_level<VAR, SCASML_MODE_GENERATE, SCASML_EQ_GRAD_DEPENDENT_NONLINEAR>(a, n, grid, s);
        case SCASML_MODE_ACCUMULATE: return launch_level<VAR, SCASML_MODE_ACCUMULATE, EQ>(a, n, grid, s);
    }
    return fail(SCASML_ERR_ARG, "picard_tree: unknown mode %d", mode);
}

__global__ void clip_kernel(float *v, int64_t n, float c, int round16) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float x = clip1(v[i], c);
        v[i] = round16 ? r16(x) : x;
    }
}

__global__ void debug_normals_kernel(uint32_t k0, uint32_t k1, uint32_t stream, uint32_t root0, uint32_t site,
                                     int d, int64_t B, float *out) {
    normal_table_to_lds();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = (d + 3) / 4;
    if (i >= B * nq) return;
    const int64_t b = i / nq;
    const int q = (int)(i % nq);
    const float4 n = normal4((uint32_t)q, site, root0 + (uint32_t)b, stream, k0, k1);
    const float v[4] = {n.x, n.y, n.z, n.w};
    for (int j = 0; j < 4; ++j)
        if (4 * q + j < d) out[b * d + 4 * q + j] = v[j];
}

__global__ void debug_jax_normals_kernel(uint32_t k0, uint32_t k1, uint64_t index0, int64_t count, float *out) {
    jax_table_to_lds();             // the path the tree kernels take: the tabulated transform
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = jax_normal_f16(k0, k1, index0 + (uint64_t)i);
}

}  // namespace scasml

using namespace scasml;

extern "C" int scasml_picard_tree(const scasml_problem *prob, const scasml_plan *plan, int mode, const float *x_t,
                                  int64_t B, int64_t site_stride, scasml_rng rng, float *points, const float *gp_vals, float *out_uz,
                                  float *out_uhat, void *stream) {
    if (!prob || !plan) return fail(SCASML_ERR_ARG, "picard_tree: null argument");
    if (B < 0) return fail(SCASML_ERR_ARG, "picard_tree: negative batch");
    if (site_stride != 0 && site_stride < B) return fail(SCASML_ERR_ARG, "picard_tree: site_stride %lld is smaller than the batch %lld", (long long)site_stride, (long long)B);
    if (B == 0) return 0;
    if (!x_t) return fail(SCASML_ERR_ARG, "picard_tree: x_t is null");
    if (prob->d < 1 || prob->d > SCASML_MAX_DIM)
        return fail(SCASML_ERR_UNSUPPORTED, "picard_tree: d=%d outside 1..%d", prob->d, SCASML_MAX_DIM);
    if (!eq_known(prob->eq_id) && !eq_mlp_only(prob->eq_id)) return fail(SCASML_ERR_UNSUPPORTED, "picard_tree: unknown equation id %d", prob->eq_id);
    if (eq_mlp_only(prob->eq_id) && (mode != SCASML_MODE_MLP || (rng.flags & SCASML_RNG_JAX_STREAM)))
        return fail(SCASML_ERR_UNSUPPORTED, "picard_tree: equation id %d (f of |z|^2) runs in SCASML_MODE_MLP on the Philox stream only", prob->eq_id);
    if (plan->variant != 0 && plan->variant != 1) return fail(SCASML_ERR_ARG, "picard_tree: variant %d", plan->variant);
    if (plan->n < 0 || plan->n > SCASML_MAX_LEVEL)
        return fail(SCASML_ERR_UNSUPPORTED, "picard_tree: level n=%d outside 0..%d", plan->n, SCASML_MAX_LEVEL);
    if (rng.world < 1 || rng.rank < 0 || rng.rank >= rng.world) return fail(SCASML_ERR_ARG, "picard_tree: bad rank/world");
    if (mode == SCASML_MODE_GENERATE && !points) return fail(SCASML_ERR_ARG, "picard_tree: GENERATE needs points");
    if (mode == SCASML_MODE_ACCUMULATE && (!gp_vals || !points))
        return fail(SCASML_ERR_ARG, "picard_tree: ACCUMULATE needs gp_vals and the points emitted by GENERATE");
    if (mode != SCASML_MODE_GENERATE && !out_uz) return fail(SCASML_ERR_ARG, "picard_tree: out_uz is null");
    for (int np = 1; np <= plan->n; ++np)
        for (int l = 0; l < np; ++l) {
            const scasml_term &t = plan->term[np][l];
            if (t.q < 1 || t.q > SCASML_MAX_Q || t.mc < 1) return fail(SCASML_ERR_ARG, "picard_tree: bad term [%d][%d]", np, l);
        }
    hipStream_t s = (hipStream_t)stream;
    if (plan->n == 0) {  // MLP.py:205-207: zeros (ScaSML: u_hat still requested by the caller through gp_eval)
        if (mode != SCASML_MODE_GENERATE) {
            if (hipMemsetAsync(out_uz, 0, sizeof(float) * B * (prob->d + 1), s) != hipSuccess)
                return fail(SCASML_ERR_HIP, "picard_tree: memset failed");
        }
        return 0;
    }
    TreeArgs a;
    a.plan = *plan;
    a.x_t = x_t;
    a.points = points;
    a.gpv = reinterpret_cast<const float4 *>(gp_vals);
    a.out_uz = out_uz;
    a.out_uhat = out_uhat;
    a.B = B;
    a.Bs = site_stride ? site_stride : B;
    a.ppr = (int64_t)plan->sites[plan->n] + 1;
    a.k0 = (uint32_t)(rng.seed & 0xFFFFFFFFu);
    a.k1 = (uint32_t)(rng.seed >> 32);
    a.stream = rng.stream;
    a.root0 = rng.root0;
    a.rank = rng.rank;
    a.world = rng.world;
    a.owner = rng.world > 1 ? rng.unit_owner : nullptr;
    a.crn = (rng.flags & SCASML_RNG_COMPAT_CRN) ? 1 : 0;
    a.f16 = (rng.flags & SCASML_RNG_COMPAT_F16) ? 1 : 0;
    a.jk = (rng.flags & SCASML_RNG_JAX_STREAM) ? rng.jax_keys : nullptr;
    if ((rng.flags & SCASML_RNG_JAX_STREAM) && !rng.jax_keys) return fail(SCASML_ERR_ARG, "picard_tree: SCASML_RNG_JAX_STREAM needs scasml_rng.jax_keys");
    a.d = prob->d;
    a.kp = scasml_point_stride(prob->d);
    a.G = ceil_pow2(a.kp / 4);
    a.logG = 0;
    while ((1 << a.logG) < a.G) ++a.logG;
    a.T = prob->T;
    a.mu = prob->mu;
    a.sigma = prob->sigma;
    a.clip = prob->clip;
    const int rpw = 64 / a.G;
    const int64_t waves = mode == SCASML_MODE_GENERATE ? (B * (a.kp / 4) + 63) / 64 : (B + rpw - 1) / rpw;   // GENERATE: flat (root, quad) lanes
    const int64_t blocks = (waves + 3) / 4;
    if (blocks > 0x7FFFFFFF) return fail(SCASML_ERR_UNSUPPORTED, "picard_tree: batch too large");
    const dim3 grid((unsigned)blocks);
    if (eq_mlp_only(prob->eq_id)) {
        constexpr int EQ2 = SCASML_EQ_QUADRATIC_GRADIENT_REACTION_DIFFUSION;
        return plan->variant == 0 ? launch_level<0, SCASML_MODE_MLP, EQ2>(a, plan->n, grid, s) : launch_level<1, SCASML_MODE_MLP, EQ2>(a, plan->n, grid, s);
    }
    int rc = SCASML_ERR_UNSUPPORTED;
    SCASML_EQ_SWITCH(prob->eq_id, rc = (plan->variant == 0 ? launch_mode<0, EQ>(a, mode, plan->n, grid, s) : launch_mode<1, EQ>(a, mode, plan->n, grid, s)));
    return rc;
}

extern "C" int scasml_debug_jax_normals(uint32_t key0, uint32_t key1, uint64_t index0, int64_t count, float *out, void *stream) {
    if (!out || count < 0) return fail(SCASML_ERR_ARG, "debug_jax_normals: bad argument");
    if (count == 0) return 0;
    hipLaunchKernelGGL(debug_jax_normals_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, key0, key1, index0, count, out);
    return check_launch("debug_jax_normals launch");
}

extern "C" int scasml_clip_round16(float *uz, int64_t count, float clip, int32_t round16, void *stream) {
    if (!uz || count < 0) return fail(SCASML_ERR_ARG, "clip: bad argument");
    if (count == 0) return 0;
    hipLaunchKernelGGL(clip_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, uz, count, clip, round16 ? 1 : 0);
    return check_launch("clip launch");
}

extern "C" int scasml_clip(float *uz, int64_t count, float clip, void *stream) { return scasml_clip_round16(uz, count, clip, 0, stream); }

// the normal transform of the 24-bit word k (a normal depends on the top 24 bits of its Philox word only), for k = k0 .. k0 + n - 1:
// tests/test_gpu_rng.py checks all 2^24 inputs against NumPy
__global__ void debug_transform_kernel(uint32_t k0, int64_t n, float *out) {
    normal_table_to_lds();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = icdf_normal((k0 + (uint32_t)i) << 8, normal_table_lds());
}

extern "C" int scasml_debug_transform(uint32_t k0, int64_t n, float *out, void *stream) {
    if (!out || n < 0 || (uint64_t)k0 + (uint64_t)n > (1ull << 24)) return fail(SCASML_ERR_ARG, "debug_transform: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(debug_transform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, k0, n, out);
    return check_launch("debug_transform launch");
}

extern "C" int scasml_debug_normals(scasml_rng rng, uint32_t site, int32_t d, int64_t B, float *out, void *stream) {
    if (!out || d < 1 || B < 0) return fail(SCASML_ERR_ARG, "debug_normals: bad argument");
    if (B == 0) return 0;
    const int64_t n = B * ((d + 3) / 4);
    hipLaunchKernelGGL(debug_normals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (uint32_t)(rng.seed & 0xFFFFFFFFu), (uint32_t)(rng.seed >> 32), rng.stream, rng.root0, site, d, B, out);
    return check_launch("debug_normals launch");
}
