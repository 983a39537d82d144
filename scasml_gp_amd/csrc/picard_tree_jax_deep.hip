// The Picard-tree kernels on the reference's own random stream (SCASML_RNG_JAX_STREAM, compat_rng = "jax"): levels 4 and 5 (BASELINE configs[3] is n = 4).
// A translation unit of its own so that the build compiles these instantiations beside picard_tree.hip's (picard_tree.hpp).
#include "picard_tree.hpp"

namespace scasml {

template <int VAR, int MODE, int EQ>
static int jax_level(const TreeArgs &a, int n, dim3 grid, hipStream_t s) {
    switch (n) {
        case 4: hipLaunchKernelGGL((picard_tree_kernel<VAR, MODE, 4, EQ, true>), grid, dim3(256), 0, s, a); break;
        case 5: hipLaunchKernelGGL((picard_tree_kernel<VAR, MODE, 5, EQ, true>), grid, dim3(256), 0, s, a); break;
        default: return fail(SCASML_ERR_UNSUPPORTED, "picard_tree: SCASML_RNG_JAX_STREAM level n=%d is not in this translation unit", n);
    }
    return check_launch("picard_tree launch");
}

template <int VAR, int EQ>
static int jax_mode(const TreeArgs &a, int mode, int n, dim3 grid, hipStream_t s) {
    switch (mode) {
        case SCASML_MODE_MLP: return jax_level<VAR, SCASML_MODE_MLP, EQ>(a, n, grid, s);
        // GENERATE evaluates neither f nor g: one instantiation (equation 0) serves every equation
        case SCASML_MODE_GENERATE: return jax_level<VAR, SCASML_MODE_GENERATE, SCASML_EQ_GRAD_DEPENDENT_NONLINEAR>(a, n, grid, s);
        case SCASML_MODE_ACCUMULATE: return jax_level<VAR, SCASML_MODE_ACCUMULATE, EQ>(a, n, grid, s);
    }
    return fail(SCASML_ERR_ARG, "picard_tree: unknown mode %d", mode);
}

int launch_tree_jax_deep(const TreeArgs &a, int variant, int mode, int eq_id, int n, dim3 grid, hipStream_t s) {
    int rc = SCASML_ERR_UNSUPPORTED;
    SCASML_EQ_SWITCH(eq_id, rc = (variant == 0 ? jax_mode<0, EQ>(a, mode, n, grid, s) : jax_mode<1, EQ>(a, mode, n, grid, s)));
    return rc;
}

}  // namespace scasml
