// Equation registry of the kernels: the device (and host) functors behind scasml_problem.eq_id.
//
// Replaces the plug-in surface of equations/equations.py:15-230 (`Equation.f / g / mu / sigma`) for the family the kernels
// cover:   u_t + mu sum_i d_i u + sigma^2/2 Lap u + f(u, s) = 0,   s = sum_i z_i,  z = sigma grad u,   u(T, x) = g(x),
// with constant mu and sigma (they travel in scasml_problem) and f depending on the gradient through its sum only -- which is
// what lets ScaSML use div_x u_hat from the fused GP evaluation instead of the full gradient (picard_tree.hip, f_eval).
// An equation is one specialisation of EqDef below; everything else -- the Picard kernels' f and g, the PDE residual of the
// GP evaluation (models/GP.py:746-769), the collocation operator F and its derivatives in the Newton kernels (:705-743) --
// is generated from it.  To add an equation: add an id to include/scasml_hip.h, a specialisation here, a case to
// SCASML_EQ_SWITCH, and its Python twin in scasml_gp_amd/equations/equations.py (host view) and oracle/equation.py (checker).
#pragma once
#include <hip/hip_runtime.h>

#include "scasml_hip.h"

namespace scasml {

template <class T>
struct FParts {
    T f, fu, fs, fuu, fus, fss;   // f(u, s) and its first and second derivatives
};

template <int EQ>
struct EqDef;

// ---- 0: Grad_Dependent_Nonlinear (equations/equations.py:232-417): f = sigma u s, g = 1 - 1 / (1 + exp(T + sum x))
template <>
struct EqDef<SCASML_EQ_GRAD_DEPENDENT_NONLINEAR> {
    static constexpr bool kGradSquare = false;   // f sees the gradient through sum_i z_i only
    template <class T>
    static __host__ __device__ __forceinline__ T f(T u, T s, T sigma, T d) {
        (void)d;
        return sigma * u * s;                                        // equations.py:303
    }
    template <class T>
    static __host__ __device__ __forceinline__ FParts<T> parts(T u, T s, T sigma, T d) {
        (void)d;
        return {sigma * u * s, sigma * s, sigma * u, T(0), sigma, T(0)};
    }
    // per-coordinate term of the terminal condition and its closing function: g = G(sum_i phi(x_i), T)
    static __device__ __forceinline__ float4 phi(float4 x) { return x; }
    static __device__ __forceinline__ float G(float sum_phi, float T) {   // equations.py:259
        return 1.0f - __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((T + sum_phi) * 1.44269504088896341f));
    }
};

// ---- 1: Cubic_Reaction_Diffusion (no reference counterpart; oracle/equation.py): f = -u (1 - u) (1 + c (1 - 2u)),
//         c = sigma^2 d / 2, mu = 0, same terminal condition; exact solution logistic(t + sum x)
template <>
struct EqDef<SCASML_EQ_CUBIC_REACTION_DIFFUSION> {
    static constexpr bool kGradSquare = false;
    template <class T>
    static __host__ __device__ __forceinline__ T f(T u, T s, T sigma, T d) {
        (void)s;
        const T c = T(0.5) * sigma * sigma * d;
        return -(u * (T(1) - u)) * (T(1) + c * (T(1) - T(2) * u));
    }
    template <class T>
    static __host__ __device__ __forceinline__ FParts<T> parts(T u, T s, T sigma, T d) {
        (void)s;
        const T c = T(0.5) * sigma * sigma * d;
        const T w = u * (T(1) - u), v = T(1) + c * (T(1) - T(2) * u), m = T(1) - T(2) * u;
        return {-w * v, -m * v + T(2) * c * w, T(0), T(2) * v + T(4) * c * m, T(0), T(0)};
    }
    static __device__ __forceinline__ float4 phi(float4 x) { return x; }
    static __device__ __forceinline__ float G(float sum_phi, float T) {
        return 1.0f - __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((T + sum_phi) * 1.44269504088896341f));
    }
};

// ---- 2: Quadratic_Gradient_Reaction_Diffusion (no reference counterpart; oracle/equation.py): the family one step wider, f(u, s1, s2) with
//         s2 = sum_i z_i^2 = |z|^2 -- f = -u (1 - u) (1 + c (1 - 2u)) + (s2 - sigma^2 d (u (1 - u))^2); on the travelling wave z_i = sigma u (1 - u),
//         so the added term vanishes and logistic(t + sum x) stays exact.  Picard kernels without a surrogate only (scasml_hip.h).
template <>
struct EqDef<SCASML_EQ_QUADRATIC_GRADIENT_REACTION_DIFFUSION> {
    static constexpr bool kGradSquare = true;
    template <class T>
    static __host__ __device__ __forceinline__ T f2(T u, T s1, T s2, T sigma, T d) {
        (void)s1;
        const T c = T(0.5) * sigma * sigma * d, w = u * (T(1) - u);
        return -w * (T(1) + c * (T(1) - T(2) * u)) + (s2 - sigma * sigma * d * w * w);
    }
    static __device__ __forceinline__ float4 phi(float4 x) { return x; }
    static __device__ __forceinline__ float G(float sum_phi, float T) {
        return 1.0f - __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((T + sum_phi) * 1.44269504088896341f));
    }
};

// equations every kernel family covers (Picard tree in all modes, GP training, fused evaluation) / those of the surrogate-free Picard tree alone
inline bool eq_mlp_only(int eq_id) { return eq_id == SCASML_EQ_QUADRATIC_GRADIENT_REACTION_DIFFUSION; }
inline bool eq_known(int eq_id) { return eq_id == SCASML_EQ_GRAD_DEPENDENT_NONLINEAR || eq_id == SCASML_EQ_CUBIC_REACTION_DIFFUSION; }

// run `stmt` with EQ bound to the compile-time id
#define SCASML_EQ_SWITCH(eq_id, stmt)                                                              \
    switch (eq_id) {                                                                               \
        case SCASML_EQ_GRAD_DEPENDENT_NONLINEAR: { constexpr int EQ = SCASML_EQ_GRAD_DEPENDENT_NONLINEAR; stmt; } break; \
        case SCASML_EQ_CUBIC_REACTION_DIFFUSION: { constexpr int EQ = SCASML_EQ_CUBIC_REACTION_DIFFUSION; stmt; } break; \
        default: break;                                                                            \
    }

// f(u, s) by runtime id (once-per-point uses: the PDE residual at the end of the GP evaluation)
template <class T>
__host__ __device__ __forceinline__ T eq_f(int eq_id, T u, T s, T sigma, T d) {
    return eq_id == SCASML_EQ_CUBIC_REACTION_DIFFUSION ? EqDef<SCASML_EQ_CUBIC_REACTION_DIFFUSION>::f(u, s, sigma, d)
                                                       : EqDef<SCASML_EQ_GRAD_DEPENDENT_NONLINEAR>::f(u, s, sigma, d);
}
template <class T>
__host__ __device__ __forceinline__ FParts<T> eq_parts(int eq_id, T u, T s, T sigma, T d) {
    return eq_id == SCASML_EQ_CUBIC_REACTION_DIFFUSION ? EqDef<SCASML_EQ_CUBIC_REACTION_DIFFUSION>::parts(u, s, sigma, d)
                                                       : EqDef<SCASML_EQ_GRAD_DEPENDENT_NONLINEAR>::parts(u, s, sigma, d);
}

// The GP's collocation operator u_t = F(z1, z3, z5), z1 = u, z3 = Lap u, z5 = div u (models/GP.py:705-719), with derivatives
struct FOp {
    double F, d1, d3, d5, F11, F15, F55;
};
__host__ __device__ __forceinline__ FOp eq_F(int eq_id, double z1, double z3, double z5, double mu, double sigma, double d) {
    const FParts<double> p = eq_parts<double>(eq_id, z1, sigma * z5, sigma, d);
    return {-mu * z5 - 0.5 * sigma * sigma * z3 - p.f, -p.fu, -0.5 * sigma * sigma, -mu - sigma * p.fs,
            -p.fuu, -sigma * p.fus, -sigma * sigma * p.fss};
}

}  // namespace scasml
