"""ctypes binding of libscasml_hip.so (include/scasml_hip.h).  Fails loudly when the library
is missing -- the product has no CPU path."""
import ctypes as C
import os

MAX_LEVEL = 5
MAX_Q = 6
MAX_DIM = 252
GP_TILE = 32
DIST_BLOCK = 256
ABI_VERSION = 7

MODE_MLP, MODE_GENERATE, MODE_ACCUMULATE = 0, 1, 2
RNG_COMPAT_CRN = 1
RNG_COMPAT_F16 = 2
RNG_JAX_STREAM = 4
EQ_GRAD_DEPENDENT_NONLINEAR = 0
EQ_CUBIC_REACTION_DIFFUSION = 1
EQ_QUADRATIC_GRADIENT_REACTION_DIFFUSION = 2      # f(u, sum z, |z|^2): surrogate-free Picard kernels only


class Problem(C.Structure):
    _fields_ = [("d", C.c_int32), ("eq_id", C.c_int32), ("T", C.c_float), ("mu", C.c_float),
                ("sigma", C.c_float), ("clip", C.c_float)]


class Rng(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("stream", C.c_uint32), ("root0", C.c_uint32),
                ("rank", C.c_int32), ("world", C.c_int32), ("flags", C.c_uint32), ("reserved", C.c_uint32),
                ("unit_owner", C.c_void_p), ("jax_keys", C.c_void_p)]


class Term(C.Structure):
    _fields_ = [("q", C.c_int32), ("mc", C.c_int32), ("sites_l", C.c_int32), ("sites_lm1", C.c_int32),
                ("dfrac", C.c_float * MAX_Q), ("cfrac", C.c_float * MAX_Q),
                ("wfrac", C.c_float * MAX_Q), ("dplus", C.c_float * MAX_Q)]


class Plan(C.Structure):
    _fields_ = [("variant", C.c_int32), ("n", C.c_int32), ("mg", C.c_int32 * (MAX_LEVEL + 1)),
                ("sites", C.c_int32 * (MAX_LEVEL + 1)), ("term", (Term * MAX_LEVEL) * (MAX_LEVEL + 1))]


class GpModel(C.Structure):
    _fields_ = [("d", C.c_int32), ("n_dom", C.c_int32), ("n_bdy", C.c_int32), ("n_pad", C.c_int32),
                ("kp", C.c_int32), ("split", C.c_int32), ("a", C.c_float), ("sigma_eq", C.c_float),
                ("mu_eq", C.c_float), ("eq_id", C.c_int32),
                ("colloc", C.c_void_p), ("colloc_frag", C.c_void_p), ("colloc_bf16", C.c_void_p), ("colloc_is_f16", C.c_int32),
                ("coef", C.c_void_p), ("x_bound", C.c_float), ("reserved", C.c_int32)]


_STRUCTS = (Problem, Rng, Term, Plan, GpModel)
_LIB = None
# SCASML_HIP_LIB: another build of the same library (development: ablation builds of one kernel)
LIB_PATH = os.environ.get("SCASML_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libscasml_hip.so")

# name -> (restype, argtypes); every symbol include/scasml_hip.h declares
SIGNATURES = {
    "scasml_abi_version": (C.c_int, []),
    "scasml_last_error": (C.c_char_p, []),
    "scasml_sizeof": (C.c_size_t, [C.c_int]),
    "scasml_points_per_root": (C.c_int64, [C.POINTER(Plan)]),
    "scasml_point_stride": (C.c_int32, [C.c_int32]),
    "scasml_picard_tree": (C.c_int, [C.POINTER(Problem), C.POINTER(Plan), C.c_int, C.c_void_p, C.c_int64, C.c_int64, Rng,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "scasml_clip": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "scasml_clip_round16": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_int32, C.c_void_p]),
    "scasml_debug_normals": (C.c_int, [Rng, C.c_uint32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "scasml_debug_jax_normals": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p]),
    "scasml_debug_transform": (C.c_int, [C.c_uint32, C.c_int64, C.c_void_p, C.c_void_p]),
    "scasml_normal_table_rows": (C.c_int32, []),
    "scasml_normal_table": (C.c_int, [C.c_void_p, C.c_int32]),
    "scasml_gp_plane_halfwords": (C.c_int64, [C.c_int32, C.c_int32]),
    "scasml_gp_coef_floats": (C.c_int64, [C.c_int32]),
    "scasml_gp_pack": (C.c_int, [C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "scasml_gp_eval": (C.c_int, [C.POINTER(GpModel), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "scasml_gp_eval_sites": (C.c_int, [C.POINTER(GpModel), C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "scasml_plan_site_kinds": (C.c_int, [C.POINTER(Plan), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "scasml_plan_deal_units": (C.c_int32, [C.POINTER(Plan), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "scasml_tile_order_blocks": (C.c_int64, [C.c_int64, C.c_int64, C.c_int32]),
    "scasml_tile_order": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "scasml_gp_gradient": (C.c_int, [C.POINTER(GpModel), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "scasml_gp_gram": (C.c_int, [C.c_int32, C.c_double, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "scasml_gp_newton_b": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "scasml_gemv": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "scasml_gp_newton_system": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "scasml_gp_gram_compat": (C.c_int, [C.c_int32, C.c_double, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                        C.c_void_p, C.c_void_p]),
    "scasml_round16_diag": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_void_p]),
    "scasml_round16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "scasml_gp_compat_pack": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "scasml_gp_eval_compat": (C.c_int, [C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p,
                                        C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "scasml_gp_gradient_compat": (C.c_int, [C.c_int32, C.c_double, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                            C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "scasml_gp_compat_model_floats": (C.c_int64, [C.c_int32, C.c_int32]),
    "scasml_gp_compat_pack_mfma": (C.c_int, [C.c_int32, C.c_float, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    "scasml_gp_eval_compat_sites": (C.c_int, [C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                              C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "scasml_gp_eval_compat_site_list": (C.c_int, [C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                                  C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                                  C.c_void_p, C.c_void_p]),
    "scasml_gp_cross_rows": (C.c_int, [C.c_int32, C.c_double, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "scasml_gp_gram_rows": (C.c_int, [C.c_int32, C.c_double, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int64,
                                      C.c_void_p, C.c_int64, C.c_void_p]),
    "scasml_gp_gram_compat_rows": (C.c_int, [C.c_int32, C.c_double, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64,
                                             C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "scasml_gemm_nt_sub": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                     C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "scasml_gp_newton_jv": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "scasml_gp_newton_jtv": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                       C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "scasml_trsm_right_lt": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "scasml_gemv_sub": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "scasml_gemv_t_ordered_scratch": (C.c_int64, [C.c_int64, C.c_int64]),
    "scasml_gemv_t_sub_ordered": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "scasml_gemv_sub_tri": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "scasml_gemv_t_sub_ordered_tri": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                                C.c_void_p]),
    "scasml_cholesky": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p]),
    "scasml_cholesky_inverse": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "scasml_trsm_lower": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
}


class ScasmlError(RuntimeError):
    pass


def load():
    """Load the library once; raise if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch bundles its own libamdhip64; import it FIRST so this library binds to the same HIP
    # runtime instance (loading the system copy first leaves two runtimes in one process and
    # ours then sees no device).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ScasmlError(
            "libscasml_hip.so is not built (%s missing). Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `python -m scasml_gp_amd._build`; "
            "scasml_gp_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if a declared symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.scasml_abi_version() != ABI_VERSION:
        raise ScasmlError("libscasml_hip.so ABI %d != binding ABI %d" % (lib.scasml_abi_version(), ABI_VERSION))
    for which, st in enumerate(_STRUCTS):
        if lib.scasml_sizeof(which) != C.sizeof(st):
            raise ScasmlError("struct %s: library says %d bytes, binding %d" % (st.__name__, lib.scasml_sizeof(which), C.sizeof(st)))
    _LIB = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise ScasmlError("%s failed (%d): %s" % (what, rc, load().scasml_last_error().decode()))


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise ScasmlError("scasml_gp_amd needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU path")
    return torch


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
