"""Build libscasml_hip.so (hipcc, gfx950) in-tree.  Used by __graft_entry__.build()."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libscasml_hip.so")
SOURCES = ["plan_host.cpp", "picard_tree_jax_deep.hip", "picard_tree.hip", "picard_tree_jax.hip", "gp_eval.hip", "gp_eval_bf16.hip", "gp_train.hip", "gp_compat.hip", "gp_eval_compat_mfma.hip", "dist_linalg.hip"]
# -ffp-contract=off: the RNG transform is specified in separately rounded IEEE mul/add
# (philox_normal.hpp); every fused multiply-add elsewhere is written as fmaf() explicitly.
# -fno-slp-vectorize: hipcc otherwise packs the scalar f32 epilogue into v_pk_fma_f32 / v_pk_mul_f32 plus
# v_mov shuffles, which costs issue slots on gfx950 (packed f32 is not faster than two scalar ops here).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-I" + os.path.join(ROOT, "include")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(ROOT, "include", "scasml_hip.h"))
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            # plan_host.cpp is plain C++ (host only): compiled AS C++ (-x c++: hipcc would otherwise treat a .cpp as HIP) without an offload
            # architecture, so nothing in it can depend on HIP -- the same file builds with g++ under the sanitizers (tests/test_host_sanitizers.py)
            flags = FLAGS if src.endswith(".hip") else [f for f in FLAGS if not f.startswith("--offload-arch")] + ["-x", "c++"]
            jobs.append([hipcc] + flags + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        list(ex.map(run, jobs))
    if force or jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build_library(verbose=True))
