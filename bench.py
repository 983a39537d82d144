#!/usr/bin/env python
"""Headline benchmark: Euler-Maruyama path-steps/s (+ relative L2 error) of
Grad_Dependent_Nonlinear d=100, solvers.ScaSML at level n = rho = 3 (BASELINE.json configs[2]).

One "step" = one ScaSML.uz_solve pass (generate points -> fused GP evaluation -> Picard
accumulation) over a device-resident batch of B synthetic evaluation points per GPU.
Multi-GPU: one process per GPU, evaluation points (independent objects) are sharded across
ranks -- no data-path collective; weak scaling.  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N ...            (plain invocation: spawns the N ranks itself, before anything touches a GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

With N > 1 the line also carries "samples_sharding": the north-star split -- the Monte-Carlo units of the root call dealt
over `sample_ranks` ranks by cost (scasml_plan_deal_units), ONE RCCL all-reduce of the (B, 1+d) partial estimators per
step -- timed on the same workload right after the root-sharded leg (strong scaling: the B roots of one GPU are shared).
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md chip table (spec)
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32, MI355X_MICROARCH.md "Peak FP32 (matrix)"
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X_MICROARCH.md chip table


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--d", type=int, default=100)
    ap.add_argument("--level", type=int, default=3, help="n = rho")
    ap.add_argument("--batch", type=int, default=None, help="evaluation points per GPU (default 2^14; 2^20 for --solver mlp)")
    ap.add_argument("--solver", choices=["scasml", "mlp"], default="scasml")
    ap.add_argument("--variant", choices=["quad", "fh"], default="quad", help="quadrature (MLP/ScaSML) or full history")
    ap.add_argument("--M", type=int, default=3, help="sample base of the full-history solvers")
    ap.add_argument("--train-domain", type=int, default=1000)
    ap.add_argument("--train-boundary", type=int, default=200)
    ap.add_argument("--compat", choices=["reference", "reference-geometry", "none"], default="reference",
                    help="reference (default): the surrogate the reference's code builds (shifted 5-index Hutchinson features, float16 entries; "
                         "GP(compat='reference'), matrix-core kernel gp_eval_compat_mfma); reference-geometry: the same fit evaluated without the "
                         "per-entry float16 roundings (factored sums, one point plane); none: the operators it documents (gp_eval_bf16)")
    ap.add_argument("--rng", choices=["philox", "jax"], default="philox",
                    help="jax: the solvers draw the reference's own random stream on the device (compat_rng='jax': jax.random's float16 normals and "
                         "uniform times under its key schedule) -- with --compat-f16 the mode that reproduces the reference's logged numbers")
    ap.add_argument("--compat-f16", action="store_true", help="the reference's solver-level float16 casts (g, f, every uz_solve return)")
    ap.add_argument("--cpu-sample", type=int, default=64, help="roots of the same workload timed on the CPU oracle")
    ap.add_argument("--shard", choices=["roots", "samples"], default="roots",
                    help="roots: each rank its own B roots, no collective (weak scaling, default); samples: every rank the same "
                         "B roots and 1/world of the Monte-Carlo units of the root call, ONE all-reduce of the partial estimators "
                         "(strong scaling; BASELINE.json north_star)")
    ap.add_argument("--max-imbalance", type=float, default=1.15,
                    help="samples leg: use the largest number of sample ranks (a divisor of the rank count) whose dealt load "
                         "max/mean stays below this; the remaining factor shards roots")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-logs-check", action="store_true", help="skip the d = 20 runs on the reference's own random stream (about a second)")
    ap.add_argument("--no-gp-train-large", action="store_true",
                    help="skip the M = 34 999 leg of the gp_train block (d = 250, 8333 + 1667 collocation points, ~30 GB, ~15 s)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development only: all ranks share cuda:0 and rendezvous over gloo (a 1-GPU box cannot host RCCL ranks)")
    return ap.parse_args()


def rel_l2(sol, exact):
    """tests/SimpleUniform.py:110-136: NaN-masked ||sol - exact||_2 / ||exact||_2."""
    sol, exact = np.asarray(sol, dtype=np.float64).ravel(), np.asarray(exact, dtype=np.float64).ravel()
    m = ~(np.isnan(sol) | np.isnan(exact))
    return float(np.linalg.norm(sol[m] - exact[m]) / np.linalg.norm(exact[m]))


def cpu_baseline(args, eq, gp, eng, n, par, x_t, x_dev, x_dom, x_bdy, steps_exec, B):
    """Time the oracle restatement (NumPy float64) on a bounded sample of the same workload, same inputs and
    Philox streams, and report the GPU-vs-CPU difference on that sample: the relative L2 error of both against the exact
    solution (tests/SimpleUniform.py:134-136) and their difference (north_star: within 1e-3)."""
    from oracle.equation import GradDependentNonlinear
    from oracle.gp import OracleGP
    from oracle.gp_compat import OracleGPCompat
    from oracle.mlp import PicardOracle
    d = args.d
    oeq = GradDependentNonlinear(d + 1)
    ogp = None
    if gp is not None:
        ogp = OracleGPCompat(oeq, gp.laplacian_idx, round_factor=False) if gp.compat == "reference" else OracleGP(oeq)
        ogp.x_t_domain = np.asarray(x_dom, dtype=np.float64)     # same trained surrogate as the GPU run
        ogp.x_t_boundary = np.asarray(x_bdy, dtype=np.float64)
        ogp.N_domain, ogp.N_boundary = len(x_dom), len(x_bdy)
        ogp.phi_dim = 4 * len(x_dom) + len(x_bdy)
        ogp.right_vector = gp.right_vector
    ns = min(args.cpu_sample if gp is not None else 4096 * args.cpu_sample, B)   # ~10-30 s of CPU work either way
    ora = PicardOracle(oeq, args.variant, gp=ogp, seed=0, stream=99, compat_f16=args.compat_f16, jax_stream=args.rng == "jax")
    ora.jax_splits = eng.jax_splits                          # the replay below (stream_id given) reads the solver's key where it stands
    t0 = time.perf_counter()
    uz_cpu = ora.uz_solve(n, par, x_t[:ns])
    t_cpu = time.perf_counter() - t0
    uz_gpu, uhat_gpu, _ = eng.solve(n, par, x_dev[:ns], stream_id=99)
    uz_gpu = uz_gpu.cpu().numpy().astype(np.float64)
    u_cpu, u_gpu = uz_cpu[:, 0], uz_gpu[:, 0]
    if gp is not None:                                     # u_solve = u_hat + u_breve (ScaSML.py:300-304)
        u_cpu = u_cpu + ogp.predict(np.asarray(x_t[:ns], dtype=np.float64))[:, 0]
        u_gpu = u_gpu + uhat_gpu.cpu().numpy().astype(np.float64)
    exact = oeq.exact_solution(np.asarray(x_t[:ns], dtype=np.float64))[:, 0]
    rel_cpu, rel_gpu = rel_l2(u_cpu, exact), rel_l2(u_gpu, exact)
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count()
    diff = np.abs(uz_gpu - uz_cpu)
    return {"value": round(ns * steps_exec / t_cpu, 1), "unit": "path-steps/s", "cores": threads, "kind": "port",
            "sample": "%d of the %d roots, same inputs and Philox streams, NumPy float64 oracle (oracle/mlp.py%s), %.1f s"
                      % (ns, B, (" + oracle/gp_compat.py" if gp.compat == "reference" else " + oracle/gp.py") if gp is not None else "", t_cpu),
            "rel_l2_gpu": round(rel_gpu, 6), "rel_l2_cpu": round(rel_cpu, 6), "abs_diff": round(abs(rel_gpu - rel_cpu), 7),
            "abs_diff_bound_north_star": 1e-3,
            "max_abs_diff_u": float(np.nanmax(diff[:, 0])), "max_abs_diff_uz": float(np.nanmax(diff)),
            "frac_elements_beyond_1e-4": round(float((diff > 1e-4).mean()), 5),
            "note": "uz is clipped to +-%g; with the as-coded surrogate u_hat and eps_PDE are float16 VALUES, so one kernel entry whose float16 "
                    "rounding is decided on a float32 value here and a float64 value there moves u_hat by a float16 ulp (2.4e-4..4.9e-4) and a z "
                    "component by that times N / (MC delta_t)" % float(eng.problem().clip) if (gp is not None and gp.compat == "reference") else None}


def kernel_source_sha1(files):
    """Hash of the sources a kernel is built from: PMC summaries under profiles/ carry the hash of the code they were taken on, and
    a summary of other code is not quoted."""
    import hashlib
    h = hashlib.sha1()
    for f in files:
        h.update(open(os.path.join(ROOT, "scasml_gp_amd", "csrc", f), "rb").read())
    return h.hexdigest()


GP_EVAL_SOURCES = {"reference": ["gp_eval_compat_mfma.hip", "gp_mfma16.hpp", "gp_common.hpp"],
                   "reference-geometry": ["gp_eval_compat_mfma.hip", "gp_mfma16.hpp", "gp_common.hpp"], "none": ["gp_eval_bf16.hip", "gp_mfma16.hpp", "gp_common.hpp"]}
PICARD_SOURCES = ["picard_tree.hip", "picard_tree.hpp", "philox_normal.hpp", "equations.hpp"]


FP64_MFMA_PEAK_TFLOPS = 78.6    # v_mfma_f64_16x16x4_f64 dense, MI355X_MICROARCH.md / SURVEY.md 8(d)


def gp_train_block(d, n_dom, n_bdy, compat=None):
    """GP training stages (models/GP.py:182-268, 487-604) with their rooflines: Gram, Cholesky (M^3/3 flop), K_p^-1 from the
    factor (2 M^3 / 3), the Newton iteration; HIP events per stage."""
    import torch
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    eq = Grad_Dependent_Nonlinear(d + 1)
    st = np.random.get_state()
    np.random.seed(1234)
    dom, bdy = eq.generate_data(n_dom, n_bdy)
    np.random.set_state(st)
    out = None
    for rep in range(2):                                  # first pass warms code objects and the allocator
        gp = GP_Grad_Dependent_Nonlinear(eq, compat=compat)
        gp.profile = rep == 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gp.GPsolver(dom, bdy, GN_steps=20)
        torch.cuda.synchronize()
        fit_s = time.perf_counter() - t0
        if rep == 1:
            M = gp.phi_dim
            ms = gp.stage_ms
            chol_tf = M ** 3 / 3.0 / (ms["cholesky"] * 1e-3) / 1e12
            inv_tf = 2.0 * M ** 3 / 3.0 / (ms["inverse"] * 1e-3) / 1e12
            N = n_dom + n_bdy
            gram_tf = (2.0 * N * N * (d + 1) + 8.0 * M * M) / (ms["gram"] * 1e-3) / 1e12
            out = {"d": d, "collocation": "%d+%d" % (n_dom, n_bdy), "surrogate": "as coded (compat='reference')" if compat else "documented operators", "M": M, "K_gb_f64": round(M * M * 8 / 1e9, 2), "fit_s": round(fit_s, 3),
                   "newton_steps": len(gp.loss_history) - 1,
                   "gram_ms": round(ms["gram"], 3), "gram_tflops": round(gram_tf, 2),
                   "cholesky_ms": round(ms["cholesky"], 3), "cholesky_tflops": round(chol_tf, 2),
                   "cholesky_frac_of_fp64_mfma_peak": round(chol_tf / FP64_MFMA_PEAK_TFLOPS, 4),
                   "inverse_ms": round(ms["inverse"], 3), "inverse_tflops": round(inv_tf, 2),
                   "inverse_frac_of_fp64_mfma_peak": round(inv_tf / FP64_MFMA_PEAK_TFLOPS, 4),
                   "peak_fp64_mfma_tflops": FP64_MFMA_PEAK_TFLOPS}
        del gp
        torch.cuda.empty_cache()
    return out


def spawn_ranks(n):
    """Plain `python bench.py --gpus N`: start the N ranks as child processes (one per GPU, rendezvous on 127.0.0.1) and
    return the worst exit code.  Nothing in this parent has touched a GPU (importing torch does not)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    return max(abs(p.wait()) for p in procs)


def sample_split(eng, n, par, world, max_imbalance):
    """(sample ranks S, root groups G, imbalance at S, imbalance if all `world` ranks shared the samples): the largest
    divisor S of `world` whose cost-dealt load max/mean stays under the bound."""
    from scasml_gp_amd.solvers._picard import deal_units
    plan = eng.plan(n, par)
    imb = {}
    for s in range(1, world + 1):
        if world % s == 0:
            load = deal_units(plan, s)[1]
            imb[s] = float(load.max() / load.mean())
    best = max(s for s, v in imb.items() if v <= max_imbalance or s == 1)
    return best, world // best, imb[best], imb[world]


def reference_logs_check():
    """The reference's own experiment at d = 20 (results/Grad_Dependent_Nonlinear/20d/SimpleUniform/SimpleUniform.log: its training set,
    its 1000 + 200 test points, n = rho = 2) on the HIP path with the reference's own random stream (compat_rng="jax": jax.random's
    float16 normals under its key schedule, drawn on the device) against the relative L2 errors that log prints.  Untimed; about a second."""
    import re
    path = os.path.join(ROOT, "tests", "golden", "reference_logged.json")
    if not os.path.exists(path):
        return None
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.MLP import MLP
    from scasml_gp_amd.solvers.MLP_full_history import MLP_full_history
    from scasml_gp_amd.solvers.ScaSML import ScaSML
    logged = json.load(open(path))
    d = 20

    def printed(kind, name):
        line = [l for l in logged[kind][str(d)]["simple_uniform"]["head"] if l.startswith(name + " rel L2")][0]
        return float(re.findall(r"-> (-?\d+\.\d+)", line)[0])
    eq = Grad_Dependent_Nonlinear(d + 1)
    eq.geometry()
    state = np.random.get_state()
    np.random.seed(1234)                                     # experiment_run.py:32
    dom, bdy = eq.generate_data(1000, 200)
    xt = np.concatenate(eq.generate_test_data(1000, 200))
    np.random.set_state(state)
    exact = eq.exact_solution(xt)
    gp = GP_Grad_Dependent_Nonlinear(eq, f16_graph=True)     # on float16 rows (its collocation and test points) the reference's kernels are float16 arithmetic
    gp.GPsolver(dom, bdy, GN_steps=20)
    kw = dict(compat_rng="jax", compat_f16=True)
    out = {"d": d, "protocol": "SimpleUniform (seed 1234), n = rho = 2 / full history n = 2, M = 3; HIP solvers on the reference's random stream, "
                               "GP(f16_graph=True): the reference's float16 op sequence on float16 rows",
           "rel_l2": {}, "logged": {}}
    for name, sol, want in (("GP", gp.predict(xt), printed("quadrature", "GP")),
                            ("MLP", MLP(eq, **kw).u_solve(2, 2, xt), printed("quadrature", "MLP")),
                            ("ScaSML", ScaSML(eq, gp, **kw).u_solve(2, 2, xt), printed("quadrature", "ScaSML")),
                            ("MLP_full_history", MLP_full_history(eq, **kw).u_solve(2, None, xt, 3), printed("full_history", "MLP"))):
        out["rel_l2"][name] = round(rel_l2(sol, exact), 7)
        out["logged"][name] = round(want, 7)
    out["max_relative_difference"] = round(max(abs(out["rel_l2"][k] - out["logged"][k]) / out["logged"][k] for k in out["logged"]), 5)
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    # SCASML_BENCH_FORCE_DIST=1: take the torch.distributed branches at WORLD_SIZE = 1 too, so that a one-GPU box can at least show
    # init_process_group("nccl"), the barrier and an RCCL all-reduce of the path's (B, 1+d) buffer executing (tests/test_gpu_bench_contract.py)
    dist_on = world > 1 or os.environ.get("SCASML_BENCH_FORCE_DIST") == "1"
    if dist_on:
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))   # "nccl" is RCCL on ROCm

    from scasml_gp_amd import tables
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.MLP import MLP
    from scasml_gp_amd.solvers.MLP_full_history import MLP_full_history
    from scasml_gp_amd.solvers.ScaSML import ScaSML
    from scasml_gp_amd.solvers.ScaSML_full_history import ScaSML_full_history

    d, n = args.d, args.level
    B = args.batch if args.batch else ((1 << 20) if args.solver == "mlp" else (1 << 14))
    par = n if args.variant == "quad" else args.M            # rho = n, or the full-history sample base M
    eq = Grad_Dependent_Nonlinear(d + 1)
    eq.geometry()

    # ---- setup (untimed): train the surrogate on 1000 + 200 collocation points -----------------
    rs = np.random.RandomState(1234)                       # reference seed, experiment_run.py:32
    state = np.random.get_state()
    np.random.set_state(rs.get_state())
    x_dom, x_bdy = eq.generate_data(args.train_domain, args.train_boundary)
    xt_h = np.concatenate(eq.generate_test_data(1000, 200))   # harness test set (float16, the stream continues: tests/SimpleUniform.py:75-86)
    np.random.set_state(state)
    gp, t_train = None, 0.0
    if args.solver == "scasml":
        compat = None if args.compat == "none" else args.compat
        gp = GP_Grad_Dependent_Nonlinear(eq, compat=compat)      # reference: its own Hutchinson index draw (threefry.py)
        t0 = time.time()
        gp.GPsolver(x_dom, x_bdy, GN_steps=20)
        torch.cuda.synchronize()
        t_train = time.time() - t0
        skw = dict(seed=0, compat_f16=args.compat_f16, compat_rng="jax" if args.rng == "jax" else None)
        solver = (ScaSML if args.variant == "quad" else ScaSML_full_history)(eq, gp, **skw)
    else:
        skw = dict(seed=0, compat_f16=args.compat_f16, compat_rng="jax" if args.rng == "jax" else None)
        solver = (MLP if args.variant == "quad" else MLP_full_history)(eq, **skw)

    # synthetic inputs: x ~ U[-0.5, 0.5]^d, t ~ U[0, 0.5), resident in HBM before timing
    def synth(seed):
        g = np.random.default_rng(seed)
        return np.concatenate([g.uniform(-0.5, 0.5, (B, d)), g.uniform(0.0, 0.5, (B, 1))], axis=1).astype(np.float32)
    x_t = synth(1234 + rank)                               # root sharding: every rank its own B roots
    x_dev = torch.from_numpy(x_t).cuda()
    x_shared = x_dev if rank == 0 else torch.from_numpy(synth(1234)).cuda()   # sample sharding: the same B roots on every rank
    eng = solver._engine
    plan = eng.plan(n, par)
    steps_exec = tables.executed_path_steps(plan)
    steps_ref = tables.reference_path_steps(args.variant, n, par, float(eq.T))

    from scasml_gp_amd import parallel

    on_host = args.rehearse_on_one_gpu                       # gloo reduces host tensors

    def timed_leg(step_fn):
        """W untimed + K timed steps, barrier + synchronize on both sides, MAX over ranks (seconds)."""
        for _ in range(args.warmup):
            step_fn()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_fn()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if on_host else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # ---- the north-star split: Monte-Carlo units over S ranks (one all-reduce), roots over the G groups -----------
    S, G, imb, imb_all = sample_split(eng, n, par, world, args.max_imbalance) if world > 1 else (1, 1, 1.0, 1.0)
    my_group, my_srank = rank // S, rank % S
    group = None
    if world > 1 and S > 1:
        for g in range(G):                                   # every rank creates every group (torch.distributed contract)
            h = dist.new_group(list(range(g * S, (g + 1) * S)))
            if g == my_group:
                group = h
    g_lo, g_cnt = parallel.root_slice(B, my_group, G)

    def samples_step(sid=None):
        if sid is None:
            sid = eng.calls
            eng.calls += 1
        out, uhat, _ = eng.solve(n, par, x_shared[g_lo:g_lo + g_cnt], root0=g_lo, rank=my_srank, world=S, stream_id=sid)
        if S > 1:
            if on_host:
                host = out.cpu()
                parallel.allreduce_partial_sums(host, group)
                out.copy_(host)
            else:
                parallel.allreduce_partial_sums(out, group)  # the single RCCL all-reduce of the path
            eng.finalize_partials(out)
        return out, uhat

    def roots_step():
        return eng.solve(n, par, x_dev, root0=rank * B)[:2]

    main_step = samples_step if args.shard == "samples" and world > 1 else roots_step
    eng.kernel_ms = {}
    for _ in range(args.warmup):
        main_step()
    eng.profile = True
    saved_warmup, args.warmup = args.warmup, 0
    elapsed = timed_leg(main_step)
    args.warmup = saved_warmup
    eng.profile = False
    kernel_ms = eng.collect_kernel_ms()                    # HIP-event durations, per kernel, averaged
    samples_leg = None
    if world > 1:
        other = roots_step if main_step is samples_step else samples_step
        t_other = timed_leg(other)
        t_s, t_r = (elapsed, t_other) if main_step is samples_step else (t_other, elapsed)
        samples_leg = {"sample_ranks": S, "root_groups": G, "unit_load_imbalance_max_over_mean": round(imb, 3),
                       "imbalance_if_all_ranks_shared_samples": round(imb_all, 3), "collective": "1 all-reduce of (B/G, 1+d) f32 per step over %d ranks" % S,
                       "scaling": "strong", "roots_total": B, "ms_per_step": round(t_s / args.steps * 1e3, 3),
                       "value": round(B * steps_exec * args.steps / t_s, 1),
                       "roots_leg": {"scaling": "weak", "roots_total": world * B, "ms_per_step": round(t_r / args.steps * 1e3, 3),
                                     "value": round(world * B * steps_exec * args.steps / t_r, 1)}}

    if world > 1:
        # the sample-sharded estimator against the unsharded one on the same roots and Philox streams (Philox is keyed by tree site,
        # so only the order of the float additions differs): max |difference| over this group's roots, worst over ranks
        sharded, _ = samples_step(sid=424242)
        whole, _, _ = eng.solve(n, par, x_shared[g_lo:g_lo + g_cnt], root0=g_lo, stream_id=424242)
        dmax = torch.tensor([float((sharded - whole).abs().max()) if g_cnt else 0.0], dtype=torch.float64, device="cpu" if on_host else "cuda")
        dist.all_reduce(dmax, op=dist.ReduceOp.MAX)
        samples_leg["max_abs_diff_vs_unsharded"] = float(dmax.item())
    rccl_selftest = None
    if dist_on and world == 1 and not on_host:
        # one rank: the all-reduce of the path's partial-sum buffer through RCCL (a copy onto itself), timed with HIP events
        buf = torch.randn((B, d + 1), dtype=torch.float32, device="cuda")
        ref = buf.clone()
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)          # (parallel.allreduce_partial_sums skips the call on one rank)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        e1.record()
        torch.cuda.synchronize()
        rccl_selftest = {"ranks": 1, "buffer": "(%d, %d) f32" % (B, d + 1), "allreduce_ms": round(e0.elapsed_time(e1) / 10, 4),
                         "unchanged": bool(torch.equal(buf, ref))}
    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        return

    # ---- accuracy on the harness protocol (untimed): 1000 + 200 test points ---------------------
    exact = eq.exact_solution(xt_h)
    u_gpu = solver.u_solve(n, par, xt_h) if args.variant == "quad" else solver.u_solve(n, None, xt_h, args.M)
    rel_gpu = rel_l2(u_gpu, exact)
    rel_gp = rel_l2(gp.predict(xt_h), exact) if gp is not None else None
    ref_logs = reference_logs_check() if not args.no_reference_logs_check else None

    # ---- roofline of the dominant kernel (fused GP evaluation, MFMA-bound) ----------------------
    n_colloc = args.train_domain + args.train_boundary
    m_feat = 4 * args.train_domain + args.train_boundary
    ppr = steps_exec + 1
    n_inf = B * ppr
    flops = n_inf * (2.0 * n_colloc * (d + 1) + 10.0 * m_feat)     # 2 N_inf N (d+1) + 10 N_inf M
    gp_ms = kernel_ms.get("gp_eval")
    traffic, traffic_source, issue, vector_roof = None, None, None, None
    # HBM bytes and issue-slot counters per launch come from separate rocprofv3 --pmc passes of this same command, condensed
    # by profiles/summarize.py (they cannot be collected inside this process): NOT measured in this run, labelled so, and quoted
    # only if they were taken on the SAME kernel source (sha1 of the files the kernel is built from)
    gp_sha = kernel_source_sha1(GP_EVAL_SOURCES[args.compat]) if gp is not None else None
    for prof in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gp_eval_pmc.json")), reverse=True):
        try:
            pj = json.load(open(prof))
            if gp is not None and pj.get("n_inf") == n_inf and pj.get("d") == d and pj.get("source_sha1") == gp_sha and pj.get("mode", "reference") == args.compat:
                traffic = pj.get("hbm_bytes_per_launch")
                traffic_source = "%s (separate rocprofv3 --pmc passes of this command on this kernel source; FETCH_SIZE doubled per MI355X_MICROARCH.md)" % os.path.relpath(prof, ROOT)
                if pj.get("valu_active_frac") is not None:
                    issue = {"valu_active_frac": round(pj["valu_active_frac"], 3), "mfma_pipe_busy_frac": round(pj["mfma_pipe_busy_frac"], 3),
                             "coexec_frac_of_mfma_busy": round(pj["coexec_frac_of_mfma_busy"], 3),
                             "cycles_per_valu_instruction": round(pj["cycles_per_valu_instruction"], 2),
                             "effective_clock_ghz": round(pj.get("effective_clock_ghz", 0.0), 3),
                             "source": os.path.relpath(prof, ROOT), "note": "vector time and matrix time ADD in this kernel (ablations and instruction-level "
                             "microbenchmarks: profiles/r03_compat_eval_experiments.txt, r04_ubench_hetero.txt, DESIGN.md 4.4): the launch is t_vector + t_matrix, "
                             "the counters' busy fractions overlap only in issue; `frac` is of the matrix roof alone, `vector` states the other term"}
                    cnt = pj.get("counters_avg_per_launch", {})
                    if cnt.get("SQ_INSTS_VALU") and pj.get("effective_clock_ghz"):
                        # the binding roof: vector wave-instructions per SIMD x the issue cost of THIS instruction mix with >= 3 waves
                        # per SIMD (tools/ubench_hetero.hip, vector-only rows: 2.02 cycles for plain float32, 3.12 for the as-coded epilogue's mix
                        # of plain / float16-conversion / exp instructions; the factored epilogue's 19 plain + 3 exp (8.3 cycles) per 22: 2.9) / the clock the chip held
                        mix = 3.12 if args.compat == "reference" else 2.9
                        clk = pj["effective_clock_ghz"] * 1e9
                        per_simd = cnt["SQ_INSTS_VALU"] / 1024.0
                        t_v = per_simd * mix / clk * 1e3
                        t_m = cnt.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / clk * 1e3
                        vector_roof = {"valu_wave_instructions_per_launch": cnt["SQ_INSTS_VALU"], "per_pair": round(cnt["SQ_INSTS_VALU"] * 64.0 / (n_inf * float((n_colloc + 31) // 32 * 32)), 2),
                                       "issue_cycles_per_instruction_of_this_mix": mix, "clock_ghz": round(pj["effective_clock_ghz"], 3),
                                       "vector_ms": round(t_v, 2), "vector_ms_at_2_cycles_per_instruction": round(per_simd * 2.0 / clk * 1e3, 2),
                                       "matrix_ms": round(t_m, 2), "sum_model_ms": round(t_v + t_m, 2), "profiled_launch_ms": round(pj["avg_ms_kernel_trace"], 2),
                                       "frac_of_sum_model": round((t_v + t_m) / pj["avg_ms_kernel_trace"], 3),
                                       "source": os.path.relpath(prof, ROOT) + " + profiles/r04_ubench_hetero.txt",
                                       "note": "a MFMA-only wave beside vector-only waves on one SIMD starves the vector waves (one instruction per 16 cycles each), and "
                                               "waves that interleave both pay ~8 issue cycles per MFMA only in a uniform stream: the kernel's launch time is the sum"}
                break
        except Exception:
            continue
    roofline = None
    if kernel_ms.get("picard_mlp"):
        # plain MLP: the whole recursion is one kernel with no HBM traffic between the root row and the result;
        # priced with the materialised-state model of SURVEY.md 8(d) (16*d bytes per path-step) for comparability
        ms = kernel_ms["picard_mlp"]
        gbs = B * steps_exec * 16.0 * d / (ms * 1e-3) / 1e9
        roofline = {"kernel": "picard_tree_kernel (MODE_MLP)", "bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": 2.0 * B * (d + 1) * 4,
                    "avg_launch_ms": round(ms, 4),
                    "note": "algorithmic bytes of the materialised model; the fused kernel keeps all path state in VGPRs, "
                            "its real traffic is the root rows in and out, and it is Philox/ALU-bound"}
    if gp_ms:
        ach = flops / (gp_ms * 1e-3) / 1e12
        from scasml_gp_amd import _lib
        kp = int(_lib.load().scasml_point_stride(d))       # the kernels' padded row length (round_up(d + 4, 16))
        n_pad = (n_colloc + 31) // 32 * 32
        if gp.compat == "reference":
            # issued 16-bit MFMA flops: per point and geometry two planes of (n_pad x kp) plus one K = 16 Hutchinson product; sites that
            # consume eps_PDE run three geometries on domain tiles and two on boundary tiles, the others two and one
            kinds = eng.site_kinds(n, par).cpu().numpy()
            n_full, n_part = int((kinds == 0).sum()) * B, int(np.isin(kinds, (1, 3, 4)).sum()) * B
            nd_pad = (args.train_domain + 31) // 32 * 32
            planes = 1 if (int(gp.eval_round16) & 4) else 2
            per_geom = lambda rows, q: 2.0 * rows * (planes * kp + (16 if q else 0))
            issued_flops = n_full * (3 * per_geom(nd_pad, True) + 2 * per_geom(n_pad - nd_pad, True)) \
                + n_part * (2 * per_geom(nd_pad, True) + per_geom(n_pad - nd_pad, False))
            issued = issued_flops / (gp_ms * 1e-3) / 1e12
            peak = MFMA_BF16_PEAK_TFLOPS
            # the same count with what the as-coded surrogate needs ALGORITHMICALLY: three distinct distance matrices (|x - y|, |x - y'|,
            # |x' - y|) where eps_PDE is consumed, two elsewhere (one on boundary rows), instead of the one of the documented operators
            flops_ac = 2.0 * (d + 1) * (n_full * (3 * args.train_domain + 2 * args.train_boundary)
                                        + n_part * (2 * args.train_domain + args.train_boundary)) + 10.0 * n_inf * m_feat
            ach_ac = flops_ac / (gp_ms * 1e-3) / 1e12
            geometry = not (int(gp.eval_round16) & 1)
            roofline = {"kernel": ("gp_eval_compat_mfma_kernel (compat='reference-geometry': the as-coded fit, 3 shifted geometries x %d fp16 plane%s, entries "
                                   "not rounded: factored sums)" % (planes, "" if planes == 1 else "s")) if geometry else
                                  "gp_eval_compat_mfma_kernel (as-coded surrogate: 3 shifted geometries x 2 fp16 planes, float16 entries)",
                        "bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                        "traffic": traffic, "traffic_source": traffic_source, "valu_issue": issue, "vector": vector_roof, "avg_launch_ms": round(gp_ms, 4),
                        "flops_per_launch": flops, "mfma_issued_tflops": round(issued, 1), "mfma_issued_frac": round(issued / peak, 4),
                        "as_coded": {"flops_per_launch": flops_ac, "achieved": round(ach_ac, 3), "frac": round(ach_ac / peak, 4),
                                     "note": "algorithmic flops of the surrogate the reference's code builds: 3 x.y products per pair where eps_PDE is "
                                             "consumed (%d of %d sites), 2 elsewhere" % (int((kinds == 0).sum()), len(kinds))},
                        "note": ("achieved = algorithmic flops of SURVEY 8(d) (2 N_inf N (d+1) + 10 N_inf M: ONE x.y product per pair); the geometry mode keeps the "
                                 "as-coded surrogate's three shifted x.y products (one fp16 plane of the point each) and drops the float16 rounding of the 13 "
                                 "entries per pair, so that the four sums factor per geometry (13 + 9 + 5 vector instructions + 3 exp per pair against ~50 + 3); "
                                 "DESIGN.md 4.5") if geometry else
                                ("achieved = algorithmic flops of SURVEY 8(d) (2 N_inf N (d+1) + 10 N_inf M: ONE x.y product per pair); the as-coded "
                                 "surrogate needs three (aligned, y shifted, x shifted) in two fp16 planes each, and 13 separately float16-rounded "
                                 "entries per pair in the epilogue (~50 vector instructions + 3 exp against 14 + 1 for the documented operators): "
                                 "the launch time is vector time plus matrix time: the chip's clock follows the MFMA density, so every instruction of either kind is paid "
                                 "for in time (valu_issue, vector; DESIGN.md 4.4)")}
        else:
            split = int(gp.eval_split)
            products = {0: 1, 2: 3, 3: 6, 22: 2 if getattr(gp, "_colloc_is_f16", False) else 3}[split]
            issued = products * 2.0 * n_inf * n_pad * kp / (gp_ms * 1e-3) / 1e12      # MFMA flops actually issued
            peak = MFMA_F32_PEAK_TFLOPS if split == 0 else MFMA_BF16_PEAK_TFLOPS
            roofline = {"kernel": "gp_eval_kernel (fp32 MFMA)" if split == 0 else ("gp_eval_bf16_kernel (2 fp16 planes, exponent-unit epilogue)" if split == 22 else "gp_eval_bf16_kernel (%d bf16 planes)" % split),
                        "bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_source, "valu_issue": issue,
                        "avg_launch_ms": round(gp_ms, 4),
                        "flops_per_launch": flops, "achieved_vs_fp32_mfma_peak": round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                        "mfma_issued_tflops": round(issued, 1), "mfma_issued_frac": round(issued / peak, 4),
                        "note": "achieved = algorithmic fp32 flops (SURVEY 8(d)); the split-precision kernel issues %dx as many "
                                "16-bit MFMA flops to keep products exact to 2^-22 (plus one K = 8 MFMA per tile for the bilinear part of the epilogue); "
                                "the vector ALUs are the busier pipe and the chip holds ~1.55 GHz here against 2.1 in the path kernels "
                                "(valu_issue; DESIGN.md 4.2)" % products}
    # the path kernels, priced with the materialised-state model of SURVEY.md 8(d): 16*d bytes per path-step
    path_ms = (kernel_ms.get("picard_generate") or 0.0) + (kernel_ms.get("picard_accumulate") or 0.0)
    path_roof = None
    from scasml_gp_amd import _lib as _l
    kp_path = _l.load().scasml_point_stride(d)
    if path_ms:
        gbs = B * steps_exec * 16.0 * d / (path_ms * 1e-3) / 1e9
        path_roof = {"kernels": "picard_tree generate+accumulate", "bound": "hbm", "achieved": round(gbs, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "model_frac": round(gbs / HBM_PEAK_GBS, 4),
                     "avg_launch_ms": round(path_ms, 4), "model": "16*d algorithmic bytes per path-step (SURVEY.md 8(d)): the materialised-state "
                     "model counts X and W read and written per step; the kernels keep W in registers and move fewer real bytes, so model_frac "
                     "overstates the HBM utilisation -- `frac` is real traffic (PMC) over peak"}
        # what the two kernels really move (rocprofv3 --pmc passes condensed by profiles/summarize.py; not measured in this run)
        pic_sha = kernel_source_sha1(PICARD_SOURCES)
        for prof in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_picard_pmc.json")), reverse=True):
            try:
                pj = json.load(open(prof))
                if pj.get("source_sha1") != pic_sha:
                    continue
                real = sum(v["hbm_bytes_per_launch"] for k, v in pj.items() if isinstance(v, dict) and v.get("grid_threads") in ((B * (int(kp_path) // 4) + 255) // 256 * 256, B * 32))
            except Exception:
                continue
            if real:
                path_roof.update({"traffic": real, "traffic_gb_per_s": round(real / (path_ms * 1e-3) / 1e9, 1),
                                  "frac": round(real / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "traffic_source": "%s (separate rocprofv3 --pmc passes on this kernel source)" % os.path.relpath(prof, ROOT)})
                break

    # ---- GP training on record: the bench's own 1000 + 200 fit and the staged size of BASELINE configs[4] ----
    gp_train = None
    if world == 1 and gp is not None:
        gp_train = [gp_train_block(d, args.train_domain, args.train_boundary, gp.compat)]
        if not args.no_gp_train_large:                       # staged configs[4]: the MFMA Gram exists for the documented operators
            gp_train.append(gp_train_block(250, 8333, 1667, None))

    # ---- CPU baseline: the oracle restatement on a bounded sample of the same workload ----------
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(args, eq, gp, eng, n, par, x_t, x_dev, x_dom, x_bdy, steps_exec, B)

    name = {("scasml", "quad"): "solvers.ScaSML (GP + Picard correction) n=rho=%d" % n,
            ("scasml", "fh"): "solvers.ScaSML_full_history n=%d M=%d" % (n, args.M),
            ("mlp", "quad"): "solvers.MLP n=rho=%d" % n,
            ("mlp", "fh"): "solvers.MLP_full_history n=%d M=%d" % (n, args.M)}[(args.solver, args.variant)]
    work_ranks = world if main_step is roots_step else 1     # samples: all ranks share the same B roots
    value = work_ranks * B * steps_exec * args.steps / elapsed
    line = {
        "metric": "Euler-Maruyama path-steps/sec + L2 rel-error, Grad_Dependent_Nonlinear d=%d n=%d" % (d, n),
        "value": round(value, 1), "unit": "path-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak" if main_step is roots_step else "strong", "rccl_ranks": dist.get_world_size() if dist_on else 1,
        "rccl_note": "no multi-GPU node has been available to this build: the N > 1 path (init_process_group('nccl'), the in-group all-reduce) is "
                     "rehearsed over gloo on one GPU only (tests/test_gpu_bench_contract.py) until a SCALE run exists; with SCASML_BENCH_FORCE_DIST=1 the "
                     "same branches run over RCCL with one rank (rccl_selftest)",
        "backend": (dist.get_backend() if dist_on else None), "rccl_selftest": rccl_selftest, "samples_sharding": samples_leg,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "Grad_Dependent_Nonlinear d=%d, %s, B=%d roots/GPU%s" % (
                       d, name, B, " (BASELINE.json configs[2])" if (args.solver, args.variant, d, n) == ("scasml", "quad", 100, 3) else ""),
                   "surrogate": ("reference's as-coded GP (compat='reference', Hutchinson indices %s)" % gp.laplacian_idx.tolist() if gp.compat == "reference"
                                 else "documented operators (compat=None)") if gp is not None else None,
                   "roots_per_gpu": B, "gp_collocation": ("%d+%d" % (args.train_domain, args.train_boundary)) if gp is not None else None,
                   "path_steps_per_root": steps_exec, "path_steps_per_root_reference_count": steps_ref,
                   "gp_point_evals_per_root": ppr,
                   "sharding": "roots across ranks, no collective" if main_step is roots_step else
                               "Monte-Carlo units of the root call over %d ranks (dealt by cost), one all-reduce of (B, 1+d) partial sums; roots over %d groups" % (S, G),
                   "note": "value counts only EXECUTED path-steps (the reference's discarded n=0 terminal draws are not "
                           "performed); with the reference's own count the same run is value_reference_count"},
        "value_reference_count": round(work_ranks * B * steps_ref * args.steps / elapsed, 1),
        "reference_logs_check": ref_logs,
        "l2_rel_error": {"solver_gpu": round(rel_gpu, 5), "gp_only": round(rel_gp, 5) if rel_gp is not None else None,
                         "points": "1000+200 harness set (np.random.seed(1234): the training draw, then this one, as tests/SimpleUniform.py)",
                         "vs_cpu_oracle": ({k: cpu[k] for k in ("rel_l2_gpu", "rel_l2_cpu", "abs_diff")} if cpu else None),
                         "surrogate": ("as coded by the reference (compat='reference')" if gp.compat == "reference" else "documented operators (compat=None)") if gp is not None else None},
        "kernel_ms": {k: round(v, 4) for k, v in kernel_ms.items()},
        "gp_train_s": round(t_train, 2),
        "roofline": roofline, "roofline_path": path_roof, "gp_train": gp_train, "cpu_baseline": cpu,
    }
    print(json.dumps(line))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
