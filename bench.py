#!/usr/bin/env python
"""Headline benchmark: Euler-Maruyama path-steps/s (+ relative L2 error) of
Grad_Dependent_Nonlinear d=100, solvers.ScaSML at level n = rho = 3 (BASELINE.json configs[2]).

One "step" = one ScaSML.uz_solve pass (generate points -> fused GP evaluation -> Picard
accumulation) over a device-resident batch of B synthetic evaluation points per GPU.
Multi-GPU: one process per GPU, evaluation points (independent objects) are sharded across
ranks -- no data-path collective; weak scaling.  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
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
    ap.add_argument("--cpu-sample", type=int, default=64, help="roots of the same workload timed on the CPU oracle")
    ap.add_argument("--shard", choices=["roots", "samples"], default="roots",
                    help="roots: each rank its own B roots, no collective (weak scaling, default); samples: every rank the same "
                         "B roots and 1/world of the Monte-Carlo units of the root call, ONE all-reduce of the partial estimators "
                         "(strong scaling; BASELINE.json north_star)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    Philox streams, and report the GPU-vs-CPU difference on that sample."""
    from oracle.equation import GradDependentNonlinear
    from oracle.gp import OracleGP
    from oracle.mlp import PicardOracle
    d = args.d
    oeq = GradDependentNonlinear(d + 1)
    ogp = None
    if gp is not None:
        ogp = OracleGP(oeq)
        ogp.x_t_domain = np.asarray(x_dom, dtype=np.float64)     # same trained surrogate as the GPU run
        ogp.x_t_boundary = np.asarray(x_bdy, dtype=np.float64)
        ogp.N_domain, ogp.N_boundary = len(x_dom), len(x_bdy)
        ogp.right_vector = gp.right_vector
    ns = min(args.cpu_sample if gp is not None else 4096 * args.cpu_sample, B)   # ~10-20 s of CPU work either way
    ora = PicardOracle(oeq, args.variant, gp=ogp, seed=0, stream=99)
    t0 = time.perf_counter()
    uz_cpu = ora.uz_solve(n, par, x_t[:ns])
    t_cpu = time.perf_counter() - t0
    uz_gpu, _, _ = eng.solve(n, par, x_dev[:ns], stream_id=99)
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count()
    return {"value": round(ns * steps_exec / t_cpu, 1), "unit": "path-steps/s", "cores": threads, "kind": "port",
            "sample": "%d of the %d roots, same inputs and Philox streams, NumPy float64 oracle (oracle/mlp.py%s), %.1f s"
                      % (ns, B, " + oracle/gp.py" if gp is not None else "", t_cpu),
            "max_abs_diff_gpu_vs_cpu": float(np.nanmax(np.abs(uz_gpu.cpu().numpy() - uz_cpu)))}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
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
    xt_h = np.concatenate(eq.generate_test_data(1000, 200)).astype(np.float32)   # harness test set
    np.random.set_state(state)
    gp, t_train = None, 0.0
    if args.solver == "scasml":
        gp = GP_Grad_Dependent_Nonlinear(eq)
        t0 = time.time()
        gp.GPsolver(x_dom, x_bdy, GN_steps=20)
        torch.cuda.synchronize()
        t_train = time.time() - t0
        solver = (ScaSML if args.variant == "quad" else ScaSML_full_history)(eq, gp, seed=0)
    else:
        solver = (MLP if args.variant == "quad" else MLP_full_history)(eq, seed=0)

    # synthetic inputs: x ~ U[-0.5, 0.5]^d, t ~ U[0, 0.5), resident in HBM before timing
    g = np.random.default_rng(1234 + (rank if args.shard == "roots" else 0))
    x_t = np.concatenate([g.uniform(-0.5, 0.5, (B, d)), g.uniform(0.0, 0.5, (B, 1))], axis=1).astype(np.float32)
    x_dev = torch.from_numpy(x_t).cuda()
    eng = solver._engine
    plan = eng.plan(n, par)
    steps_exec = tables.executed_path_steps(plan)
    steps_ref = tables.reference_path_steps(args.variant, n, par, float(eq.T))

    from scasml_gp_amd import parallel

    def one_step():
        if args.shard == "samples" and world > 1:
            sid = eng.calls
            eng.calls += 1
            out, uhat, _ = eng.solve(n, par, x_dev, rank=rank, world=world, stream_id=sid)
            if args.rehearse_on_one_gpu:                     # gloo reduces host tensors
                host = out.cpu()
                parallel.allreduce_partial_sums(host)
                out.copy_(host)
            else:
                parallel.allreduce_partial_sums(out)         # the single RCCL all-reduce of the path
            return eng.finalize_partials(out), uhat
        out, uhat, _ = eng.solve(n, par, x_dev, root0=rank * B)
        return out, uhat

    for _ in range(args.warmup):
        one_step()
    eng.profile = True
    eng.kernel_ms = {}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, uhat = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    eng.profile = False
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = eng.collect_kernel_ms()                    # HIP-event durations, per kernel, averaged

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- accuracy on the harness protocol (untimed): 1000 + 200 test points ---------------------
    exact = eq.exact_solution(xt_h)
    u_gpu = solver.u_solve(n, par, xt_h) if args.variant == "quad" else solver.u_solve(n, None, xt_h, args.M)
    rel_gpu = rel_l2(u_gpu, exact)
    rel_gp = rel_l2(gp.predict(xt_h), exact) if gp is not None else None

    # ---- roofline of the dominant kernel (fused GP evaluation, MFMA-bound) ----------------------
    n_colloc = args.train_domain + args.train_boundary
    m_feat = 4 * args.train_domain + args.train_boundary
    ppr = steps_exec + 1
    n_inf = B * ppr
    flops = n_inf * (2.0 * n_colloc * (d + 1) + 10.0 * m_feat)     # 2 N_inf N (d+1) + 10 N_inf M
    gp_ms = kernel_ms.get("gp_eval")
    traffic = None
    # HBM bytes per launch come from separate rocprofv3 --pmc passes of this same command, condensed by
    # profiles/summarize.py (they cannot be collected inside this process)
    prof = os.path.join(ROOT, "profiles", "r01_gp_eval_pmc.json")
    if os.path.exists(prof):
        try:
            pj = json.load(open(prof))
            if pj.get("n_inf") == n_inf and pj.get("d") == d and pj.get("split") == int(gp.eval_split):
                traffic = pj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
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
        split = int(gp.eval_split)
        kp = (d + 2 + 15) // 16 * 16
        n_pad = (n_colloc + 31) // 32 * 32
        products = {0: 1, 2: 3, 3: 6, 22: 2 if getattr(gp, "_colloc_is_f16", False) else 3}[split]
        issued = products * 2.0 * n_inf * n_pad * kp / (gp_ms * 1e-3) / 1e12      # MFMA flops actually issued
        peak = MFMA_F32_PEAK_TFLOPS if split == 0 else MFMA_BF16_PEAK_TFLOPS
        roofline = {"kernel": "gp_eval_kernel (fp32 MFMA)" if split == 0 else ("gp_eval_bf16_kernel (2 fp16 planes, exponent-unit epilogue)" if split == 22 else "gp_eval_bf16_kernel (%d bf16 planes)" % split),
                    "bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "avg_launch_ms": round(gp_ms, 4),
                    "flops_per_launch": flops, "achieved_vs_fp32_mfma_peak": round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                    "mfma_issued_tflops": round(issued, 1), "mfma_issued_frac": round(issued / peak, 4),
                    "note": "achieved = algorithmic fp32 flops (SURVEY 8(d)); the split-precision kernel issues %dx as many "
                            "16-bit MFMA flops to keep products exact to 2^-22; the kernel is instruction-issue-bound: on gfx950 "
                            "MFMA and VALU issue time add (DESIGN.md 4.2)" % products}
    # the path kernels, priced with the materialised-state model of SURVEY.md 8(d): 16*d bytes per path-step
    path_ms = (kernel_ms.get("picard_generate") or 0.0) + (kernel_ms.get("picard_accumulate") or 0.0)
    path_roof = None
    if path_ms:
        gbs = B * steps_exec * 16.0 * d / (path_ms * 1e-3) / 1e9
        path_roof = {"kernels": "picard_tree generate+accumulate", "bound": "hbm", "achieved": round(gbs, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                     "avg_launch_ms": round(path_ms, 4)}

    # ---- CPU baseline: the oracle restatement on a bounded sample of the same workload ----------
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(args, eq, gp, eng, n, par, x_t, x_dev, x_dom, x_bdy, steps_exec, B)

    name = {("scasml", "quad"): "solvers.ScaSML (GP + Picard correction) n=rho=%d" % n,
            ("scasml", "fh"): "solvers.ScaSML_full_history n=%d M=%d" % (n, args.M),
            ("mlp", "quad"): "solvers.MLP n=rho=%d" % n,
            ("mlp", "fh"): "solvers.MLP_full_history n=%d M=%d" % (n, args.M)}[(args.solver, args.variant)]
    work_ranks = world if args.shard == "roots" else 1       # samples: all ranks share the same B roots
    value = work_ranks * B * steps_exec * args.steps / elapsed
    line = {
        "metric": "Euler-Maruyama path-steps/sec + L2 rel-error, Grad_Dependent_Nonlinear d=%d n=%d" % (d, n),
        "value": round(value, 1), "unit": "path-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak" if args.shard == "roots" else "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "Grad_Dependent_Nonlinear d=%d, %s, B=%d roots/GPU%s" % (
                       d, name, B, " (BASELINE.json configs[2])" if (args.solver, args.variant, d, n) == ("scasml", "quad", 100, 3) else ""),
                   "roots_per_gpu": B, "gp_collocation": ("%d+%d" % (args.train_domain, args.train_boundary)) if gp is not None else None,
                   "path_steps_per_root": steps_exec, "path_steps_per_root_reference_count": steps_ref,
                   "gp_point_evals_per_root": ppr,
                   "sharding": "roots across ranks, no collective" if args.shard == "roots" else
                               "Monte-Carlo units of the root call across ranks, one all-reduce of (B, 1+d) partial sums",
                   "note": "value counts only EXECUTED path-steps (the reference's discarded n=0 terminal draws are not "
                           "performed); with the reference's own count the same run is value_reference_count"},
        "value_reference_count": round(work_ranks * B * steps_ref * args.steps / elapsed, 1),
        "l2_rel_error": {"solver_gpu": round(rel_gpu, 5), "gp_only": round(rel_gp, 5) if rel_gp is not None else None,
                         "points": "1000+200 harness set",
                         "logged_reference_d20": "0.069 (results/.../20d/RepeatedExperiment.log:21)"},
        "kernel_ms": {k: round(v, 4) for k, v in kernel_ms.items()},
        "gp_train_s": round(t_train, 2),
        "roofline": roofline, "roofline_path": path_roof, "cpu_baseline": cpu,
    }
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
