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

With SCASML_BENCH_DIST_GP=1 (opt-in) an N > 1 run also fits the block-row distributed GP (M = 7001) over its own process group and reports
"dist_gp_check": right_vector against the single-GPU fit -- the first thing to run on a multi-GPU node.

At N = 1 the line also carries "other_runs": the other BASELINE configurations and modes (configs[1], configs[3], the
reference-stream parity mode, the geometry mode), 5 timed steps each in this same process, and "gp_train": the stages of the
surrogate's fit with their FP64-MFMA fractions.  Layout of this file: workloads -> timing -> rooflines -> the line (main).
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
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA, MI355X_MICROARCH.md chip table
FP64_MFMA_PEAK_TFLOPS = 78.6    # v_mfma_f64_16x16x4_f64 dense, MI355X_MICROARCH.md / SURVEY.md 8(d)

GP_EVAL_SOURCES = {"reference": ["gp_eval_compat_mfma.hip", "gp_mfma16.hpp", "gp_common.hpp"],
                   "reference-geometry": ["gp_eval_compat_mfma.hip", "gp_mfma16.hpp", "gp_common.hpp"],
                   "none": ["gp_eval_bf16.hip", "gp_mfma16.hpp", "gp_common.hpp"]}
PICARD_SOURCES = ["picard_tree.hip", "picard_tree.hpp", "philox_normal.hpp", "equations.hpp"]


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
    ap.add_argument("--min-sample-efficiency", type=float, default=0.9,
                    help="samples leg: use the largest number of sample ranks S (a divisor of the rank count) whose PREDICTED strong-scaling "
                         "efficiency -- unsharded step / (ranks x (slowest rank's modelled time + the stated all-reduce cost)) -- reaches this "
                         "(BASELINE.json north_star: 0.9); the remaining factor shards roots.  0: all ranks share the samples")
    ap.add_argument("--allreduce-busbw-gbs", type=float, default=100.0,
                    help="stated cost of the one all-reduce in that prediction: bus bandwidth of RCCL's all-reduce at this message size (GB/s) ...")
    ap.add_argument("--allreduce-latency-us", type=float, default=40.0, help="... and its fixed latency; no multi-GPU node has measured either yet")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-fit", action="store_true", help="skip the oracle's own GP fit (cpu_baseline.fit: ~20 s of CPU at the reference's 1000 + 200 points)")
    ap.add_argument("--no-reference-logs-check", action="store_true", help="skip the d = 20 runs on the reference's own random stream (about a second)")
    ap.add_argument("--no-other-runs", action="store_true", help="skip the other_runs block (the other BASELINE configs and modes, 5 timed steps each)")
    ap.add_argument("--no-gp-train-large", action="store_true",
                    help="skip the M = 34 999 leg of the gp_train block (d = 250, 8333 + 1667 collocation points, ~30 GB, ~15 s)")
    ap.add_argument("--gp-train-xl", action="store_true",
                    help="add the M = 70 001 legs to the gp_train block (d = 250, 16 667 + 3 333 collocation points, 39 GB per matrix: past 2^31 "
                         "elements; both surrogates, one fit each, about two minutes)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development only: all ranks share cuda:0 and rendezvous over gloo (a 1-GPU box cannot host RCCL ranks)")
    return ap.parse_args()


def rel_l2(sol, exact):
    """tests/SimpleUniform.py:110-136: NaN-masked ||sol - exact||_2 / ||exact||_2."""
    sol, exact = np.asarray(sol, dtype=np.float64).ravel(), np.asarray(exact, dtype=np.float64).ravel()
    m = ~(np.isnan(sol) | np.isnan(exact))
    return float(np.linalg.norm(sol[m] - exact[m]) / np.linalg.norm(exact[m]))


# =========================================================================================== ranks
class Ranks:
    """One process per GPU: WORLD_SIZE / RANK / LOCAL_RANK from the launcher; "nccl" is RCCL on ROCm."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.dist = dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = 0 if args.rehearse_on_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world))
        torch.cuda.set_device(local)
        self.on_host = args.rehearse_on_one_gpu                 # gloo reduces host tensors
        # SCASML_BENCH_FORCE_DIST=1: take the torch.distributed branches at WORLD_SIZE = 1 too, so that a one-GPU box can at least show
        # init_process_group("nccl"), the barrier and an RCCL all-reduce of the path's (B, 1+d) buffer executing (tests/test_gpu_bench_contract.py)
        self.on = self.world > 1 or os.environ.get("SCASML_BENCH_FORCE_DIST") == "1"
        if self.on:
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
                os.environ.setdefault(k, v)
            if self.on_host:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    def max_seconds(self, dt):
        if not self.on:
            return dt
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if self.on_host else "cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.on:
            self.dist.destroy_process_group()


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


# =========================================================================================== workloads
class Workload:
    """One solver configuration with its synthetic roots resident in HBM: x ~ U[-0.5, 0.5]^d, t ~ U[0, 0.5)."""

    def __init__(self, eq, gp, solver_kind, variant, n, M, B, rank, rng="philox", compat_f16=False):
        import torch
        from scasml_gp_amd import tables
        from scasml_gp_amd.solvers.MLP import MLP
        from scasml_gp_amd.solvers.MLP_full_history import MLP_full_history
        from scasml_gp_amd.solvers.ScaSML import ScaSML
        from scasml_gp_amd.solvers.ScaSML_full_history import ScaSML_full_history
        self.eq, self.gp, self.kind, self.variant, self.n, self.M, self.B = eq, gp, solver_kind, variant, n, M, B
        self.d = eq.n_input - 1
        self.par = n if variant == "quad" else M                 # rho = n, or the full-history sample base M
        kw = dict(seed=0, compat_f16=compat_f16, compat_rng="jax" if rng == "jax" else None)
        if solver_kind == "scasml":
            self.solver = (ScaSML if variant == "quad" else ScaSML_full_history)(eq, gp, **kw)
        else:
            self.solver = (MLP if variant == "quad" else MLP_full_history)(eq, **kw)
        self.eng = self.solver._engine
        self.plan = self.eng.plan(n, self.par)
        self.steps_exec = tables.executed_path_steps(self.plan)
        self.steps_ref = tables.reference_path_steps(variant, n, self.par, float(eq.T))
        self.rank = rank
        self.x_t = self.synth(1234 + rank)                       # root sharding: every rank its own B roots
        self.x_dev = torch.from_numpy(self.x_t).cuda()
        self.name = {("scasml", "quad"): "solvers.ScaSML (GP + Picard correction) n=rho=%d" % n,
                     ("scasml", "fh"): "solvers.ScaSML_full_history n=%d M=%d" % (n, M),
                     ("mlp", "quad"): "solvers.MLP n=rho=%d" % n,
                     ("mlp", "fh"): "solvers.MLP_full_history n=%d M=%d" % (n, M)}[(solver_kind, variant)]

    def synth(self, seed):
        g = np.random.default_rng(seed)
        return np.concatenate([g.uniform(-0.5, 0.5, (self.B, self.d)), g.uniform(0.0, 0.5, (self.B, 1))], axis=1).astype(np.float32)

    def step(self):
        return self.eng.solve(self.n, self.par, self.x_dev, root0=self.rank * self.B)[:2]


def harness_sets(eq, n_dom, n_bdy):
    """The reference's draw order under its seed (experiment_run.py:32, tests/SimpleUniform.py:73-86): training set, then 1000 + 200 test points."""
    state = np.random.get_state()
    np.random.seed(1234)
    x_dom, x_bdy = eq.generate_data(n_dom, n_bdy)
    xt = np.concatenate(eq.generate_test_data(1000, 200))
    np.random.set_state(state)
    return x_dom, x_bdy, xt


def fit_surrogate(eq, x_dom, x_bdy, compat):
    import torch
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    gp = GP_Grad_Dependent_Nonlinear(eq, compat=None if compat == "none" else compat)   # reference: its own Hutchinson index draw (threefry.py)
    t0 = time.time()
    gp.GPsolver(x_dom, x_bdy, GN_steps=20)
    torch.cuda.synchronize()
    return gp, time.time() - t0


# =========================================================================================== timing
def timed_leg(ranks, step_fn, steps, warmup):
    """W untimed + K timed steps, barrier + synchronize on both sides, MAX over ranks (seconds)."""
    import torch
    for _ in range(warmup):
        step_fn()
    if ranks.on:
        ranks.dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize()
    if ranks.on:
        ranks.dist.barrier()
    return ranks.max_seconds(time.perf_counter() - t0)


def measure(ranks, wl, step_fn, steps, warmup):
    """-> (seconds of the K timed steps, HIP-event kernel durations averaged per launch): the events are recorded on the launch stream
    inside the timed region itself (PicardEngine._timed)."""
    wl.eng.kernel_ms = {}
    for _ in range(warmup):
        step_fn()
    wl.eng.profile = True
    elapsed = timed_leg(ranks, step_fn, steps, 0)
    wl.eng.profile = False
    return elapsed, wl.eng.collect_kernel_ms()


def other_runs(ranks, args, eq100, gp100, x_dom, x_bdy):
    """The other BASELINE configurations and modes, 5 timed steps each after 2 warm-up steps, in this process: same
    timing protocol and step definition as the headline; `value` counts executed path-steps."""
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    out = []

    def run(label, wl, note=None):
        elapsed, kms = measure(ranks, wl, wl.step, 5, 2)
        out.append({"workload": "Grad_Dependent_Nonlinear d=%d, %s, B=%d roots%s" % (wl.d, wl.name, wl.B, label), "steps": 5, "warmup": 2,
                    "ms_per_step": round(elapsed / 5 * 1e3, 3), "value": round(wl.B * wl.steps_exec * 5 / elapsed, 1), "unit": "path-steps/s",
                    "path_steps_per_root": wl.steps_exec, "path_steps_per_root_reference_count": wl.steps_ref,
                    "kernel_ms": {k: round(v, 4) for k, v in kms.items()}, "note": note})
        del wl

    eq20 = Grad_Dependent_Nonlinear(21)
    eq20.geometry()
    run(" (BASELINE.json configs[1])", Workload(eq20, None, "mlp", "quad", 2, 3, 1 << 20, ranks.rank))
    if gp100 is not None:
        surrogate = "as coded" if gp100.compat == "reference" else "documented operators"
        run(" (BASELINE.json configs[3] on one GPU; surrogate %s)" % surrogate, Workload(eq100, gp100, "scasml", "fh", 4, 3, 1 << 14, ranks.rank))
        run(" (configs[2] in the parity mode: --rng jax --compat-f16)", Workload(eq100, gp100, "scasml", "quad", 3, 3, 1 << 14, ranks.rank, rng="jax", compat_f16=True),
            "the reference's own random stream (jax.random float16 normals under its key schedule, drawn on the device) and its solver-level float16 casts: "
            "the mode that reproduces the reference's logs (reference_logs_check)")
        if gp100.compat == "reference" and not gp100.eval_geometry:
            geo = GP_Grad_Dependent_Nonlinear(eq100, compat="reference-geometry", laplacian_idx=gp100.laplacian_idx)
            geo.load_right_vector(x_dom, x_bdy, gp100.right_vector)          # the SAME fit, evaluated without the per-entry float16 roundings
            run(" (configs[2] with compat='reference-geometry')", Workload(eq100, geo, "scasml", "quad", 3, 3, 1 << 14, ranks.rank),
                "opt-in: the as-coded fit, the hot evaluation with factored sums and one fp16 plane of the point (GP relative L2 moves by <= 1e-5)")
    return out


def other_runs_summary(others):
    """[[short label, ms_per_step, value]] of other_runs, well under 600 characters."""
    if not others:
        return None
    out = []
    for o in others:
        w = o["workload"]
        label = ("configs[1] MLP d=20 n=2 B=2^20" if "configs[1]" in w else
                 "configs[3] ScaSML_fh d=100 n=4 M=3 B=2^14" if "configs[3]" in w else
                 "configs[2] parity mode (jax stream + f16 casts)" if "parity mode" in w else
                 "configs[2] reference-geometry" if "reference-geometry" in w else
                 "configs[4] staged: d=250 ScaSML n=3 B=2^10, 20000 colloc" if "configs[4]" in w else w[:48])
        out.append([label, o["ms_per_step"], float("%.4g" % o["value"])])
    return out


# =========================================================================================== multi-GPU legs
# A sample-sharded rank's step beyond its dealt share of the unsharded step (ms, one MI355X, headline shape, profiles/r06_sample_sharding_rank_times.txt):
# three launches whose ramp and tail do not shrink with the share, ACCUMULATE's un-pipelined root call (8 ranks: 2.80 measured, 2.68 by share alone)
SAMPLE_RANK_FIXED_MS = 0.12


def allreduce_model_ms(nbytes, ranks, args):
    """The STATED cost of one all-reduce of `nbytes` over `ranks` ranks (no multi-GPU node has measured it): latency + 2 (S - 1) / S x bytes / bus bandwidth."""
    if ranks <= 1:
        return 0.0
    return args.allreduce_latency_us * 1e-3 + 2.0 * (ranks - 1) / ranks * nbytes / (args.allreduce_busbw_gbs * 1e9) * 1e3


def sample_split_table(eng, n, par, world, unsharded_ms, roots, d, args):
    """Per divisor S of `world` (S sample ranks x G = world / S root groups): the dealt load (scasml_plan_deal_units with the surrogate's site
    costs: second evaluations of shared node points and replayed path steps included), the slowest rank's modelled step -- its share of the
    measured unsharded step of these `roots` roots, cut G ways by the root groups, plus the fixed cost of a sharded step -- the stated
    all-reduce cost for (roots / G)(1 + d) floats, and the predicted strong-scaling efficiency unsharded / (world x (rank + all-reduce))."""
    from scasml_gp_amd.solvers._picard import deal_units, site_cost
    plan, cost = eng.plan(n, par), site_cost(eng.gp)
    whole = float(deal_units(plan, 1, cost)[1][0])
    table = []
    for s in range(1, world + 1):
        if world % s:
            continue
        g = world // s
        load = deal_units(plan, s, cost)[1]
        rank_ms = unsharded_ms / g * float(load.max()) / whole + (SAMPLE_RANK_FIXED_MS if s > 1 else 0.0)
        ar_ms = allreduce_model_ms((roots // g) * (d + 1) * 4, s, args)
        table.append({"sample_ranks": s, "root_groups": g, "dealt_load_max_over_mean": round(float(load.max() / load.mean()), 4),
                      "dealt_load_sum_over_unsharded": round(float(load.sum()) / whole, 4), "modelled_rank_ms": round(rank_ms, 4),
                      "allreduce_ms_stated": round(ar_ms, 4), "predicted_efficiency": round(unsharded_ms / (world * (rank_ms + ar_ms)), 4)})
    return table


class SampleSharding:
    """The north-star split: Monte-Carlo units of the root call over S ranks (one all-reduce), roots over the G groups.  S is chosen by
    PREDICTED efficiency from a short calibration of the unsharded step on every rank (the slowest rank's time, so all ranks agree)."""

    def __init__(self, ranks, wl, args):
        import torch
        from scasml_gp_amd import parallel
        self.ranks, self.wl, self.parallel = ranks, wl, parallel
        world, rank = ranks.world, ranks.rank
        self.S, self.G, self.table, self.calibration_ms = 1, 1, None, None
        if world > 1:
            self.calibration_ms = timed_leg(ranks, wl.step, 3, 2) / 3 * 1e3
            self.table = sample_split_table(wl.eng, wl.n, wl.par, world, self.calibration_ms, wl.B, wl.d, args)
            ok = [row for row in self.table if row["predicted_efficiency"] >= args.min_sample_efficiency or row["sample_ranks"] == 1]
            self.S = max(row["sample_ranks"] for row in ok)
            self.G = world // self.S
        self.row = next((row for row in self.table if row["sample_ranks"] == self.S), None) if self.table else None
        self.group_id, self.srank = rank // self.S, rank % self.S
        self.group = None
        if world > 1 and self.S > 1:
            for g in range(self.G):                               # every rank creates every group (torch.distributed contract)
                h = ranks.dist.new_group(list(range(g * self.S, (g + 1) * self.S)))
                if g == self.group_id:
                    self.group = h
        self.lo, self.cnt = parallel.root_slice(wl.B, self.group_id, self.G)
        self.x_shared = wl.x_dev if rank == 0 else torch.from_numpy(wl.synth(1234)).cuda()   # the same B roots on every rank
        self.compute_events = None                                # [(start, end)] of eng.solve alone, when the leg is being timed per rank

    def step(self, sid=None):
        import torch
        wl, eng = self.wl, self.wl.eng
        if sid is None:
            sid = eng.calls
            eng.calls += 1
        if self.compute_events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        out, uhat, _ = eng.solve(wl.n, wl.par, self.x_shared[self.lo:self.lo + self.cnt], root0=self.lo, rank=self.srank, world=self.S, stream_id=sid)
        if self.compute_events is not None:
            e1.record()
            self.compute_events.append((e0, e1))
        if self.S > 1:
            if self.ranks.on_host:
                host = out.cpu()
                self.parallel.allreduce_partial_sums(host, self.group)
                out.copy_(host)
            else:
                self.parallel.allreduce_partial_sums(out, self.group)   # the single RCCL all-reduce of the path
            eng.finalize_partials(out)
        return out, uhat

    def rank_compute_ms(self):
        """Every rank's own compute time per step of the leg just timed (HIP events around its three kernels, the all-reduce outside):
        gathered so that rank 0 can print the measured times next to the dealt loads."""
        import torch
        torch.cuda.synchronize()
        ev, self.compute_events = self.compute_events, None
        mine = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
        t = torch.zeros(self.ranks.world, dtype=torch.float64, device="cpu" if self.ranks.on_host else "cuda")
        t[self.ranks.rank] = mine
        self.ranks.dist.all_reduce(t, op=self.ranks.dist.ReduceOp.SUM)
        return [round(float(v), 4) for v in t.tolist()]

    def report(self, t_s, t_r, steps, rank_ms):
        wl, world = self.wl, self.ranks.world
        ms_s, ms_r = t_s / steps * 1e3, t_r / steps * 1e3
        return {"sample_ranks": self.S, "root_groups": self.G,
                "chosen_by": "largest S with predicted_efficiency >= the requested minimum (S = 1: roots only)",
                "unit_load_imbalance_max_over_mean": self.row["dealt_load_max_over_mean"], "predicted_efficiency": self.row["predicted_efficiency"],
                "predicted_from": {"unsharded_step_ms_calibration": round(self.calibration_ms, 3), "fixed_ms_per_sharded_rank_step": SAMPLE_RANK_FIXED_MS,
                                   "per_S": self.table},
                "collective": "1 all-reduce of (B/G, 1+d) f32 per step over %d ranks" % self.S,
                "scaling": "strong", "roots_total": wl.B, "ms_per_step": round(ms_s, 3),
                "value": round(wl.B * wl.steps_exec * steps / t_s, 1),
                # measured: the B roots of ONE GPU's unsharded step (the roots leg's per-GPU work) shared by all `world` ranks
                "efficiency_vs_unsharded_step": round(ms_r / (world * ms_s), 4),
                "rank_compute_ms": rank_ms, "rank_compute_ms_max_over_mean": round(max(rank_ms) / (sum(rank_ms) / len(rank_ms)), 4) if rank_ms else None,
                "roots_leg": {"scaling": "weak", "roots_total": world * wl.B, "ms_per_step": round(ms_r, 3),
                              "value": round(world * wl.B * wl.steps_exec * steps / t_r, 1)}}

    def max_abs_diff_vs_unsharded(self):
        """The sample-sharded estimator against the unsharded one on the same roots and Philox streams (Philox is keyed by tree site, so only
        the order of the float additions differs): max |difference| over this group's roots, worst over ranks."""
        import torch
        wl = self.wl
        sharded, _ = self.step(sid=424242)
        whole, _, _ = wl.eng.solve(wl.n, wl.par, self.x_shared[self.lo:self.lo + self.cnt], root0=self.lo, stream_id=424242)
        dmax = torch.tensor([float((sharded - whole).abs().max()) if self.cnt else 0.0], dtype=torch.float64, device="cpu" if self.ranks.on_host else "cuda")
        self.ranks.dist.all_reduce(dmax, op=self.ranks.dist.ReduceOp.MAX)
        return float(dmax.item())


def dist_gp_check(ranks):
    """SCASML_BENCH_DIST_GP=1, N > 1: the block-row distributed fit of the as-coded surrogate (scasml_gp_amd/dist_gp.py; d = 20, 1600 + 601
    collocation points, M = 7001: 28 block rows over the ranks) over the run's own process group -- RCCL on a multi-GPU node -- against the
    single-GPU fit of the same data on rank 0.  Untimed side check, off by default (a first contact with real xGMI should not be able to take
    the headline line down with it); every rank takes part."""
    import torch
    from scasml_gp_amd.dist_gp import Comm, DistributedGP
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    eq = Grad_Dependent_Nonlinear(21)
    dom, bdy, _ = harness_sets(eq, 1600, 601)
    gp = GP_Grad_Dependent_Nonlinear(eq)
    cm = Comm()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fit = DistributedGP(gp, cm)
    fit.fit(dom, bdy, GN_steps=20)
    torch.cuda.synchronize()
    out = {"M": gp.phi_dim, "ranks": cm.world, "backend": cm.backend, "fit_s": round(time.perf_counter() - t0, 2), "newton_steps": len(gp.loss_history) - 1,
           "cg_products": int(sum(fit.cg_iterations)), "collective_calls": dict(cm.calls), "collective_gb_per_rank": round(cm.bytes_moved / 1e9, 3)}
    if ranks.rank == 0:
        one = GP_Grad_Dependent_Nonlinear(eq)
        one.GPsolver(dom, bdy, GN_steps=20)
        out["right_vector_rel_diff_vs_single_gpu"] = float(np.abs(gp.right_vector - one.right_vector).max() / np.abs(one.right_vector).max())
        out["newton_steps_single_gpu"] = len(one.loss_history) - 1
    return out


def rccl_selftest(ranks, B, d):
    """One rank: the all-reduce of the path's partial-sum buffer through RCCL (a copy onto itself), timed with HIP events."""
    import torch
    dist = ranks.dist
    buf = torch.randn((B, d + 1), dtype=torch.float32, device="cuda")
    ref = buf.clone()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)              # (parallel.allreduce_partial_sums skips the call on one rank)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    e1.record()
    torch.cuda.synchronize()
    sub = dist.new_group([0])                               # the sample-sharded leg reduces inside sub-groups of the world: one of those, through RCCL
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=sub)
    t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                # the MAX-over-ranks of the timed region
    torch.cuda.synchronize()
    return {"ranks": 1, "buffer": "(%d, %d) f32" % (B, d + 1), "allreduce_ms": round(e0.elapsed_time(e1) / 10, 4),
            "unchanged": bool(torch.equal(buf, ref)) and float(t.item()) == 1.5, "subgroup_allreduce": True}


# =========================================================================================== CPU baseline and accuracy
def cpu_baseline(args, wl, x_dom, x_bdy):
    """Time the oracle restatement (NumPy float64) on a bounded sample of the same workload, same inputs and
    Philox streams, and report the GPU-vs-CPU difference on that sample: the relative L2 error of both against the exact
    solution (tests/SimpleUniform.py:134-136) and their difference (north_star: within 1e-3)."""
    from oracle.equation import GradDependentNonlinear
    from oracle.gp import OracleGP
    from oracle.gp_compat import OracleGPCompat
    from oracle.mlp import PicardOracle
    d, gp, eng, B = wl.d, wl.gp, wl.eng, wl.B
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
    ora = PicardOracle(oeq, wl.variant, gp=ogp, seed=0, stream=99, compat_f16=args.compat_f16, jax_stream=args.rng == "jax")
    ora.jax_splits = eng.jax_splits                          # the replay below (stream_id given) reads the solver's key where it stands
    t0 = time.perf_counter()
    uz_cpu = ora.uz_solve(wl.n, wl.par, wl.x_t[:ns])
    t_cpu = time.perf_counter() - t0
    uz_gpu, uhat_gpu, _ = eng.solve(wl.n, wl.par, wl.x_dev[:ns], stream_id=99)
    uz_gpu = uz_gpu.cpu().numpy().astype(np.float64)
    u_cpu, u_gpu = uz_cpu[:, 0], uz_gpu[:, 0]
    if gp is not None:                                     # u_solve = u_hat + u_breve (ScaSML.py:300-304)
        u_cpu = u_cpu + ogp.predict(np.asarray(wl.x_t[:ns], dtype=np.float64))[:, 0]
        u_gpu = u_gpu + uhat_gpu.cpu().numpy().astype(np.float64)
    exact = oeq.exact_solution(np.asarray(wl.x_t[:ns], dtype=np.float64))[:, 0]
    rel_cpu, rel_gpu = rel_l2(u_cpu, exact), rel_l2(u_gpu, exact)
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count()
    diff = np.abs(uz_gpu - uz_cpu)
    fit = None
    if gp is not None and not args.no_cpu_fit and 4 * len(x_dom) + len(x_bdy) <= 6000:
        # the sample above runs on the device's right_vector (same trained surrogate on both sides); here the oracle FITS on its own -- Gram, factor,
        # Newton (models/GP.py:182-268, 487-604 restated) -- and the two fits are compared: the training half of the path, timed and checked
        ofit = OracleGPCompat(oeq, gp.laplacian_idx, round_factor=False) if gp.compat == "reference" else OracleGP(oeq)
        t0 = time.perf_counter()
        ofit.GPsolver(np.asarray(x_dom, dtype=np.float64), np.asarray(x_bdy, dtype=np.float64), GN_steps=20)
        t_fit = time.perf_counter() - t0
        rv_c, rv_g = np.asarray(ofit.right_vector, dtype=np.float64).ravel(), np.asarray(gp.right_vector, dtype=np.float64).ravel()
        xs = np.asarray(wl.x_t[:1024], dtype=np.float64)
        ex = oeq.exact_solution(xs)[:, 0]
        fit = {"M": int(4 * len(x_dom) + len(x_bdy)), "cpu_s": round(t_fit, 2), "newton_steps_cpu": len(ofit.loss_history) - 1,
               "newton_steps_gpu": len(gp.loss_history) - 1,
               "right_vector_max_diff_over_max": float(np.abs(rv_c - rv_g).max() / np.abs(rv_c).max()),
               "final_loss_rel_diff": float(abs(ofit.loss_history[-1] - gp.loss_history[-1]) / abs(ofit.loss_history[-1])),
               "gp_rel_l2_cpu_fit": round(rel_l2(ofit.predict(xs)[:, 0], ex), 6),
               "gp_rel_l2_gpu_fit": round(rel_l2(np.asarray(gp.predict(np.asarray(wl.x_t[:1024])), dtype=np.float64)[:, 0], ex), 6),
               "note": "the oracle's own fit on the same training set (NumPy float64, eigh factor) against the device fit (gp_train[0].fit_s); predictions on "
                       "the first 1024 roots"}
    return {"value": round(ns * wl.steps_exec / t_cpu, 1), "unit": "path-steps/s", "cores": threads, "kind": "port", "fit": fit,
            "sample": "%d of the %d roots, same inputs and Philox streams, NumPy float64 oracle (oracle/mlp.py%s), %.1f s"
                      % (ns, B, (" + oracle/gp_compat.py" if gp.compat == "reference" else " + oracle/gp.py") if gp is not None else "", t_cpu),
            "rel_l2_gpu": round(rel_gpu, 6), "rel_l2_cpu": round(rel_cpu, 6), "abs_diff": round(abs(rel_gpu - rel_cpu), 7),
            "abs_diff_bound_north_star": 1e-3,
            "max_abs_diff_u": float(np.nanmax(diff[:, 0])), "max_abs_diff_uz": float(np.nanmax(diff)),
            "frac_elements_beyond_1e-4": round(float((diff > 1e-4).mean()), 5),
            "note": "uz is clipped to +-%g; with the as-coded surrogate u_hat and eps_PDE are float16 VALUES, so one kernel entry whose float16 "
                    "rounding is decided on a float32 value here and a float64 value there moves u_hat by a float16 ulp (2.4e-4..4.9e-4) and a z "
                    "component by that times N / (MC delta_t)" % float(eng.problem().clip) if (gp is not None and gp.compat == "reference") else None}


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
    dom, bdy, xt = harness_sets(eq, 1000, 200)
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


# =========================================================================================== rooflines
def kernel_source_sha1(files):
    """Hash of the sources a kernel is built from: PMC summaries under profiles/ carry the hash of the code they were taken on, and
    a summary of other code is not quoted."""
    import hashlib
    h = hashlib.sha1()
    for f in files:
        h.update(open(os.path.join(ROOT, "scasml_gp_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def pmc_summaries(pattern):
    """profiles/r*_<pattern>.json, newest round first.  HBM bytes and issue-slot counters per launch come from separate rocprofv3 --pmc passes
    of this same command, condensed by profiles/summarize.py (they cannot be collected inside this process): NOT measured in this run, labelled
    so, and quoted only if they were taken on the SAME kernel source."""
    for prof in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            yield os.path.relpath(prof, ROOT), json.load(open(prof))
        except Exception:
            continue


def gp_eval_counters(args, n_inf, n_colloc):
    """(traffic, traffic_source, valu_issue, vector) of the GP evaluation kernel from the newest matching PMC summary, or Nones."""
    sha = kernel_source_sha1(GP_EVAL_SOURCES[args.compat])
    for rel, pj in pmc_summaries("r*_gp_eval_pmc.json"):
        if not (pj.get("n_inf") == n_inf and pj.get("d") == args.d and pj.get("source_sha1") == sha and pj.get("mode", "reference") == args.compat):
            continue
        source = "%s (separate rocprofv3 --pmc passes of this command on this kernel source; FETCH_SIZE doubled per MI355X_MICROARCH.md)" % rel
        issue, vector = None, None
        if pj.get("valu_active_frac") is not None:
            issue = {"valu_active_frac": round(pj["valu_active_frac"], 3), "mfma_pipe_busy_frac": round(pj["mfma_pipe_busy_frac"], 3),
                     "coexec_frac_of_mfma_busy": round(pj["coexec_frac_of_mfma_busy"], 3),
                     "cycles_per_valu_instruction": round(pj["cycles_per_valu_instruction"], 2),
                     "effective_clock_ghz": round(pj.get("effective_clock_ghz", 0.0), 3), "source": rel,
                     "note": "vector time and matrix time ADD in this kernel (ablations and instruction-level microbenchmarks: profiles/HISTORY.md, "
                             "profiles/r04_ubench_hetero.txt): the launch is t_vector + t_matrix, the counters' busy fractions overlap only in issue; "
                             "`frac` is of the matrix roof alone, `vector` states the other term"}
            cnt = pj.get("counters_avg_per_launch", {})
            if cnt.get("SQ_INSTS_VALU") and pj.get("effective_clock_ghz"):
                # the binding roof: vector wave-instructions per SIMD x the issue cost of THIS instruction mix with >= 3 waves per SIMD
                # (tools/ubench_hetero.hip, vector-only rows: 2.02 cycles for plain float32, 3.12 for the as-coded epilogue's mix of plain /
                # float16-conversion / exp instructions; the factored epilogue's 19 plain + 3 exp (8.3 cycles) per 22: 2.9) / the clock the chip held
                mix = 3.12 if args.compat == "reference" else 2.9
                clk = pj["effective_clock_ghz"] * 1e9
                per_simd = cnt["SQ_INSTS_VALU"] / 1024.0
                t_v = per_simd * mix / clk * 1e3
                t_m = cnt.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / clk * 1e3
                vector = {"valu_wave_instructions_per_launch": cnt["SQ_INSTS_VALU"],
                          "per_pair": round(cnt["SQ_INSTS_VALU"] * 64.0 / (n_inf * float((n_colloc + 31) // 32 * 32)), 2),
                          "issue_cycles_per_instruction_of_this_mix": mix, "clock_ghz": round(pj["effective_clock_ghz"], 3),
                          "vector_ms": round(t_v, 2), "vector_ms_at_2_cycles_per_instruction": round(per_simd * 2.0 / clk * 1e3, 2),
                          "matrix_ms": round(t_m, 2), "sum_model_ms": round(t_v + t_m, 2), "profiled_launch_ms": round(pj["avg_ms_kernel_trace"], 2),
                          "frac_of_sum_model": round((t_v + t_m) / pj["avg_ms_kernel_trace"], 3), "source": rel + " + profiles/r04_ubench_hetero.txt",
                          "note": "a MFMA-only wave beside vector-only waves on one SIMD starves the vector waves (one instruction per 16 cycles each), and "
                                  "waves that interleave both pay ~8 issue cycles per MFMA only in a uniform stream: the kernel's launch time is the sum"}
        return pj.get("hbm_bytes_per_launch"), source, issue, vector
    return None, None, None, None


def mlp_roofline(wl, kernel_ms):
    """Plain MLP: the whole recursion is one kernel with no HBM traffic between the root row and the result; priced with the
    materialised-state model of SURVEY.md 8(d) (16*d bytes per path-step) for comparability."""
    ms = kernel_ms["picard_mlp"]
    gbs = wl.B * wl.steps_exec * 16.0 * wl.d / (ms * 1e-3) / 1e9
    return {"kernel": "picard_tree_kernel (MODE_MLP)", "bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": 2.0 * wl.B * (wl.d + 1) * 4, "avg_launch_ms": round(ms, 4),
            "note": "algorithmic bytes of the materialised model; the fused kernel keeps all path state in VGPRs, "
                    "its real traffic is the root rows in and out, and it is Philox/ALU-bound"}


def gp_eval_roofline(args, wl, gp_ms):
    """The dominant kernel (fused GP evaluation, MFMA-bound): achieved = ALGORITHMIC flops of SURVEY.md 8(d) per launch
    (2 N_inf N (d+1) + 10 N_inf M) over the HIP-event launch duration measured in the timed region."""
    from scasml_gp_amd import _lib
    gp, d, B = wl.gp, wl.d, wl.B
    n_dom, n_bdy = args.train_domain, args.train_boundary
    n_colloc, m_feat = n_dom + n_bdy, 4 * n_dom + n_bdy
    n_inf = B * (wl.steps_exec + 1)
    flops = n_inf * (2.0 * n_colloc * (d + 1) + 10.0 * m_feat)
    traffic, traffic_source, issue, vector = gp_eval_counters(args, n_inf, n_colloc)
    ach = flops / (gp_ms * 1e-3) / 1e12
    kp = int(_lib.load().scasml_point_stride(d))           # the kernels' padded row length (round_up(d + 4, 16))
    n_pad = (n_colloc + 31) // 32 * 32
    common = {"bound": "mfma", "achieved": round(ach, 3), "unit": "TFLOP/s", "traffic": traffic, "traffic_source": traffic_source,
              "valu_issue": issue, "avg_launch_ms": round(gp_ms, 4), "flops_per_launch": flops}
    if gp.compat == "reference":
        # issued 16-bit MFMA flops: per point and geometry `planes` planes of (n_pad x kp) plus one K = 16 Hutchinson product; sites that
        # consume eps_PDE run three geometries on domain tiles and two on boundary tiles, the others two and one
        kinds = wl.eng.site_kinds(wl.n, wl.par).cpu().numpy()
        n_full, n_part = int((kinds == 0).sum()) * B, int(np.isin(kinds, (1, 3, 4)).sum()) * B
        nd_pad = (n_dom + 31) // 32 * 32
        planes = 1 if (int(gp.eval_round16) & 4) else 2
        per_geom = lambda rows, q: 2.0 * rows * (planes * kp + (16 if q else 0))
        issued = (n_full * (3 * per_geom(nd_pad, True) + 2 * per_geom(n_pad - nd_pad, True))
                  + n_part * (2 * per_geom(nd_pad, True) + per_geom(n_pad - nd_pad, False))) / (gp_ms * 1e-3) / 1e12
        peak = MFMA_BF16_PEAK_TFLOPS
        # what the as-coded surrogate needs ALGORITHMICALLY: three distinct distance matrices (|x - y|, |x - y'|, |x' - y|) where eps_PDE is
        # consumed, two elsewhere (one on boundary rows), instead of the one of the documented operators
        flops_ac = 2.0 * (d + 1) * (n_full * (3 * n_dom + 2 * n_bdy) + n_part * (2 * n_dom + n_bdy)) + 10.0 * n_inf * m_feat
        ach_ac = flops_ac / (gp_ms * 1e-3) / 1e12
        geometry = not (int(gp.eval_round16) & 1)
        common.update({
            "kernel": ("gp_eval_compat_mfma_kernel (compat='reference-geometry': the as-coded fit, 3 shifted geometries x %d fp16 plane%s, entries "
                       "not rounded: factored sums)" % (planes, "" if planes == 1 else "s")) if geometry else
                      "gp_eval_compat_mfma_kernel (as-coded surrogate: 3 shifted geometries x 2 fp16 planes, float16 entries)",
            "peak": peak, "frac": round(ach / peak, 4), "vector": vector,
            # what binds this kernel is the SUM of its vector time and its matrix time, not the matrix roof (`frac`, kept as the contract's A / P):
            # the primary fraction is modelled (vector + matrix) time over the profiled launch, from the PMC passes on this kernel source
            "bound": "valu+mfma (sum model)", "primary_fraction": "frac_of_sum_model",
            "frac_of_sum_model": vector["frac_of_sum_model"] if vector else None,
            "frac_of_matrix_roof": round(ach / peak, 4),
            "mfma_issued_tflops": round(issued, 1), "mfma_issued_frac": round(issued / peak, 4),
            "as_coded": {"flops_per_launch": flops_ac, "achieved": round(ach_ac, 3), "frac": round(ach_ac / peak, 4),
                         "note": "algorithmic flops of the surrogate the reference's code builds: 3 x.y products per pair where eps_PDE is "
                                 "consumed (%d of %d sites), 2 elsewhere" % (int((kinds == 0).sum()), len(kinds))},
            "note": ("achieved = algorithmic flops of SURVEY 8(d) (ONE x.y product per pair); the geometry mode keeps the as-coded surrogate's three "
                     "shifted x.y products (one fp16 plane of the point each) and drops the float16 rounding of the 13 entries per pair, so that the "
                     "four sums factor per geometry (13 + 9 + 5 vector instructions + 3 exp per pair against ~50 + 3); DESIGN.md 4") if geometry else
                    ("achieved = algorithmic flops of SURVEY 8(d) (ONE x.y product per pair); the as-coded surrogate needs three (aligned, y shifted, "
                     "x shifted) in two fp16 planes each, and 13 separately float16-rounded entries per pair in the epilogue (~50 vector instructions "
                     "+ 3 exp against 14 + 1 for the documented operators): the launch time is vector time plus matrix time (valu_issue, vector; DESIGN.md 4)")})
        return common
    split = int(gp.eval_split)
    products = {0: 1, 2: 3, 3: 6, 22: 2 if getattr(gp, "_colloc_is_f16", False) else 3}[split]
    issued = products * 2.0 * n_inf * n_pad * kp / (gp_ms * 1e-3) / 1e12      # MFMA flops actually issued
    peak = MFMA_F32_PEAK_TFLOPS if split == 0 else MFMA_BF16_PEAK_TFLOPS
    common.update({
        "kernel": "gp_eval_kernel (fp32 MFMA)" if split == 0 else ("gp_eval_bf16_kernel (2 fp16 planes, exponent-unit epilogue)" if split == 22 else "gp_eval_bf16_kernel (%d bf16 planes)" % split),
        "peak": peak, "frac": round(ach / peak, 4), "achieved_vs_fp32_mfma_peak": round(ach / MFMA_F32_PEAK_TFLOPS, 4),
        "mfma_issued_tflops": round(issued, 1), "mfma_issued_frac": round(issued / peak, 4),
        "note": "achieved = algorithmic fp32 flops (SURVEY 8(d)); the split-precision kernel issues %dx as many 16-bit MFMA flops to keep products "
                "exact to 2^-22 (plus one K = 8 MFMA per tile for the bilinear part of the epilogue); the vector ALUs are the busier pipe" % products})
    return common


def path_roofline(wl, kernel_ms):
    """The path kernels (GENERATE + ACCUMULATE), priced with the materialised-state model of SURVEY.md 8(d): 16*d bytes per path-step;
    `frac` is what they really move (PMC passes condensed by profiles/summarize.py; not measured in this run) over the HBM peak."""
    from scasml_gp_amd import _lib
    path_ms = (kernel_ms.get("picard_generate") or 0.0) + (kernel_ms.get("picard_accumulate") or 0.0)
    if not path_ms:
        return None
    B, d = wl.B, wl.d
    gbs = B * wl.steps_exec * 16.0 * d / (path_ms * 1e-3) / 1e9
    roof = {"kernels": "picard_tree generate+accumulate", "bound": "hbm", "achieved": round(gbs, 2),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "model_frac": round(gbs / HBM_PEAK_GBS, 4), "avg_launch_ms": round(path_ms, 4),
            "model": "16*d algorithmic bytes per path-step (SURVEY.md 8(d)): the materialised-state model counts X and W read and written per step; "
                     "the kernels keep W in registers and move fewer real bytes, so model_frac overstates the HBM utilisation -- `frac` is real "
                     "traffic (PMC) over peak"}
    kp = int(_lib.load().scasml_point_stride(d))
    grids = ((B * (kp // 4) + 255) // 256 * 256, B * 32)
    sha = kernel_source_sha1(PICARD_SOURCES)
    for rel, pj in pmc_summaries("r*_picard_pmc.json"):
        if pj.get("source_sha1") != sha:
            continue
        real = sum(v["hbm_bytes_per_launch"] for v in pj.values() if isinstance(v, dict) and v.get("grid_threads") in grids)
        if real:
            roof.update({"traffic": real, "traffic_gb_per_s": round(real / (path_ms * 1e-3) / 1e9, 1),
                         "frac": round(real / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic_source": "%s (separate rocprofv3 --pmc passes on this kernel source)" % rel})
            break
    return roof


def xl_solver_roofline(flops, gp_ms):
    """The evaluation kernel of configs[4]'s solver half (gp_eval_compat_mfma_kernel<16, 2, true, 2>: 16 K-steps, 625 collocation tiles, a 34.6 MB model):
    achieved from this run's HIP events; traffic and the vector + matrix sum model from the PMC passes of tools/xl_solver_counters.sh, quoted only
    for the same kernel source."""
    ach = flops / (gp_ms * 1e-3) / 1e12
    roof = {"kernel": "gp_eval_compat_mfma_kernel<16, 2, true, 2>", "bound": "valu+mfma (sum model)", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "flops_per_launch": flops, "avg_launch_ms": round(gp_ms, 3), "traffic": None}
    path = os.path.join(ROOT, "profiles", "r06_gp_eval_pmc_d250.json")
    if os.path.exists(path):
        pj = json.load(open(path))
        if pj.get("source_sha1") == kernel_source_sha1(GP_EVAL_SOURCES["reference"]):
            roof.update({"traffic": pj["hbm_bytes_per_launch"], "l2_hit_rate": round(pj["l2_hit_rate"], 3), "sum_model": pj["sum_model"],
                         "mfma_pipe_busy_frac": round(pj["mfma_pipe_busy_frac"], 3), "valu_active_frac": round(pj["valu_active_frac"], 3),
                         "traffic_source": "profiles/r06_gp_eval_pmc_d250.json (separate rocprofv3 --pmc passes on this kernel source; FETCH_SIZE doubled, Infinity-Cache hits counted)"})
    return roof


def gp_train_block(d, n_dom, n_bdy, compat=None, reps=2, keep=None):
    """GP training stages (models/GP.py:182-268, 487-604) with their rooflines: Gram, Cholesky (M^3/3 flop), K_p^-1 from the
    factor (2 M^3 / 3), the Newton iteration; HIP events per stage.  The first of two passes warms code objects and the allocator."""
    import torch
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    eq = Grad_Dependent_Nonlinear(d + 1)
    dom, bdy, _ = harness_sets(eq, n_dom, n_bdy)
    out = None
    for rep in range(reps):
        gp = GP_Grad_Dependent_Nonlinear(eq, compat=compat)
        gp.profile = rep == reps - 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gp.GPsolver(dom, bdy, GN_steps=20)
        torch.cuda.synchronize()
        fit_s = time.perf_counter() - t0
        if gp.profile:
            M, ms, N = gp.phi_dim, gp.stage_ms, n_dom + n_bdy
            chol_tf = M ** 3 / 3.0 / (ms["cholesky"] * 1e-3) / 1e12
            inv_tf = 2.0 * M ** 3 / 3.0 / (ms["inverse"] * 1e-3) / 1e12
            gram_tf = (2.0 * N * N * (d + 1) + 8.0 * M * M) / (ms["gram"] * 1e-3) / 1e12
            out = {"d": d, "collocation": "%d+%d" % (n_dom, n_bdy), "surrogate": "as coded (compat='reference')" if compat else "documented operators",
                   "M": M, "K_gb_f64": round(M * M * 8 / 1e9, 2), "fit_s": round(fit_s, 3), "newton_steps": len(gp.loss_history) - 1,
                   "gram_ms": round(ms["gram"], 3), "gram_tflops": round(gram_tf, 2),
                   "cholesky_ms": round(ms["cholesky"], 3), "cholesky_tflops": round(chol_tf, 2),
                   "cholesky_frac_of_fp64_mfma_peak": round(chol_tf / FP64_MFMA_PEAK_TFLOPS, 4),
                   "inverse_ms": round(ms["inverse"], 3), "inverse_tflops": round(inv_tf, 2),
                   "inverse_frac_of_fp64_mfma_peak": round(inv_tf / FP64_MFMA_PEAK_TFLOPS, 4),
                   "peak_fp64_mfma_tflops": FP64_MFMA_PEAK_TFLOPS, "warm": reps > 1}
            if keep is not None:                          # the fitted surrogate goes on to a solver run (configs[4] staged)
                gp._L_pad = gp.cholesky_phi_phi_perturb = None
                keep.update(eq=eq, gp=gp)
        del gp
        torch.cuda.empty_cache()
    return out


def gp_train_blocks(args, gp, ranks, others):
    """The bench's own fit, the staged size of BASELINE configs[4] (M = 34 999) and, on request, M = 70 001 (past 2^31 matrix elements) -- whose
    as-coded surrogate then serves configs[4]'s SOLVER half at a fifth of its collocation count: ScaSML n = rho = 3 at d = 250 on 1024 roots
    ("report Gram + Cholesky TFLOP/s and solver steps/s separately"), appended to `others`."""
    blocks = [gp_train_block(args.d, args.train_domain, args.train_boundary, gp.compat)]
    if not args.no_gp_train_large:                       # staged configs[4]: the MFMA Gram exists for the documented operators
        blocks.append(gp_train_block(250, 8333, 1667, None))
    if args.gp_train_xl:
        blocks.append(gp_train_block(250, 16667, 3333, None, reps=1))
        kept = {}
        blocks.append(gp_train_block(250, 16667, 3333, "reference", reps=1, keep=kept))
        if others is not None and kept:
            wl = Workload(kept["eq"], kept["gp"], "scasml", "quad", 3, 3, 1 << 10, ranks.rank)
            elapsed, kms = measure(ranks, wl, wl.step, 5, 2)
            n_inf = wl.B * (wl.steps_exec + 1)
            flops = n_inf * (2.0 * 20000 * 251 + 10.0 * 70001)
            others.append({"workload": "Grad_Dependent_Nonlinear d=250, %s, B=%d roots (BASELINE.json configs[4] staged: the solver on the as-coded surrogate "
                                       "of 16667+3333 collocation points, M = 70 001)" % (wl.name, wl.B), "steps": 5, "warmup": 2,
                           "ms_per_step": round(elapsed / 5 * 1e3, 3), "value": round(wl.B * wl.steps_exec * 5 / elapsed, 1), "unit": "path-steps/s",
                           "path_steps_per_root": wl.steps_exec, "path_steps_per_root_reference_count": wl.steps_ref,
                           "kernel_ms": {k: round(v, 4) for k, v in kms.items()},
                           "gp_eval_algorithmic_tflops": round(flops / (kms["gp_eval"] * 1e-3) / 1e12, 1),
                           "gp_eval_frac_of_fp16_mfma_peak": round(flops / (kms["gp_eval"] * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                           "roofline": xl_solver_roofline(flops, kms["gp_eval"]),
                           "note": "16 K-steps of 16 per x.y product (d = 250) and 20 000 collocation rows per point: 1.36e10 (point, row) pairs per step, "
                                   "as many as the headline's"})
    return blocks


# =========================================================================================== the line
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    ranks = Ranks(args)
    world, rank, d, n = ranks.world, ranks.rank, args.d, args.level
    B = args.batch if args.batch else ((1 << 20) if args.solver == "mlp" else (1 << 14))
    eq = Grad_Dependent_Nonlinear(d + 1)
    eq.geometry()

    # ---- setup (untimed): train the surrogate on 1000 + 200 collocation points; synthetic roots resident in HBM
    x_dom, x_bdy, xt_h = harness_sets(eq, args.train_domain, args.train_boundary)
    gp, t_train = fit_surrogate(eq, x_dom, x_bdy, args.compat) if args.solver == "scasml" else (None, 0.0)
    wl = Workload(eq, gp, args.solver, args.variant, n, args.M, B, rank, rng=args.rng, compat_f16=args.compat_f16)
    sharding = SampleSharding(ranks, wl, args)
    by_samples = args.shard == "samples" and world > 1
    main_step = sharding.step if by_samples else wl.step

    # ---- the timed region: W warm-up + K steps of the hot path, barrier + synchronize on both sides, MAX over ranks
    elapsed, kernel_ms = measure(ranks, wl, main_step, args.steps, args.warmup)
    samples_leg = None
    if world > 1:
        t_other = timed_leg(ranks, wl.step if by_samples else sharding.step, args.steps, args.warmup)
        t_s, t_r = (elapsed, t_other) if by_samples else (t_other, elapsed)
        sharding.compute_events = []                          # a third, short pass: every rank's own compute time per sharded step
        timed_leg(ranks, sharding.step, min(args.steps, 5), 0)
        samples_leg = sharding.report(t_s, t_r, args.steps, sharding.rank_compute_ms())
        samples_leg["max_abs_diff_vs_unsharded"] = sharding.max_abs_diff_vs_unsharded()
    selftest = rccl_selftest(ranks, B, d) if (ranks.on and world == 1 and not ranks.on_host) else None
    dist_gp = dist_gp_check(ranks) if (world > 1 and os.environ.get("SCASML_BENCH_DIST_GP") == "1") else None
    if rank != 0:
        ranks.close()
        return

    # ---- accuracy on the harness protocol (untimed): 1000 + 200 test points
    exact = eq.exact_solution(xt_h)
    u_gpu = wl.solver.u_solve(n, wl.par, xt_h) if args.variant == "quad" else wl.solver.u_solve(n, None, xt_h, args.M)
    rel_gpu = rel_l2(u_gpu, exact)
    rel_gp = rel_l2(gp.predict(xt_h), exact) if gp is not None else None
    ref_logs = reference_logs_check() if not args.no_reference_logs_check else None

    # ---- rooflines from the HIP-event durations of the timed region; the other configurations; training stages; CPU baseline
    roofline = mlp_roofline(wl, kernel_ms) if kernel_ms.get("picard_mlp") else None
    if kernel_ms.get("gp_eval"):
        roofline = gp_eval_roofline(args, wl, kernel_ms["gp_eval"])
    others = other_runs(ranks, args, eq if d == 100 else None, gp if d == 100 else None, x_dom, x_bdy) if (world == 1 and not args.no_other_runs) else None
    gp_train = gp_train_blocks(args, gp, ranks, others) if (world == 1 and gp is not None) else None
    cpu = cpu_baseline(args, wl, x_dom, x_bdy) if (world == 1 and not args.no_cpu_baseline) else None

    work_ranks = 1 if by_samples else world                  # samples: all ranks share the same B roots
    value = work_ranks * B * wl.steps_exec * args.steps / elapsed
    surrogate = None
    if gp is not None:
        surrogate = "reference's as-coded GP (compat='reference', Hutchinson indices %s)" % gp.laplacian_idx.tolist() if gp.compat == "reference" \
            else "documented operators (compat=None)"
    line = {
        "metric": "Euler-Maruyama path-steps/sec + L2 rel-error, Grad_Dependent_Nonlinear d=%d n=%d" % (d, n),
        "value": round(value, 1), "unit": "path-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "strong" if by_samples else "weak",
        # N > 1: the headline `value` above is the collective-free ROOTS leg unless --shard samples; the north-star leg (Monte-Carlo samples
        # of the root call shared by the ranks, one all-reduce per step) is this block -- not to be mistaken for the headline's scaling
        "north_star_samples_leg": ({k: samples_leg[k] for k in ("value", "ms_per_step", "scaling", "sample_ranks", "root_groups",
                                                                 "efficiency_vs_unsharded_step", "predicted_efficiency")}
                                   if samples_leg else None),
        "rccl_ranks": ranks.dist.get_world_size() if ranks.on else 1,
        "rccl_note": "no multi-GPU node has been available to this build: the N > 1 path (init_process_group('nccl'), the in-group all-reduce) is "
                     "rehearsed over gloo on one GPU only (tests/test_gpu_bench_contract.py) until a SCALE run exists; with SCASML_BENCH_FORCE_DIST=1 the "
                     "same branches run over RCCL with one rank (rccl_selftest)",
        "backend": (ranks.dist.get_backend() if ranks.on else None), "rccl_selftest": selftest, "samples_sharding": samples_leg, "dist_gp_check": dist_gp,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "Grad_Dependent_Nonlinear d=%d, %s, B=%d roots/GPU%s" % (
                       d, wl.name, B, " (BASELINE.json configs[2])" if (args.solver, args.variant, d, n) == ("scasml", "quad", 100, 3) else ""),
                   "surrogate": surrogate, "roots_per_gpu": B,
                   "gp_collocation": ("%d+%d" % (args.train_domain, args.train_boundary)) if gp is not None else None,
                   "path_steps_per_root": wl.steps_exec, "path_steps_per_root_reference_count": wl.steps_ref,
                   "gp_point_evals_per_root": wl.steps_exec + 1,
                   "sharding": "Monte-Carlo units of the root call over %d ranks (dealt by cost), one all-reduce of (B, 1+d) partial sums; roots over %d groups"
                               % (sharding.S, sharding.G) if by_samples else "roots across ranks, no collective",
                   "note": "value counts only EXECUTED path-steps (the reference's discarded n=0 terminal draws are not "
                           "performed); with the reference's own count the same run is value_reference_count"},
        "value_reference_count": round(work_ranks * B * wl.steps_ref * args.steps / elapsed, 1),
        "reference_logs_check": ref_logs,
        "l2_rel_error": {"solver_gpu": round(rel_gpu, 5), "gp_only": round(rel_gp, 5) if rel_gp is not None else None,
                         "points": "1000+200 harness set (np.random.seed(1234): the training draw, then this one, as tests/SimpleUniform.py)",
                         "vs_cpu_oracle": ({k: cpu[k] for k in ("rel_l2_gpu", "rel_l2_cpu", "abs_diff")} if cpu else None),
                         "surrogate": ("as coded by the reference (compat='reference')" if gp.compat == "reference" else "documented operators (compat=None)") if gp is not None else None},
        "kernel_ms": {k: round(v, 4) for k, v in kernel_ms.items()},
        "gp_train_s": round(t_train, 2),
        "roofline": roofline, "roofline_path": path_roofline(wl, kernel_ms), "other_runs": others, "gp_train": gp_train, "cpu_baseline": cpu,
        # LAST key, compact: the driver's record keeps the tail of this line -- [label, ms_per_step, path-steps/s] of every other run
        "other_runs_summary": other_runs_summary(others),
    }
    print(json.dumps(line))
    ranks.close()


if __name__ == "__main__":
    main()
