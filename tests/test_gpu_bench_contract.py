"""bench.py must keep printing exactly one JSON line with the driver's contract (metric, value, unit, n_gpus,
steps, warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload, roofline,
cpu_baseline); run on a tiny workload."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--compat", "none"], ["--solver", "mlp", "--d", "20", "--level", "2"]])
def test_bench_prints_one_contract_line(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "256",
           "--train-domain", "96", "--train-boundary", "32", "--cpu-sample", "2"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str)):
        assert isinstance(j[key], typ), key
    assert j["vs_baseline"] is None and j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["unit"] == "path-steps/s" and j["value"] > 0 and j["data"] == "synthetic" and "workload" in j["config"]
    roof = j["roofline"]
    # the as-coded evaluation kernel is bound by vector time PLUS matrix time: labelled so, with `frac` still achieved / (matrix) peak
    assert roof["bound"] in ("hbm", "mfma", "valu+mfma (sum model)") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert (roof["bound"] == "valu+mfma (sum model)") == (not extra) and (extra or roof["primary_fraction"] == "frac_of_sum_model")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and "traffic" in roof
    cpu = j["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["unit"] == "path-steps/s" and cpu["sample"]
    # north_star: L2 relative error within 1e-3 of the reference CPU path, on the same roots
    assert cpu["abs_diff"] <= 1e-3 and abs(cpu["rel_l2_gpu"] - cpu["rel_l2_cpu"]) <= 1e-3 and cpu["max_abs_diff_u"] < 1e-3
    if "--solver" not in extra:                         # the oracle's OWN fit on the same training set beside the device fit (timed, compared)
        fit = cpu["fit"]
        assert fit["M"] == 4 * 96 + 32 and fit["cpu_s"] > 0 and fit["newton_steps_cpu"] == fit["newton_steps_gpu"]
        assert fit["right_vector_max_diff_over_max"] <= 1e-6 and abs(fit["gp_rel_l2_cpu_fit"] - fit["gp_rel_l2_gpu_fit"]) <= 1e-3
    chk = j["reference_logs_check"]                     # the reference's own d = 20 experiment on its own random stream, against its logs
    assert chk["d"] == 20 and set(chk["rel_l2"]) == {"GP", "MLP", "ScaSML", "MLP_full_history"} and chk["max_relative_difference"] <= 6e-3
    assert abs(chk["rel_l2"]["MLP"] - chk["logged"]["MLP"]) <= 5e-4 * chk["logged"]["MLP"]
    if not extra:                                     # the default is the reference's as-coded surrogate
        assert "as-coded" in j["config"]["surrogate"] and j["l2_rel_error"]["vs_cpu_oracle"]["abs_diff"] <= 1e-3
    # the other BASELINE configurations and modes, timed in the same process (5 steps each)
    runs = j["other_runs"]
    assert "configs[1]" in runs[0]["workload"] and "B=1048576" in runs[0]["workload"] and runs[0]["kernel_ms"]["picard_mlp"] > 0
    # ... and their compact summary as the LAST key of the line (the driver's record keeps the line's tail)
    assert list(j)[-1] == "other_runs_summary" and len(j["other_runs_summary"]) == len(runs) and len(json.dumps(j["other_runs_summary"])) < 600
    assert all(abs(a[1] - r["ms_per_step"]) < 1e-9 for a, r in zip(j["other_runs_summary"], runs)) and j["north_star_samples_leg"] is None
    if "--solver" not in extra:
        assert len(runs) == (4 if not extra else 3) and "configs[3]" in runs[1]["workload"] and "parity" in runs[2]["workload"]
        assert all(r["value"] > 0 and r["ms_per_step"] > 0 and r["steps"] == 5 and r["kernel_ms"]["gp_eval"] > 0 for r in runs[1:])
        assert not extra or "reference-geometry" not in runs[-1]["workload"]


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,variant,level,expect_s", [(4, "fh", "3", 4), (4, "quad", "3", 4), (2, "quad", "2", 2)])
def test_plain_invocation_spawns_its_ranks_and_reports_both_shardings(ranks, variant, level, expect_s):
    """`python bench.py --gpus N` with no launcher around it: the parent spawns the N ranks (here sharing the one GPU of
    the test box and meeting over gloo: --rehearse-on-one-gpu), rank 0 prints the one line, which carries the root-sharded
    (weak) headline and the samples_sharding block of the north-star split with its dealt-load imbalance."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--batch", "96",
           "--train-domain", "96", "--train-boundary", "32", "--d", "20", "--variant", variant, "--level", level,
           "--rehearse-on-one-gpu", "--min-sample-efficiency", "0"]   # 0: all ranks share the samples (a 96-root step is all fixed cost)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    if ranks == 2:
        env["SCASML_BENCH_DIST_GP"] = "1"             # + the distributed GP fit over the run's own process group (opt-in side check)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == ranks and j["rccl_ranks"] == ranks and j["scaling"] == "weak" and j["cpu_baseline"] is None
    s = j["samples_sharding"]
    assert s["sample_ranks"] == expect_s and s["sample_ranks"] * s["root_groups"] == ranks
    # the number of sample ranks is chosen by PREDICTED efficiency: the table of every divisor S, its modelled rank time and stated all-reduce cost
    per_s = s["predicted_from"]["per_S"]
    assert [row["sample_ranks"] for row in per_s] == [x for x in range(1, ranks + 1) if ranks % x == 0]
    assert all(0 < row["predicted_efficiency"] <= 1.0 and row["modelled_rank_ms"] > 0 and row["dealt_load_max_over_mean"] >= 1.0 for row in per_s)
    assert per_s[0]["allreduce_ms_stated"] == 0 and per_s[-1]["allreduce_ms_stated"] > 0 and per_s[-1]["dealt_load_sum_over_unsharded"] >= 1.0
    assert s["predicted_efficiency"] == per_s[-1]["predicted_efficiency"] and s["unit_load_imbalance_max_over_mean"] == per_s[-1]["dealt_load_max_over_mean"]
    assert s["predicted_from"]["unsharded_step_ms_calibration"] > 0 and 0 < s["efficiency_vs_unsharded_step"]
    assert len(s["rank_compute_ms"]) == ranks and min(s["rank_compute_ms"]) > 0 and s["rank_compute_ms_max_over_mean"] >= 1.0
    # ... and the leg stands next to the headline at top level, so that the collective-free roots leg is not read as the north-star number
    top = j["north_star_samples_leg"]
    assert top["value"] == s["value"] and top["sample_ranks"] == s["sample_ranks"] and top["scaling"] == "strong"
    assert top["efficiency_vs_unsharded_step"] == s["efficiency_vs_unsharded_step"] and top["predicted_efficiency"] == s["predicted_efficiency"]
    assert s["value"] > 0 and s["roots_leg"]["value"] > 0 and s["scaling"] == "strong"
    # the sharded estimator IS the unsharded one: same sites, same draws, same surrogate values; only the order of the additions differs
    assert s["max_abs_diff_vs_unsharded"] <= 2e-5, s["max_abs_diff_vs_unsharded"]
    assert abs(s["roots_leg"]["value"] - j["value"]) / j["value"] < 0.5      # same leg, timed twice
    if ranks == 2:
        g = j["dist_gp_check"]
        assert g["M"] == 7001 and g["ranks"] == 2 and g["backend"] == "gloo" and g["collective_calls"]["all_gather"] > 0
        assert g["right_vector_rel_diff_vs_single_gpu"] <= 1e-6 and g["newton_steps"] == g["newton_steps_single_gpu"]
    else:
        assert j["dist_gp_check"] is None


@pytest.mark.gpu
def test_sample_sharded_result_does_not_depend_on_the_rank_count():
    """World = 8 rehearsal of the north-star split on one GPU: the partial estimators of 8 sample ranks (units dealt by
    cost) add up to the single-rank result -- ScaSML_full_history n = 4, M = 3 (BASELINE configs[3] at a small batch)."""
    import numpy as np
    import torch
    from oracle.equation import sample_points
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.ScaSML_full_history import ScaSML_full_history
    d = 100
    dom, bdy = sample_points(np.random.default_rng(0), d, 96, 32)
    eq = Grad_Dependent_Nonlinear(d + 1)
    gp = GP_Grad_Dependent_Nonlinear(eq, compat=None)
    gp.GPsolver(dom, bdy)
    eng = ScaSML_full_history(eq, gp, seed=1)._engine
    xt = np.concatenate(sample_points(np.random.default_rng(1), d, 24, 8))
    full, _, _ = eng.solve(4, 3, xt, stream_id=0)
    for world in (2, 8):
        total = None
        for r in range(world):
            part, _, _ = eng.solve(4, 3, xt, rank=r, world=world, stream_id=0)
            total = part.clone() if total is None else total + part
        assert torch.allclose(eng.finalize_partials(total), full, atol=2e-5, rtol=1e-5), world
    owner, _, load = eng.unit_owners(4, 3, 8)
    assert len(owner) == 240 and load.max() / load.mean() < 1.25


@pytest.mark.gpu
def test_the_rccl_branches_execute_on_one_rank():
    """A one-GPU box cannot host two RCCL ranks; what it can show is that the branches the N > 1 run takes -- init_process_group("nccl"),
    the barriers and MAX-reduction around the timed region, an all-reduce of the path's (B, 1+d) partial-sum buffer, the teardown --
    execute on this image, with WORLD_SIZE = 1 (SCASML_BENCH_FORCE_DIST=1)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "256",
           "--train-domain", "96", "--train-boundary", "32", "--no-cpu-baseline", "--no-other-runs"]
    env = dict(os.environ, SCASML_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200),
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["backend"] == "nccl" and j["rccl_ranks"] == 1 and j["value"] > 0
    st = j["rccl_selftest"]
    assert st["ranks"] == 1 and st["unchanged"] and st["allreduce_ms"] > 0 and st["subgroup_allreduce"]
