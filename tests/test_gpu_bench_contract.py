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
@pytest.mark.parametrize("extra", [[], ["--solver", "mlp", "--d", "20", "--level", "2"]])
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
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and "traffic" in roof
    cpu = j["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["unit"] == "path-steps/s" and cpu["sample"]
    assert cpu["max_abs_diff_gpu_vs_cpu"] < 1e-3
