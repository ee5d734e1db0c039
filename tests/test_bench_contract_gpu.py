"""The driver's contract for bench.py (one JSON line on stdout, fixed keys, roofline + cpu_baseline objects), checked on a small
scene; and the N-rank entry point rehearsed with two ranks on the one GPU of the test box (host-staged exchange over gloo)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline")


def _one_line(out):
    lines = [l for l in out.strip().splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout, got %d" % len(lines)
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_single_gpu(hip_lib):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "40,40,40", "--steps", "4", "--warmup", "2",
                        "--spin-up", "60", "--resting-steps", "10"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    for k in KEYS + ("cpu_baseline",):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "particle-steps/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    n = d["config"]["particles"]
    assert abs(d["value"] - n * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) <= 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["peak"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    # round 3: the timed region follows an untimed spin-up under a CFL-respecting time step, both declared in the line
    assert d["config"]["spin_up_steps"] == 60 and d["config"]["first_timed_step"] == 62 and abs(d["config"]["dt"] - 2.5e-4) < 1e-9
    assert d["cfl_ok"] is True and d["developed"]["cfl_dt_limit"] >= d["developed"]["dt"]
    assert d["developed"]["first_step"] == 62 and d["resting"]["first_step"] == 20 and d["resting"]["steps"] == 10
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1


@pytest.mark.gpu
def test_bench_line_two_ranks_one_gpu_gloo(hip_lib):
    env = dict(os.environ, NEREUS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "48,40,40", "--steps", "4",
                        "--warmup", "2", "--spin-up", "120"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    for k in KEYS + ("backend",):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["backend"].startswith("gloo")
    assert d["config"]["particles"] == 2 * 48 * 40 * 40
    # the same window rule as the one-GPU line: spin-up (with count-balanced re-cuts every 50 steps) under the CFL-stable dt
    assert d["config"]["spin_up_steps"] == 120 and d["config"]["rebalance_every"] == 50 and d["cfl_ok"] is True
