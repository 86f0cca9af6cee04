"""GPU: bench.py prints ONE JSON line with the fields the driver reads, on a small workload (a child
process, as the driver runs it)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(*extra, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gaussians", "20000", "--width", "320", "--height", "200",
           "--views", "2", "--steps", "2", "--warmup", "1", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields():
    d = run_bench()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["value"] > 0 and d["unit"] == "renders/s" and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 2e-3)) < 0.01 * d["value"]       # views * steps / elapsed
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["achieved"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and c["sample"]


def test_bench_forward_only_and_training_iteration_lines():
    f = run_bench("--forward-only", "--no-cpu-baseline")
    assert f["roofline"]["kernel"].startswith("composite_fwd") and "cpu_baseline" not in f
    t = run_bench("--train-iter", "--no-cpu-baseline")
    assert "Adam" in t["config"]["workload"] and t["value"] > 0


def test_the_line_stays_alone_on_stdout_with_a_process_group():
    """The multi-rank code path rehearsed with one rank (process group over RCCL, both collectives, barrier): RCCL
    announces its version on standard output when its communicator comes up; bench.py's own output must still be
    the one JSON line."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29633",
               GSR_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    d = run_bench("--gpus", "1", "--no-cpu-baseline", env=env)
    assert d["n_gpus"] == 1 and d["exchange_ms_per_step"] is not None and d["exchange_ms_per_step"] > 0
    assert "RCCL all-reduce" in d["config"]["workload"]
