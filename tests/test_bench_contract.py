"""bench.py keeps the driver's contract: one JSON line with the required fields (a reduced workload here)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [1, 3])
def test_bench_prints_one_contract_line(pipeline):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--bricks", "6", "--steps", "2", "--warmup", "1",
                        "--cpu-seconds", "1", "--pipeline", str(pipeline)], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in REQUIRED:
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["unit"] == "Mvoxels/s" and j["dtype"] == "u8" and j["vs_baseline"] is None and j["scaling"] == "weak"
    assert j["value"] > 0 and "workload" in j["config"]
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
