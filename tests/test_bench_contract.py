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
    assert j["value_no_compact"] is None or j["value_no_compact"] > 0
    assert j["roofline_encode"]["bound"] == "hbm" and j["roofline_encode"]["achieved"] > 0
    assert j["value"] > 0 and "workload" in j["config"]
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    assert cb["host_cores"] >= 1 and cb["all_cores"]["threads_used"] >= 1 and cb["all_cores"]["value"] > 0


def test_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher: the parent starts two ranks itself (torch.distributed.run on
    127.0.0.1) before touching a GPU.  VRHIP_BENCH_DRYRUN=1 keeps the ranks off the GPU (gloo), so the launch,
    rendezvous and max-over-ranks plumbing runs on the CPU."""
    env = dict(os.environ, VRHIP_BENCH_DRYRUN="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, cwd=ROOT, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["dry_run"] is True
    assert j["max_rank_seconds"] >= 0.02          # the slower rank's time (MAX over ranks)
    # N > 1 defaults to BASELINE config 4: ONE volume, every one of its 960 bricks on exactly one rank
    assert j["scaling"] == "strong" and j["partition_ok"] is True and j["bricks_per_rank"] == [480, 480]


@pytest.mark.parametrize("n", [3, 4])
def test_strong_scaling_partition_other_rank_counts(n):
    env = dict(os.environ, VRHIP_BENCH_DRYRUN="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, cwd=ROOT, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["partition_ok"] is True and sum(j["bricks_per_rank"]) == 960 and len(j["bricks_per_rank"]) == n


def test_gpus_flag_must_match_world_size():
    env = dict(os.environ, VRHIP_BENCH_DRYRUN="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       cwd=ROOT, timeout=120, env=env)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
