#!/usr/bin/env python3
"""bench.py -- kd-tree compress+decode throughput (Mvoxels/s) of the HIP hot path, with
the decode kernel's HBM roofline and the CPU oracle timed beside it.

  python bench.py --gpus N --steps K --warmup W
N > 1: launched by torch.distributed.run, one rank per GPU; bricks are sharded across
ranks with no data-path collective (weak scaling: every rank encodes+decodes the same
number of bricks).  A "step" = one build() + one levelCut() of the rank's brick batch.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def make_bricks(kind, n_bricks, dims, seed=12345):
    """Synthetic uint8 bricks.  dims = (X, Y, Z); returns [B][Z][Y][X]."""
    X, Y, Z = dims
    out = np.empty((n_bricks, Z, Y, X), np.uint8)
    z, y, x = np.meshgrid(np.arange(Z, dtype=np.float32), np.arange(Y, dtype=np.float32),
                          np.arange(X, dtype=np.float32), indexing="ij")
    for b in range(n_bricks):
        rng = np.random.default_rng(seed + b)
        if kind == "sphere_n3":
            r = np.sqrt((x - X / 2) ** 2 + (y - Y / 2) ** 2 + (z - Z / 2) ** 2) / (min(X, Y, Z) / 2)
            v = np.floor(255.0 * np.maximum(0.0, 1.0 - r)) + rng.integers(0, 8, (Z, Y, X))
        elif kind == "rm_like":
            # two-fluid interface with a perturbed mixing layer; noise only inside the layer
            h = Z / 2 + (Z / 16.0) * (np.sin(x * (2 * np.pi * 3 / X) + b) + np.cos(y * (2 * np.pi * 5 / Y) - b)
                                       + 0.5 * np.sin((x + y) * (2 * np.pi * 7 / X)))
            d = (z - h) / 6.0
            mix = np.exp(-d * d)
            v = 128 + 120 * np.tanh(d) + mix * rng.integers(-12, 13, (Z, Y, X))
        elif kind == "random":
            v = rng.integers(0, 256, (Z, Y, X))
        else:
            raise ValueError(kind)
        out[b] = np.clip(v, 0, 255).astype(np.uint8)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bricks", type=int, default=16, help="bricks per GPU")
    ap.add_argument("--dims", type=int, nargs=3, default=[256, 256, 256])
    ap.add_argument("--kind", default="rm_like")
    ap.add_argument("--tolerance", type=int, default=1)      # main.cpp:254
    ap.add_argument("--max-epochs", type=int, default=2)     # main.cpp:253
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import __graft_entry__ as g
    g.build()
    import volumerenderer_amd as vr
    from volumerenderer_amd import _lib

    dims = tuple(args.dims)
    B = args.bricks
    V = dims[0] * dims[1] * dims[2]
    host = make_bricks(args.kind, B, dims, seed=12345 + 1000 * rank)
    vox = torch.from_numpy(host).cuda().reshape(-1)            # inputs resident in HBM before timing
    out = torch.empty_like(vox)
    bs = vr.BrickSet(B, dims, args.tolerance, args.max_epochs)
    L = _lib.lib()
    stream = torch.cuda.current_stream()

    def step():
        bs.build(vox)
        bs.decode(out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enc_ms = dec_ms = 0.0
    for _ in range(args.steps):
        step()
        tm = bs.last_timings()                                  # hipEvent timings on the launch stream
        enc_ms += tm["BUILD"] + tm["COMPRESS"] + tm["PRUNE"] + tm["CONVERT"]
        dec_ms += tm["DECODE"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    total_vox = float(V) * B * world * args.steps
    value = total_vox / dt / 1e6

    # decode kernel roofline: algorithmic bytes = V + ceil(numActiveNodes/4) + (maxTreeDepth+1) per brick
    alg = 0
    tokens = 0
    for b in range(B):
        inf = bs.info(b)
        alg += V + inf["tree_bytes"] + inf["max_tree_depth"] + 1
        tokens += inf["num_active_nodes"]
    dec_avg_s = dec_ms / args.steps / 1e3
    achieved = alg / dec_avg_s / 1e9 if dec_avg_s > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": "k_decode", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "alg_bytes_per_launch": alg, "avg_launch_ms": round(dec_avg_s * 1e3, 4)}

    res = {"metric": "Mvoxels/s kd-tree compress+decode", "value": round(value, 2), "unit": "Mvoxels/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "%d bricks/GPU of %dx%dx%d uint8 (%s), tolerance %d, maxEpochs %d, build+levelCut"
                      % (B, dims[0], dims[1], dims[2], args.kind, args.tolerance, args.max_epochs),
                      "tokens_per_voxel": round(tokens / float(V * B), 3)},
           "encode_ms": round(enc_ms / args.steps, 3), "decode_ms": round(dec_ms / args.steps, 3),
           "roofline": roofline}

    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import oracle as O                         # CPU baseline leg: the oracle as the reference's port
        n_done, t_cpu = 0, 0.0
        small = host[0]
        while t_cpu < args.cpu_seconds and n_done < B:
            c0 = time.perf_counter()
            t = O.OracleTree(host[n_done].copy(), tolerance=args.tolerance, max_epochs=args.max_epochs).build()
            t.levelCut()
            t_cpu += time.perf_counter() - c0
            n_done += 1
        res["cpu_baseline"] = {"value": round(n_done * V / t_cpu / 1e6, 3), "unit": "Mvoxels/s", "cores": 1,
                               "kind": "port", "sample": "%d of the %d bricks, serial build(false)+levelCut, 1 thread"
                               % (n_done, B)}
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
