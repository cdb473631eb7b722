#!/usr/bin/env python3
"""bench.py -- kd-tree compress+decode throughput (Mvoxels/s) of the HIP hot path, with
the decode kernel's HBM roofline, the 1080p ray-cast rate and the CPU oracle beside it.

  python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric): one 2048x2048x1920 uint8 volume = the reference's 8x8x15
grid of 256x256x128 bricks (main.cpp:78-79), one kd-tree per brick, all 960 trees built and
decoded by one batched launch sequence.  A "step" = build() + levelCut() of the whole
volume (+ one 1080p ray-cast frame of the decoded volume, timed separately).
N > 1: one rank per GPU over RCCL; the codec shards with no data-path collective.  Default = BASELINE config 4
(--scaling strong): ONE volume, its 960 bricks dealt to the ranks by slabs of the brick grid, value = the volume's
voxels / the slowest rank's time; then every rank ray-marches its slab and the frame is composited sort-last over RCCL
(vr_compositor_composite) and compared with the one-GPU frame.  --scaling weak: one volume (timestep) per rank.  Launched either by
torch.distributed.run (RANK / WORLD_SIZE in the environment) or plainly as `python bench.py
--gpus N`: the parent then starts the N ranks itself, before it touches the GPU, and relays
rank 0's JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
PMC_TRAFFIC = "profiles/r03_pmc_hbm_traffic.json"   # separate rocprofv3 --pmc passes of this same command (profiles/refresh_profiles.sh)


def make_bricks(kind, n_bricks, dims, seed=12345):
    """Small host-side synthetic bricks (tests, cpu baseline).  dims = (X, Y, Z); [B][Z][Y][X]."""
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


def make_volume_gpu(torch, gdims, bdims, seed, kind="rm_volume"):
    """Richtmyer-Meshkov-like two-fluid volume generated on the GPU, returned brick by brick
    ([B][Z][Y][X] uint8, brick b at grid (i,j,k) = fillVolumeBrickMap order, main.cpp:599-619).

    heavy fluid below a perturbed interface, light above, a mixing layer ~ Z/6 thick with
    multi-scale perturbations and 2-bit sensor noise inside the layer; pure-fluid regions
    are constant."""
    GX, GY, GZ = gdims
    X, Y, Z = bdims
    I, J, K = GX // X, GY // Y, GZ // Z
    dev = "cuda"
    ph = [float(v) for v in np.random.default_rng(seed).random(12) * (2 * math.pi)]   # host RNG: phases only
    xs = torch.arange(GX, device=dev, dtype=torch.float32)
    ys = torch.arange(GY, device=dev, dtype=torch.float32)
    # interface height h(x, y): bubbles and spikes at three scales
    hx = (torch.sin(xs * (2 * math.pi * 3 / GX) + ph[0]) + 0.5 * torch.sin(xs * (2 * math.pi * 11 / GX) + ph[1])
          + 0.25 * torch.sin(xs * (2 * math.pi * 37 / GX) + ph[2]))
    hy = (torch.cos(ys * (2 * math.pi * 4 / GY) + ph[3]) + 0.5 * torch.sin(ys * (2 * math.pi * 13 / GY) + ph[4])
          + 0.25 * torch.cos(ys * (2 * math.pi * 29 / GY) + ph[5]))
    h = GZ * 0.5 + (GZ / 14.0) * (hx[None, :] + hy[:, None] + 0.6 * torch.sin((xs[None, :] + ys[:, None]) * (2 * math.pi * 7 / GX) + ph[6]))
    thick = GZ / 24.0
    vol = torch.empty((GZ, GY, GX), dtype=torch.uint8, device=dev)
    xi = torch.arange(GX, device=dev, dtype=torch.int64)[None, None, :]
    yi = torch.arange(GY, device=dev, dtype=torch.int64)[None, :, None]
    xx, yy = xs[None, None, :], ys[None, :, None]
    CH = 32                                   # z-chunk: few, large kernels (a counter-collecting
    for z0 in range(0, GZ, CH):               # profiler dies on tens of thousands of tiny dispatches)
        z1 = min(GZ, z0 + CH)
        zz = torch.arange(z0, z1, device=dev, dtype=torch.float32)[:, None, None]
        d = (zz - h[None, :, :]) / thick
        if kind != "rm_volume":
            raise ValueError(kind)
        mix = torch.exp(-d * d)
        turb = (torch.sin(xx * 0.37 + zz * 0.21 + ph[7]) * torch.cos(yy * 0.29 - zz * 0.17 + ph[8])
                + 0.5 * torch.sin(xx * 0.83 + yy * 0.71 + zz * 0.59 + ph[9]))
        # 2-bit noise from an integer hash of the global voxel coordinate (no device RNG: reproducible
        # on every device)
        zi = torch.arange(z0, z1, device=dev, dtype=torch.int64)[:, None, None]
        hsh = (xi * 73856093) ^ (yi * 19349663) ^ (zi * 83492791) ^ (seed * 2654435761)
        hsh = (hsh ^ (hsh >> 13)) * 1274126177
        noise = ((hsh >> 16) & 3).float()
        v = 128.0 + 120.0 * torch.tanh(d + 0.8 * mix * turb) + mix * noise
        vol[z0:z1] = v.clamp_(0, 255).to(torch.uint8)
        del d, mix, turb, hsh, noise, v
    # global volume -> contiguous bricks (inverse of VolumeReader::LoadBricksToTexture placement)
    import volumerenderer_amd as vr
    ijk = np.array([[b % I, (b // I) % J, b // (I * J)] for b in range(I * J * K)], np.int64)
    out = vr.disassemble_bricks(vol.reshape(-1), bdims, ijk, (I, J, K))
    del vol
    return out.reshape(I * J * K, Z, Y, X)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run the same command line under torch.distributed.run
    (one rank per GPU, rendezvous on 127.0.0.1) as a child process and pass its output and exit code on."""
    import socket
    import subprocess
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    """VRHIP_BENCH_DRYRUN=1: the launch / rendezvous / reduction plumbing of an N-rank run with no GPU work
    (CPU test of `--gpus N`; gloo).  Prints the contract line with value null."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # the strong-scaling partition: every brick of the volume on exactly one rank
    from volumerenderer_amd import distributed as D
    gdims, bdims = tuple(args.volume), tuple(args.dims)
    grid = tuple(gdims[a] // bdims[a] for a in range(3))
    ids, slab = D.shard_bricks_by_slab(grid, rank, world, axis=1)
    per_rank = [ids]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, ids)
    if rank == 0:
        flat = sorted(sum(per_rank, []))
        print(json.dumps({"metric": "Mvoxels/s kd-tree compress+decode", "value": None, "unit": "Mvoxels/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                          "scaling": args.scaling if world > 1 else "weak",
                          "bricks_per_rank": [len(x) for x in per_rank],
                          "partition_ok": flat == list(range(grid[0] * grid[1] * grid[2])),
                          "max_rank_seconds": round(dt, 4)}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def usable_cores():
    """Host cores this process may really use: the affinity mask, cut down to the cgroup's CPU quota if it has one
    (a GPU box hands each job a share of a large host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(math.ceil(float(txt[0]) / float(txt[1])))))
            else:
                q = float(txt[0])
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(math.ceil(q / per))))
            break
        except Exception:
            continue
    return n


def cpu_baseline(vox4, B, V, tolerance, max_epochs, seconds):
    """The CPU oracle (the reference's algorithm restated, kind "port") timed on this box's host cores on a bounded
    sample of the same bricks: one thread (the reference's effective parallelism: COMPRESS / CONVERT / levelCut are
    serial, R.cpp:242,293,631,726) and all usable cores with brick-level parallelism (ctypes releases the GIL)."""
    import threading
    from oracle import oracle as O
    O.lib()
    mid = B // 2                                           # bricks around the interface: the representative ones
    cand = [(mid + (q + 1) // 2 * (1 if q % 2 else -1)) % B for q in range(B)]

    def one(hb):
        t = O.OracleTree(hb, tolerance=tolerance, max_epochs=max_epochs).build()
        t.levelCut()

    n1, t1 = 0, 0.0
    while t1 < seconds * 0.4 and n1 < B:
        hb = vox4[cand[n1]].cpu().numpy()
        c0 = time.perf_counter()
        one(hb)
        t1 += time.perf_counter() - c0
        n1 += 1
    # a GPU box gives one job a CPU share of its host (16 cores per GPU on this pool) without necessarily showing a
    # quota: the affinity mask says 256.  Measured there: 256 threads reach 17.8x one thread, so more than 32 threads
    # only add memory (one ~0.3 GB oracle tree each) and tail latency; VRHIP_BENCH_CPU_THREADS overrides
    cores = min(usable_cores(), int(os.environ.get("VRHIP_BENCH_CPU_THREADS", "32")))
    # all-cores leg: a fixed wall-clock budget; every thread takes the next brick until the deadline has passed
    # (bricks in flight at the deadline are finished and counted)
    budget = max(2.0, seconds * 0.6)
    per_brick = t1 / max(1, n1)
    want = min(B, max(cores, int(cores * budget / max(per_brick, 1e-3)) + cores))
    host = [vox4[cand[q % B]].cpu().numpy() for q in range(want)]
    nxt, done = [0], [0]
    lock = threading.Lock()
    c0 = time.perf_counter()

    def worker():
        while True:
            with lock:
                q = nxt[0]
                if q >= len(host) or time.perf_counter() - c0 > budget:
                    return
                nxt[0] += 1
            one(host[q])
            with lock:
                done[0] += 1

    th = [threading.Thread(target=worker) for _ in range(cores)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    tall = time.perf_counter() - c0
    host_n = done[0]
    return {"value": round(n1 * V / t1 / 1e6, 3), "unit": "Mvoxels/s", "cores": 1, "kind": "port",
            "sample": "%d bricks from the middle of the volume, serial build(false)+levelCut of the CPU oracle, "
                      "1 thread, %.1f s" % (n1, t1),
            "host_cores": os.cpu_count(), "usable_cores": cores,
            "all_cores": {"value": round(host_n * V / tall / 1e6, 3), "unit": "Mvoxels/s", "threads_used": cores,
                          "sample": "%d bricks, one oracle tree per thread at a time, %.1f s wall" % (host_n, tall)}}


def config5_leg(torch, vr, vox4, gdims, bdims, grid, args, timesteps=4, frames_per_stage=4):
    """BASELINE config 5 at full size: `timesteps` volumes (different seeds) stream from pinned host memory through a
    MidRangeTree set (both 2-bit streams = the 4-bit packing): upload of t+1 beside the build / decodes / frames of t;
    every timestep is decoded at cuts D-6 (one value per 4x4x4 block), D (no grown branches) and maxTreeDepth (levelCut)
    with the half-range stream decoded at the first cut, 1080p frames on a 1-degree-per-frame orbit after each stage.
    Also times the MidRangeTree build + decode alone (the 4-bit path's throughput)."""
    from volumerenderer_amd.pipeline import TimestepStreamer
    B = vox4.shape[0]
    V = bdims[0] * bdims[1] * bdims[2]
    st = TimestepStreamer(B, bdims, args.tolerance, args.max_epochs, variant=2)
    D = st.bs.build(vox4.reshape(-1)).info(0)["orig_tree_depth"]
    out = st.out[0]
    torch.cuda.synchronize()
    ms = []
    for _ in range(3):                       # MidRangeTree build + levelCut, serial (4-bit stream pair)
        st.bs.build(vox4.reshape(-1)); st.bs.decode(out)
        torch.cuda.synchronize()
        tm = st.bs.last_timings()
        ms.append(tm["BUILD"] + tm["COMPRESS"] + tm["PRUNE"] + tm["CONVERT"] + tm["DECODE"])
    mr_ms = min(ms)
    host = [torch.empty(B * V, dtype=torch.uint8, pin_memory=True) for _ in range(timesteps)]
    host[0].copy_(vox4.reshape(-1))
    for t in range(1, timesteps):
        host[t].copy_(make_volume_gpu(torch, gdims, bdims, seed=12345 + 77 * t).reshape(-1))
    torch.cuda.synchronize()
    ijk_all = np.array([[b % grid[0], (b // grid[0]) % grid[1], b // (grid[0] * grid[1])] for b in range(B)], np.int64)
    whole = torch.empty(B * V, dtype=torch.uint8, device="cuda")
    rng = torch.empty(B * V, dtype=torch.uint8, device="cuda")
    img = torch.empty((1080, 1920, 4), dtype=torch.float32, device="cuda")
    cam, P = vr.default_camera(), vr.default_params(1920, 1080, (256, 256, 128))
    cuts = [D - 6, D, D + 7]
    theta = [0.0]
    nframes = [0]

    def on_stage(t, k, cut, vol, stream):
        if k == 0:
            st.bs.decode_range(rng, cut_depth=cut, stream=stream)       # [mid - range, mid + range] preview bounds
        vr.assemble_bricks(vol, bdims, ijk_all, grid, out=whole, stream=stream)
        for _ in range(frames_per_stage):
            th = math.radians(theta[0]); theta[0] += 1.0
            cam.pos[:] = (0.75 * math.sin(th), 0.0, -0.75 * math.cos(th))
            cam.front[:] = (-math.sin(th), 0.0, math.cos(th))
            vr.raycast(whole, gdims, cam, P, img, stream=stream)
            nframes[0] += 1

    st.run_progressive(host[:1], cuts, on_stage)          # warm-up (first touch of every buffer)
    nframes[0] = 0
    torch.cuda.synchronize()
    w0 = time.perf_counter()
    ev = st.run_progressive(host, cuts, on_stage)
    wall = time.perf_counter() - w0
    first = [e["uploaded"].elapsed_time(e["stages"][0]) for e in ev]
    refine = [e["stages"][0].elapsed_time(e["stages"][-1]) for e in ev]
    build = [e["uploaded"].elapsed_time(e["built"]) for e in ev]
    return {"timesteps": timesteps, "bricks_per_timestep": B, "variant": "MidRangeTree (2 x 2-bit streams = 4-bit packing)",
            "cuts": cuts, "frames": nframes[0], "wall_s": round(wall, 3), "fps_1080p_incl_build_and_decodes": round(nframes[0] / wall, 1),
            "timesteps_per_s": round(timesteps / wall, 2),
            "upload_to_first_frame_ms": round(sum(first) / len(first), 2), "first_to_full_refine_ms": round(sum(refine) / len(refine), 2),
            "build_ms_in_stream": round(sum(build) / len(build), 2),
            "midrange_build_plus_decode_ms": round(mr_ms, 2), "midrange_Mvoxels_per_s": round(B * V / mr_ms / 1e3, 1)}


def general_extents_leg(torch, vr):
    """The reference program's own volume (main.cpp:242-251): 384 bricks of 256x256x64 assembled into 2048x2048x768 and ONE
    31-level VolumeKdtree over it (table-driven geometry, 64-bit token offsets): build + levelCut, milliseconds."""
    gdims, bdims = (2048, 2048, 768), (256, 256, 64)
    grid = tuple(gdims[k] // bdims[k] for k in range(3))
    vox4 = make_volume_gpu(torch, gdims, bdims, seed=777)
    B = vox4.shape[0]
    ijk = np.array([[b % grid[0], (b // grid[0]) % grid[1], b // (grid[0] * grid[1])] for b in range(B)], np.int64)
    vol = vr.assemble_bricks(vox4.reshape(-1), bdims, ijk, grid)
    del vox4
    t = vr.BrickSet(1, gdims, 1, 2)
    out = torch.empty_like(vol)
    best = None
    for _ in range(2):
        t.build(vol); t.decode(out)
        torch.cuda.synchronize()
        tm = t.last_timings()
        ms = (tm["BUILD"] + tm["COMPRESS"] + tm["PRUNE"] + tm["CONVERT"], tm["DECODE"])
        best = ms if best is None or sum(ms) < sum(best) else best
    inf = t.info(0)
    return {"volume": list(gdims), "orig_tree_depth": inf["orig_tree_depth"], "num_active_nodes": inf["num_active_nodes"],
            "build_ms": round(best[0], 2), "levelcut_ms": round(best[1], 2),
            "Mvoxels_per_s": round(gdims[0] * gdims[1] * gdims[2] / (best[0] + best[1]) / 1e3, 1)}


def disk_stage_leg(torch, vr, vox4, gdims, bdims, grid, args, timesteps=2):
    """config 5's disk stage at full size: `timesteps` x 960 raw brick files (VolumeReader<T>::LoadVolumeFromBinaryFile's
    format: X*Y*Z bytes each) on local disk -> pinned host memory (reader thread) -> device -> build -> levelCut -> one
    1080p frame; the time from the first file read to the first frame, and the whole run."""
    import shutil
    import tempfile
    from volumerenderer_amd.pipeline import BrickFileSource, TimestepStreamer
    B = vox4.shape[0]
    d = tempfile.mkdtemp(prefix="vrhip_bricks_")
    try:
        host = vox4.cpu().numpy()
        for t in range(timesteps):
            for b in range(B):
                host[(b + 7 * t) % B].tofile(os.path.join(d, "d_%d_%d" % (270 + t, b)))      # (timestep t: the bricks rotated)
        del host
        src = BrickFileSource(lambda b, t: os.path.join(d, "d_%d_%d" % (t, b)), B, bdims, [270 + t for t in range(timesteps)])
        st = TimestepStreamer(B, bdims, args.tolerance, args.max_epochs)
        ijk_all = np.array([[b % grid[0], (b // grid[0]) % grid[1], b // (grid[0] * grid[1])] for b in range(B)], np.int64)
        whole = torch.empty(B * bdims[0] * bdims[1] * bdims[2], dtype=torch.uint8, device="cuda")
        img = torch.empty((1080, 1920, 4), dtype=torch.float32, device="cuda")
        cam, P = vr.default_camera(), vr.default_params(1920, 1080, (256, 256, 128))
        first = []

        def on_decoded(t, vol, stream):
            vr.assemble_bricks(vol, bdims, ijk_all, grid, out=whole, stream=stream)
            vr.raycast(whole, gdims, cam, P, img, stream=stream)
            if not first:
                stream.synchronize()
                first.append(time.perf_counter())

        torch.cuda.synchronize()
        w0 = time.perf_counter()
        st.run(src, on_decoded=on_decoded)
        wall = time.perf_counter() - w0
        return {"timesteps": timesteps, "brick_files": timesteps * B, "bytes_per_timestep": int(B * bdims[0] * bdims[1] * bdims[2]),
                "disk_to_first_frame_ms": round((first[0] - w0) * 1e3, 1), "wall_s": round(wall, 3)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--volume", type=int, nargs=3, default=[2048, 2048, 1920], help="global volume per GPU")
    ap.add_argument("--dims", type=int, nargs=3, default=[256, 256, 128], help="brick dims (main.cpp:78)")
    ap.add_argument("--bricks", type=int, default=0, help="use only the first N bricks (0 = whole volume)")
    ap.add_argument("--kind", default="rm_volume")
    ap.add_argument("--tolerance", type=int, default=1)      # main.cpp:254
    ap.add_argument("--max-epochs", type=int, default=2)     # main.cpp:253
    ap.add_argument("--cpu-seconds", type=float, default=24.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-render", action="store_true")
    ap.add_argument("--no-stream", action="store_true", help="skip the MidRangeTree / config-5 streaming leg")
    ap.add_argument("--pipeline", type=int, default=0, choices=[0, 1, 2, 3, 4, 6, 8],
                    help="bricksets in flight, each on its own stream with its own output volume.  0 (default) = 3 (~59 GB "
                         "each of the 288 GB for a whole volume; 4 where a rank holds a fraction of the volume): build + levelCut of steps k, k+1, k+2 run beside each other (the streaming "
                         "use: the next timesteps compress while this one decodes; falls back to 2 if the third set "
                         "does not fit); 1: strictly serial")
    ap.add_argument("--level-loop-streams", type=int, default=4, choices=[1, 2, 3, 4],
                    help="vr_brickset_set_concurrency for the strictly serial pass: brick ranges whose level loops run side "
                         "by side on internal streams (4 = the most the library takes and the fastest single build; the "
                         "library's own default is 2, which leaves a hardware queue to a caller's copy stream; 1 for clean "
                         "per-kernel profiles).  The pipelined sets use 1 (2 for a quarter volume or less)")
    ap.add_argument("--pipeline-level-loop-streams", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="vr_brickset_set_concurrency of the pipelined sets (0 = 1 for half a volume or more, 2 below)")
    ap.add_argument("--no-extra-timing", action="store_true", help="skip the second timed pass (value_no_compact)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong (default, BASELINE config 4) = ONE volume, its bricks dealt to the ranks by slabs of the "
                         "brick grid along y, value = the volume's voxels / the slowest rank's time; weak = one volume per rank")
    ap.add_argument("--no-composite", action="store_true",
                    help="N > 1: skip the sort-last composited 1080p frames over RCCL (they run by default, after the timed "
                         "region, under the process group's timeout: a collective that fails ends the run non-zero)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks as children (torch.distributed.run) BEFORE this
        # process initialises HIP -- a process that holds the GPU must neither exec nor fork workers
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        sys.exit(2)
    # VRHIP_BENCH_REHEARSAL=1: rehearse the N > 1 code path on a one-GPU box (every rank on device 0, gloo)
    rehearsal = os.environ.get("VRHIP_BENCH_REHEARSAL") == "1"
    if os.environ.get("VRHIP_BENCH_DRYRUN") == "1":
        return dry_run(args, rank, world)

    import __graft_entry__ as g
    g.build()                      # before anything touches the GPU (a rebuild execs hipcc)
    import torch
    import torch.distributed as dist
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        import datetime
        # a collective that cannot complete ends the run (non-zero exit) after this long instead of hanging it
        pg_timeout = datetime.timedelta(seconds=int(os.environ.get("VRHIP_BENCH_PG_TIMEOUT", "300")))
        if rehearsal:
            dist.init_process_group("gloo", timeout=pg_timeout)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=pg_timeout)

    import volumerenderer_amd as vr

    bdims = tuple(args.dims)
    gdims = tuple(args.volume)
    V = bdims[0] * bdims[1] * bdims[2]
    grid = tuple(gdims[a] // bdims[a] for a in range(3))
    strong = world > 1 and args.scaling == "strong" and args.kind == "rm_volume" and not args.bricks
    my_ids, my_slab = None, None
    B_total = None
    if strong:
        # BASELINE config 4: ONE volume (the same seed on every rank), its bricks dealt to the ranks by slabs of the brick
        # grid along y (8 brick rows: even shares for 2, 4 and 8 ranks; fillVolumeBrickMap order, main.cpp:599-619).  No
        # data-path collective: every rank builds and decodes its own bricks.
        from volumerenderer_amd import distributed as D
        vox_all = make_volume_gpu(torch, gdims, bdims, seed=12345, kind=args.kind)
        B_total = vox_all.shape[0]
        my_ids, my_slab = D.shard_bricks_by_slab(grid, rank, world, axis=1)
        vox4 = vox_all[torch.tensor(my_ids, device="cuda")].contiguous()
        del vox_all
        torch.cuda.empty_cache()
    elif args.kind in ("rm_volume",):
        vox4 = make_volume_gpu(torch, gdims, bdims, seed=12345 + 1000 * rank, kind=args.kind)  # rank = timestep
        if args.bricks:
            # keep the bricks around the interface first (they are the expensive ones)
            order = sorted(range(vox4.shape[0]), key=lambda b: abs(b // (grid[0] * grid[1]) - grid[2] // 2))
            vox4 = vox4[torch.tensor(order[:args.bricks], device="cuda")].contiguous()
    else:
        vox4 = torch.from_numpy(make_bricks(args.kind, args.bricks or 16, bdims, seed=12345 + 1000 * rank)).cuda()
    B = vox4.shape[0]
    vox = vox4.reshape(-1)                                      # inputs resident in HBM before timing
    out = torch.empty_like(vox)
    # pipeline >= 2: several bricksets in flight (see run_steps); the serial per-kernel pass below reuses the
    # first of them, so no further set (65 GB at the full volume) is allocated after the pipelined ones
    # bricksets in flight: 3 for a whole volume per GPU (a fourth does not fit beside them usefully: 36.5 instead of 35.9 ms);
    # 4 for a rank that holds a fraction of the volume (strong scaling), where a build's latency-bound control steps
    # weigh more (one rank's y-slab at N = 2 / 8: 17.6 -> 17.0 / 5.3 -> 5.0 ms per step; 5 or 6 sets: slower again)
    NS = args.pipeline or (3 if B >= 960 else 4)
    # level-loop forks inside a pipelined set: none for a whole or half volume (the other sets fill the gaps), two brick
    # ranges for a quarter or less (one rank's y-slab at N = 4 / 8: 10.6 -> 9.9 / 6.2 -> 5.8 ms per step; N = 2: slower)
    lls_pipe = args.pipeline_level_loop_streams or (2 if B <= 240 else 1)
    sets = []
    torch.cuda.synchronize()
    free_before_sets = torch.cuda.mem_get_info()[0]      # the library allocates with hipMalloc, outside torch's caching allocator
    while True:
        try:
            sets = [vr.BrickSet(B, bdims, args.tolerance, args.max_epochs) for _ in range(NS)]
            for s_ in sets:            # several sets in flight fill each other's gaps: no fork inside a build (vrhip.h)
                s_.set_concurrency(lls_pipe if NS >= 2 else args.level_loop_streams)
            for s_ in sets:            # setup, not a step: allocate and first-touch every set's buffers
                s_.build(vox); s_.decode(out)
            torch.cuda.synchronize()
            break
        except vr.VrError:             # out of device memory: one set less
            if NS == 1:
                raise
            sets = []
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            NS -= 1
    bs = sets[0]
    # device memory a brickset holds once it has built and decoded (encoder arrays, gapped + contiguous stream, decode
    # side-cars, control blocks): measured, not summed from a table
    device_bytes_per_set = (free_before_sets - torch.cuda.mem_get_info()[0]) // NS
    # every set on its own stream, decoding into its own output volume: build + levelCut of step k run beside those of
    # steps k+1 .. k+NS-1 (measured against two build streams + one decode stream: 36.1 instead of 37.6 ms per step)
    streams = [torch.cuda.Stream() for _ in range(NS)]
    outs = [out]
    try:
        for _ in range(NS - 1):
            outs.append(torch.empty_like(out))
    except RuntimeError:               # no room for further output volumes: the sets share one (same bytes every step)
        outs = [out] * NS
    outs = (outs + [out] * NS)[:NS]

    def run_steps(n):
        """n x (build + levelCut) of the whole volume; every launch of every step is inside the caller's timed
        region.  With NS >= 2 bricksets, step k uses set k % NS on stream k % NS: its build and its levelCut run beside
        those of the following steps; a brickset is rebuilt only after its own decode (stream order)."""
        if NS == 1:
            for _ in range(n):
                bs.build(vox)
                bs.decode(out)
            return
        for k in range(n):
            i = k % NS
            sets[i].build(vox, stream=streams[i])
            sets[i].decode(outs[i], stream=streams[i])
        for st_ in streams:                        # the default stream (serial pass below) follows the pipelined steps
            torch.cuda.current_stream().wait_stream(st_)

    run_steps(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device="cpu" if rehearsal else "cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    total_vox = float(V) * (B_total if strong else B * world) * args.steps
    value = total_vox / dt / 1e6
    # the same steps without the contiguous copy of the compressed stream at the end of build() (vr_brickset_set_compaction:
    # a pipeline that only decodes on the device may leave it to the first get_tree / save): reported beside `value`, never as it
    value_nc = None
    if not args.no_extra_timing:
        for s_ in sets:
            s_.set_compaction(False)
        run_steps(min(args.warmup, NS))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        run_steps(args.steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt_nc = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([dt_nc], device="cpu" if rehearsal else "cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_nc = float(tt.item())
        value_nc = total_vox / dt_nc / 1e6
        for s_ in sets:
            s_.set_compaction(True)
    outs = None                        # the extra output volumes are not needed any more
    torch.cuda.empty_cache()

    # per-kernel timing outside the timed region, strictly serial: hipEvents on the launch stream
    # (vr_brickset_last_timings)
    bs.set_concurrency(args.level_loop_streams)      # default 4: the fastest single build() (vr_brickset_set_concurrency)
    enc_ms, dec_ms = [], []
    for i in range(4):
        bs.build(vox)
        bs.decode(out)
        torch.cuda.synchronize()
        tm = bs.last_timings()
        if i == 0:
            continue        # the first serial pass after the pipelined region still pays for its cold caches
        enc_ms.append(tm["BUILD"] + tm["COMPRESS"] + tm["PRUNE"] + tm["CONVERT"])
        dec_ms.append(tm["DECODE"])
        phases = tm
    # decode kernel roofline: algorithmic bytes = V + ceil(numActiveNodes/4) + (maxTreeDepth+1) per brick
    alg = 0
    tokens = 0
    n_const = 0                       # bricks of a single value: encoded in closed form (k_const_finish)
    for b in range(B):
        inf = bs.info(b)
        alg += V + inf["tree_bytes"] + inf["max_tree_depth"] + 1
        tokens += inf["num_active_nodes"]
        n_const += inf["num_active_nodes"] <= 3
    dec_avg_s = sum(dec_ms) / len(dec_ms) / 1e3
    achieved = alg / dec_avg_s / 1e9
    # HBM traffic of the decode launch from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE /
    # --pmc WRITE_SIZE runs of this same command, profiles/): only quoted when the workload is the profiled one
    traffic = None
    try:
        pm = json.load(open(os.path.join(ROOT, PMC_TRAFFIC)))
        if args.kind == "rm_volume" and not args.bricks and gdims == (2048, 2048, 1920) and bdims == (256, 256, 128) \
                and args.tolerance == 1 and args.max_epochs == 2 and not strong:
            traffic = pm["decode_traffic_bytes_per_launch"]
    except Exception:
        pass
    roofline = {"bound": "hbm", "kernel": "k_decode_region", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_source": PMC_TRAFFIC + " (bytes per launch)" if traffic else None,
                "alg_bytes_per_launch": alg, "avg_launch_ms": round(dec_avg_s * 1e3, 4)}

    # the encoder as a whole against the same roof: algorithmic bytes of a build = the voxels read + the stream written
    enc_avg_s = sum(enc_ms) / len(enc_ms) / 1e3
    enc_alg = alg                       # V + tree bytes (+ distanceMap) per brick: the same sum, read and written the other way round
    enc_traffic = None
    try:
        if traffic is not None:
            enc_traffic = pm.get("encode_traffic_bytes_per_build")
    except Exception:
        pass
    roofline_encode = {"bound": "hbm", "kernels": "one build() = pyramid + level loop + prune/emit + index + compaction",
                       "achieved": round(enc_alg / enc_avg_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(enc_alg / enc_avg_s / 1e9 / HBM_PEAK_GBS, 5), "traffic": enc_traffic,
                       "alg_bytes_per_build": enc_alg, "avg_build_ms": round(enc_avg_s * 1e3, 3)}

    res = {"metric": "Mvoxels/s kd-tree compress+decode", "value": round(value, 2), "unit": "Mvoxels/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": ("ONE %dx%dx%d uint8 volume, its %d bricks of %dx%dx%d dealt to %d ranks by slabs of the brick grid "
                                   "along y (%d bricks on rank 0), " % (gdims[0], gdims[1], gdims[2], B_total, bdims[0], bdims[1], bdims[2], world, B)
                                   if strong else
                                   "%dx%dx%d uint8 volume per GPU as %d bricks of %dx%dx%d, " % (gdims[0], gdims[1], gdims[2], B, bdims[0], bdims[1], bdims[2]))
                                  + "%s, tolerance %d, maxEpochs %d, VolumeKdtree build + levelCut" % (args.kind, args.tolerance, args.max_epochs),
                      "pipeline": ("%d bricksets in flight, each on its own stream: build + levelCut of step k run "
                                   "beside those of the following steps" % NS if NS >= 2 else "serial"),
                      "level_loop_streams": {"pipelined": lls_pipe if NS >= 2 else args.level_loop_streams, "serial": args.level_loop_streams},
                      "constant_bricks": int(n_const),
                      "tokens_per_voxel": round(tokens / float(V * B), 3),
                      "compression_ratio": round(float(V * B) / (tokens / 4.0), 2)},
           "device_bytes_per_set": int(device_bytes_per_set), "device_bytes_per_set_over_input": round(device_bytes_per_set / float(V * B), 3),
           "device_bytes_pipeline": int(device_bytes_per_set * NS + vox.numel() * (1 + NS)),
           "serial_ms_per_step": round(sum(enc_ms) / len(enc_ms) + dec_avg_s * 1e3, 3),
           "encode_ms": round(sum(enc_ms) / len(enc_ms), 3), "decode_ms": round(dec_avg_s * 1e3, 3),
           "phases_ms": {k: round(v, 3) for k, v in phases.items()},
           "value_no_compact": round(value_nc, 2) if value_nc else None,
           "backend": (dist.get_backend() if world > 1 else None), "rccl_ranks": (dist.get_world_size() if world > 1 else 1),
           "roofline": roofline, "roofline_encode": roofline_encode}

    if not args.no_render and not args.bricks and rank == 0:
        # 1080p frames of the decoded volume on a camera orbit: raycaster.frag and isosurface.frag, each also with the
        # empty-space skip grid (bit-identical frames, tests/test_gpu_render_pins.py)
        ijk_all = np.array([[b % grid[0], (b // grid[0]) % grid[1], b // (grid[0] * grid[1])] for b in range(B)], np.int64)
        vol = vr.assemble_bricks(out, bdims, ijk_all, grid)
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        sgrid = vr.build_skip_grid(vol, gdims, 8)
        torch.cuda.synchronize()
        res["skip_grid_build_ms"] = round((time.perf_counter() - g0) * 1e3, 2)
        cam = vr.default_camera()
        img = torch.empty((1080, 1920, 4), dtype=torch.float32, device="cuda")
        frames = 36

        def orbit_fps(P):
            fps = 0.0
            for warm in (True, False):
                torch.cuda.synchronize()
                r0 = time.perf_counter()
                for f in range(frames):
                    th = math.radians(f * 10.0)
                    cam.pos[:] = (0.75 * math.sin(th), 0.0, -0.75 * math.cos(th))
                    cam.front[:] = (-math.sin(th), 0.0, math.cos(th))
                    vr.raycast(vol, gdims, cam, P, img)
                torch.cuda.synchronize()
                fps = frames / (time.perf_counter() - r0)
            return round(fps, 1)

        P = vr.default_params(1920, 1080, (256, 256, 128))
        res["raycast_1080p_fps"] = orbit_fps(P)
        res["raycast_1080p_fps_skip_grid"] = orbit_fps(vr.use_skip_grid(vr.default_params(1920, 1080, (256, 256, 128)), sgrid, 8))
        Pi = vr.default_params(1920, 1080, (256, 256, 128), vr.RENDER_ISOSURFACE, 40.0 / 255.0)     # main.cpp:52,334
        res["isosurface_1080p_fps"] = orbit_fps(Pi)
        res["isosurface_1080p_fps_skip_grid"] = orbit_fps(vr.use_skip_grid(
            vr.default_params(1920, 1080, (256, 256, 128), vr.RENDER_ISOSURFACE, 40.0 / 255.0), sgrid, 8))
        del vol, sgrid

    if world > 1 and not args.no_composite and not args.bricks and args.kind == "rm_volume":
        # sort-last frames over RCCL (the path's one real exchange): rank r ray-marches ITS slab of the decoded volume
        # (strong: the y-slab of the bricks it decoded, plus one halo plane from each neighbour; weak: z-slab r of its own
        # volume) into a partial (c, tau) image; vr_compositor_composite: direct-send exchange (grouped ncclSend / ncclRecv),
        # per-pixel ordered combine, gather on rank 0.  The leg runs in stages; after each, the ranks agree (all-reduce MIN)
        # that every one of them got through it, so an error that is local to a rank -- out of memory, RCCL not bindable,
        # a bug -- makes all of them skip the rest and is reported in the line (`composite_error`) instead of leaving the
        # others in a collective.  A collective that cannot complete is the process group's timeout's business: the run
        # then ends non-zero (no retry, no re-exec).
        from volumerenderer_amd import distributed as D
        ax = 1 if strong else 2
        stage_err = [None]

        def agree(tag, fn):
            """fn() on every rank, then MIN over the ranks of "it worked": what fn returned, or None if anybody failed"""
            out_, ok_ = None, 1
            if stage_err[0] is None:
                try:
                    out_ = fn()
                except Exception as ex:          # noqa: BLE001 -- reported, never swallowed
                    ok_ = 0
                    stage_err[0] = "%s: %r" % (tag, ex)
            else:
                ok_ = 0
            flag = torch.tensor([ok_], device="cpu" if rehearsal else "cuda", dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0 and stage_err[0] is None:
                stage_err[0] = "%s: failed on another rank" % tag
            return out_ if stage_err[0] is None else None

        def stage_slab():
            if strong:
                lo_b, hi_b = my_slab
                ylo, yhi = lo_b * bdims[1], hi_b * bdims[1]
                ijk = np.array([[b % grid[0], (b // grid[0]) % grid[1] - lo_b, b // (grid[0] * grid[1])] for b in my_ids], np.int64)
                slab_ = vr.assemble_bricks(out, bdims, ijk, (grid[0], hi_b - lo_b, grid[2])).reshape(gdims[2], yhi - ylo, gdims[0])
                return slab_, ylo, yhi
            vol = vr.assemble_bricks(out, bdims, np.array([[b % grid[0], (b // grid[0]) % grid[1], b // (grid[0] * grid[1])]
                                                            for b in range(B)], np.int64), grid)
            zlo, zhi = D.shard_range(gdims[2], rank, world)
            a0_, a1_ = max(0, zlo - 1), min(gdims[2], zhi + 1)
            return vol.reshape(gdims[2], gdims[1], gdims[0])[a0_:a1_].contiguous(), zlo, zhi

        got = agree("assemble the slab", stage_slab)
        slab = sub_dims = None
        a0 = lo_f = hi_f = ext = 0
        if got is not None:
            slab, lo_f, hi_f = got
            if strong:
                def stage_halo():
                    # halo planes: my first plane to the rank below, my last one to the rank above
                    lo_halo = torch.empty((gdims[2], 1, gdims[0]), dtype=torch.uint8, device="cuda") if rank > 0 else None
                    hi_halo = torch.empty((gdims[2], 1, gdims[0]), dtype=torch.uint8, device="cuda") if rank < world - 1 else None
                    first, last = slab[:, :1].contiguous(), slab[:, -1:].contiguous()
                    ops = []
                    if rank > 0:
                        ops += [dist.P2POp(dist.isend, first, rank - 1), dist.P2POp(dist.irecv, lo_halo, rank - 1)]
                    if rank < world - 1:
                        ops += [dist.P2POp(dist.isend, last, rank + 1), dist.P2POp(dist.irecv, hi_halo, rank + 1)]
                    for req in dist.batch_isend_irecv(ops):
                        req.wait()
                    return torch.cat([t_ for t_ in (lo_halo, slab, hi_halo) if t_ is not None], dim=1).contiguous()
                slab = agree("halo exchange", stage_halo)
                a0 = lo_f - (1 if rank > 0 else 0)
                ext = gdims[1]
                if slab is not None:
                    sub_dims = (gdims[0], slab.shape[1], gdims[2])
            else:
                a0 = max(0, lo_f - 1)
                ext = gdims[2]
                sub_dims = (gdims[0], gdims[1], slab.shape[0])
        cam = vr.default_camera()
        P = vr.default_params(1920, 1080, (256, 256, 128), vr.RENDER_PARTIAL)
        part = frame = None
        if stage_err[0] is None:
            bmin, bmax, org = [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0, 0, 0]
            bmin[ax] = lo_f / ext
            bmax[ax] = hi_f / ext if rank < world - 1 else 2.0
            org[ax] = a0
            P.box_min[:] = tuple(bmin); P.box_max[:] = tuple(bmax)
            P.global_dims[:] = gdims
            P.vol_origin[:] = tuple(org)

        def stage_bind():
            # every rank checks that it can bind RCCL and allocate its images BEFORE any of them enters the communicator's
            # collective initialisation
            from volumerenderer_amd import _lib
            import ctypes as C_
            probe = (C_.c_uint8 * 128)()
            _lib.check(_lib.lib().vr_rccl_unique_id(probe), "vr_rccl_unique_id")
            part_ = torch.empty((1080, 1920, 4), dtype=torch.float32, device="cuda")
            frame_ = torch.empty((1080, 1920, 4), dtype=torch.float32, device="cuda") if rank == 0 else None
            return part_, frame_

        got = agree("bind RCCL", stage_bind)
        if got is not None:
            part, frame = got
        frames = 36

        def stage_frames():
            cfps_ = 0.0
            for warm in (True, False):
                torch.cuda.synchronize()
                dist.barrier()
                r0 = time.perf_counter()
                for f in range(frames):
                    th = math.radians(f * 10.0)
                    cam.pos[:] = (0.75 * math.sin(th), 0.0, -0.75 * math.cos(th))
                    cam.front[:] = (-math.sin(th), 0.0, math.cos(th))
                    vr.raycast(slab.reshape(-1), sub_dims, cam, P, part)
                    D.composite_sort_last(part, cam, P, axis=ax, out=frame)
                torch.cuda.synchronize()
                dist.barrier()
                cfps_ = frames / (time.perf_counter() - r0)
            return cfps_

        cfps = agree("composited frames", stage_frames)
        if stage_err[0] is not None:
            res["composite_error"] = stage_err[0][:300]
        else:
            res["composited_1080p_fps"] = round(cfps, 1)
            res["composite"] = {"axis": "xyz"[ax], "slab_voxels_rank0": [int(v) for v in sub_dims], "exchange": "vr_compositor_composite (C ABI): grouped "
                                "ncclSend/ncclRecv direct send, k_composite_slabs, gather on rank 0"}
        if strong and rank == 0 and stage_err[0] is None:
            # the same frame from one GPU: rank 0 decodes the whole volume once more (outside every timed region) and
            # marches it without the early exit the slabs cannot honour (raycaster.frag:76)
            del slab
            sets, bs = None, None
            torch.cuda.empty_cache()
            try:
                vox_all = make_volume_gpu(torch, gdims, bdims, seed=12345, kind=args.kind)
                full_set = vr.BrickSet(B_total, bdims, args.tolerance, args.max_epochs)
                full_set.set_compaction(False)
                dec_all = full_set.build(vox_all.reshape(-1)).decode()
                vol = vr.assemble_bricks(dec_all, bdims, np.array([[b % grid[0], (b // grid[0]) % grid[1], b // (grid[0] * grid[1])]
                                                                    for b in range(B_total)], np.int64), grid)
                Pf = vr.default_params(1920, 1080, (256, 256, 128))
                Pf.no_early_exit = 1
                ref = vr.raycast(vol, gdims, cam, Pf)               # the last camera of the orbit: `frame` holds that composite
                res["composite_max_abs_diff"] = round(float((ref - frame).abs().max().item()), 6)
                del vox_all, full_set, dec_all, vol, ref
            except Exception as ex:
                res["composite_max_abs_diff"] = None
                res["composite_reference_error"] = repr(ex)[:200]

    if rank == 0 and world == 1 and not args.no_stream and not args.bricks and args.kind == "rm_volume":
        del sets, bs
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        try:
            res["config5"] = config5_leg(torch, vr, vox4, gdims, bdims, grid, args)
        except Exception as ex:       # an extra: never lets the headline line go missing
            res["config5"] = {"error": repr(ex)[:300]}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        try:
            res["config5_disk_stage"] = disk_stage_leg(torch, vr, vox4, gdims, bdims, grid, args)
        except Exception as ex:
            res["config5_disk_stage"] = {"error": repr(ex)[:300]}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        try:
            res["general_extents"] = general_extents_leg(torch, vr)
        except Exception as ex:
            res["general_extents"] = {"error": repr(ex)[:300]}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_cpu:
        res["cpu_baseline"] = cpu_baseline(vox4, B, V, args.tolerance, args.max_epochs, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
