"""one rank's share of the strong-scaling bench at N ranks (a y-slab of the brick grid), pipelined over 3 sets: ms per step"""
import sys, time
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
g.build()
import torch, bench
import volumerenderer_amd as vr
from volumerenderer_amd import distributed as D
bd, gd, grid = (256, 256, 128), (2048, 2048, 1920), (8, 8, 15)
vox4 = bench.make_volume_gpu(torch, gd, bd, seed=12345)
for world, NS, LLS in ((1, 3, 1), (2, 4, 1), (4, 4, 2), (8, 4, 2)):
    worst = 0.0
    for rank in ((0,) if world == 1 else (1, world - 2)):
        ids, _ = D.shard_bricks_by_slab(grid, rank, world, axis=1)
        v = vox4[torch.tensor(ids, device="cuda")].contiguous().reshape(-1)
        sets = [vr.BrickSet(len(ids), bd, 1, 2) for _ in range(NS)]
        outs = [torch.empty_like(v) for _ in range(NS)]
        streams = [torch.cuda.Stream() for _ in range(NS)]
        for s_ in sets:
            s_.set_concurrency(LLS); s_.build(v); s_.decode(outs[0])
        torch.cuda.synchronize()
        def run(n):
            for k in range(n):
                i = k % NS
                sets[i].build(v, stream=streams[i]); sets[i].decode(outs[i], stream=streams[i])
            torch.cuda.synchronize()
        run(3)
        t0 = time.perf_counter(); run(12); dt = (time.perf_counter() - t0) / 12 * 1e3
        worst = max(worst, dt)
        print("world %d rank %d NS %d LLS %d: %d bricks, %.2f ms per step" % (world, rank, NS, LLS, len(ids), dt), flush=True)
        del sets, outs, v
        torch.cuda.empty_cache()
    print("== world %d: slowest sampled rank %.2f ms -> %.0f Mvoxels/s, efficiency vs world 1 printed by hand" % (world, worst, 8053.06368 / worst * 1e3), flush=True)
