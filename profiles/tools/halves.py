"""one full set vs two half sets on two streams (single build+decode wall time)"""
import sys, time
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
g.build()
import torch
import volumerenderer_amd as vr
import bench
bd, gd = (256, 256, 128), (2048, 2048, 1920)
vox4 = bench.make_volume_gpu(torch, gd, bd, seed=12345)
B = vox4.shape[0]; V = bd[0] * bd[1] * bd[2]
vox = vox4.reshape(-1); out = torch.empty_like(vox)
def timeit(f, n=4):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
full = vr.BrickSet(B, bd, 1, 2)
def run_full():
    full.build(vox); full.decode(out)
print("one set: %.2f ms" % timeit(run_full), flush=True)
del full
for P in (2, 3, 4):
    cuts = [B * p // P for p in range(P + 1)]
    # interleave bricks so that each part gets a similar mix: parts take every P-th brick
    sets = [vr.BrickSet(cuts[p + 1] - cuts[p], bd, 1, 2) for p in range(P)]
    st = [torch.cuda.Stream() for _ in range(P)]
    def run_parts():
        for p in range(P):
            sets[p].build(vox[cuts[p] * V:cuts[p + 1] * V], stream=st[p])
            sets[p].decode(out[cuts[p] * V:cuts[p + 1] * V], stream=st[p])
        for s_ in st: torch.cuda.current_stream().wait_stream(s_)
    print("%d part sets on %d streams: %.2f ms" % (P, P, timeit(run_parts)), flush=True)
    del sets
