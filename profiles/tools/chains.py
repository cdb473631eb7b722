"""pipelined throughput with NS sets x P parts, every (set, part) on its own stream"""
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
for NS, P in ((3, 1), (3, 2), (2, 2), (3, 4), (2, 4)):
    cuts = [B * p // P for p in range(P + 1)]
    sets = [[vr.BrickSet(cuts[p + 1] - cuts[p], bd, 1, 2) for p in range(P)] for _ in range(NS)]
    st = [[torch.cuda.Stream() for p in range(P)] for _ in range(NS)]
    def run(n):
        for k in range(n):
            i = k % NS
            for p in range(P):
                sets[i][p].build(vox[cuts[p] * V:cuts[p + 1] * V], stream=st[i][p])
                sets[i][p].decode(out[cuts[p] * V:cuts[p + 1] * V], stream=st[i][p])
        for row in st:
            for s_ in row: torch.cuda.current_stream().wait_stream(s_)
    run(NS); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(6); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
    print("sets %d x parts %d: %.2f ms/step = %.1f Gvox/s" % (NS, P, dt * 1e3, B * V / dt / 1e9), flush=True)
    del sets
    torch.cuda.synchronize()
