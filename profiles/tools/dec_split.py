"""decode cost of live vs pruned regions: bench volume, block-constant volume, noisy volume (240 bricks each)"""
import sys, os
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
g.build()
import torch
import volumerenderer_amd as vr
import bench
bd, gd = (256, 256, 128), (2048, 2048, 1920)
vox4 = bench.make_volume_gpu(torch, gd, bd, seed=12345)
nb = 240
sel = torch.arange(360, 360 + nb, device="cuda")
vols = {"bench": vox4[sel].contiguous()}
del vox4
c = torch.randint(0, 256, (nb, bd[2] // 16, 1, bd[1] // 16, 1, bd[0] // 16, 1), device="cuda", dtype=torch.uint8)
vols["const16"] = c.expand(nb, bd[2] // 16, 16, bd[1] // 16, 16, bd[0] // 16, 16).reshape(nb, bd[2], bd[1], bd[0]).contiguous()
c = torch.randint(0, 256, (nb, bd[2] // 64, 1, bd[1] // 64, 1, bd[0] // 64, 1), device="cuda", dtype=torch.uint8)
vols["const64"] = c.expand(nb, bd[2] // 64, 64, bd[1] // 64, 64, bd[0] // 64, 64).reshape(nb, bd[2], bd[1], bd[0]).contiguous()
hv = torch.zeros((nb, bd[2], bd[1], bd[0]), device="cuda", dtype=torch.uint8); hv[:, bd[2] // 2:] = 200
vols["halves"] = hv
vols["noise"] = torch.randint(0, 256, (nb, bd[2], bd[1], bd[0]), device="cuda", dtype=torch.uint8)
sm = vols["bench"].float()
vols["smooth+-3"] = (sm + torch.randint(-3, 4, sm.shape, device="cuda")).clamp(0, 255).to(torch.uint8)
del sm
V = bd[0] * bd[1] * bd[2]
for name, v4 in vols.items():
    vox = v4.reshape(-1)
    out = torch.empty_like(vox)
    bs = vr.BrickSet(nb, bd, 1, 2)
    bs.build(vox); torch.cuda.synchronize()
    tok = sum(bs.info(b)["num_active_nodes"] for b in range(nb)) / (nb * V)
    ms = []
    for i in range(5):
        bs.decode(out); torch.cuda.synchronize(); ms.append(bs.last_timings()["DECODE"])
    print("%-10s tokens/voxel %.3f decode %.3f ms -> %.1f ps/voxel, x4 = %.2f ms per 960 bricks" % (name, tok, min(ms), min(ms) * 1e9 / (nb * V), min(ms) * 4), flush=True)
    del bs, out
