"""encode-only phase timing on the bench volume: python profiles/tools/enc_time.py"""
import sys, os
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
g.build()
import torch
import volumerenderer_amd as vr
import bench
bd, gd = (256, 256, 128), (2048, 2048, 1920)
vox4 = bench.make_volume_gpu(torch, gd, bd, seed=12345)
B = vox4.shape[0]
vox = vox4.reshape(-1)
bs = vr.BrickSet(B, bd, 1, 2)
bs.set_concurrency(int(os.environ.get("LLS", "1")))
acc = {}
for i in range(5):
    bs.build(vox); torch.cuda.synchronize()
    t = bs.last_timings()
    if i >= 2:
        for k, v in t.items(): acc.setdefault(k, []).append(v)
print(os.environ.get("TAG", ""), {k: round(min(v), 3) for k, v in acc.items() if k != "DECODE"}, "sum", round(sum(min(v) for k, v in acc.items() if k != "DECODE"), 3), flush=True)
