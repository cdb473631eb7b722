"""encode-only timing on the bench volume"""
import sys, os, time
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
g.build()
import numpy as np, torch
import volumerenderer_amd as vr
import bench
bd, gd = (256, 256, 128), (2048, 2048, 1920)
vox4 = bench.make_volume_gpu(torch, gd, bd, seed=12345)
vox = vox4.reshape(-1)
bs = vr.BrickSet(vox4.shape[0], bd, 1, 2)
bs.build(vox); torch.cuda.synchronize()
acc = {}
for i in range(int(os.environ.get('ENC_REPS', 4))):
    bs.build(vox); torch.cuda.synchronize()
    tm = bs.last_timings()
    for k in ("BUILD", "COMPRESS", "PRUNE", "CONVERT"): acc.setdefault(k, []).append(tm[k])
print(os.environ.get("TAG", ""), {k: round(min(v), 3) for k, v in acc.items()}, "total %.3f" % sum(min(v) for v in acc.values()), flush=True)
fb = [bs.info(b)["est_exact_segments"] for b in range(vox4.shape[0])]
print("est exact segments: total %d, max %d" % (sum(fb), max(fb)), flush=True)
