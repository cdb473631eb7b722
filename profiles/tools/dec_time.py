"""decode-only timing on the bench volume: python profiles/tools/dec_time.py [nbricks]
times the default kernel (k_decode_region) and, unless QUICK=1, round 2's k_decode_quad and round 1's k_decode_fine
on the same set (vr_brickset_set_switch), and checks that they agree"""
import sys, os, time
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
g.build()
import numpy as np, torch
import volumerenderer_amd as vr
import bench
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 0
bd, gd = (256, 256, 128), (2048, 2048, 1920)
vox4 = bench.make_volume_gpu(torch, gd, bd, seed=12345)
if nb:
    grid = (8, 8, 15)
    order = sorted(range(vox4.shape[0]), key=lambda b: abs(b // 64 - 7))
    vox4 = vox4[torch.tensor(order[:nb], device="cuda")].contiguous()
B = vox4.shape[0]
vox = vox4.reshape(-1)
out = torch.empty_like(vox)
bs = vr.BrickSet(B, bd, 1, 2)
bs.build(vox); torch.cuda.synchronize()
V = bd[0] * bd[1] * bd[2]
alg = sum(V + bs.info(b)["tree_bytes"] + 31 for b in range(B))
def run(tag):
    ms = []
    for i in range(6):
        bs.decode(out); torch.cuda.synchronize(); ms.append(bs.last_timings()["DECODE"])
    print("%-10s decode ms min %.3f med %.3f  -> %.0f GB/s (frac %.3f)" % (tag, min(ms), sorted(ms)[3], alg / min(ms) / 1e6, alg / min(ms) / 1e6 / 8000), flush=True)
    return out.clone()
ref = run(os.environ.get("TAG", "region"))
if os.environ.get("QUICK") != "1":
    for name in ("decode_quad", "decode_fine_v1"):
        bs.set_switch(name, 1)
        o = run(name)
        bs.set_switch(name, 0)
        print("   equal to the default kernel's output:", bool(torch.equal(o, ref)), flush=True)
        del o
