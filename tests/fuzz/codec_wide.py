import sys, time
from common import gen, ROOT
import __graft_entry__ as g
g.build()
import numpy as np, torch
import volumerenderer_amd as vr
from oracle import oracle

shapes = [(4, 8, 128), (8, 8, 128), (4, 16, 256), (16, 8, 128), (8, 32, 128), (4, 8, 512)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); bad = 0
for it in range(n):
    shape = shapes[rng.integers(0, len(shapes))]
    kind = int(rng.integers(0, 5)); tol = int(rng.choice([0, 1, 1, 2, 5, 9])); ep = int(rng.choice([1, 2, 2, 3, 5])); var = int(rng.choice([0, 0, 1]))
    vol = gen(rng, shape, kind)
    z, y, x = shape
    ref = oracle.OracleTree(vol.copy(), tolerance=tol, max_epochs=ep, guarded=bool(var)).build()
    bs = vr.BrickSet(1, (x, y, z), tol, ep, var)
    bs.build(vol.copy())
    info = bs.info(0)
    st = ref.leaf_stats()
    ok = (info["num_active_nodes"] == ref.numActiveNodes and list(bs.distance_map(0)) == list(ref.distanceMap)
          and np.array_equal(bs.tree(0), ref.tree) and info["num_reverts"] == ref.numReverts
          and info["max_error_before"] == st["max_before"] and info["max_error_after"] == st["max_after"]
          and abs(info["mean_l1_after"] - st["l1_after"]) < 1e-12
          and np.array_equal(bs.decode().cpu().numpy().reshape(shape), ref.levelCut()))
    if ok:
        import os
        bs.set_switch("decode_walk", 1)
        ok = np.array_equal(bs.decode().cpu().numpy().reshape(shape), ref.levelCut())
        bs.set_switch("decode_walk", 0)
        fs = vr.BrickSet(1, (x, y, z), tol, ep, var)
        fs.set_tree(0, ref.tree, ref.numActiveNodes, ref.distanceMap)
        ok = ok and np.array_equal(fs.decode().cpu().numpy().reshape(shape), ref.levelCut())
    if ok and ref.origTreeDepth >= 8:
        cut = int(rng.integers(2, ref.origTreeDepth))
        ok = np.array_equal(bs.decode(cut_depth=cut).cpu().numpy().reshape(shape), ref.levelCutProgressive(cut))
    if not ok:
        bad += 1
        print("MISMATCH", it, shape, kind, tol, ep, var, flush=True)
print("cases", n, "mismatches", bad, "%.1f s" % (time.time() - t0))
