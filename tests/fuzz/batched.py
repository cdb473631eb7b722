"""large batched sets (the level loop forks over brick ranges at >= 32 bricks): every brick vs the oracle"""
import sys, time
from common import gen, ROOT
import __graft_entry__ as g
g.build()
import numpy as np, torch
import volumerenderer_amd as vr
from oracle import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0; t0 = time.time()
for it in range(n):
    shape = [(16, 16, 16), (16, 32, 32), (32, 32, 32), (16, 16, 64)][int(rng.integers(0, 4))]
    nb = int(rng.choice([33, 48, 64, 70, 97]))
    tol = int(rng.choice([0, 1, 1, 2, 5])); ep = int(rng.choice([1, 2, 2, 3])); var = int(rng.choice([0, 1, 2]))
    streams = int(rng.choice([1, 2, 3, 4]))
    vols = [gen(rng, shape, int(rng.integers(0, 5))) for _ in range(nb)]
    z, y, x = shape
    bs = vr.BrickSet(nb, (x, y, z), tol, ep, var).set_concurrency(streams)
    bs.build(np.stack(vols))
    dec = bs.decode().cpu().numpy().reshape((nb,) + shape)
    for i in rng.choice(nb, 12, replace=False):
        i = int(i)
        ref = oracle.OracleTree(vols[i].copy(), tolerance=tol, max_epochs=ep, guarded=var >= 1, midrange=var == 2).build()
        ok = np.array_equal(bs.tree(i), ref.tree) and list(bs.distance_map(i)) == list(ref.distanceMap) and np.array_equal(dec[i], ref.levelCut())
        ok = ok and bs.info(i)["num_reverts"] == ref.numReverts
        if ok and var == 2:
            ok = np.array_equal(bs.tree_range(i), ref.tree_range) and list(bs.distance_map_range(i)) == list(ref.distanceMap_range)
        if not ok:
            bad += 1
            print("MISMATCH", it, shape, nb, i, tol, ep, var, streams, flush=True)
print("cases", n, "mismatches", bad, "%.1f s" % (time.time() - t0))
