import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as g
g.build()
import numpy as np, torch
import volumerenderer_amd as vr
from oracle import oracle

def gen(rng, shape, kind):
    z, y, x = shape
    if kind == 0: return rng.integers(0, 256, shape, dtype=np.uint8)
    if kind == 1:
        zz, yy, xx = np.meshgrid(np.arange(z), np.arange(y), np.arange(x), indexing="ij")
        v = 128 + 100 * np.sin(xx * rng.uniform(0.05, 0.6)) * np.cos(yy * rng.uniform(0.05, 0.6)) + zz * rng.uniform(-2, 2) + rng.integers(0, rng.integers(1, 6), shape)
        return np.clip(v, 0, 255).astype(np.uint8)
    if kind == 2:   # piecewise constant boxes + noise patches
        v = np.full(shape, int(rng.integers(0, 256)), np.int64)
        for _ in range(rng.integers(1, 6)):
            a = [sorted(rng.integers(0, s + 1, 2)) for s in shape]
            v[a[0][0]:a[0][1], a[1][0]:a[1][1], a[2][0]:a[2][1]] = rng.integers(0, 256)
        for _ in range(rng.integers(0, 3)):
            a = [sorted(rng.integers(0, s + 1, 2)) for s in shape]
            sub = v[a[0][0]:a[0][1], a[1][0]:a[1][1], a[2][0]:a[2][1]]
            sub += rng.integers(-rng.integers(1, 40), 40, sub.shape)
        return np.clip(v, 0, 255).astype(np.uint8)
    if kind == 3:   # saturated ends with noise (clamp cases)
        v = np.where(rng.random(shape) < 0.5, rng.integers(0, 12, shape), rng.integers(244, 256, shape))
        v = np.where(rng.random(shape) < 0.2, rng.integers(0, 256, shape), v)
        return v.astype(np.uint8)
    return np.full(shape, int(rng.integers(0, 256)), np.uint8)

shapes = [(32, 64, 128), (64, 64, 128), (16, 128, 256), (64, 32, 256)]
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
        os.environ["VRHIP_DECODE_WALK"] = "1"
        ok = np.array_equal(bs.decode().cpu().numpy().reshape(shape), ref.levelCut())
        del os.environ["VRHIP_DECODE_WALK"]
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
