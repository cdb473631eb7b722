"""fuzz: general extents, batched sets with mixed bricks (incl. constant), foreign streams with cuts, save/open, error helpers"""
import sys, time, os, tempfile
from common import gen, ROOT
import __graft_entry__ as g
g.build()
import numpy as np, torch
import volumerenderer_amd as vr
from oracle import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); bad = 0
tmp = tempfile.mkdtemp()
def report(tag, what):
    global bad
    bad += 1
    print("MISMATCH", tag, what, flush=True)
for it in range(n):
    mode = it % 3
    tol = int(rng.choice([0, 1, 1, 2, 5])); ep = int(rng.choice([1, 2, 2, 3])); kind = int(rng.integers(0, 5))
    if mode == 0:      # general extents
        shape = tuple(int(v) for v in (rng.integers(2, 40), rng.integers(2, 40), rng.integers(2, 70)))
        vol = gen(rng, shape, kind)
        z, y, x = shape
        var = int(rng.choice([0, 0, 2]))
        ref = oracle.OracleTree(vol.copy(), tolerance=tol, max_epochs=ep, midrange=var == 2, guarded=var == 2).build()
        bs = vr.BrickSet(1, (x, y, z), tol, ep, var)
        bs.build(vol.copy())
        ok = np.array_equal(bs.tree(0), ref.tree) and list(bs.distance_map(0)) == list(ref.distanceMap)
        ok = ok and np.array_equal(bs.decode().cpu().numpy().reshape(shape), ref.levelCut())
        if ok and ref.maxTreeDepth > 2:
            cut = int(rng.integers(0, ref.maxTreeDepth))
            ok = np.array_equal(bs.decode(cut_depth=cut).cpu().numpy().reshape(shape), ref.levelCutProgressive(cut))
        if ok and var == 2:
            ok = np.array_equal(bs.tree_range(0), ref.tree_range) and np.array_equal(bs.decode_range(cut_depth=-1).cpu().numpy().reshape(shape), ref.levelCutRange(None))
        if ok:
            p = os.path.join(tmp, "f.bin"); bs.save(p)
            fs = vr.BrickSet.open(p, var)
            ok = np.array_equal(fs.decode().cpu().numpy().reshape(shape), ref.levelCut())
        if not ok: report("general", (shape, kind, tol, ep, var))
    elif mode == 1:    # batched mixed bricks
        shape = [(32, 32, 32), (16, 32, 64), (64, 64, 64), (8, 16, 128)][int(rng.integers(0, 4))]
        nb = int(rng.integers(2, 7))
        vols = [gen(rng, shape, int(rng.integers(0, 5))) for _ in range(nb)]
        z, y, x = shape
        bs = vr.BrickSet(nb, (x, y, z), tol, ep)
        bs.build(np.stack(vols))
        dec = bs.decode().cpu().numpy().reshape((nb,) + shape)
        cut = None
        for i, v in enumerate(vols):
            ref = oracle.OracleTree(v.copy(), tolerance=tol, max_epochs=ep).build()
            ok = np.array_equal(bs.tree(i), ref.tree) and list(bs.distance_map(i)) == list(ref.distanceMap) and np.array_equal(dec[i], ref.levelCut())
            ok = ok and bs.info(i)["num_reverts"] == ref.numReverts
            if cut is None: cut = int(rng.integers(1, ref.maxTreeDepth))
            if ok:
                dc = bs.decode(cut_depth=cut).cpu().numpy().reshape((nb,) + shape)[i]
                ok = np.array_equal(dc, ref.levelCutProgressive(cut))
            if not ok: report("batched", (shape, nb, i, tol, ep, cut))
    else:              # foreign stream + cuts
        shape = [(16, 16, 16), (32, 32, 32), (16, 16, 256), (8, 8, 512), (64, 64, 32)][int(rng.integers(0, 5))]
        vol = gen(rng, shape, kind)
        z, y, x = shape
        ref = oracle.OracleTree(vol.copy(), tolerance=tol, max_epochs=ep).build()
        fs = vr.BrickSet(1, (x, y, z), tol, ep)
        fs.set_tree(0, ref.tree, ref.numActiveNodes, ref.distanceMap)
        ok = np.array_equal(fs.decode().cpu().numpy().reshape(shape), ref.levelCut())
        for _ in range(3):
            cut = int(rng.integers(0, ref.maxTreeDepth))
            ok = ok and np.array_equal(fs.decode(cut_depth=cut).cpu().numpy().reshape(shape), ref.levelCutProgressive(cut))
        if not ok: report("foreign", (shape, kind, tol, ep))
print("cases", n, "mismatches", bad, "%.1f s" % (time.time() - t0))
