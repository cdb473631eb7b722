"""MidRangeTree fuzz: both streams, packed4, decodes vs the oracle"""
import os, sys, time
from common import gen, ROOT
import __graft_entry__ as g
g.build()
import numpy as np, torch
import volumerenderer_amd as vr
from oracle import oracle
shapes = [(16, 16, 16), (32, 32, 32), (16, 32, 64), (64, 64, 64), (8, 16, 128), (16, 16, 256), (32, 64, 128)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); bad = 0
for it in range(n):
    shape = shapes[rng.integers(0, len(shapes))]
    kind = int(rng.integers(0, 5)); tol = int(rng.choice([0, 1, 1, 2, 5])); ep = int(rng.choice([1, 2, 2, 3]))
    vol = gen(rng, shape, kind)
    z, y, x = shape
    ref = oracle.OracleTree(vol.copy(), tolerance=tol, max_epochs=ep, midrange=True, guarded=True).build()
    bs = vr.BrickSet(1, (x, y, z), tol, ep, 2)
    bs.build(vol.copy())
    ok = (np.array_equal(bs.tree(0), ref.tree) and np.array_equal(bs.tree_range(0), ref.tree_range)
          and list(bs.distance_map(0)) == list(ref.distanceMap) and list(bs.distance_map_range(0)) == list(ref.distanceMap_range)
          and np.array_equal(bs.packed4(0), ref.convertToByteArray())
          and np.array_equal(bs.decode().cpu().numpy().reshape(shape), ref.levelCut())
          and np.array_equal(bs.decode_range(cut_depth=-1).cpu().numpy().reshape(shape), ref.levelCutRange(None)))
    if not ok:
        bad += 1
        parts = dict(tree=np.array_equal(bs.tree(0), ref.tree), rng=np.array_equal(bs.tree_range(0), ref.tree_range),
                     dm=list(bs.distance_map(0)) == list(ref.distanceMap), dmr=list(bs.distance_map_range(0)) == list(ref.distanceMap_range),
                     p4=np.array_equal(bs.packed4(0), ref.convertToByteArray()),
                     dec=np.array_equal(bs.decode().cpu().numpy().reshape(shape), ref.levelCut()),
                     decr=np.array_equal(bs.decode_range(cut_depth=-1).cpu().numpy().reshape(shape), ref.levelCutRange(None)))
        print("MISMATCH", it, shape, kind, tol, ep, parts, flush=True)
        if not parts["rng"]:
            a, b = bs.tree_range(0), ref.tree_range
            print("  range stream lengths", len(a), len(b), "first diff byte", int(np.argmax(a[:min(len(a), len(b))] != b[:min(len(a), len(b))])), "ndiff", int((a[:min(len(a), len(b))] != b[:min(len(a), len(b))]).sum()))
        if not parts["decr"]:
            da = bs.decode_range(cut_depth=-1).cpu().numpy().reshape(shape); db = ref.levelCutRange(None)
            w = np.argwhere(da != db)
            print("  range decode: ndiff", len(w), "first", w[:3].tolist(), da[tuple(w[0])], db[tuple(w[0])])
        if not parts["dec"]:
            da = bs.decode().cpu().numpy().reshape(shape); db = ref.levelCut()
            w = np.argwhere(da != db)
            print("  decode: ndiff", len(w), "first", w[:3].tolist(), da[tuple(w[0])], db[tuple(w[0])])
        if len(sys.argv) > 3:
            np.save(os.path.join(ROOT, "gpurun_out", "mr_case_%d.npy") % it, vol)
print("cases", n, "mismatches", bad, "%.1f s" % (time.time() - t0))
