"""bench-shaped bricks (256x256x128, real field from the middle of the volume) at several tolerance / epoch / variant settings vs the oracle"""
import sys, time
from common import gen, ROOT
import __graft_entry__ as g
g.build()
import numpy as np, torch
import volumerenderer_amd as vr
from oracle import oracle
import bench
bd, gd = (256, 256, 128), (2048, 2048, 1920)
vox4 = bench.make_volume_gpu(torch, gd, bd, seed=12345)
pick = [448 + 7, 448 + 36, 512 + 9, 384 + 60, 576 + 27, 3]       # bricks around the interface + one far away
sel = vox4[torch.tensor(pick, device="cuda")].contiguous()
del vox4
host = sel.cpu().numpy()
bad = 0; t0 = time.time()
for tol, ep, var in ((1, 2, 2), (3, 2, 2), (2, 3, 0)):
    bs = vr.BrickSet(len(pick), bd, tol, ep, var)
    bs.build(sel.reshape(-1))
    dec = bs.decode().cpu().numpy().reshape(host.shape)
    cutd = 21
    cut = bs.decode(cut_depth=cutd).cpu().numpy().reshape(host.shape)
    rdec = bs.decode_range(cut_depth=-1).cpu().numpy().reshape(host.shape) if var == 2 else None
    for b in range(len(pick)):
        ref = oracle.OracleTree(host[b].copy(), tolerance=tol, max_epochs=ep, guarded=var >= 1, midrange=var == 2).build()
        info = bs.info(b)
        ok = (info["num_active_nodes"] == ref.numActiveNodes and list(bs.distance_map(b)) == list(ref.distanceMap)
              and np.array_equal(bs.tree(b), ref.tree) and np.array_equal(dec[b], ref.levelCut())
              and np.array_equal(cut[b], ref.levelCutProgressive(cutd)) and info["num_reverts"] == ref.numReverts)
        if ok and var == 2:
            ok = np.array_equal(bs.tree_range(b), ref.tree_range) and np.array_equal(rdec[b], ref.levelCutRange(None))
        if not ok:
            bad += 1
            parts = dict(n=info["num_active_nodes"] == ref.numActiveNodes, dm=list(bs.distance_map(b)) == list(ref.distanceMap),
                         tree=np.array_equal(bs.tree(b), ref.tree), dec=np.array_equal(dec[b], ref.levelCut()),
                         cut=np.array_equal(cut[b], ref.levelCutProgressive(cutd)), rev=info["num_reverts"] == ref.numReverts)
            if var == 2:
                parts.update(rng=np.array_equal(bs.tree_range(b), ref.tree_range), dmr=list(bs.distance_map_range(b)) == list(ref.distanceMap_range),
                             rdec=np.array_equal(rdec[b], ref.levelCutRange(None)))
                if not parts["rdec"]:
                    w = np.argwhere(rdec[b] != ref.levelCutRange(None))
                    print("  rdec ndiff", len(w), "first", w[:2].tolist(), "last", w[-1].tolist())
                if not parts["rng"]:
                    a_, b_ = bs.tree_range(b), ref.tree_range
                    m_ = min(len(a_), len(b_)); d_ = np.flatnonzero(a_[:m_] != b_[:m_])
                    print("  range stream len", len(a_), len(b_), "ndiff", len(d_), "first", d_[:3])
            print("MISMATCH", (tol, ep, var), "brick", pick[b], parts, flush=True)
    print("setting", (tol, ep, var), "done, %.0f s" % (time.time() - t0), flush=True)
print("mismatches", bad)
