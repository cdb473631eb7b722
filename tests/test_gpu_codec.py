"""GPU parity: the HIP encode / decode path (through the C ABI) against the CPU oracle.

Bit-exact: tree bytes, distanceMap, numActiveNodes, decoded voxels."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vr():
    import torch
    assert torch.cuda.is_available()
    import __graft_entry__ as g
    g.build()
    import volumerenderer_amd as vr
    return vr


def rm_like(shape, seed=3):
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    h = shape[0] / 2 + 3 * np.sin(x * 0.4) + 2 * np.cos(y * 0.23)
    v = 128 + 120 * np.tanh((z - h) / 3.0) + rng.integers(0, 3, shape)
    return np.clip(v, 0, 255).astype(np.uint8)


def check_case(vr, O, vol, tol, ep, variant=0):
    z, y, x = vol.shape
    ref = O.OracleTree(vol.copy(), tolerance=tol, max_epochs=ep).build()
    bs = vr.BrickSet(1, (x, y, z), tol, ep, variant)
    bs.build(vol.copy())
    info = bs.info(0)
    assert info["orig_tree_depth"] == ref.origTreeDepth and info["max_tree_depth"] == ref.maxTreeDepth
    assert list(bs.distance_map(0)) == list(ref.distanceMap)
    assert info["num_active_nodes"] == ref.numActiveNodes
    assert np.array_equal(bs.tree(0), ref.tree)
    assert info["num_reverts"] == ref.numReverts
    # the zero-run rewrite (R.cpp:662-669,686-688): the oracle implements and counts it, the GPU emitters only count
    # the branches that would need it -- both must say "never" (a non-zero count would mean diverging streams)
    assert info["zero_run_rewrites"] == 0 and ref.zeroRunRewrites == 0
    st = ref.leaf_stats()
    assert info["max_error_before"] == st["max_before"] and info["max_error_after"] == st["max_after"]
    assert abs(info["mean_l1_after"] - st["l1_after"]) < 1e-12
    dec = bs.decode().cpu().numpy().reshape(z, y, x)
    assert np.array_equal(dec, ref.levelCut())
    return ref, bs


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64])
def test_sphere_n3_cubes(vr, oracle, n):
    check_case(vr, oracle, oracle.gen_sphere(n, 7), 1, 2)


@pytest.mark.parametrize("shape", [(8, 16, 16), (16, 8, 32), (4, 64, 2), (32, 32, 16), (1, 1, 8), (1, 16, 1)])
def test_anisotropic_bricks(vr, oracle, shape):
    rng = np.random.default_rng(11)
    check_case(vr, oracle, rng.integers(0, 256, shape, dtype=np.uint8), 2, 2)
    check_case(vr, oracle, rm_like(shape), 1, 2)


@pytest.mark.parametrize("tol", [0, 1, 2, 4, 6, 12])
@pytest.mark.parametrize("ep", [1, 2, 5])
def test_tolerance_epoch_sweep(vr, oracle, tol, ep):
    rng = np.random.default_rng(tol * 10 + ep)
    check_case(vr, oracle, rng.integers(0, 256, (16, 16, 16), dtype=np.uint8), tol, ep)
    check_case(vr, oracle, oracle.gen_sphere(32, 3), tol, ep)


def test_reverts_reproduced(vr, oracle):
    """Defect C-2: stale codes after a gradient-descent revert."""
    hit = 0
    for seed in range(6):
        rng = np.random.default_rng(100 + seed)
        ref, _ = check_case(vr, oracle, rng.integers(0, 256, (32, 32, 32), dtype=np.uint8), 0, 5)
        hit += ref.numReverts
    ref, _ = check_case(vr, oracle, oracle.gen_sphere(128, 7), 1, 2)   # the survey's revert case
    assert ref.numReverts >= 1
    assert hit + ref.numReverts >= 1


def test_degenerate_volumes(vr, oracle):
    for v in (0, 37, 255):
        ref, bs = check_case(vr, oracle, np.full((16, 16, 16), v, np.uint8), 1, 2)
        assert bs.info(0)["num_active_nodes"] == (1 if v == 0 else 3)
    check_case(vr, oracle, np.zeros((1, 1, 1), np.uint8) + 9, 1, 2)   # D = 0
    check_case(vr, oracle, oracle.gen_sphere(16, 7), 1, 0)            # maxEpochs = 0


@pytest.mark.parametrize("shape", [(2, 1, 1), (8, 8, 8), (16, 32, 64), (64, 64, 64)])
def test_constant_brick_closed_form(vr, oracle, shape):
    """Constant bricks take k_const_finish (tolerance >= 1, maxEpochs >= 1) or the general path
    (tolerance 0 / maxEpochs 0): both must equal the oracle, alone and inside a mixed batch."""
    for v in (0, 1, 128, 255):
        for tol, ep in ((1, 2), (6, 5), (1, 1), (0, 2), (3, 0)):
            check_case(vr, oracle, np.full(shape, v, np.uint8), tol, ep)
    z, y, x = shape
    rng = np.random.default_rng(9)
    vols = [np.full(shape, 0, np.uint8), rng.integers(0, 256, shape, dtype=np.uint8), np.full(shape, 77, np.uint8),
            rm_like(shape), np.full(shape, 255, np.uint8)]
    bs = vr.BrickSet(len(vols), (x, y, z), 1, 2)
    bs.build(np.stack(vols))
    dec = bs.decode().cpu().numpy().reshape(len(vols), z, y, x)
    D = bs.info(0)["orig_tree_depth"]
    cut = bs.decode(cut_depth=max(D - 3, 0)).cpu().numpy().reshape(len(vols), z, y, x)
    for i, v in enumerate(vols):
        ref = oracle.OracleTree(v.copy(), tolerance=1, max_epochs=2).build()
        assert bs.info(i)["num_active_nodes"] == ref.numActiveNodes
        assert np.array_equal(bs.tree(i), ref.tree)
        assert list(bs.distance_map(i)) == list(ref.distanceMap)
        assert np.array_equal(dec[i], ref.levelCut())
        assert np.array_equal(cut[i], ref.levelCutProgressive(max(D - 3, 0)))
        assert bs.info(i)["num_reverts"] == ref.numReverts


@pytest.mark.parametrize("shape", [(8, 8, 8), (16, 32, 64), (64, 64, 64)])
def test_constant_bricks_in_a_midrange_set(vr, oracle, shape):
    """MidRangeTree: constant bricks take the closed form for BOTH streams ([1][3][3] / [0][3][3], all-zero range
    distances) next to ordinary bricks in one batch; tolerance 0 keeps them on the general path."""
    z, y, x = shape
    rng = np.random.default_rng(11)
    vols = [np.full(shape, 0, np.uint8), rng.integers(0, 256, shape, dtype=np.uint8), np.full(shape, 77, np.uint8),
            rm_like(shape), np.full(shape, 255, np.uint8)]
    for tol, ep in ((1, 2), (3, 1), (0, 2)):
        bs = vr.BrickSet(len(vols), (x, y, z), tol, ep, 2)
        bs.build(np.stack(vols))
        dec = bs.decode().cpu().numpy().reshape(len(vols), z, y, x)
        rdec = bs.decode_range(cut_depth=-1).cpu().numpy().reshape(len(vols), z, y, x)
        D = bs.info(0)["orig_tree_depth"]
        rcut = bs.decode_range(cut_depth=max(D - 4, 0)).cpu().numpy().reshape(len(vols), z, y, x)
        for i, v in enumerate(vols):
            ref = oracle.OracleTree(v.copy(), tolerance=tol, max_epochs=ep, midrange=True, guarded=True).build()
            what = (tol, ep, i)
            assert bs.info(i)["num_active_nodes"] == ref.numActiveNodes and bs.info(i)["num_reverts"] == ref.numReverts, what
            assert np.array_equal(bs.tree(i), ref.tree) and np.array_equal(bs.tree_range(i), ref.tree_range), what
            assert list(bs.distance_map(i)) == list(ref.distanceMap), what
            assert list(bs.distance_map_range(i)) == list(ref.distanceMap_range), what
            assert np.array_equal(bs.packed4(i), ref.convertToByteArray()), what
            assert np.array_equal(dec[i], ref.levelCut()) and np.array_equal(rdec[i], ref.levelCutRange(None)), what
            assert np.array_equal(rcut[i], ref.levelCutRange(max(D - 4, 0))), what


def test_guarded_variant_same_bytes(vr, oracle):
    vol = oracle.gen_sphere(32, 7)
    a = vr.BrickSet(1, (32, 32, 32), 1, 3, 0).build(vol.copy())
    b = vr.BrickSet(1, (32, 32, 32), 1, 3, 1).build(vol.copy())
    assert np.array_equal(a.tree(0), b.tree(0))


def test_batched_bricks_match_single(vr, oracle):
    import torch
    rng = np.random.default_rng(5)
    vols = [oracle.gen_sphere(32, 7), rng.integers(0, 256, (32, 32, 32), dtype=np.uint8),
            np.full((32, 32, 32), 200, np.uint8), rm_like((32, 32, 32)), oracle.gen_sphere(32, 0)]
    bs = vr.BrickSet(len(vols), (32, 32, 32), 1, 2)
    bs.build(np.stack(vols))
    dec = bs.decode().cpu().numpy().reshape(len(vols), 32, 32, 32)
    for i, v in enumerate(vols):
        ref = oracle.OracleTree(v.copy(), tolerance=1, max_epochs=2).build()
        assert bs.info(i)["num_active_nodes"] == ref.numActiveNodes
        assert np.array_equal(bs.tree(i), ref.tree)
        assert list(bs.distance_map(i)) == list(ref.distanceMap)
        assert np.array_equal(dec[i], ref.levelCut())


def test_golden_file_decodes_on_gpu(vr, oracle, tmp_path):
    """open() of the tree file the REFERENCE wrote; decode == oracle; save() byte-identical."""
    import json, os
    gold = os.path.join(os.path.dirname(__file__), "golden")
    ka = json.load(open(os.path.join(gold, "survey_known_answers.json")))["volume_kdtree"][0]
    t = vr.VolumeKdtree().open(os.path.join(gold, ka["saved_file"]))
    assert t.numActiveNodes == ka["numActiveNodes"]
    dec = t.levelCut(t.maxTreeDepth).cpu().numpy()
    assert "%016x" % oracle.fnv1a64(dec) == ka["voxels_fnv"]
    vol = oracle.gen_sphere(16, 7)
    k = vr.VolumeKdtree(vol.copy(), 16, 16, 16)
    k.setMaxEpochs(2); k.setErrorTolerance(1)
    k.build()
    assert "%016x" % oracle.fnv1a64(k.tree) == ka["tree_fnv"]
    assert list(k.distanceMap) == ka["distanceMap"]
    p = str(tmp_path / "t.bin")
    k.save(p)
    assert open(p, "rb").read() == open(os.path.join(gold, ka["saved_file"]), "rb").read()
    k.levelCut(k.maxTreeDepth)
    assert k.measureMaxError() == oracle.measure_max_error(dec.reshape(16, 16, 16), vol)
    assert abs(k.measureMeanError() - oracle.measure_mean_error(dec.reshape(16, 16, 16), vol)) < 1e-12


def test_midrange_tree(vr, oracle):
    vol = oracle.gen_sphere(32, 7)
    ref = oracle.OracleTree(vol.copy(), tolerance=1, max_epochs=1, midrange=True, guarded=True).build()
    t = vr.MidRangeTree(vol.copy(), 32, 32, 32)
    t.setMaxEpochs(1); t.setErrorTolerance(1)
    t.build()
    assert t.numActiveNodes == ref.numActiveNodes == 172562
    assert np.array_equal(t.tree, ref.tree)
    assert np.array_equal(t.tree_range, ref.tree_range)
    assert list(t.distanceMap_range) == list(ref.distanceMap_range)
    info, st = t._bs.info(0), ref.leaf_stats()        # D = 15: the 12-level prune kernel with a range stream
    assert info["max_error_before"] == st["max_before"] and info["max_error_after"] == st["max_after"]
    assert abs(info["mean_l1_after"] - st["l1_after"]) < 1e-12
    pk = t.convertToByteArray()
    assert np.array_equal(pk, ref.convertToByteArray())
    assert "%016x" % oracle.fnv1a64(pk) == "d363bd19dd90d0fd"     # SURVEY Appendix B known answer
    rng = np.random.default_rng(2)
    v2 = rng.integers(0, 256, (16, 16, 8), dtype=np.uint8)
    r2 = oracle.OracleTree(v2.copy(), tolerance=2, max_epochs=3, midrange=True, guarded=True).build()
    t2 = vr.MidRangeTree(v2.copy(), 8, 16, 16)
    t2.setMaxEpochs(3); t2.setErrorTolerance(2)
    t2.build()
    assert np.array_equal(t2.tree, r2.tree) and np.array_equal(t2.tree_range, r2.tree_range)
    assert np.array_equal(t2.convertToByteArray(), r2.convertToByteArray())


@pytest.mark.parametrize("gen", ["sphere_n3", "sphere_n0", "random"])
def test_full_brick_256(vr, oracle, gen):
    """BASELINE config 1/2: one 256^3 brick, hashes recorded from the reference."""
    mask = {"sphere_n3": 7, "sphere_n0": 0, "random": 256}[gen]
    vol = oracle.gen_sphere(256, mask)
    ref, bs = check_case(vr, oracle, vol, 1, 2)
    if gen == "sphere_n3":
        assert "%016x" % oracle.fnv1a64(bs.tree(0)) == "aaa282422610f044"
        assert "%016x" % oracle.fnv1a64(bs.decode().cpu().numpy()) == "e97ae40e1e4bb47c"
    if gen == "sphere_n0":
        assert bs.info(0)["num_active_nodes"] == 21373869


def test_config3_error_tolerance_sweep(vr, oracle):
    """BASELINE config 3 shape (2x2x2 grid of bricks, one tree per brick, tolerance sweep): the DECODED
    max/mean error (vr_measure_error on levelCut output) and the encoder's self-reported leaf error are
    reported separately and both equal the oracle's (they differ when a revert happened, defect C-2)."""
    import torch
    n = 64
    rng = np.random.default_rng(9)
    bricks = []
    for b in range(8):
        base = oracle.gen_sphere(n, 7 if b % 2 else 3, seed=12345 + b).astype(np.int32)
        bricks.append(np.clip(base + rng.integers(-b, b + 1, base.shape), 0, 255).astype(np.uint8))
    vol = np.stack(bricks)
    for tol in (0, 1, 2, 4, 6, 12):
        bs = vr.BrickSet(8, (n, n, n), tol, 5)
        bs.build(vol)
        dec = bs.decode()
        for b in range(8):
            ref = oracle.OracleTree(bricks[b].copy(), tolerance=tol, max_epochs=5).build()
            rdec = ref.levelCut()
            got = dec[b * n ** 3:(b + 1) * n ** 3]
            assert np.array_equal(got.cpu().numpy().reshape(n, n, n), rdec)
            mx, mean = vr.measure_error(got, bricks[b])
            assert mx == oracle.measure_max_error(rdec, bricks[b])
            assert abs(mean - oracle.measure_mean_error(rdec, bricks[b])) < 1e-12
            st = ref.leaf_stats()
            info = bs.info(b)
            assert info["max_error_after"] == st["max_after"]
            assert abs(info["mean_l1_after"] - st["l1_after"]) < 1e-12
            assert info["num_reverts"] == ref.numReverts
            if ref.numReverts == 0:
                assert mx <= max(tol, 0) or tol == 0 and mx == 0
        err = vr.query_error(dec, vol).cpu().numpy()
        assert err.max() == max(oracle.measure_max_error(dec[b * n ** 3:(b + 1) * n ** 3].cpu().numpy(), bricks[b]) for b in range(8))


def test_progressive_cut_matches_oracle(vr, oracle, tmp_path):
    """vr_brickset_decode(cut_depth < maxTreeDepth): defined progressive semantics (not the reference's
    de-synchronising walk, C-4) against oracle.levelCutProgressive -- tile kernel (128^3), lane kernel
    (32^3, 16x8x32) and a foreign stream reopened from a file."""
    rng = np.random.default_rng(3)
    cases = [oracle.gen_sphere(128, 7), oracle.gen_sphere(32, 3), rng.integers(0, 256, (32, 8, 16), dtype=np.uint8)]
    for vol in cases:
        z, y, x = vol.shape
        ref = oracle.OracleTree(vol.copy(), tolerance=1, max_epochs=2).build()
        bs = vr.BrickSet(1, (x, y, z), 1, 2).build(vol.copy())
        D, M = ref.origTreeDepth, ref.maxTreeDepth
        prev = None
        for cut in sorted({0, 1, 5, D - 7, D - 6, D - 5, D - 1, D, D + 1, D + 3, M - 1, M}):
            if cut < 0:
                continue
            got = bs.decode(cut_depth=cut).cpu().numpy().reshape(z, y, x)
            assert np.array_equal(got, ref.levelCutProgressive(cut)), "cut %d" % cut
        assert np.array_equal(bs.decode(cut_depth=M).cpu().numpy().reshape(z, y, x), ref.levelCut())
        # foreign stream (file written by save(), reopened): same answers from the bytes alone
        p = str(tmp_path / "t.bin")
        bs.save(p)
        fs = vr.BrickSet.open(p)
        for cut in (0, 3, D - 6, D - 2, M):
            if cut < 0:
                continue
            got = fs.decode(cut_depth=cut).cpu().numpy().reshape(z, y, x)
            assert np.array_equal(got, ref.levelCutProgressive(cut)), "foreign cut %d" % cut


def test_fused_emit_equals_two_pass_emit(vr, oracle, monkeypatch):
    """The default D >= 12 path (k_prune_emit12 + k_concat12) and the older per-quad emitter kept behind
    VRHIP_NO_FUSED_EMIT must produce the same bytes, index and statistics."""
    rng = np.random.default_rng(21)
    vols = [rm_like((32, 64, 32)), rng.integers(0, 256, (32, 64, 32), dtype=np.uint8), oracle.gen_sphere(32, 3).repeat(2, axis=1)]
    res = []
    for env in (None, "1"):
        if env: monkeypatch.setenv("VRHIP_NO_FUSED_EMIT", env)
        bs = vr.BrickSet(len(vols), (32, 64, 32), 1, 2)
        bs.build(np.stack(vols))
        dec = bs.decode().cpu().numpy()
        res.append([(bs.tree(i).tobytes(), tuple(sorted(bs.info(i).items()))) for i in range(len(vols))] + [dec.tobytes()])
    monkeypatch.delenv("VRHIP_NO_FUSED_EMIT")
    assert res[0] == res[1]
    ref = oracle.OracleTree(vols[1].copy(), tolerance=1, max_epochs=2).build()
    assert res[0][1][0] == ref.tree.tobytes()


def test_midrange_file_roundtrip(vr, oracle, tmp_path):
    """MidRangeTree::save layout byte-identical to the oracle's restatement of M.cpp:753-785; open() returns
    exactly what was saved (the reference's own reader does not, see test_midrange_file_layout)."""
    vol = oracle.gen_sphere(32, 7)
    ref = oracle.OracleTree(vol.copy(), tolerance=1, max_epochs=1, midrange=True, guarded=True).build()
    t = vr.MidRangeTree(vol.copy(), 32, 32, 32)
    t.setMaxEpochs(1); t.setErrorTolerance(1)
    t.build()
    p, q = str(tmp_path / "gpu.bin"), str(tmp_path / "ref.bin")
    t.save(p); ref.save(q)
    assert open(p, "rb").read() == open(q, "rb").read()
    u = vr.MidRangeTree().open(q)
    assert u.numActiveNodes == ref.numActiveNodes and (u.X, u.Y, u.Z) == (32, 32, 32)
    assert np.array_equal(u.tree, ref.tree) and np.array_equal(u.tree_range, ref.tree_range)
    assert list(u.distanceMap) == list(ref.distanceMap) and list(u.distanceMap_range) == list(ref.distanceMap_range)
    assert np.array_equal(u.convertToByteArray(), ref.convertToByteArray())
    assert np.array_equal(u.levelCut().cpu().numpy().reshape(32, 32, 32), ref.levelCut())
    r = str(tmp_path / "again.bin")
    u.save(r)
    assert open(r, "rb").read() == open(q, "rb").read()
    with pytest.raises(vr.VrError):
        vr.MidRangeTree().open(str(tmp_path / "missing.bin"))
    with pytest.raises(vr.VrError):
        vr.MidRangeTree().open(os.path.join(os.path.dirname(__file__), "golden", "ref_sphere_n3_16_tol1_ep2.tree.bin"))


def test_bench_workload_bricks_match_oracle(vr, oracle):
    """The bench's own brick shape and field (256x256x128 = depth 23, the fused 12-level kernels,
    XCD-swizzled pyramid, tile decode): three bricks through the interface, bit-exact with the oracle."""
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    bd, gd = (256, 256, 128), (256, 256, 384)
    vox4 = bench.make_volume_gpu(torch, gd, bd, seed=12345, kind="rm_volume")
    host = vox4.cpu().numpy()
    bs = vr.BrickSet(3, bd, 1, 2)
    bs.build(vox4.reshape(-1))
    dec = bs.decode().cpu().numpy().reshape(host.shape)
    cut = bs.decode(cut_depth=20).cpu().numpy().reshape(host.shape)
    assert len({host[b].tobytes() for b in range(3)}) == 3 and host[1].min() != host[1].max()
    for b in range(3):
        ref = oracle.OracleTree(host[b].copy(), tolerance=1, max_epochs=2).build()
        info = bs.info(b)
        assert info["orig_tree_depth"] == 23 and info["num_active_nodes"] == ref.numActiveNodes
        assert list(bs.distance_map(b)) == list(ref.distanceMap)
        assert oracle.fnv1a64(bs.tree(b)) == oracle.fnv1a64(ref.tree)
        assert np.array_equal(dec[b], ref.levelCut())
        assert np.array_equal(cut[b], ref.levelCutProgressive(20))
        st = ref.leaf_stats()
        assert info["max_error_before"] == st["max_before"] and info["max_error_after"] == st["max_after"]
        assert info["num_reverts"] == ref.numReverts


@pytest.mark.parametrize("seed", [0, 1])
def test_estimator_threshold_drift(vr, oracle, seed):
    """Slabs of very different noise amplitude make the running-mean threshold drift far and fast inside a
    level: the candidate windows are left repeatedly (later rounds, the exact last-resort walk).  The result
    must still be the serial one."""
    rng = np.random.default_rng(seed)
    n = 64
    amp = np.array([0, 60, 2, 120, 0, 8, 200, 1] * (n // 8))[:n]
    rng.shuffle(amp)
    vol = np.zeros((n, n, n), np.int64) + 128
    for z in range(n):
        if amp[z]:
            vol[z] += rng.integers(-amp[z] // 2, amp[z] // 2 + 1, (n, n))
    # Morton order interleaves z with x, y: the drift happens along every level, not once
    vol = np.clip(vol, 0, 255).astype(np.uint8)
    for tol, ep in ((1, 2), (4, 5)):
        check_case(vr, oracle, vol, tol, ep)
    check_case(vr, oracle, np.ascontiguousarray(vol.transpose(2, 1, 0)), 1, 2)


def _fuzz_volume(rng, shape, kind):
    z, y, x = shape
    if kind == 0:
        return rng.integers(0, 256, shape, dtype=np.uint8)
    if kind == 1:
        zz, yy, xx = np.meshgrid(np.arange(z), np.arange(y), np.arange(x), indexing="ij")
        v = (128 + 100 * np.sin(xx * rng.uniform(0.05, 0.6)) * np.cos(yy * rng.uniform(0.05, 0.6)) + zz * rng.uniform(-2, 2)
             + rng.integers(0, rng.integers(1, 6), shape))
        return np.clip(v, 0, 255).astype(np.uint8)
    if kind == 2:     # piecewise constant boxes with noisy patches
        v = np.full(shape, int(rng.integers(0, 256)), np.int64)
        for _ in range(rng.integers(1, 6)):
            a = [sorted(rng.integers(0, s + 1, 2)) for s in shape]
            v[a[0][0]:a[0][1], a[1][0]:a[1][1], a[2][0]:a[2][1]] = rng.integers(0, 256)
        for _ in range(rng.integers(0, 3)):
            a = [sorted(rng.integers(0, s + 1, 2)) for s in shape]
            sub = v[a[0][0]:a[0][1], a[1][0]:a[1][1], a[2][0]:a[2][1]]
            sub += rng.integers(-rng.integers(1, 40), 40, sub.shape)
        return np.clip(v, 0, 255).astype(np.uint8)
    if kind == 3:     # saturated ends with noise: the clamps at 0 / 255 matter
        v = np.where(rng.random(shape) < 0.5, rng.integers(0, 12, shape), rng.integers(244, 256, shape))
        return np.where(rng.random(shape) < 0.2, rng.integers(0, 256, shape), v).astype(np.uint8)
    return np.full(shape, int(rng.integers(0, 256)), np.uint8)


def test_randomized_parity(vr, oracle):
    """Seeded fuzz over shapes (both sides of the 12-level kernels), data kinds, tolerance, epochs and variant;
    scratch/fuzz.py is the same loop with thousands of cases (4400 run clean on an MI355X)."""
    shapes = [(16, 16, 16), (32, 16, 32), (32, 32, 32), (16, 32, 64), (64, 32, 16), (64, 64, 64), (8, 8, 8), (4, 32, 2), (64, 64, 32)]
    rng = np.random.default_rng(2026)
    for _ in range(120):
        shape = shapes[rng.integers(0, len(shapes))]
        kind, tol = int(rng.integers(0, 5)), int(rng.choice([0, 1, 1, 2, 5, 9]))
        ep, var = int(rng.choice([1, 2, 2, 3, 5])), int(rng.choice([0, 0, 1]))
        vol = _fuzz_volume(rng, shape, kind)
        z, y, x = shape
        ref = oracle.OracleTree(vol.copy(), tolerance=tol, max_epochs=ep, guarded=bool(var)).build()
        bs = vr.BrickSet(1, (x, y, z), tol, ep, var)
        bs.build(vol.copy())
        info, st, what = bs.info(0), ref.leaf_stats(), (shape, kind, tol, ep, var)
        assert info["num_active_nodes"] == ref.numActiveNodes, what
        assert list(bs.distance_map(0)) == list(ref.distanceMap), what
        assert np.array_equal(bs.tree(0), ref.tree), what
        assert info["num_reverts"] == ref.numReverts, what
        assert info["max_error_before"] == st["max_before"] and info["max_error_after"] == st["max_after"], what
        assert np.array_equal(bs.decode().cpu().numpy().reshape(shape), ref.levelCut()), what
        if ref.origTreeDepth >= 8:
            cut = int(rng.integers(2, ref.origTreeDepth))
            assert np.array_equal(bs.decode(cut_depth=cut).cpu().numpy().reshape(shape), ref.levelCutProgressive(cut)), what


def test_reopened_file_decodes_through_the_tile_kernel(vr, oracle, tmp_path):
    """A brick wide enough for k_decode_tile (X >= 128), saved, reopened (side-car index rebuilt on the host from
    the bytes alone) and decoded: equal to the oracle and to the decode of the tree that was built in place."""
    shape = (16, 32, 128)
    vol = rm_like(shape)
    t = vr.VolumeKdtree(vol.copy(), 128, 32, 16)
    t.setMaxEpochs(2); t.setErrorTolerance(1)
    t.build()
    want = t.levelCut().cpu().numpy().copy()
    p = str(tmp_path / "wide.bin")
    t.save(p)
    u = vr.VolumeKdtree().open(p)
    got = u.levelCut(u.maxTreeDepth).cpu().numpy()
    ref = oracle.OracleTree(vol.copy(), tolerance=1, max_epochs=2).build()
    assert np.array_equal(got, want) and np.array_equal(got.reshape(shape), ref.levelCut())
    cut = u.levelCut(9).cpu().numpy()          # progressive cut above the index level, foreign stream
    assert np.array_equal(cut.reshape(shape), ref.levelCutProgressive(9))


def _mixed_volume(rng, shape):
    """Constant boxes, noisy patches and saturated ends: pruned nodes of every size next to long grown branches."""
    v = np.full(shape, int(rng.integers(0, 256)), np.int64)
    for _ in range(int(rng.integers(2, 7))):
        a = [sorted(rng.integers(0, s + 1, 2)) for s in shape]
        v[a[0][0]:a[0][1], a[1][0]:a[1][1], a[2][0]:a[2][1]] = rng.integers(0, 256)
    for _ in range(int(rng.integers(1, 4))):
        a = [sorted(rng.integers(0, s + 1, 2)) for s in shape]
        sub = v[a[0][0]:a[0][1], a[1][0]:a[1][1], a[2][0]:a[2][1]]
        sub += rng.integers(-int(rng.integers(1, 60)), 60, sub.shape)
    if rng.random() < 0.5:
        m = rng.random(shape) < 0.05
        v[m] = rng.integers(0, 256, int(m.sum()))
    return np.clip(v, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("shape", [(64, 128, 128), (128, 128, 256), (128, 128, 128), (16, 16, 128), (64, 256, 256)])
def test_region_decode_all_axis_orders(vr, oracle, shape):
    """k_decode_region (a wave per 16^3 emit block, a workgroup per 128 x 16 x 16 region) wherever the twelve deepest
    levels are four equal (a, b, c) triples: x second in a triple (the bench's 256 x 256 x 128 shape), x deepest
    (256 x 128 x 128), x first (cubes).  Against the oracle's levelCut / levelCutProgressive at every cut the kernel
    serves, against round 2's k_decode_quad and the walking k_decode_tile, on volumes with pruned regions of every
    size next to long grown branches, with several bricks per launch.  (16, 16, 128) does not qualify and must
    still decode (the older kernels)."""
    rng = np.random.default_rng(4000 + shape[0] + shape[2])
    z, y, x = shape
    vols = [rm_like(shape, 5), _mixed_volume(rng, shape), _mixed_volume(rng, shape)]
    vols.append(np.full(shape, 77, np.uint8))                                    # a constant brick inside the batch
    vols.append(rng.integers(0, 256, shape, dtype=np.uint8) if z * y * x <= (1 << 21) else _mixed_volume(rng, shape))
    tol, ep = (1, 2) if shape[0] != 128 else (2, 3)
    bs = vr.BrickSet(len(vols), (x, y, z), tol, ep)
    bs.build(np.stack(vols))
    refs = [oracle.OracleTree(v.copy(), tolerance=tol, max_epochs=ep).build() for v in vols]
    D, M = refs[0].origTreeDepth, refs[0].maxTreeDepth
    for b, ref in enumerate(refs):
        assert np.array_equal(bs.tree(b), ref.tree), b
    for cut in [None, M - 1, D + 4, D + 1, D, D - 1, D - 2, D - 3]:
        got = (bs.decode() if cut is None else bs.decode(cut_depth=cut)).cpu().numpy().reshape((len(vols),) + shape)
        bs.set_switch("decode_quad", 1)
        quad = (bs.decode() if cut is None else bs.decode(cut_depth=cut)).cpu().numpy().reshape((len(vols),) + shape)
        bs.set_switch("decode_quad", 0)
        assert np.array_equal(got, quad), cut
        for b, ref in enumerate(refs):
            want = ref.levelCut() if cut is None else ref.levelCutProgressive(cut)
            assert np.array_equal(got[b], want), (b, cut)
    bs.set_switch("decode_walk", 1)
    walk = bs.decode().cpu().numpy().reshape((len(vols),) + shape)
    bs.set_switch("decode_walk", 0)
    assert np.array_equal(walk, bs.decode().cpu().numpy().reshape((len(vols),) + shape))
    # the same bytes as a foreign stream (open()): contiguous layout, side-cars from the host parse of the bytes
    fs = vr.BrickSet(1, (x, y, z), tol, ep)
    fs.set_tree(0, refs[1].tree, refs[1].numActiveNodes, refs[1].distanceMap)
    assert np.array_equal(fs.decode().cpu().numpy().reshape(shape), refs[1].levelCut())
    assert np.array_equal(fs.decode(cut_depth=D - 2).cpu().numpy().reshape(shape), refs[1].levelCutProgressive(D - 2))


@pytest.mark.parametrize("shape", [(4, 8, 128), (8, 16, 128), (16, 16, 256), (8, 8, 512)])
def test_fine_decode_equals_walk_decode_and_oracle(vr, oracle, monkeypatch, shape):
    """k_decode_fine (one lane per four voxels, token offsets from the fused encoder's per-4-leaf counts) against
    k_decode_tile (one lane walks 64 voxels; VRHIP_DECODE_WALK=1) and against the oracle's levelCut, full depth
    and progressive cuts on both sides of the index level."""
    rng = np.random.default_rng(1000 + shape[0] * shape[2])
    z, y, x = shape
    for case in range(6):
        tol = int(rng.choice([1, 2, 6, 9])); ep = int(rng.choice([1, 2, 5]))
        vol = _mixed_volume(rng, shape) if case else rng.integers(0, 256, shape, dtype=np.uint8)
        ref = oracle.OracleTree(vol.copy(), tolerance=tol, max_epochs=ep).build()
        bs = vr.BrickSet(1, (x, y, z), tol, ep)
        bs.build(vol.copy())
        assert np.array_equal(bs.tree(0), ref.tree)
        D = ref.origTreeDepth
        M = ref.maxTreeDepth
        cuts = [None, D, D - 1, D - 2, D - 3, D - 4, D - 5, D - 6, D - 7, 3, D + 1, D + 4, M - 1,
                int(rng.integers(1, ref.maxTreeDepth + 1))]
        for cut in cuts:
            want = ref.levelCut() if cut is None else ref.levelCutProgressive(cut)
            # default: k_decode_quad for cuts >= D-3 (one table lookup per voxel leaf), k_decode_fine above
            quad = (bs.decode() if cut is None else bs.decode(cut_depth=cut)).cpu().numpy().reshape(shape)
            bs.set_switch("decode_fine_v1", 1)
            fine = (bs.decode() if cut is None else bs.decode(cut_depth=cut)).cpu().numpy().reshape(shape)
            bs.set_switch("decode_fine_v1", 0)
            bs.set_switch("decode_walk", 1)
            walk = (bs.decode() if cut is None else bs.decode(cut_depth=cut)).cpu().numpy().reshape(shape)
            bs.set_switch("decode_walk", 0)
            assert np.array_equal(quad, want), (case, tol, ep, cut)
            assert np.array_equal(fine, want), (case, tol, ep, cut)
            assert np.array_equal(walk, want), (case, tol, ep, cut)
        # the same bytes installed as a foreign stream (what open() does): the per-4-leaf counts then come from
        # the host-side parse of the bytes, not from the encoder
        fs = vr.BrickSet(1, (x, y, z), tol, ep)
        fs.set_tree(0, ref.tree, ref.numActiveNodes, ref.distanceMap)
        assert np.array_equal(fs.decode().cpu().numpy().reshape(shape), ref.levelCut()), (case, tol, ep, "foreign")
        for cut in (D - 2, D - 3, D - 4, D + 2):
            assert np.array_equal(fs.decode(cut_depth=cut).cpu().numpy().reshape(shape), ref.levelCutProgressive(cut)), cut


def test_hashed_kdtree_interface(vr, oracle):
    """HashedKdtree keeps the reference's interface (HashedKdtree.h:26-143); parity with the reference class is
    unpinned (it cannot run), so what is checked is that the interface delivers the VolumeKdtree path's results at
    the class's own tolerance 4, bit-exact against the oracle, and that its error helpers agree with the oracle's."""
    vol = rm_like((32, 32, 64), seed=11)
    h = vr.HashedKdtree(vol.copy(), 64, 32, 32)
    assert h.tolerance == 4
    h.build()
    ref = oracle.OracleTree(vol.copy(), tolerance=4, max_epochs=5).build()
    assert np.array_equal(h.treeData, ref.tree) and list(h.distanceMap) == list(ref.distanceMap)
    dec = h.levelCut(h.treeDepth).cpu().numpy().reshape(vol.shape)
    assert np.array_equal(dec, ref.levelCut()) and h.queryDepth == ref.maxTreeDepth
    assert h.measureMaxError() == oracle.measure_max_error(ref.levelCut(), vol)
    assert h.numCollisions == 0


@pytest.mark.parametrize("shape", [(32, 32, 32), (16, 16, 8), (8, 16, 128)])
def test_midrange_range_stream_decode(vr, oracle, shape):
    """MidRangeTree's half-range stream decoded at full depth and at progressive cuts (the reference
    builds that stream but never decodes it, so the oracle is the progressive walk applied to tree_range /
    distanceMap_range); together with levelCut it bounds every voxel by [mid - range, mid + range]."""
    rng = np.random.default_rng(5 + shape[2])
    vol = rm_like(shape, seed=4) if shape[2] != 8 else rng.integers(0, 256, shape, dtype=np.uint8)
    z, y, x = shape
    ref = oracle.OracleTree(vol.copy(), tolerance=2, max_epochs=2, midrange=True, guarded=True).build()
    t = vr.MidRangeTree(vol.copy(), x, y, z)
    t.setMaxEpochs(2); t.setErrorTolerance(2)
    t.build()
    assert np.array_equal(t.tree_range, ref.tree_range)
    D = ref.origTreeDepth
    for cut in [None, D, D - 3, max(1, D - 7), 2]:
        got = t.levelCutRange(cut).cpu().numpy().reshape(shape)
        assert np.array_equal(got, ref.levelCutRange(cut)), cut
    mid = t.levelCut(D - 3).cpu().numpy().reshape(shape).astype(np.int32)
    assert np.array_equal(mid, ref.levelCutProgressive(D - 3))
    v = vr.VolumeKdtree(vol.copy(), x, y, z)
    v.build()
    with pytest.raises(vr.VrError):
        v._bs.decode_range()                           # not a MidRangeTree: VR_ERR_STATE


def test_encoder_out_of_memory_is_reported_every_time(vr):
    """A brickset whose stream buffers fit but whose encoder side buffers do not: build() must return
    VR_ERR_OOM -- on every call (a half-allocated set once launched kernels on null buffers the second time) --
    and a set that does fit must still work afterwards."""
    import torch
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    dims = (256, 256, 128)
    V = dims[0] * dims[1] * dims[2]
    B = int(free // (6 * V))                 # input V + create() ~2.4 V per brick fit; the encoder's ~5.6 V more do not
    vox = torch.zeros(B * V, dtype=torch.uint8, device="cuda")
    bs = vr.BrickSet(B, dims, 1, 2)
    for _ in range(2):
        with pytest.raises(vr.VrError) as ei:
            bs.build(vox)
        assert ei.value.status == -3        # VR_ERR_OOM
    del vox
    del bs
    torch.cuda.synchronize()
    small = vr.BrickSet(1, (32, 32, 32), 1, 2)
    v = np.random.default_rng(5).integers(0, 256, (32, 32, 32), dtype=np.uint8)
    small.build(v)
    assert small.info(0)["num_active_nodes"] > 0


@pytest.mark.parametrize("shape", [(12, 6, 5), (40, 16, 24), (48, 64, 96), (3, 1, 1), (7, 9, 2), (2, 2, 2048)])
def test_general_extents_match_oracle(vr, oracle, shape, tmp_path):
    """Extents that are not powers of two (R.cpp:151-162: unequal boxes, a split axis that differs from node to node,
    leaves of two cells or none, R.cpp:194-195 / 759-766) and axes beyond 1024: tree bytes, distanceMap, statistics
    and the decoded voxels -- full depth and progressive -- are the oracle's, also through a saved file."""
    rng = np.random.default_rng(sum(shape))
    z, y, x = shape
    for vol, tol, ep in ((rng.integers(0, 256, shape, dtype=np.uint8), 2, 2), (rm_like(shape), 1, 2), (rm_like(shape, 5), 0, 5)):
        ref, bs = check_case(vr, oracle, vol, tol, ep)
        D, M = ref.origTreeDepth, ref.maxTreeDepth
        for cut in sorted({0, 1, D // 2, D - 1, D, D + 2, M - 1}):
            if 0 <= cut < M:
                got = bs.decode(cut_depth=cut).cpu().numpy().reshape(shape)
                assert np.array_equal(got, ref.levelCutProgressive(cut)), cut
        p = str(tmp_path / "g.bin")
        bs.save(p)
        fs = vr.BrickSet.open(p)
        assert np.array_equal(fs.decode().cpu().numpy().reshape(shape), ref.levelCut())
    # MidRangeTree on the same geometry
    vol = rm_like(shape, 9)
    ref = oracle.OracleTree(vol.copy(), tolerance=1, max_epochs=2, guarded=True, midrange=True).build()
    ms = vr.BrickSet(1, (x, y, z), 1, 2, 2)
    ms.build(vol.copy())
    assert np.array_equal(ms.tree(0), ref.tree) and np.array_equal(ms.tree_range(0), ref.tree_range)
    assert np.array_equal(ms.decode().cpu().numpy().reshape(shape), ref.levelCut())


def test_general_extents_big_brick(vr, oracle):
    """256 x 256 x 96 (z not a power of two: D = 22, most z leaves span one cell, every third pair of them two)."""
    shape = (96, 256, 256)
    vol = rm_like(shape, 2)
    ref, bs = check_case(vr, oracle, vol, 1, 2)
    # batched: two different bricks of that shape in one set
    v2 = np.stack([vol, rm_like(shape, 4)])
    b2 = vr.BrickSet(2, (256, 256, 96), 1, 2).build(v2)
    dec = b2.decode().cpu().numpy().reshape(2, *shape)
    assert np.array_equal(dec[0], ref.levelCut())
    r1 = oracle.OracleTree(v2[1].copy(), tolerance=1, max_epochs=2).build()
    assert np.array_equal(b2.tree(1), r1.tree) and np.array_equal(dec[1], r1.levelCut())


@pytest.mark.parametrize("shape,tol,ep", [((128, 256, 256), 1, 2), ((16, 16, 256), 2, 3), ((8, 16, 128), 6, 5), ((16, 32, 64), 0, 2)])
def test_midrange_on_the_fused_kernels(vr, oracle, shape, tol, ep):
    """MidRangeTree through k_prune_emit12 / k_concat12 (second launch for the range stream, M.cpp:871-982) and the
    mid stream through k_decode_quad: both 2-bit streams, both distanceMaps, the 4-bit packing (M.cpp:1095-1128), the
    decoded voxels and the decoded half-range stream against the oracle -- at the bench brick size and on volumes with
    saturated ends (branches the table does not cover) and pruned boxes of every size."""
    rng = np.random.default_rng(shape[0] + tol)
    z, y, x = shape
    vols = [rm_like(shape, 7)] if shape[0] == 128 else [_mixed_volume(rng, shape) for _ in range(3)] + [rng.integers(0, 256, shape, dtype=np.uint8)]
    for vol in vols:
        ref = oracle.OracleTree(vol.copy(), tolerance=tol, max_epochs=ep, midrange=True, guarded=True).build()
        bs = vr.BrickSet(1, (x, y, z), tol, ep, 2)
        bs.build(vol.copy())
        assert bs.info(0)["num_active_nodes"] == ref.numActiveNodes and bs.info(0)["num_reverts"] == ref.numReverts
        assert list(bs.distance_map(0)) == list(ref.distanceMap) and list(bs.distance_map_range(0)) == list(ref.distanceMap_range)
        assert np.array_equal(bs.tree(0), ref.tree)
        assert np.array_equal(bs.tree_range(0), ref.tree_range)
        assert np.array_equal(bs.packed4(0), ref.convertToByteArray())
        assert np.array_equal(bs.decode().cpu().numpy().reshape(shape), ref.levelCut())
        D = ref.origTreeDepth
        for cut in (None, D, D - 6, 4):
            got = bs.decode_range(cut_depth=-1 if cut is None else cut).cpu().numpy().reshape(shape)
            assert np.array_equal(got, ref.levelCutRange(cut)), cut


@pytest.mark.parametrize("shape", [(40, 16, 24), (16, 32, 128), (96, 64, 48)])
def test_64_bit_token_offsets_small(vr, oracle, shape, monkeypatch):
    """Trees deeper than 28 levels (the reference's own 2048 x 2048 x 768 volume is 31 deep, main.cpp:242-251) can hold
    more than 2^32 tokens: the emitter then scans in 64 bits and the decode index is kept relative to its 4096-leaf
    block.  VRHIP_FORCE_IDX64 takes small bricks through exactly that code, where the oracle can check every byte."""
    monkeypatch.setenv("VRHIP_FORCE_IDX64", "1")
    rng = np.random.default_rng(shape[2])
    for vol, tol, ep in ((rng.integers(0, 256, shape, dtype=np.uint8), 1, 2), (rm_like(shape), 1, 2), (_mixed_volume(rng, shape), 4, 5)):
        ref, bs = check_case(vr, oracle, vol, tol, ep)
        D = ref.origTreeDepth
        for cut in (D - 7, D - 1, D + 3):
            assert np.array_equal(bs.decode(cut_depth=cut).cpu().numpy().reshape(shape), ref.levelCutProgressive(cut)), cut
    z, y, x = shape
    vol = rm_like(shape, 8)
    ref = oracle.OracleTree(vol.copy(), tolerance=1, max_epochs=2, guarded=True, midrange=True).build()
    ms = vr.BrickSet(1, (x, y, z), 1, 2, 2).build(vol.copy())
    assert np.array_equal(ms.tree(0), ref.tree) and np.array_equal(ms.tree_range(0), ref.tree_range)
    assert np.array_equal(ms.decode().cpu().numpy().reshape(shape), ref.levelCut())
    assert np.array_equal(ms.decode_range().cpu().numpy().reshape(shape), ref.levelCutRange(None))


@pytest.mark.parametrize("tol,ep", [(1, 1), (1, 2), (2, 3), (4, 2)])
def test_constant_blocks_skipped_by_the_level_loop(vr, oracle, monkeypatch, tol, ep):
    """SkipBlocks (kd_encode.hip): 16x16x16 boxes of one value each -- some reproduced exactly by depth D-3 (their
    neighbours hold the same value), most not -- next to noisy boxes.  The level loop leaves the exact ones alone at
    depths D-1 and D; stream, distances and statistics must not notice (oracle), nor differ from a build with the
    shortcut switched off."""
    rng = np.random.default_rng(77 + tol * 10 + ep)
    shape = (32, 64, 64)                  # D = 17: 32 blocks of 4096 leaves
    z, y, x = shape
    coarse = rng.integers(0, 256, (z // 16, y // 16, x // 16)).astype(np.uint8)
    coarse[:, :2, :2] = 90                # a 32 x 32 region of equal boxes: exact early, skipped
    coarse[0, 2:, 2:] = 255               # saturated boxes
    vol = np.repeat(np.repeat(np.repeat(coarse, 16, 0), 16, 1), 16, 2)
    noisy = rng.random((z // 16, y // 16, x // 16)) < 0.25
    mask = np.repeat(np.repeat(np.repeat(noisy, 16, 0), 16, 1), 16, 2)
    vol = np.where(mask, rng.integers(0, 256, shape), vol).astype(np.uint8)
    ref, bs = check_case(vr, oracle, vol, tol, ep)
    monkeypatch.setenv("VRHIP_NO_SKIP_BLOCKS", "1")
    plain = vr.BrickSet(1, (x, y, z), tol, ep)
    plain.build(vol.copy())
    monkeypatch.delenv("VRHIP_NO_SKIP_BLOCKS")
    assert np.array_equal(plain.tree(0), bs.tree(0))
    assert list(plain.distance_map(0)) == list(bs.distance_map(0))
    assert plain.info(0) == bs.info(0)


def test_midrange_range_decode_below_pruned_blocks(vr, oracle):
    """Regression (found by scratch fuzzing, 2 of 250 cases): a 4096-leaf block that the mid stream prunes as a whole
    keeps, in the half-range stream's code ARRAY, whatever the range level loop chose below its root; the range
    stream itself ends there (M.cpp:864-865).  The scalars decode_range starts from must stop at the pruned node."""
    cases = np.load(os.path.join(os.path.dirname(__file__), "golden", "midrange_range_decode_cases.npz"))
    for name, tol, ep in (("case53_tol5_ep2", 5, 2), ("case149_tol2_ep1", 2, 1)):
        vol = cases[name]
        z, y, x = vol.shape
        ref = oracle.OracleTree(vol.copy(), tolerance=tol, max_epochs=ep, midrange=True, guarded=True).build()
        bs = vr.BrickSet(1, (x, y, z), tol, ep, 2)
        bs.build(vol.copy())
        assert np.array_equal(bs.tree(0), ref.tree) and np.array_equal(bs.tree_range(0), ref.tree_range)
        D = ref.origTreeDepth
        for cut in (None, D, D - 3, D - 6, D - 7, D - 12, 4):
            got = bs.decode_range(cut_depth=-1 if cut is None else cut).cpu().numpy().reshape(vol.shape)
            assert np.array_equal(got, ref.levelCutRange(cut)), (name, cut)


def test_level_loop_concurrency_does_not_change_results(vr, oracle):
    """vr_brickset_set_concurrency: the level loops of 1 / 2 / 4 brick ranges side by side on internal streams give the
    same streams (and the oracle's), also with constant bricks at range boundaries and for repeated builds."""
    rng = np.random.default_rng(21)
    shape = (16, 32, 32)
    z, y, x = shape
    vols = []
    for i in range(70):
        k = i % 5
        vols.append(np.full(shape, (i * 37) & 255, np.uint8) if k == 4 else (rm_like(shape, i) if k == 3 else rng.integers(0, 256, shape, dtype=np.uint8)))
    stack = np.stack(vols)
    got = {}
    for n in (1, 2, 4, 3):
        bs = vr.BrickSet(len(vols), (x, y, z), 1, 2).set_concurrency(n)
        for _ in range(2):
            bs.build(stack)
        got[n] = [bs.tree(i).tobytes() for i in range(len(vols))], [tuple(bs.distance_map(i)) for i in range(len(vols))], \
            bs.decode().cpu().numpy().copy()
    for n in (2, 4, 3):
        assert got[n][0] == got[1][0] and got[n][1] == got[1][1] and np.array_equal(got[n][2], got[1][2]), n
    for i in (0, 4, 17, 34, 35, 52, 69):
        ref = oracle.OracleTree(vols[i].copy(), tolerance=1, max_epochs=2).build()
        assert got[2][0][i] == ref.tree.tobytes() and list(got[2][1][i]) == list(ref.distanceMap)
    with pytest.raises(vr.VrError):
        vr.BrickSet(1, (x, y, z), 1, 2).set_concurrency(5)


def test_leafless_builds_across_epoch_settings_on_one_handle(vr, oracle):
    """A fused build stores nothing of the leaf level and recomputes it in the prune from the distances the level loop
    ended with (Ctrl::finalReconDist / finalCodesDist, which differ after reverted epochs); an epoch one step beside
    the previous fill reads that fill's central-difference partials instead of running a fill.  One handle through
    setMaxEpochs 5 -> 0 -> 2 -> 1 (0 switches the handle to the storing mode and back: the encoder's arrays are made
    again) and tolerances 0 / 1 / 3, volumes with and without reverts: every build equals the oracle's."""
    rng = np.random.default_rng(77)
    vols = [rng.integers(0, 256, (32, 32, 32), dtype=np.uint8), oracle.gen_sphere(32, 7), rm_like((16, 32, 64), 5),
            (rng.integers(0, 4, (32, 32, 32)) + 1).astype(np.uint8)]           # the last one: small values, clamps at 0 matter
    reverts = 0
    for vol in vols:
        z, y, x = vol.shape
        bs = vr.BrickSet(1, (x, y, z), 1, 5)
        for ep, tol in ((5, 0), (0, 1), (2, 1), (1, 3), (5, 1), (0, 0), (3, 0)):
            bs.set_max_epochs(ep)
            bs.set_error_tolerance(tol)
            bs.build(vol.copy())
            ref = oracle.OracleTree(vol.copy(), tolerance=tol, max_epochs=ep).build()
            reverts += ref.numReverts
            info = bs.info(0)
            assert info["num_active_nodes"] == ref.numActiveNodes and info["num_reverts"] == ref.numReverts, (ep, tol)
            assert list(bs.distance_map(0)) == list(ref.distanceMap), (ep, tol)
            assert np.array_equal(bs.tree(0), ref.tree), (ep, tol)
            assert np.array_equal(bs.decode().cpu().numpy().reshape(z, y, x), ref.levelCut()), (ep, tol)
    assert reverts >= 1
