"""Pins the CPU oracle (oracle/kdtree_oracle.c) to outputs of the reference itself.

The only reference outputs that exist for this path are the known-answer values
the survey recorded and the tree file the
reference's own save() wrote for the 16^3 case; see tests/golden/
survey_known_answers.json for provenance.  Bit-exact."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KA = json.load(open(os.path.join(GOLD, "survey_known_answers.json")))


@pytest.mark.parametrize("case", KA["volume_kdtree"], ids=lambda c: "n%d" % c["n"])
def test_volume_kdtree_known_answers(oracle, case, tmp_path):
    O = oracle
    vol = O.gen_sphere(case["n"], case["noise_mask"])
    t = O.OracleTree(vol.copy(), tolerance=case["tolerance"], max_epochs=case["max_epochs"]).build()
    if "origTreeDepth" in case:
        assert t.origTreeDepth == case["origTreeDepth"]
        assert t.maxTreeDepth == case["maxTreeDepth"]
    if "numActiveNodes" in case:
        assert t.numActiveNodes == case["numActiveNodes"]
        assert len(t.tree) == case["tree_bytes"]
    if "distanceMap" in case:
        assert list(map(int, t.distanceMap)) == case["distanceMap"]
    assert "%016x" % O.fnv1a64(t.tree) == case["tree_fnv"]
    out = t.levelCut()
    if "voxels_fnv" in case:
        assert "%016x" % O.fnv1a64(out) == case["voxels_fnv"]
    if "decoded_max_error" in case:
        assert O.measure_max_error(out, vol) == case["decoded_max_error"]
    if "saved_file" in case:
        # byte-identical to the file the reference's save() wrote (R.cpp:521-552)
        p = str(tmp_path / "t.bin")
        t.save(p)
        ref = open(os.path.join(GOLD, case["saved_file"]), "rb").read()
        assert len(ref) == case["saved_file_bytes"]
        assert open(p, "rb").read() == ref
        # open() of the reference-written file: 8-byte over-allocation (C-6), same voxels
        u = O.OracleTree.open(os.path.join(GOLD, case["saved_file"]))
        assert len(u.tree) == case["tree_bytes"] + 8
        assert u.numActiveNodes == case["numActiveNodes"]
        assert np.array_equal(u.levelCut(), out)


@pytest.mark.parametrize("case", KA["mid_range_tree"], ids=lambda c: "n%d" % c["n"])
def test_mid_range_tree_known_answers(oracle, case):
    O = oracle
    vol = O.gen_sphere(case["n"], case["noise_mask"])
    t = O.OracleTree(vol.copy(), tolerance=case["tolerance"], max_epochs=case["max_epochs"],
                     midrange=True, guarded=True).build()
    assert t.numActiveNodes == case["numActiveNodes"]
    assert "%016x" % O.fnv1a64(t.tree) == case["tree_fnv"]
    assert "%016x" % O.fnv1a64(t.tree_range) == case["tree_range_fnv"]
    packed = t.convertToByteArray()
    assert len(packed) == case["packed_bytes"]
    assert "%016x" % O.fnv1a64(packed) == case["packed_fnv"]
    assert O.measure_max_error(t.levelCut(), vol) == case["decoded_max_error"]


def test_guarded_variant_is_output_identical(oracle):
    """VolumeKdtree.cpp:333 only skips work whose result is never used."""
    O = oracle
    vol = O.gen_sphere(32, 7)
    for ep in (1, 2, 5):
        a = O.OracleTree(vol.copy(), tolerance=2, max_epochs=ep).build()
        b = O.OracleTree(vol.copy(), tolerance=2, max_epochs=ep, guarded=True).build()
        assert np.array_equal(a.tree, b.tree) and np.array_equal(a.distanceMap, b.distanceMap)


def test_oracle_properties(oracle):
    O = oracle
    rng = np.random.default_rng(7)
    # all-zero volume: a single pruned root token
    z = np.zeros((8, 8, 8), np.uint8)
    t = O.OracleTree(z.copy(), tolerance=1, max_epochs=2).build()
    assert t.numActiveNodes == 1 and list(t.tree) == [3]
    assert np.array_equal(t.levelCut(), z)
    # constant volume: root code 1 + two pruned children
    c = np.full((8, 8, 8), 37, np.uint8)
    t = O.OracleTree(c.copy(), tolerance=1, max_epochs=2).build()
    assert t.numActiveNodes == 3 and int(t.distanceMap[0]) == 37
    assert np.array_equal(t.levelCut(), c)
    # maxEpochs == 1: no revert possible -> decoded error <= tolerance
    for tol in (0, 1, 3, 6):
        v = rng.integers(0, 256, (16, 16, 8), dtype=np.uint8)
        t = O.OracleTree(v.copy(), tolerance=tol, max_epochs=1).build()
        assert t.numReverts == 0
        assert O.measure_max_error(t.levelCut(), v) <= tol
        assert t.zeroRunRewrites == 0
    # non power-of-two extents run (C-10): decode is well formed
    v = rng.integers(0, 256, (5, 6, 12), dtype=np.uint8)
    t = O.OracleTree(v.copy(), tolerance=1, max_epochs=2).build()
    t.levelCut()


def test_encode_node_closed_form_exhaustive():
    """The packed HIP kernels (kd_common.h enc_pair*) use a reduced form of encodeNode (R.cpp:457-502):
    the candidate on the far side of the parent never beats "keep" and the clamp only shortens the
    step.  Checked here over every (truth, parent, distance)."""
    t = np.arange(256, dtype=np.int32)[:, None, None]
    p = np.arange(256, dtype=np.int32)[None, :, None]
    d = np.arange(256, dtype=np.int32)[None, None, :]
    none = np.abs(p - t) + 0 * d
    add = np.minimum(p + d, 255); ae = np.abs(add - t)
    sub = np.maximum(p - d, 0); se = np.abs(sub - t)
    m = np.minimum(np.minimum(none, ae), se)
    code = np.where(m == none, 0, np.where(m == ae, 1, 2))
    recon = np.where(m == none, p + 0 * d, np.where(m == ae, add, sub))
    pd = np.abs(t - p) + 0 * d
    up = (t > p) + 0 * d
    h = np.where(up, 255 - t, t) + 0 * d
    x = np.minimum(d - pd, h)
    take = np.abs(x) < pd
    assert np.array_equal(m, np.minimum(pd, np.abs(x)))
    assert np.array_equal(code, np.where(take, np.where(up, 1, 2), 0))
    assert np.array_equal(recon, np.where(take, np.where(up, t + x, t - x), p + 0 * d))


def test_midrange_file_layout(oracle, tmp_path):
    """MidRangeTree::save (M.cpp:753-785): 88-byte header, both distance maps, both streams; and the
    reference reader's defect (M.cpp:815: three of four int64 fields subtracted, then halved)."""
    vol = oracle.gen_sphere(16, 7)
    t = oracle.OracleTree(vol.copy(), tolerance=1, max_epochs=1, midrange=True, guarded=True).build()
    p = str(tmp_path / "mr.bin")
    t.save(p)
    raw = open(p, "rb").read()
    T, m = len(t.tree), t.maxTreeDepth + 1
    assert len(raw) == 88 + 2 * m + 2 * T
    assert raw[88:88 + m] == bytes(t.distanceMap) and raw[88 + m:88 + 2 * m] == bytes(t.distanceMap_range)
    assert raw[88 + 2 * m:88 + 2 * m + T] == t.tree.tobytes() and raw[88 + 2 * m + T:] == t.tree_range.tobytes()
    r = oracle.OracleTree.open_midrange(p)
    assert len(r.tree) == T + 4                                       # over-sized by 8 / 2
    assert r.tree[:T].tobytes() == t.tree.tobytes()                   # the mid stream survives ...
    assert r.tree[T:].tobytes() == t.tree_range[:4].tobytes()         # ... followed by the range stream's head
    assert r.tree_range[:T - 4].tobytes() == t.tree_range[4:].tobytes()   # and the range stream comes back shifted
