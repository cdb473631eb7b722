"""The header-only C++ facade (include/vrhip/*.hpp): a host program written against the
reference's class names compiles with plain g++ (CPU check) and, on the GPU box, produces the
same tree file as the CPU oracle (parity through the C++ boundary)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compile(tmp_path):
    import __graft_entry__ as g
    g.build()
    exe = str(tmp_path / "main_pipeline")
    lib = os.path.join(ROOT, "volumerenderer_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "main_pipeline.cpp"), "-L" + lib, "-lvrhip",
                           "-Wl,-rpath," + lib, "-o", exe])
    return exe


def test_facade_compiles_and_fails_loudly_without_gpu(tmp_path):
    exe = _compile(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    d = tmp_path / "bricks"
    d.mkdir()
    r = subprocess.run([exe, str(d)], capture_output=True, text=True)
    assert r.returncode != 0                      # no CPU fallback: the facade throws VR_ERR_NO_DEVICE
    assert "no usable HIP device" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_facade_pipeline_matches_oracle(tmp_path, oracle):
    exe = _compile(tmp_path)
    d = tmp_path / "bricks"
    d.mkdir()
    r = subprocess.run([exe, str(d)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "reopen: voxels equal 1" in r.stdout
    assert "hashed: max error" in r.stdout          # HashedKdtree.hpp: the interface over the VolumeKdtree path
    # rebuild the same volume from the brick files the program wrote and run the oracle on it
    X, Y, Z = 64, 64, 32
    vol = np.zeros((2 * Z, 2 * Y, 2 * X), np.uint8)
    for b in range(8):
        i, j, k = b % 2, (b // 2) % 2, b // 4
        brick = np.fromfile(str(d / ("d_273_%d" % b)), np.uint8).reshape(Z, Y, X)
        vol[k * Z:(k + 1) * Z, j * Y:(j + 1) * Y, i * X:(i + 1) * X] = brick       # VolumeReader.h:184-198
    ref = oracle.OracleTree(vol.copy(), tolerance=1, max_epochs=2).build()
    p = str(tmp_path / "ref.bin")
    ref.save(p)
    assert open(p, "rb").read() == open(str(d / "tree_1tolerance.bin"), "rb").read()
    dec = ref.levelCut()
    assert ("MAX ERROR: %d" % oracle.measure_max_error(dec, vol)) in r.stdout
