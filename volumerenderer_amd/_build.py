"""Builds volumerenderer_amd/libvrhip.so with hipcc for gfx950 (in-tree, no JIT cache)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvrhip.so")
SOURCES = ["kd_encode.hip", "kd_decode.hip", "raymarch.hip", "capi.hip", "compositor.hip"]
HEADERS = ["kd_common.h", "brickset.h", os.path.join("..", "..", "include", "vrhip.h")]
# -ffp-contract=off: the gradient-descent control kernel and the ray marcher must round
# exactly like the reference's scalar code (no FMA contraction).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-value", "-Wno-unused-result", "-ldl"]
# experiments: extra -D switches for tuning runs (part of the staleness hash, so a change rebuilds)
FLAGS += os.environ.get("VRHIP_EXTRA_HIPCC_FLAGS", "").split()


STAMP = LIB + ".srchash"


def _source_hash():
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in SOURCES + HEADERS:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()


def _stale():
    # content hash, not mtimes: the snapshot shipped to a GPU box does not keep mtimes, and a
    # needless rebuild there would exec hipcc from a process that may already hold the GPU
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    return open(STAMP).read().strip() != _source_hash()


def build(force=False, verbose=False):
    if not (force or _stale()):
        return LIB
    import fcntl
    # several ranks of one node may get here together: one builds, the others wait and re-check
    with open(LIB + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if force or _stale():
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            tmp = LIB + ".tmp.%d" % os.getpid()
            cmd = [hipcc] + FLAGS + ["-o", tmp] + SOURCES
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd, cwd=CSRC)
            os.replace(tmp, LIB)
            with open(STAMP, "w") as f:
                f.write(_source_hash())
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
