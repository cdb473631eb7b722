"""Volume generators shared by the fuzz drivers in this directory (not collected by pytest: run them by hand on a GPU
box, e.g. `python tests/fuzz/codec.py 1200 777`; every driver prints `mismatches N`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np


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

