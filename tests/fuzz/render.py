"""render fuzz: random volumes / cameras / modes: HIP vs the C restatement (2e-3), and skip grid on/off bit-identical"""
import sys, time, math
from common import gen, ROOT
import __graft_entry__ as g
g.build()
import numpy as np, torch
import volumerenderer_amd as vr
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0; t0 = time.time(); worst = 0.0
for it in range(n):
    X, Y, Z = [int(v) for v in rng.choice([16, 24, 32, 40, 48, 64, 128], 3)]
    kind = it % 4
    if kind == 0: vol = rng.integers(0, 256, (Z, Y, X), dtype=np.uint8)
    elif kind == 1:
        zz, yy, xx = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
        r = np.sqrt((xx - X / 2) ** 2 + (yy - Y / 2) ** 2 + (zz - Z / 2) ** 2)
        vol = np.clip(220 - 255 * r / (min(X, Y, Z) / 2), 0, 255).astype(np.uint8)
    elif kind == 2:
        vol = np.zeros((Z, Y, X), np.uint8)
        a = [sorted(rng.integers(0, s + 1, 2)) for s in (Z, Y, X)]
        vol[a[0][0]:a[0][1], a[1][0]:a[1][1], a[2][0]:a[2][1]] = rng.integers(1, 256)
        m = rng.random((Z, Y, X)) < 0.02
        vol[m] = rng.integers(0, 256, int(m.sum()))
    else:
        vol = np.clip(rng.normal(90, 60, (Z, Y, X)), 0, 255).astype(np.uint8)
    th, ph = rng.uniform(0, 2 * math.pi), rng.uniform(-1.2, 1.2)
    rad = rng.choice([0.2, 0.6, 0.9, 1.4])
    pos = (rad * math.cos(ph) * math.sin(th), rad * math.sin(ph), -rad * math.cos(ph) * math.cos(th))
    jit = rng.normal(0, 0.15, 3)
    front = tuple(-p + j for p, j in zip(pos, jit))
    mode = int(rng.integers(0, 2)); iso = float(rng.choice([2, 40, 100, 180, 250])) / 255.0
    w, h = int(rng.choice([96, 160, 200])), int(rng.choice([64, 96, 120]))
    sd = (256, 256, 128) if rng.random() < 0.5 else (X, Y, Z)
    cg, co = vr.default_camera(), O.default_camera()
    fov = float(rng.choice([35.0, 50.0, 70.0]))
    for c in (cg, co):
        c.pos[:] = pos; c.front[:] = front; c.fov_deg = fov
    Pg, Po = vr.default_params(w, h, sd, mode, iso), O.default_params(w, h, sd, mode, iso)
    dvol = torch.from_numpy(vol).cuda().reshape(-1)
    got = vr.raycast(dvol, (X, Y, Z), cg, Pg).cpu().numpy()
    want = O.render(vol, co, Po)
    dd = np.abs(got - want).max(axis=2)
    frac = float((dd > 2e-3).mean())
    d = float(np.median(dd[dd > 2e-3])) if frac > 0 else float(dd.max()); worst = max(worst, frac)
    cell = int(rng.choice([4, 8]))
    grid = vr.build_skip_grid(dvol, (X, Y, Z), cell)
    vr.use_skip_grid(Pg, grid, cell)
    sk = vr.raycast(dvol, (X, Y, Z), cg, Pg).cpu().numpy()
    ok = frac <= 2e-3 and np.array_equal(got, sk)
    if frac > 0:
        # are the differing pixels on the silhouette (a neighbour in `want` is background while the pixel is not, or v.v.)?
        bg = (want[..., 3] == 0) if mode == 0 else (np.abs(want[..., :3] - want[0, 0, :3]).max(axis=2) == 0)
        edge = np.zeros_like(bg)
        edge[1:, :] |= bg[1:, :] != bg[:-1, :]; edge[:-1, :] |= bg[1:, :] != bg[:-1, :]
        edge[:, 1:] |= bg[:, 1:] != bg[:, :-1]; edge[:, :-1] |= bg[:, 1:] != bg[:, :-1]
        onedge = float(edge[dd > 2e-3].mean())
        print("  case", it, "mode", mode, "differing pixels", int((dd > 2e-3).sum()), "of", dd.size, "on silhouette %.2f" % onedge, flush=True)
    if not ok:
        bad += 1
        print("MISMATCH", it, (X, Y, Z), kind, mode, iso, pos, "frac %.2e" % frac, "skip identical", np.array_equal(got, sk), flush=True)
print("cases", n, "mismatches", bad, "worst fraction %.2e" % worst, "%.1f s" % (time.time() - t0))
