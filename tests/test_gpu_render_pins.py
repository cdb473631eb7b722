"""Independent pins for the ray-march kernels.

The C restatement in oracle/raymarch_oracle.c and csrc/raymarch.hip were written side by side, so agreement between
them cannot catch a shared misreading of the shaders.  The checks here come from other directions:

  * closed forms -- trilinear filtering of a linear ramp is the ramp itself; the iso-surface of a ramp along z has the
    normal (0, 0, -1), so its Blinn-Phong colour (isosurface.frag:64-75) is a function of the ray direction alone;
    a volume that only varies along z is a 1-D profile along every ray;
  * a NumPy float64 marcher written from the GLSL text (raycaster.vert:10-21, raycaster.frag:18-86,
    isosurface.frag:23-159, main.cpp:396-397 for the matrices), vectorised over rays, used pixel by pixel on sampled
    rows of a 1080p frame;
  * bit-identity of frames rendered with and without the empty-space skip grid.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vr():
    import __graft_entry__ as g
    g.build()
    import volumerenderer_amd as vr
    return vr


# ---- NumPy restatement (float64) -------------------------------------------------------------------------------------
def rays(pos, front, up, fov_deg, W, H, rows=None, near=0.1, far=100.0):
    """Per pixel centre: (covered, vUV, dir).  gl_Position = P*V*M*v with M = identity, V = lookAt, P = perspectiveFov
    (main.cpp:396-397): a pixel's ray through the unit cube [-0.5, 0.5]^3; the nearest cube-surface point in front of
    the near plane wins (depth test LESS, no culling, main.cpp:367-369); vUV = vertex + 0.5 (raycaster.vert:17)."""
    f = np.asarray(front, float); f /= np.linalg.norm(f)
    s = np.cross(f, np.asarray(up, float)); s /= np.linalg.norm(s)
    u = np.cross(s, f)
    ty = math.tan(math.radians(fov_deg) / 2); tx = ty * W / H
    ys = np.arange(H) if rows is None else np.asarray(rows)
    px, py = np.meshgrid(np.arange(W), ys)
    nx = 2 * (px + 0.5) / W - 1
    ny = 1 - 2 * (py + 0.5) / H
    d = f[None, None, :] + nx[..., None] * tx * s + ny[..., None] * ty * u          # view-space z = 1 along f
    cp = np.asarray(pos, float)
    with np.errstate(divide="ignore", invalid="ignore"):
        lo = (-0.5 - cp) / d
        hi = (0.5 - cp) / d
    t0 = np.minimum(lo, hi); t1 = np.maximum(lo, hi)
    par = d == 0
    t0 = np.where(par, -np.inf, t0); t1 = np.where(par, np.inf, t1)
    miss = (par & ((cp < -0.5) | (cp > 0.5))).any(-1)
    tn, tf = t0.max(-1), t1.min(-1)
    th = np.where(tn >= near, tn, tf)
    covered = ~miss & (tn <= tf) & (th >= near) & (th <= far)
    vuv = cp + th[..., None] * d + 0.5
    g = vuv - 0.5 - cp
    g /= np.linalg.norm(g, axis=-1, keepdims=True)
    return covered, vuv, g


def tex3d(vol, p):
    """texture(volume, p).r: R8 normalised, GL_LINEAR, clamp to edge (VolumeReader.h:120-127)."""
    Z, Y, X = vol.shape
    out = None
    c = [p[..., 0] * X - 0.5, p[..., 1] * Y - 0.5, p[..., 2] * Z - 0.5]
    i0 = [np.floor(v).astype(np.int64) for v in c]
    fr = [v - i for v, i in zip(c, i0)]
    n = [X, Y, Z]
    a = [np.clip(i, 0, m - 1) for i, m in zip(i0, n)]
    b = [np.clip(i + 1, 0, m - 1) for i, m in zip(i0, n)]
    v = vol.astype(np.float64) / 255.0
    def at(ix, iy, iz): return v[iz, iy, ix]
    c00 = at(a[0], a[1], a[2]) * (1 - fr[0]) + at(b[0], a[1], a[2]) * fr[0]
    c10 = at(a[0], b[1], a[2]) * (1 - fr[0]) + at(b[0], b[1], a[2]) * fr[0]
    c01 = at(a[0], a[1], b[2]) * (1 - fr[0]) + at(b[0], a[1], b[2]) * fr[0]
    c11 = at(a[0], b[1], b[2]) * (1 - fr[0]) + at(b[0], b[1], b[2]) * fr[0]
    c0 = c00 * (1 - fr[1]) + c10 * fr[1]
    c1 = c01 * (1 - fr[1]) + c11 * fr[1]
    return c0 * (1 - fr[2]) + c1 * fr[2]


def inside(p):
    """stop = dot(sign(p - 0), sign(1 - p)) < 3 (raycaster.frag:51): strictly inside on every axis."""
    return ((p > 0) & (p < 1)).all(-1)


def march_composite(vol, covered, vuv, g, step, max_samples=300):
    """raycaster.frag:33-85 per ray."""
    st = g * np.asarray(step, float)
    pos = vuv.copy()
    rgb = np.zeros(covered.shape); A = np.zeros(covered.shape)
    live = covered.copy()
    for _ in range(max_samples):
        pos = pos + st
        live = live & inside(pos)
        if not live.any():
            break
        s = tex3d(vol, np.where(live[..., None], pos, 0.5))
        pa = s - s * A
        rgb = np.where(live, rgb + pa * s, rgb)
        A = np.where(live, A + 0.6 * pa, A)
        live = live & ~(A > 0.99)
    out = np.ones(covered.shape + (4,))
    out[..., 0] = np.where(covered, 1 - rgb, 1.0)
    out[..., 1] = out[..., 0]
    out[..., 3] = np.where(covered, A, 1.0)
    return out


def march_iso(vol, covered, vuv, g, step, iso, max_samples=300):
    """isosurface.frag:77-159 per ray (Bisection :23-42, GetGradient :47-62, PhongLighting :64-75)."""
    st = g * np.asarray(step, float)
    pos = vuv.copy()
    col = np.ones(covered.shape + (4,))
    live = covered.copy()
    for _ in range(max_samples):
        pos = pos + st
        live = live & inside(pos)
        if not live.any():
            break
        safe = np.where(live[..., None], pos, 0.5)
        s1, s2 = tex3d(vol, safe), tex3d(vol, safe + st)
        hit = live & (s1 - iso < 0) & (s2 - iso >= 0)
        if hit.any():
            l, r = pos.copy(), pos + st
            for _b in range(4):
                m = (l + r) / 2
                below = tex3d(vol, np.where(hit[..., None], m, 0.5)) < iso
                l = np.where(below[..., None], m, l)
                r = np.where(below[..., None], r, m)
            tc = np.where(hit[..., None], (l + r) / 2, 0.5)
            D = 0.01
            N = np.stack([(tex3d(vol, tc - [D, 0, 0]) - tex3d(vol, tc + [D, 0, 0])) / 2,
                          (tex3d(vol, tc - [0, D, 0]) - tex3d(vol, tc + [0, D, 0])) / 2,
                          (tex3d(vol, tc - [0, 0, D]) - tex3d(vol, tc + [0, 0, D])) / 2], -1)
            nl = np.linalg.norm(N, axis=-1, keepdims=True)
            N = np.where(nl > 0, N / np.where(nl > 0, nl, 1), 0.0)
            V = -g
            diff = np.maximum((V * N).sum(-1), 0)
            Hh = V + V
            Hh = Hh / np.linalg.norm(Hh, axis=-1, keepdims=True)
            spec = np.maximum(1e-5, (Hh * N).sum(-1)) ** 250
            shade = np.minimum(1.0, diff[..., None] * np.array([0.39, 0.58, 0.93]) + spec[..., None])
            col[..., :3] = np.where(hit[..., None], shade, col[..., :3])
            live = live & ~hit
    return col


def _cam(vr, pos, front, fov=50.0):
    c = vr.default_camera()
    c.pos[:] = pos; c.front[:] = front; c.fov_deg = fov
    return c


# ---- closed forms ----------------------------------------------------------------------------------------------------
def test_trilinear_of_a_ramp_is_the_ramp(vr):
    """One sample per ray (max_samples = 1): A = 0.6 s (raycaster.frag:72) reveals the fetched value.  For
    v = 2x + 3y + z the trilinear fetch at p is (2 X(p) + 3 Y(p) + Z(p)) / 255 with X(p) = clamp(p.x * GX - 0.5, 0, GX-1)
    (texel centres at (i + 0.5) / G, clamp to edge) -- whatever the eight taps are."""
    X, Y, Z = 48, 24, 40
    z, y, x = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    vol = (2 * x + 3 * y + z).astype(np.uint8)
    assert (2 * x + 3 * y + z).max() <= 255
    W, H = 96, 72
    for pos, front in (((0.2, -0.3, -0.9), (-0.2, 0.3, 1.0)), ((0.9, 0.8, 0.7), (-1.0, -0.9, -0.8))):
        cam = _cam(vr, pos, front)
        P = vr.default_params(W, H, (X, Y, Z))
        P.max_samples = 1
        got = vr.raycast(vol, (X, Y, Z), cam, P).cpu().numpy().astype(np.float64)
        cov, vuv, g = rays(pos, front, (0, 1, 0), 50.0, W, H)
        p = vuv + g * np.array([1 / X, 1 / Y, 1 / Z])
        ok = cov & inside(p)
        c = [np.clip(p[..., k] * n - 0.5, 0, n - 1) for k, n in enumerate((X, Y, Z))]
        want = (2 * c[0] + 3 * c[1] + c[2]) / 255.0
        assert ok.sum() > 500
        assert np.abs(got[..., 3][ok] / 0.6 - want[ok]).max() < 2e-6
        assert np.abs((1 - got[..., 0])[ok] - want[ok] ** 2).max() < 2e-6          # rgb = s * s (raycaster.frag:70)
        assert (got[~cov] == 1.0).all()                                          # clear colour where the cube is not


def test_compositing_of_a_z_profile_in_closed_form(vr):
    """A volume that varies only along z: along any ray the fetched values are the 1-D linear interpolation of the
    profile at z_i = (vUV.z + i * step.z * dir.z), so T <- T (1 - 0.6 s), C <- C + T s^2 (raycaster.frag:69-72) can be
    run per ray from the profile alone, including the 300-sample cap, the strict inside test and the alpha > 0.99 exit."""
    rng = np.random.default_rng(5)
    X, Y, Z = 16, 16, 64
    prof = rng.integers(0, 200, Z).astype(np.uint8)
    prof[20:30] = 0                                   # an empty slab inside
    vol = np.broadcast_to(prof[:, None, None], (Z, Y, X)).copy()
    W, H = 80, 60
    pos, front = (0.25, 0.15, -0.8), (-0.3, -0.2, 1.0)
    cam = _cam(vr, pos, front)
    P = vr.default_params(W, H, (256, 256, 128))       # the reference's step: BRICK_DIM, not the volume's dims (main.cpp:330-331)
    got = vr.raycast(vol, (X, Y, Z), cam, P).cpu().numpy().astype(np.float64)
    cov, vuv, g = rays(pos, front, (0, 1, 0), 50.0, W, H)
    st = g * np.array([1 / 256, 1 / 256, 1 / 128])
    p = vuv.copy()
    A = np.zeros(cov.shape); C = np.zeros(cov.shape); live = cov.copy()
    v = prof.astype(np.float64) / 255
    for _ in range(300):
        p = p + st
        live &= inside(p)
        zc = p[..., 2] * Z - 0.5
        z0 = np.floor(zc).astype(int); fz = zc - z0
        s = v[np.clip(z0, 0, Z - 1)] * (1 - fz) + v[np.clip(z0 + 1, 0, Z - 1)] * fz
        pa = s - s * A
        C = np.where(live, C + pa * s, C)
        A = np.where(live, A + 0.6 * pa, A)
        live &= ~(A > 0.99)
    assert cov.sum() > 1000
    assert np.abs(got[..., 3][cov] - A[cov]).max() < 2e-5
    assert np.abs((1 - got[..., 0])[cov] - C[cov]).max() < 2e-5
    assert (got[..., 2] == 1.0).all()                   # b = 255 clamps to 1 (raycaster.frag:84)


def test_isosurface_of_a_ramp_in_closed_form(vr):
    """v = 3 z: the first upward crossing of iso, refined by four bisections, lies on the plane f = iso within
    step / 32; the central-difference gradient (f(-d) - f(+d)) / 2 points along -z whatever the hit position is, so
    N = (0, 0, -1), L = V = H = -dir and the pixel is min(1, max(dir.z, 0) * (0.39, 0.58, 0.93) + max(1e-5, dir.z)^250)
    (isosurface.frag:64-75, 142-155) wherever the ray reaches the plane inside the cube."""
    X, Y, Z = 32, 32, 64
    vol = np.broadcast_to((3 * np.arange(Z))[:, None, None], (Z, Y, X)).astype(np.uint8).copy()
    iso = 0.4
    W, H = 96, 64
    pos, front = (0.1, -0.05, -0.9), (-0.1, 0.05, 1.0)
    cam = _cam(vr, pos, front)
    P = vr.default_params(W, H, (X, Y, Z), 1, iso)
    got = vr.raycast(vol, (X, Y, Z), cam, P).cpu().numpy().astype(np.float64)
    cov, vuv, g = rays(pos, front, (0, 1, 0), 50.0, W, H)
    # where f = iso: 3 (z* Z - 0.5) / 255 = iso
    zstar = (iso * 255 / 3 + 0.5) / Z
    # the march advances by dir (*) step_size, component-wise (raycaster.frag:31): in texture space that is not the
    # direction of dir unless the volume is a cube
    st = g * np.array([1 / X, 1 / Y, 1 / Z])
    n = (zstar - vuv[..., 2]) / st[..., 2]                    # steps to the plane
    hitp = vuv + n[..., None] * st
    reach = cov & (g[..., 2] > 0) & (vuv[..., 2] < zstar - 2 / Z) & (n < 295) & ((hitp[..., :2] > 0.02) & (hitp[..., :2] < 0.98)).all(-1)
    dz = g[..., 2]
    want = np.minimum(1.0, np.maximum(dz, 0)[..., None] * np.array([0.39, 0.58, 0.93]) + np.maximum(1e-5, dz)[..., None] ** 250)
    assert reach.sum() > 800
    assert np.abs(got[..., :3][reach] - want[reach]).max() < 5e-5
    assert (got[..., 3][reach] == 1.0).all()
    # and the NumPy marcher (hit search, bisection and gradient through the sampler) says the same everywhere
    ref = march_iso(vol, cov, vuv, g, (1 / X, 1 / Y, 1 / Z), iso)
    assert np.abs(got - ref).max() < 2e-4


# ---- pixel by pixel against the NumPy marcher ------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [0, 1])
def test_random_volume_matches_numpy_marcher(vr, mode):
    rng = np.random.default_rng(11 + mode)
    vol = rng.integers(0, 256, (24, 20, 28), dtype=np.uint8)
    vol[8:16] //= 8                                     # a dim slab: long rays, no early exit
    W, H = 72, 54
    pos, front = (0.7, 0.5, -0.8), (-0.6, -0.45, 0.9)
    cam = _cam(vr, pos, front)
    P = vr.default_params(W, H, (256, 256, 128), mode, 0.45)
    got = vr.raycast(vol, (28, 20, 24), cam, P).cpu().numpy().astype(np.float64)
    cov, vuv, g = rays(pos, front, (0, 1, 0), 50.0, W, H)
    step = (1 / 256, 1 / 256, 1 / 128)
    ref = march_composite(vol, cov, vuv, g, step) if mode == 0 else march_iso(vol, cov, vuv, g, step, 0.45)
    d = np.abs(got - ref)
    # float32 vs float64: a hit decided by a sample within rounding of iso may land one step apart on a few pixels
    assert (d > 2e-3).mean() <= (0.0 if mode == 0 else 0.003), float((d > 2e-3).mean())
    assert np.median(d) < 1e-5


def test_1080p_frame_pixel_by_pixel_on_sampled_rows(vr, oracle):
    """BASELINE config 2 shape: a decoded brick, a 1920 x 1080 frame, every pixel of eight rows against the marcher."""
    vol = oracle.gen_sphere(128, 7)
    t = vr.VolumeKdtree(vol.copy(), 128, 128, 128)
    t.setMaxEpochs(2); t.setErrorTolerance(1)
    t.build()
    dec = t.levelCut(t.maxTreeDepth)
    host = dec.cpu().numpy().reshape(128, 128, 128)
    W, H = 1920, 1080
    rows = [3, 200, 411, 539, 540, 700, 901, 1076]
    for mode, iso in ((0, 0.0), (1, 60 / 255.0)):
        cam, P = vr.default_camera(), vr.default_params(W, H, (128, 128, 128), mode, iso)
        img = vr.raycast(dec, (128, 128, 128), cam, P).cpu().numpy().astype(np.float64)
        cov, vuv, g = rays((0, 0, -0.75), (0, 0, 1), (0, 1, 0), 50.0, W, H, rows=rows)
        step = (1 / 128,) * 3
        ref = march_composite(host, cov, vuv, g, step) if mode == 0 else march_iso(host, cov, vuv, g, step, iso)
        d = np.abs(img[rows] - ref)
        assert (d > 2e-3).mean() < (1e-4 if mode == 0 else 0.003), (mode, float((d > 2e-3).mean()))
        assert cov.any()


# ---- empty-space skipping --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cell", [4, 8])
def test_skip_grid_frames_are_bit_identical(vr, cell):
    """Frames with the (min, max) skip grid attached equal the frames without it bit for bit, in both shaders, from
    several cameras (incl. one inside the cube): a skipped sample is one whose contribution is exactly nothing."""
    import torch
    rng = np.random.default_rng(cell)
    X, Y, Z = 96, 80, 72
    vol = np.zeros((Z, Y, X), np.uint8)
    zz, yy, xx = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    r = np.sqrt((xx - 40) ** 2 + (yy - 38) ** 2 + (zz - 30) ** 2)
    vol[r < 22] = (200 - 6 * r[r < 22]).astype(np.uint8)                 # a ball ...
    vol[50:60, 10:30, 60:90] = rng.integers(0, 256, (10, 20, 30))        # ... a noisy box ...
    vol[5:9, :, :] = 3                                                   # ... and a faint sheet; the rest is empty
    dvol = torch.from_numpy(vol).cuda().reshape(-1)
    grid = vr.build_skip_grid(dvol, (X, Y, Z), cell)
    g = grid.cpu().numpy().reshape(-1, 2)
    n = [(q + cell - 1) // cell for q in (X, Y, Z)]
    # the grid itself: exact (min, max) over [c*S, c*S + S] per axis
    for cx, cy, cz in ((0, 0, 0), (n[0] - 1, n[1] - 1, n[2] - 1), (5, 4, 3), (n[0] // 2, 1, n[2] - 2)):
        blk = vol[cz * cell:cz * cell + cell + 1, cy * cell:cy * cell + cell + 1, cx * cell:cx * cell + cell + 1]
        assert tuple(g[cx + n[0] * (cy + n[1] * cz)]) == (blk.min(), blk.max())
    assert (g[:, 1] == 0).mean() > 0.3                                   # plenty to skip
    for pos, front in (((0, 0, -0.75), (0, 0, 1)), ((0.8, 0.6, -0.7), (-0.7, -0.5, 0.8)), ((0.1, 0.0, 0.1), (0.3, 0.2, -1.0))):
        cam = _cam(vr, pos, front)
        for mode, iso in ((0, 0.0), (1, 40 / 255.0), (1, 150 / 255.0), (1, 2 / 255.0)):
            P = vr.default_params(320, 200, (256, 256, 128), mode, iso)
            plain = vr.raycast(dvol, (X, Y, Z), cam, P).cpu().numpy()
            vr.use_skip_grid(P, grid, cell)
            skipped = vr.raycast(dvol, (X, Y, Z), cam, P).cpu().numpy()
            assert np.array_equal(plain, skipped), (pos, mode, iso)
            assert mode != 0 or pos[2] > 0 or (plain[..., 0] < 1).any()      # (a camera inside the cube starts at the exit face)


@pytest.mark.parametrize("dims", [(128, 24, 17), (256, 40, 24), (384, 9, 8)])
def test_skip_grid_strip_kernel_equals_definition(vr, monkeypatch, dims):
    """k_skip_grid8 (rows that are multiples of 128 voxels, 8-voxel cells: one wave per strip of 16 cells) against
    the definition -- (min, max) over [8c, 8c + 8] per axis, clamped to the volume -- for every cell, and against the
    one-wave-per-cell kernel."""
    import torch
    X, Y, Z = dims
    rng = np.random.default_rng(X + Y)
    vol = rng.integers(0, 256, (Z, Y, X), dtype=np.uint8)
    vol[:, :, 100:140] = 7                    # a constant band across a strip boundary
    vol[2:6, 3:9, :] = rng.integers(100, 110, (4, 6, X))
    dvol = torch.from_numpy(vol).cuda().reshape(-1)
    g = vr.build_skip_grid(dvol, (X, Y, Z), 8).cpu().numpy().reshape(-1, 2)
    n = [(q + 7) // 8 for q in (X, Y, Z)]
    want = np.empty((n[2], n[1], n[0], 2), np.uint8)
    for cz in range(n[2]):
        for cy in range(n[1]):
            for cx in range(n[0]):
                blk = vol[cz * 8:cz * 8 + 9, cy * 8:cy * 8 + 9, cx * 8:cx * 8 + 9]
                want[cz, cy, cx] = (blk.min(), blk.max())
    assert np.array_equal(g, want.reshape(-1, 2))
    from volumerenderer_amd import _lib
    assert _lib.lib().vr_debug_set(b"skip_grid_v1", 1) == 0
    try:
        g1 = vr.build_skip_grid(dvol, (X, Y, Z), 8).cpu().numpy().reshape(-1, 2)
    finally:
        _lib.lib().vr_debug_set(b"skip_grid_v1", 0)
    assert np.array_equal(g1, g)
