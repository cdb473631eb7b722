"""GPU parity of the ray-march kernels against the CPU float restatement of
raycaster.frag / isosurface.frag (oracle/raymarch_oracle.c).

Tolerance: |delta| <= 2e-3 per channel on float RGBA in [0,1] (about half an LSB of
an 8-bit framebuffer; covers sqrt/pow/division rounding differences).  Parity with a
real OpenGL driver is unpinned (no GL here; the reference holds no golden images)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 2e-3


@pytest.fixture(scope="module")
def vr():
    import __graft_entry__ as g
    g.build()
    import volumerenderer_amd as vr
    return vr


def _cams(vr, O, pos, front, fov=50.0):
    a, b = vr.default_camera(), O.default_camera()
    for c in (a, b):
        c.pos[:] = pos
        c.front[:] = front
        c.fov_deg = fov
    return a, b


def _both(vr, O, vol, w, h, mode, pos=(0, 0, -0.75), front=(0, 0, 1), step_dims=None, iso=40 / 255.0, **kw):
    z, y, x = vol.shape
    sd = step_dims or (x, y, z)
    cg, co = _cams(vr, O, pos, front)
    Pg, Po = vr.default_params(w, h, sd, mode, iso), O.default_params(w, h, sd, mode, iso)
    for k, v in kw.items():
        setattr(Pg, k, v)
        setattr(Po, k, v)
    got = vr.raycast(vol.copy(), (x, y, z), cg, Pg).cpu().numpy()
    want = O.render(vol, co, Po)
    return got, want


@pytest.mark.parametrize("mode", [0, 1])
def test_sphere_matches_oracle(vr, oracle, mode):
    vol = oracle.gen_sphere(64, 3)
    got, want = _both(vr, oracle, vol, 160, 120, mode)
    assert np.abs(got - want).max() <= TOL
    assert (want[..., 0] < 0.999).any()      # the cube is on screen


@pytest.mark.parametrize("pos,front", [((0.9, 0.4, -0.9), (-0.6, -0.3, 0.7)), ((0.0, 1.2, 0.1), (0.0, -1.0, -0.05)),
                                       ((0.1, 0.05, -0.2), (0.2, 0.1, 1.0))])   # last: camera inside the cube
def test_camera_positions(vr, oracle, pos, front):
    rng = np.random.default_rng(1)
    vol = rng.integers(0, 256, (32, 48, 16), dtype=np.uint8)   # anisotropic + reference's BRICK_DIM-style step
    for mode in (0, 1):
        got, want = _both(vr, oracle, vol, 200, 96, mode, pos, front, step_dims=(256, 256, 128), iso=0.5)
        assert np.abs(got - want).max() <= TOL


def test_analytic_constant_volume(vr, oracle):
    """Constant volume s: after n samples A_n = (1 - (1-0.6 s)^n)/0.6... closed form of raycaster.frag:69-72."""
    s = 51 / 255.0
    vol = np.full((16, 16, 16), 51, np.uint8)
    got, _ = _both(vr, oracle, vol, 64, 64, 0)
    # near-centre ray (pixel centres are half a pixel off axis): enters at vUV.z = 0 and advances
    # gd.z/16 < 1/16 per step, so samples k = 1..16 are all strictly inside the cube
    n = 16
    T = (1 - 0.6 * s) ** n
    A = 1 - T
    rgb = s * s * (1 - T) / (0.6 * s)
    px = got[32, 32]
    assert abs(px[3] - A) < 1e-4 and abs(px[0] - (1 - rgb)) < 1e-4 and px[2] == 1.0


def test_empty_volume_and_miss_are_white(vr, oracle):
    vol = np.zeros((8, 8, 8), np.uint8)
    got, want = _both(vr, oracle, vol, 64, 48, 0)
    assert np.abs(got - want).max() <= 1e-6
    assert np.allclose(got[24, 32, :3], 1.0) and got[24, 32, 3] == 0.0   # covered, nothing accumulated
    got, want = _both(vr, oracle, vol, 64, 48, 0, pos=(0, 0, -3.0))
    assert np.abs(got - want).max() <= 1e-6
    assert np.allclose(got[0, 0], 1.0)                 # corner pixel misses the cube: clear colour (1,1,1,1)
    assert got[24, 32, 3] == 0.0
    got, want = _both(vr, oracle, vol, 64, 48, 0, pos=(0, 0, -0.75), front=(0, 0, -1))   # looking away
    assert np.allclose(got, 1.0) and np.allclose(want, 1.0)


def test_partial_images_composite_to_full(vr, oracle):
    """Sort-last: z-slab partial (c, tau) images composited in view order == single pass without early exit."""
    import torch
    vol = oracle.gen_sphere(32, 3)
    z, y, x = vol.shape
    cg, co = _cams(vr, oracle, (0.3, 0.2, -0.9), (-0.25, -0.15, 1.0))
    w, h = 128, 96
    full = vr.default_params(w, h, (x, y, z), 0)
    full.no_early_exit = 1
    want = vr.raycast(vol.copy(), (x, y, z), cg, full).cpu().numpy()
    R = 4
    acc = None
    for r in range(R):                                  # dir.z > 0 for every pixel: slab order = view order
        z0, z1 = r * z // R, (r + 1) * z // R
        lo, hi = max(0, z0 - 1), min(z, z1 + 1)         # one halo layer each side
        P = vr.default_params(w, h, (x, y, z), 2)
        P.box_min[:] = (0.0, 0.0, z0 / z)
        P.box_max[:] = (1.0, 1.0, z1 / z if r < R - 1 else 2.0)
        P.global_dims[:] = (x, y, z)
        P.vol_origin[:] = (0, 0, lo)
        part = vr.raycast(np.ascontiguousarray(vol[lo:hi]), (x, y, hi - lo), cg, P)
        # the oracle's partial mode agrees too
        Po = oracle.default_params(w, h, (x, y, z), 2)
        Po.box_min[:] = P.box_min[:]; Po.box_max[:] = P.box_max[:]
        Po.global_dims[:] = (x, y, z); Po.vol_origin[:] = (0, 0, lo)
        assert np.abs(part.cpu().numpy() - oracle.render(np.ascontiguousarray(vol[lo:hi]), co, Po)).max() <= TOL
        acc = part if acc is None else vr.composite_over(acc, part)
    got = vr.composite_finish(acc).cpu().numpy()
    assert np.abs(got - want).max() <= TOL
    # early exit changes the result by at most 0.01/0.6
    ee = vr.raycast(vol.copy(), (x, y, z), cg, vr.default_params(w, h, (x, y, z), 0)).cpu().numpy()
    assert np.abs(ee - want).max() <= 0.017


def test_assemble_bricks_matches_reference_indexing(vr, oracle):
    import torch
    rng = np.random.default_rng(0)
    X, Y, Z, I, J, K = 16, 8, 4, 2, 3, 2
    nb = I * J * K
    bricks = rng.integers(0, 256, (nb, Z, Y, X), dtype=np.uint8)
    bmap = vr.fill_volume_brick_map(I, J, K)
    ijk = np.array([bmap[b] for b in range(nb)], np.int64)
    vol = vr.assemble_bricks(bricks, (X, Y, Z), ijk, (I, J, K)).cpu().numpy().reshape(K * Z, J * Y, I * X)
    want = np.zeros_like(vol)
    for b in range(nb):                                 # VolumeReader.h:184-198
        i, j, k = bmap[b]
        want[k * Z:(k + 1) * Z, j * Y:(j + 1) * Y, i * X:(i + 1) * X] = bricks[b]
    assert np.array_equal(vol, want)
    back = vr.disassemble_bricks(vol, (X, Y, Z), ijk, (I, J, K)).cpu().numpy().reshape(nb, Z, Y, X)
    assert np.array_equal(back, bricks)


def test_1080p_frame_of_decoded_brick(vr, oracle):
    """BASELINE config 2 shape: decoded brick -> 1920x1080 frame; spot-check rows against the oracle."""
    vol = oracle.gen_sphere(128, 7)
    t = vr.VolumeKdtree(vol.copy(), 128, 128, 128)
    t.setMaxEpochs(2); t.setErrorTolerance(1)
    t.build()
    dec = t.levelCut(t.maxTreeDepth)
    cam, P = vr.default_camera(), vr.default_params(1920, 1080, (128, 128, 128))
    img = vr.raycast(dec, (128, 128, 128), cam, P).cpu().numpy()
    assert img.shape == (1080, 1920, 4) and np.isfinite(img).all()
    sub_w, sub_h = 240, 135                          # oracle at 1/8 resolution samples the same pixel centres? no: compare statistics
    small = oracle.render(dec.cpu().numpy().reshape(128, 128, 128), oracle.default_camera(),
                          oracle.default_params(sub_w, sub_h, (128, 128, 128)))
    assert abs(img[..., 0].mean() - small[..., 0].mean()) < 5e-3


def test_composite_slabs_kernel_matches_oracle(vr, oracle):
    """vr_composite_slabs (the per-tile combine of the direct-send exchange) against the oracle,
    with a camera for which the view order along z differs between pixels."""
    import ctypes as C
    import torch
    from volumerenderer_amd import distributed as D
    vol = oracle.gen_sphere(32, 3)
    w, h, R = 96, 64, 4
    cg, co = _cams(vr, oracle, (0.9, 0.1, 0.02), (-1.0, -0.1, 0.0))     # looking along -x: dir.z changes sign
    parts = []
    for r in range(R):
        lo, hi = D.shard_range(32, r, R)
        a, b = max(0, lo - 1), min(32, hi + 1)
        P = vr.default_params(w, h, (32, 32, 32), 2)
        P.box_min[:] = (0.0, 0.0, lo / 32)
        P.box_max[:] = (1.0, 1.0, hi / 32 if r < R - 1 else 2.0)
        P.global_dims[:] = (32, 32, 32)
        P.vol_origin[:] = (0, 0, a)
        parts.append(vr.raycast(np.ascontiguousarray(vol[a:b]), (32, 32, b - a), cg, P))
    stack = torch.stack([p.reshape(-1, 4) for p in parts], 0).contiguous()
    full = vr.default_params(w, h, (32, 32, 32), 0)
    got = D._gpu_combine(stack, 0, 2, cg, full).cpu().numpy()
    want = oracle.composite_slabs(stack.cpu().numpy(), 0, 2, co, w, h)
    assert np.abs(got - want).max() <= 1e-6
    ref = vr.default_params(w, h, (32, 32, 32), 0)
    ref.no_early_exit = 1
    single = vr.raycast(vol.copy(), (32, 32, 32), cg, ref).cpu().numpy().reshape(-1, 4)
    assert np.abs(got - single).max() <= TOL
    # a tile in the middle of the frame (first_pixel > 0)
    lo, hi = 20 * w, 41 * w
    tile = D._gpu_combine(stack[:, lo:hi].contiguous(), lo, 2, cg, full).cpu().numpy()
    assert np.abs(tile - got[lo:hi]).max() <= 1e-6


def test_c_abi_compositor_single_rank_and_rccl_binding(vr, oracle):
    """vr_compositor_* (the C ABI a C++ host composites with): a one-rank compositor needs no communicator and must equal
    the colour transfer of the partial image; the RCCL binding (dlopen at run time) must hand out an ncclUniqueId on a
    box that has RCCL; composite_sort_last without a process group takes the same path."""
    import ctypes as C
    import torch
    from volumerenderer_amd import _lib
    from volumerenderer_amd import distributed as D
    L = _lib.lib()
    vol = oracle.gen_sphere(32, 3)
    w, h = 96, 64
    cg, co = _cams(vr, oracle, (0.3, 0.2, -0.9), (-0.2, -0.1, 1.0))
    P = vr.default_params(w, h, (32, 32, 32), 2)
    part = vr.raycast(vol.copy(), (32, 32, 32), cg, P)
    hnd = C.c_void_p()
    assert L.vr_compositor_create(C.byref(hnd), None, 0, 1, w, h) == 0
    out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    full = vr.default_params(w, h, (32, 32, 32), 0)
    assert L.vr_compositor_composite(hnd, C.c_void_p(part.data_ptr()), 2, C.byref(cg), C.byref(full), C.c_void_p(out.data_ptr()), None) == 0
    torch.cuda.synchronize()
    want = vr.composite_finish(part).cpu().numpy().reshape(h, w, 4)
    assert np.abs(out.cpu().numpy() - want).max() <= 1e-6
    bad = vr.default_params(w + 1, h, (32, 32, 32), 0)
    assert L.vr_compositor_composite(hnd, C.c_void_p(part.data_ptr()), 2, C.byref(cg), C.byref(bad), C.c_void_p(out.data_ptr()), None) == -1
    assert L.vr_compositor_destroy(hnd) == 0
    frame = D.composite_sort_last(part, cg, full, axis=2)
    assert np.abs(frame.cpu().numpy() - want).max() <= 1e-6
    uid = (C.c_uint8 * 128)()
    assert L.vr_rccl_unique_id(uid) == 0 and any(uid)         # RCCL found and bound (the image has it)
    assert L.vr_compositor_create(C.byref(hnd), None, 0, 2, w, h) == -1      # more than one rank needs the shared id
    assert L.vr_compositor_create(C.byref(hnd), uid, 2, 2, w, h) == -1       # rank out of range


def test_config3_isosurface_of_brick_grid(vr, oracle):
    """BASELINE config 3: 2x2x2 bricks assembled into one volume, iso-surface shader at several iso values."""
    n = 32
    bricks = np.stack([oracle.gen_sphere(n, 3, seed=100 + b) for b in range(8)])
    bmap = vr.fill_volume_brick_map(2, 2, 2)
    ijk = np.array([bmap[b] for b in range(8)], np.int64)
    vol = vr.assemble_bricks(bricks, (n, n, n), ijk, (2, 2, 2))
    host = vol.cpu().numpy().reshape(2 * n, 2 * n, 2 * n)
    for iso in (40 / 255.0, 80 / 255.0, 120 / 255.0):
        cg, co = _cams(vr, oracle, (0.2, 0.3, -0.9), (-0.2, -0.3, 1.0))
        Pg, Po = vr.default_params(120, 90, (256, 256, 128), 1, iso), oracle.default_params(120, 90, (256, 256, 128), 1, iso)
        got = vr.raycast(vol, (2 * n, 2 * n, 2 * n), cg, Pg).cpu().numpy()
        want = oracle.render(host, co, Po)
        assert np.abs(got - want).max() <= TOL
        assert (want[..., 0] < 0.99).any()
