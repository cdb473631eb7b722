"""Headless viewer: the reference's camera / key / mouse semantics (main.cpp:462-578)
as a windowless state machine, in C++ (include/vrhip/Viewer.hpp) and mirrored in Python."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(v):
    rows = []

    def show(what):
        rows.append((what, [float(x) for x in v.cameraPos] + [float(x) for x in v.cameraFront] +
                     [float(v.yaw), float(v.pitch), float(v.fov), float(v.currIsoVal), float(v.shouldClose)]))
    show("start")
    v.key("UP", True); v.advance(0.016); show("up")
    v.key("UP", False); v.key("LEFT", True); v.advance(0.033); show("left")
    v.key("LEFT", False)
    v.mouse(800.0, 600.0, True); v.mouse(830.0, 570.0, True); show("drag")
    v.mouse(830.0, 400.0, True); show("drag2")
    v.mouse(0.0, 0.0, False); v.mouse(100.0, 100.0, True); show("regrab")
    v.key("DOWN", True); v.key("RIGHT", True); v.advance(0.02); show("downright")
    v.key("DOWN", False); v.key("RIGHT", False)
    v.scroll(3.0); show("scroll")
    for _ in range(60): v.scroll(1.0)
    show("scrollmin")
    v.scroll(-100.0); show("scrollmax")
    for _ in range(10): v.key("0", True)
    show("isomin")
    for _ in range(60): v.key("1", True)
    show("isomax")
    v.key("ENTER", True); show("reset")
    v.key("ESCAPE", True); show("escape")
    return rows


def test_viewer_semantics_cpp_equals_python(tmp_path):
    import __graft_entry__ as g
    g.build()
    from volumerenderer_amd.viewer import HeadlessViewer
    exe = str(tmp_path / "viewer_script")
    lib = os.path.join(ROOT, "volumerenderer_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "viewer_script.cpp"), "-L" + lib, "-lvrhip",
                           "-Wl,-rpath," + lib, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    rows = _script(HeadlessViewer(1600, 1200))
    assert len(out) == len(rows)
    for line, (what, vals) in zip(out, rows):
        name, rest = line.split(" ", 1)
        got = [float(x) for x in rest.replace("|", " ").split()]
        assert name == what
        assert np.allclose(got, vals, rtol=0, atol=2e-6), (what, got, vals)
    d = dict(rows)
    assert d["start"][:3] == [0.0, 0.0, -0.75] and d["start"][8] == 50.0 and d["start"][9] == 40.0   # main.cpp:33-52
    assert abs(d["up"][2] - (-0.75 + 2.5 * 0.016)) < 1e-6                    # cameraSpeed = 2.5 * deltaTime
    assert d["drag2"][7] == 89.0                                             # pitch clamp
    assert d["scrollmin"][8] == 1.0 and d["scrollmax"][8] == 50.0            # fov in [1, fovStart]
    assert d["isomin"][9] == 0.0 and d["isomax"][9] == 255.0
    assert d["reset"][:8] == d["start"][:8] and d["escape"][10] == 1.0


@pytest.mark.gpu
def test_viewer_frame_and_dump(tmp_path, oracle):
    import torch
    import volumerenderer_amd as vr
    from volumerenderer_amd.viewer import HeadlessViewer
    vol = oracle.gen_sphere(32, 0)
    v = HeadlessViewer(160, 120)
    v.key("UP", True); v.advance(0.05); v.mouse(80.0, 60.0, True); v.mouse(84.0, 58.0, True)
    img = v.draw(torch.from_numpy(vol).cuda().reshape(-1), (32, 32, 32), brick_dims=(32, 32, 32))
    ref = vr.raycast(torch.from_numpy(vol).cuda().reshape(-1), (32, 32, 32), v.camera(),
                     vr.default_params(160, 120, (32, 32, 32), iso=40.0 / 255.0))
    assert torch.equal(img, ref)
    co = oracle.default_camera()
    cg = v.camera()
    co.pos[:] = tuple(cg.pos); co.front[:] = tuple(cg.front); co.up[:] = tuple(cg.up); co.fov_deg = cg.fov_deg
    want = oracle.render(vol, co, oracle.default_params(160, 120, (32, 32, 32), 0, 40.0 / 255.0))
    assert np.abs(img.cpu().numpy() - want).max() <= 2e-3          # pixel tolerance of SURVEY 8d
    p = str(tmp_path / "frame.ppm")
    HeadlessViewer.dump_ppm(p, img)
    raw = open(p, "rb").read()
    assert raw.startswith(b"P6\n160 120\n255\n") and len(raw) == len(b"P6\n160 120\n255\n") + 160 * 120 * 3
