"""BASELINE config 5 shape: several timesteps streamed through build -> levelCut -> frame with the
next upload overlapped; results must equal the sequential path and the oracle."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_streamed_timesteps_match_sequential_and_oracle(oracle):
    import torch
    import __graft_entry__ as g
    g.build()
    import volumerenderer_amd as vr
    from volumerenderer_amd.pipeline import TimestepStreamer
    n, B, T = 64, 4, 4
    steps = [np.stack([oracle.gen_sphere(n, 7, seed=1000 * t + b) for b in range(B)]) for t in range(T)]
    pinned = [torch.from_numpy(s).reshape(-1).pin_memory() for s in steps]
    frames = {}
    cam, P = vr.default_camera(), vr.default_params(160, 120, (n, n, n))

    def on_decoded(t, vol, stream):
        frames[t] = (vol[:n ** 3].clone(), vr.raycast(vol[:n ** 3], (n, n, n), cam, P, stream=stream))

    st = TimestepStreamer(B, (n, n, n), 1, 2)
    infos = st.run(pinned, on_decoded, overlap=True, collect_info=True)
    dec_overlap = {t: frames[t][0].cpu().numpy() for t in range(T)}
    img_overlap = {t: frames[t][1].cpu().numpy() for t in range(T)}
    frames.clear()
    infos2 = st.run(pinned, on_decoded, overlap=False, collect_info=True)
    for t in range(T):
        assert np.array_equal(dec_overlap[t], frames[t][0].cpu().numpy())
        assert np.array_equal(img_overlap[t], frames[t][1].cpu().numpy())
        assert [i["num_active_nodes"] for i in infos[t]] == [i["num_active_nodes"] for i in infos2[t]]
        ref = oracle.OracleTree(steps[t][0].copy(), tolerance=1, max_epochs=2).build()
        assert infos[t][0]["num_active_nodes"] == ref.numActiveNodes
        assert np.array_equal(dec_overlap[t].reshape(n, n, n), ref.levelCut())


def test_disk_stage_streams_brick_files(oracle, tmp_path):
    """The disk stage (VolumeReader.h:244-289 per brick, main.cpp:581-597 naming): raw brick files -> pinned staging
    -> device, read by a thread that runs ahead of the uploads; the decoded volumes equal the in-memory path's and
    a file of the wrong size raises like the reference does (VolumeReader.h:258-260)."""
    import torch
    import __graft_entry__ as g
    g.build()
    import volumerenderer_amd as vr
    from volumerenderer_amd.pipeline import BrickFileSource, TimestepStreamer
    n, B, T = 32, 6, 5
    steps = [np.stack([oracle.gen_sphere(n, 7, seed=77 * t + b) for b in range(B)]) for t in range(T)]
    for t in range(T):
        for b in range(B):
            steps[t][b].tofile(tmp_path / ("bout%02d_%04d.raw" % (b, 270 + t)))
    find = lambda brick, time: str(tmp_path / ("bout%02d_%04d.raw" % (brick, time)))
    src = BrickFileSource(find, B, (n, n, n), [270 + t for t in range(T)])
    got = {}
    st = TimestepStreamer(B, (n, n, n), 1, 2)
    st.run(src, lambda t, vol, stream: got.__setitem__(t, vol.clone()))
    mem = {}
    st.run([torch.from_numpy(s).reshape(-1).pin_memory() for s in steps], lambda t, vol, stream: mem.__setitem__(t, vol.clone()))
    for t in range(T):
        assert torch.equal(got[t], mem[t])
        ref = oracle.OracleTree(steps[t][B - 1].copy(), tolerance=1, max_epochs=2).build()
        assert np.array_equal(got[t].cpu().numpy().reshape(B, n, n, n)[B - 1], ref.levelCut())
    (tmp_path / "bout03_0272.raw").write_bytes(b"short")
    with pytest.raises(RuntimeError, match="File size does not match"):
        TimestepStreamer(B, (n, n, n), 1, 2).run(src)


def test_progressive_refinement_during_an_orbit(oracle):
    """BASELINE config 5 in miniature: 4 timesteps x 8 bricks of a MidRangeTree set streamed; every timestep is decoded
    at cuts D-6 -> D -> maxTreeDepth while the camera orbits 1 degree per frame; each stage's volume is the oracle's
    progressive cut, the half-range stream decodes next to it, and no stage waits for the host."""
    import math
    import torch
    import __graft_entry__ as g
    g.build()
    import volumerenderer_amd as vr
    from volumerenderer_amd.pipeline import TimestepStreamer
    n, B, T = 32, 8, 4
    steps = [np.stack([oracle.gen_sphere(n, 7, seed=500 * t + b) for b in range(B)]) for t in range(T)]
    pinned = [torch.from_numpy(s).reshape(-1).pin_memory() for s in steps]
    st = TimestepStreamer(B, (n, n, n), 1, 2, variant=2)
    D = 15
    cuts = [D - 6, D, D + 7]
    bmap = vr.fill_volume_brick_map(2, 2, 2)
    ijk = np.array([bmap[b] for b in range(B)], np.int64)
    cam, P = vr.default_camera(), vr.default_params(160, 120, (256, 256, 128))
    vols, frames, ranges = {}, [], {}
    theta = [0.0]

    def on_stage(t, k, cut, vol, stream):
        vols[(t, k)] = vol.clone()
        if k == 0:
            ranges[t] = st.bs.decode_range(cut_depth=cut, stream=stream).clone()
        whole = vr.assemble_bricks(vol, (n, n, n), ijk, (2, 2, 2), stream=stream)
        for _ in range(3):                              # three frames of the orbit per refinement stage
            th = math.radians(theta[0]); theta[0] += 1.0
            cam.pos[:] = (0.75 * math.sin(th), 0.0, -0.75 * math.cos(th))
            cam.front[:] = (-math.sin(th), 0.0, math.cos(th))
            frames.append(vr.raycast(whole, (2 * n, 2 * n, 2 * n), cam, P, stream=stream))

    ev = st.run_progressive(pinned, cuts, on_stage)
    assert len(frames) == T * len(cuts) * 3 and all(torch.isfinite(f).all() for f in frames)
    for t in range(T):
        for b in (0, B - 1):
            ref = oracle.OracleTree(steps[t][b].copy(), tolerance=1, max_epochs=2, midrange=True, guarded=True).build()
            for k, cut in enumerate(cuts):
                got = vols[(t, k)].cpu().numpy().reshape(B, n, n, n)[b]
                assert np.array_equal(got, ref.levelCutProgressive(cut)), (t, b, cut)
            assert np.array_equal(ranges[t].cpu().numpy().reshape(B, n, n, n)[b], ref.levelCutRange(cuts[0]))
        # stages complete in order, after the build, after the upload
        assert ev[t]["uploaded"].elapsed_time(ev[t]["built"]) > 0
        assert ev[t]["built"].elapsed_time(ev[t]["stages"][0]) > 0
        assert ev[t]["stages"][0].elapsed_time(ev[t]["stages"][2]) > 0
