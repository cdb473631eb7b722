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
    infos = st.run(pinned, on_decoded, overlap=True)
    dec_overlap = {t: frames[t][0].cpu().numpy() for t in range(T)}
    img_overlap = {t: frames[t][1].cpu().numpy() for t in range(T)}
    frames.clear()
    infos2 = st.run(pinned, on_decoded, overlap=False)
    for t in range(T):
        assert np.array_equal(dec_overlap[t], frames[t][0].cpu().numpy())
        assert np.array_equal(img_overlap[t], frames[t][1].cpu().numpy())
        assert [i["num_active_nodes"] for i in infos[t]] == [i["num_active_nodes"] for i in infos2[t]]
        ref = oracle.OracleTree(steps[t][0].copy(), tolerance=1, max_epochs=2).build()
        assert infos[t][0]["num_active_nodes"] == ref.numActiveNodes
        assert np.array_equal(dec_overlap[t].reshape(n, n, n), ref.levelCut())
