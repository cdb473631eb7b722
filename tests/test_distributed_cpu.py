"""world_size-2 (and 3) gloo tests of the multi-GPU host logic on CPU tensors: brick
sharding and the direct-send sort-last compositing exchange.  The per-tile combine is the
oracle's (the HIP kernel needs a GPU); partial images come from the oracle's renderer, so
the composited frame must equal the single-process frame."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from volumerenderer_amd import distributed as D
        vol = O.gen_sphere(32, 3)
        z = vol.shape[0]
        cam = O.default_camera()
        cam.pos[:] = (0.35, 0.2, 0.9)       # looking down -z: view order = descending slab index
        cam.front[:] = (-0.3, -0.2, -1.0)
        lo, hi = D.shard_range(z, rank, world)
        P = O.default_params(w, h, (32, 32, 32), 2)
        P.box_min[:] = (0.0, 0.0, lo / z)
        P.box_max[:] = (1.0, 1.0, hi / z if rank < world - 1 else 2.0)
        P.global_dims[:] = (32, 32, 32)
        a, b = max(0, lo - 1), min(z, hi + 1)
        P.vol_origin[:] = (0, 0, a)
        part = torch.from_numpy(O.render(np.ascontiguousarray(vol[a:b]), cam, P))

        def combine(parts, first_pixel, axis, cam_, params_):
            return torch.from_numpy(O.composite_slabs(parts.numpy(), first_pixel, axis, cam_, w, h))

        frame = D.composite_sort_last(part, cam, P, axis=2, combine=combine)
        if rank == 0:
            np.save(result_path, frame.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sort_last_compositing_matches_single_pass(world, tmp_path, oracle):
    w, h = 64, 45                          # 45 rows: uneven tiles
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, out), nprocs=world, join=True)
    got = np.load(out)
    O = oracle
    vol = O.gen_sphere(32, 3)
    cam = O.default_camera()
    cam.pos[:] = (0.35, 0.2, 0.9)
    cam.front[:] = (-0.3, -0.2, -1.0)
    P = O.default_params(w, h, (32, 32, 32), 0)
    P.no_early_exit = 1
    want = O.render(vol, cam, P)
    assert np.abs(got - want).max() <= 2e-3


def test_sharding_helpers():
    from volumerenderer_amd import distributed as D
    for n in (1, 7, 15, 960):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = D.shard_range(n, r, world)
                seen += list(range(lo, hi))
            assert seen == list(range(n))
    ids = [D.shard_bricks_by_slab((8, 8, 15), r, 8)[0] for r in range(8)]
    assert sorted(sum(ids, [])) == list(range(960))
    assert all(len(x) in (64, 128) for x in ids)          # 15 slabs over 8 ranks: 7 ranks x 2, 1 rank x 1
    assert D.tile_rows(1080, 8)[0] == (0, 135) and D.tile_rows(1080, 8)[-1] == (945, 1080)
