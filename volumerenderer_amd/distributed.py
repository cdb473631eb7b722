"""Multi-GPU host logic: one process per GPU over torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm).

  * The codec shards with NO data-path collective: bricks (or whole timesteps) are dealt
    to ranks, every rank builds / decodes its own trees.
  * Rendering has one real exchange step: sort-last compositing of the per-rank partial
    (c, tau) images.  Direct send: the frame is cut into R row tiles, one grouped send/recv moves
    tile t of every rank's partial image to rank t (1080p: 33 MB per partial, 4.1 MB per
    peer at R = 8 -- latency-, not bandwidth-bound on the 7 x ~153 GB/s xGMI links), rank
    t composites its R partials per pixel in view order (vr_composite_slabs), and the
    finished tiles are gathered on rank 0.  A plain all-reduce cannot be used: "over" is
    associative but not commutative.

The combine step is injectable so that the exchange pattern can be exercised on CPU tensors
with the gloo backend (tests/test_distributed_cpu.py); on the GPU it is the C-ABI kernel.
"""
import ctypes as C

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous block partition [lo, hi) of n_items over world ranks (first ranks get the remainder)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_bricks_by_slab(grid, rank, world, axis=2):
    """Bricks of an I x J x K grid owned by `rank` when the grid is cut into `world` slabs along
    `axis` (brick numbering of fillVolumeBrickMap, main.cpp:599-619: i fastest).  Returns
    (brick ids, (slab_lo, slab_hi))."""
    I, J, K = grid
    lo, hi = shard_range(grid[axis], rank, world)
    ids = []
    for b in range(I * J * K):
        ijk = (b % I, (b // I) % J, b // (I * J))
        if lo <= ijk[axis] < hi:
            ids.append(b)
    return ids, (lo, hi)


def tile_rows(height, world):
    """Row range of every rank's image tile."""
    return [shard_range(height, r, world) for r in range(world)]


def _gpu_combine(parts, first_pixel, axis, cam, params):
    from . import _lib
    from .codec import _stream_ptr
    from ._lib import check
    out = torch.empty((parts.shape[1], 4), dtype=torch.float32, device=parts.device)
    check(_lib.lib().vr_composite_slabs(C.c_void_p(parts.data_ptr()), parts.shape[0], parts.shape[1], int(first_pixel),
                                        int(axis), C.byref(cam), C.byref(params), C.c_void_p(out.data_ptr()),
                                        _stream_ptr()), "vr_composite_slabs")
    return out


_compositors = {}


def _compositor(group, world, rank, W, H):
    """The C-ABI compositor of (group, frame size): its RCCL communicator is created once from an ncclUniqueId that
    rank 0 asks the library for and the process group passes on."""
    key = (id(group), world, rank, W, H)
    h = _compositors.get(key)
    if h is None:
        from . import _lib
        from ._lib import check
        L = _lib.lib()
        uid = (C.c_uint8 * 128)()
        if world > 1:
            box = [None]
            if rank == 0:
                check(L.vr_rccl_unique_id(uid), "vr_rccl_unique_id")
                box[0] = bytes(uid)
            dist.broadcast_object_list(box, src=0, group=group)
            uid = (C.c_uint8 * 128).from_buffer_copy(box[0])
        h = C.c_void_p()
        check(L.vr_compositor_create(C.byref(h), uid, rank, world, W, H), "vr_compositor_create")
        _compositors[key] = h
    return h


def composite_sort_last(partial, cam, params, axis=2, group=None, combine=None, out=None):
    """partial: this rank's (c, tau, covered, 0) image [H][W][4] float32 (rank r holds slab r along
    `axis`).  Returns the finished RGBA frame [H][W][4] on rank 0 (None elsewhere).

    Device tensors take the C-ABI compositor (vr_compositor_composite: grouped RCCL send/recv, k_composite_slabs,
    gather -- what a C++ host calls).  With an injected `combine` (the CPU tests: gloo, the oracle's combine) the same
    exchange runs over torch.distributed point-to-point operations."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    H, W = partial.shape[0], partial.shape[1]
    if combine is None and partial.is_cuda:
        from . import _lib
        from .codec import _stream_ptr
        from ._lib import check
        h = _compositor(group, world, rank, W, H)
        frame = None
        if rank == 0:
            frame = out if out is not None else torch.empty((H, W, 4), dtype=torch.float32, device=partial.device)
        check(_lib.lib().vr_compositor_composite(h, C.c_void_p(partial.data_ptr()), int(axis), C.byref(cam), C.byref(params),
                                                 C.c_void_p(frame.data_ptr()) if rank == 0 else None, _stream_ptr()),
              "vr_compositor_composite")
        return frame
    combine = combine or _gpu_combine
    rows = tile_rows(H, world)
    if world == 1:
        return combine(partial.reshape(1, H * W, 4), 0, axis, cam, params).reshape(H, W, 4)
    # send tile t to rank t, receive my tile from everybody (slab order = rank order)
    send = [partial[lo:hi].reshape(-1, 4).contiguous() for lo, hi in rows]
    my_lo, my_hi = rows[rank]
    npix = (my_hi - my_lo) * W
    recv = [torch.empty((npix, 4), dtype=partial.dtype, device=partial.device) for _ in range(world)]
    recv[rank].copy_(send[rank])
    # grouped point-to-point = ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on RCCL; unlike
    # all_to_all it also exists on gloo, which the CPU tests use
    ops = []
    for peer in range(world):
        if peer == rank:
            continue
        ops.append(dist.P2POp(dist.isend, send[peer], peer, group))
        ops.append(dist.P2POp(dist.irecv, recv[peer], peer, group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    parts = torch.stack(recv, 0)
    tile = combine(parts, my_lo * W, axis, cam, params)
    # gather the finished tiles on rank 0 (tiles may differ by one row: pad to the largest)
    max_rows = max(hi - lo for lo, hi in rows)
    padded = torch.zeros((max_rows * W, 4), dtype=tile.dtype, device=tile.device)
    padded[:npix] = tile
    gathered = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, gathered, dst=0, group=group)
    if rank != 0:
        return None
    frame = torch.empty((H, W, 4), dtype=tile.dtype, device=tile.device)
    for r, (lo, hi) in enumerate(rows):
        frame[lo:hi] = gathered[r][:(hi - lo) * W].reshape(hi - lo, W, 4)
    return frame
