"""Timestep streaming (BASELINE config 5): disk -> pinned host memory -> device -> build ->
progressive levelCut -> frame, with the stages of consecutive timesteps overlapped on separate HIP streams and
double-buffered volumes.  The reference does these strictly one after another (main.cpp:242-290):
LoadBricksToTexture -> build -> levelCut -> glTexImage3D -> draw.

torch is plumbing here (pinned host memory, streams, events); all compute is libvrhip.so.  Nothing in this module
synchronises the device per timestep unless the caller asks for the trees' public members (collect_info=True), so the
build of timestep t+1 can start while the frames of timestep t are still being drawn."""
import os
import threading

import numpy as np
import torch

from . import _lib
from .codec import BrickSet


class BrickFileSource:
    """The disk stage: VolumeReader<T>::LoadVolumeFromBinaryFile for every brick of a timestep (VolumeReader.h:244-289,
    called from LoadBricksToTexture, :172-182).  find_source_file(brick, timestep) -> path, as the reference's
    findSourceFile (main.cpp:581-597).  A file whose size is not X*Y*Z bytes raises, as :254-261 does."""

    def __init__(self, find_source_file, num_bricks, brick_dims, timesteps):
        self.find = find_source_file
        self.num_bricks = int(num_bricks)
        self.brick_bytes = int(brick_dims[0]) * int(brick_dims[1]) * int(brick_dims[2])
        self.timesteps = list(timesteps)

    def __len__(self):
        return len(self.timesteps)

    def read_into(self, i, pinned):
        """Timestep self.timesteps[i] -> pinned uint8 host tensor (num_bricks * brick_bytes), brick after brick."""
        dst = pinned.numpy()
        for b in range(self.num_bricks):
            path = self.find(b, self.timesteps[i])
            if os.path.getsize(path) != self.brick_bytes:
                raise RuntimeError("File size does not match expected dataset size!")       # VolumeReader.h:258-260
            with open(path, "rb") as f:
                got = f.readinto(dst[b * self.brick_bytes:(b + 1) * self.brick_bytes])
            if got != self.brick_bytes:
                raise RuntimeError("short read: %s" % path)


class TimestepStreamer:
    def __init__(self, num_bricks, brick_dims, tolerance=1, max_epochs=2, variant=_lib.VARIANT_RECOVER):
        self.bs = BrickSet(num_bricks, brick_dims, tolerance, max_epochs, variant)
        self.n = num_bricks * brick_dims[0] * brick_dims[1] * brick_dims[2]
        self.vox = [torch.empty(self.n, dtype=torch.uint8, device="cuda") for _ in range(2)]
        self.out = [torch.empty(self.n, dtype=torch.uint8, device="cuda") for _ in range(2)]
        self.copy_stream = torch.cuda.Stream()
        self.compute_stream = torch.cuda.Stream()
        self._stage = None      # pinned staging buffers of the disk stage

    # ---- where a timestep's bytes come from: pinned host tensors, or brick files through two pinned staging buffers
    def _host_feed(self, source, consumed_by_copy):
        """Returns get(t) -> pinned host tensor holding timestep t (blocks until the disk stage has read it)."""
        if not isinstance(source, BrickFileSource):
            return (lambda t: source[t]), None
        if self._stage is None:
            self._stage = [torch.empty(self.n, dtype=torch.uint8).pin_memory() for _ in range(2)]
        ready = [threading.Event() for _ in range(len(source))]
        err = []

        def reader():
            try:
                for t in range(len(source)):
                    if t >= 2:
                        consumed_by_copy[t - 2].wait()          # the copy out of this staging buffer has been issued ...
                        self._copied[t - 2].synchronize()       # ... and has finished
                    source.read_into(t, self._stage[t & 1])
                    ready[t].set()
            except Exception as ex:      # surfaces in the consumer
                err.append(ex)
                for e in ready:
                    e.set()

        th = threading.Thread(target=reader, daemon=True)
        th.start()

        def get(t):
            ready[t].wait()
            if err:
                raise err[0]
            return self._stage[t & 1]

        return get, th

    def run(self, source, on_decoded=None, overlap=True, collect_info=False, cut_depth=-1):
        """source: list of pinned uint8 host tensors (one per timestep) or a BrickFileSource.  on_decoded(t, volume,
        stream) is called with the decoded device volume while `stream` is current (launch the frames there).
        collect_info=True returns the per-timestep tree infos -- at the price of one device synchronisation per
        timestep (the reference prints them inside build(), R.cpp:39-41,138-139)."""
        T = len(source)
        infos = []
        uploaded = [torch.cuda.Event() for _ in range(T)]
        consumed = [torch.cuda.Event() for _ in range(T)]
        issued = [threading.Event() for _ in range(T)]
        self._copied = uploaded
        get, th = self._host_feed(source, issued)

        def upload(t):
            host = get(t)
            with torch.cuda.stream(self.copy_stream):
                if t >= 2:
                    self.copy_stream.wait_event(consumed[t - 2])      # the device buffer's previous user has finished
                self.vox[t & 1].copy_(host, non_blocking=True)
                uploaded[t].record(self.copy_stream)
            issued[t].set()

        upload(0)
        for t in range(T):
            if overlap and t + 1 < T:
                upload(t + 1)
            with torch.cuda.stream(self.compute_stream):
                self.compute_stream.wait_event(uploaded[t])
                self.bs.build(self.vox[t & 1], stream=self.compute_stream)
                self.bs.decode(self.out[t & 1], cut_depth=cut_depth, stream=self.compute_stream)
                if on_decoded is not None:
                    on_decoded(t, self.out[t & 1], self.compute_stream)
                consumed[t].record(self.compute_stream)
            if collect_info:
                infos.append([self.bs.info(b) for b in range(self.bs.num_bricks)])
            if not overlap and t + 1 < T:
                upload(t + 1)
                self.copy_stream.synchronize()
        torch.cuda.synchronize()
        if th is not None:
            th.join()
        return infos

    def run_progressive(self, source, cuts, on_stage, overlap=True):
        """BASELINE config 5: every timestep is built once and decoded at each depth of `cuts` in turn (coarse to fine:
        a progressive cut costs one decode launch, the stream is not rebuilt); on_stage(t, k, cut, volume, stream) draws
        its frames after stage k.  Returns per timestep the hipEvents (uploaded, built, [stage done ...]) for
        time-to-first-frame / time-to-refine."""
        T = len(source)
        uploaded = [torch.cuda.Event(enable_timing=True) for _ in range(T)]
        built = [torch.cuda.Event(enable_timing=True) for _ in range(T)]
        staged = [[torch.cuda.Event(enable_timing=True) for _ in cuts] for _ in range(T)]
        consumed = [torch.cuda.Event() for _ in range(T)]
        issued = [threading.Event() for _ in range(T)]
        self._copied = uploaded
        get, th = self._host_feed(source, issued)

        def upload(t):
            host = get(t)
            with torch.cuda.stream(self.copy_stream):
                if t >= 2:
                    self.copy_stream.wait_event(consumed[t - 2])
                self.vox[t & 1].copy_(host, non_blocking=True)
                uploaded[t].record(self.copy_stream)
            issued[t].set()

        upload(0)
        for t in range(T):
            if overlap and t + 1 < T:
                upload(t + 1)
            with torch.cuda.stream(self.compute_stream):
                self.compute_stream.wait_event(uploaded[t])
                self.bs.build(self.vox[t & 1], stream=self.compute_stream)
                built[t].record(self.compute_stream)
                for k, cut in enumerate(cuts):
                    self.bs.decode(self.out[t & 1], cut_depth=cut, stream=self.compute_stream)
                    on_stage(t, k, cut, self.out[t & 1], self.compute_stream)
                    staged[t][k].record(self.compute_stream)
                consumed[t].record(self.compute_stream)
            if not overlap and t + 1 < T:
                upload(t + 1)
                self.copy_stream.synchronize()
        torch.cuda.synchronize()
        if th is not None:
            th.join()
        return [{"uploaded": uploaded[t], "built": built[t], "stages": staged[t]} for t in range(T)]
