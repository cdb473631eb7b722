"""Host-side mirror of the reference's codec classes over the C ABI (include/vrhip.h).

  BrickSet       a batch of per-brick trees (one kd-tree per brick, batched launches)
  VolumeKdtree   same names / argument meaning as the reference class
                 (volume_renderer/VolumeKdtree_recover.h:51-175), a BrickSet of 1
  MidRangeTree   volume_renderer/MidRangeTree.h:51-271 (second stream + 4-bit packing)

PyTorch appears here only as plumbing: device buffers (torch.uint8 CUDA tensors),
streams.  All compute is in libvrhip.so; nothing here falls back to the CPU.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import VrError, check


def _stream_ptr(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def _as_dev_u8(x):
    """numpy / torch, host or device -> contiguous CUDA uint8 tensor (the boundary
    hands over host buffers in the reference: std::vector<byte>&)."""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint8))
    if not isinstance(x, torch.Tensor) or x.dtype != torch.uint8:
        raise TypeError("expected a uint8 numpy array or torch tensor")
    if not x.is_cuda:
        x = x.cuda()
    return x.contiguous()


class BrickSet:
    def __init__(self, num_bricks, dims, tolerance=6, max_epochs=5, variant=_lib.VARIANT_RECOVER, _handle=None):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            d = (C.c_int64 * 3)(*[int(v) for v in dims])
            check(self._L.vr_brickset_create(C.byref(self._h), int(num_bricks), d, int(tolerance), int(max_epochs),
                                             int(variant)), "vr_brickset_create")
        self.num_bricks = int(num_bricks)
        self.dims = tuple(int(v) for v in dims)
        self.voxels_per_brick = self.dims[0] * self.dims[1] * self.dims[2]
        self._keep = None

    def __del__(self):
        try:
            if self._h:
                self._L.vr_brickset_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_error_tolerance(self, tol):
        check(self._L.vr_brickset_set_error_tolerance(self._h, int(tol)), "setErrorTolerance")

    def set_max_epochs(self, e):
        check(self._L.vr_brickset_set_max_epochs(self._h, int(e)), "setMaxEpochs")

    def set_compaction(self, on_build=True):
        """build() ends with the reference's contiguous stream (default) or leaves it to the first get_tree / save."""
        check(self._L.vr_brickset_set_compaction(self._h, 1 if on_build else 0), "vr_brickset_set_compaction")

    def set_switch(self, name, value=1):
        """Debugging switch of this set (vr_brickset_set_switch): which kernel serves the next calls."""
        check(self._L.vr_brickset_set_switch(self._h, name.encode(), int(value)), "vr_brickset_set_switch(%s)" % name)

    def build(self, voxels, stream=None):
        v = _as_dev_u8(voxels)
        if v.numel() != self.num_bricks * self.voxels_per_brick:
            raise ValueError("voxel buffer has %d bytes, expected %d" % (v.numel(), self.num_bricks * self.voxels_per_brick))
        self._keep = v  # the launches are asynchronous: keep the input alive
        check(self._L.vr_brickset_build(self._h, C.c_void_p(v.data_ptr()), _stream_ptr(stream)), "vr_brickset_build")
        return self

    def info(self, brick=0):
        ti = _lib.TreeInfo()
        check(self._L.vr_brickset_info(self._h, int(brick), C.byref(ti)), "vr_brickset_info")
        return {k: getattr(ti, k) for k, _ in ti._fields_}

    def tree(self, brick=0):
        n = self.info(brick)["tree_bytes"]
        out = np.empty(n, np.uint8)
        check(self._L.vr_brickset_get_tree(self._h, int(brick), out.ctypes.data, n), "vr_brickset_get_tree")
        return out

    def distance_map(self, brick=0):
        n = self.info(brick)["max_tree_depth"] + 1
        out = np.empty(n, np.uint8)
        check(self._L.vr_brickset_get_distance_map(self._h, int(brick), out.ctypes.data, n), "get_distance_map")
        return out

    def tree_range(self, brick=0):
        n = self.info(brick)["tree_bytes"]
        out = np.empty(n, np.uint8)
        check(self._L.vr_brickset_get_tree_range(self._h, int(brick), out.ctypes.data, n), "get_tree_range")
        return out

    def distance_map_range(self, brick=0):
        n = self.info(brick)["max_tree_depth"] + 1
        out = np.empty(n, np.uint8)
        check(self._L.vr_brickset_get_distance_map_range(self._h, int(brick), out.ctypes.data, n), "get_distance_map_range")
        return out

    def packed4(self, brick=0):
        n = C.c_int64()
        check(self._L.vr_brickset_get_packed4(self._h, int(brick), None, 0, C.byref(n)), "get_packed4")
        out = np.empty(n.value, np.uint8)
        check(self._L.vr_brickset_get_packed4(self._h, int(brick), out.ctypes.data, n.value, C.byref(n)), "get_packed4")
        return out

    def decode(self, out=None, cut_depth=-1, stream=None):
        if out is None:
            out = torch.empty(self.num_bricks * self.voxels_per_brick, dtype=torch.uint8, device="cuda")
        if not (out.is_cuda and out.dtype == torch.uint8 and out.is_contiguous()
                and out.numel() == self.num_bricks * self.voxels_per_brick):
            raise ValueError("bad output buffer")
        check(self._L.vr_brickset_decode(self._h, int(cut_depth), C.c_void_p(out.data_ptr()), _stream_ptr(stream)),
              "vr_brickset_decode")
        return out

    def decode_range(self, out=None, cut_depth=-1, stream=None):
        """MidRangeTree sets: the half-range stream decoded like the mid stream (vr_brickset_decode_range)."""
        if out is None:
            out = torch.empty(self.num_bricks * self.voxels_per_brick, dtype=torch.uint8, device="cuda")
        if not (out.is_cuda and out.dtype == torch.uint8 and out.is_contiguous()
                and out.numel() == self.num_bricks * self.voxels_per_brick):
            raise ValueError("bad output buffer")
        check(self._L.vr_brickset_decode_range(self._h, int(cut_depth), C.c_void_p(out.data_ptr()), _stream_ptr(stream)),
              "vr_brickset_decode_range")
        return out

    def set_tree(self, brick, tree_bytes, num_active_nodes, distance_map):
        t = np.ascontiguousarray(tree_bytes, np.uint8)
        d = np.ascontiguousarray(distance_map, np.uint8)
        check(self._L.vr_brickset_set_tree(self._h, int(brick), t.ctypes.data, t.size, int(num_active_nodes),
                                           d.ctypes.data, d.size), "vr_brickset_set_tree")

    def save(self, path, brick=0):
        check(self._L.vr_brickset_save(self._h, int(brick), os.fsencode(path)), "vr_brickset_save")

    @classmethod
    def open(cls, path, variant=_lib.VARIANT_RECOVER):
        L = _lib.lib()
        h = C.c_void_p()
        check(L.vr_brickset_open_variant(C.byref(h), os.fsencode(path), int(variant)), "vr_brickset_open")
        ti = _lib.TreeInfo()
        check(L.vr_brickset_info(h, 0, C.byref(ti)), "vr_brickset_info")
        return cls(1, (ti.X, ti.Y, ti.Z), _handle=h)

    def set_concurrency(self, level_loop_streams):
        """Brick ranges whose level loops run side by side inside one build (1 = off, 2 = default, <= 4): see vrhip.h."""
        check(self._L.vr_brickset_set_concurrency(self._h, int(level_loop_streams)), "set_concurrency")
        return self

    def last_timings(self):
        ms = (C.c_float * 5)()
        check(self._L.vr_brickset_last_timings(self._h, ms), "last_timings")
        return dict(zip(("BUILD", "COMPRESS", "PRUNE", "CONVERT", "DECODE"), [float(v) for v in ms]))


def measure_error(decoded, original, stream=None):
    """measureMaxError / measureMeanError (R.cpp:386-401) with the original passed explicitly."""
    a, b = _as_dev_u8(decoded), _as_dev_u8(original)
    mx, mean = C.c_int32(), C.c_double()
    check(_lib.lib().vr_measure_error(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), a.numel(),
                                      C.byref(mx), C.byref(mean), _stream_ptr(stream)), "vr_measure_error")
    return mx.value, mean.value


def query_error(decoded, original, stream=None):
    """queryError (R.cpp:404-411)."""
    a, b = _as_dev_u8(decoded), _as_dev_u8(original)
    out = torch.empty_like(a)
    check(_lib.lib().vr_query_error(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), a.numel(),
                                    C.c_void_p(out.data_ptr()), _stream_ptr(stream)), "vr_query_error")
    return out


class VolumeKdtree:
    """volume_renderer/VolumeKdtree_recover.h:51-175 -- same method names and argument meaning.

    Differences forced by the defects listed in SURVEY.md Appendix C: the input is
    not destroyed by build() (C-7: the error helpers need it); levelCut() below
    maxTreeDepth is a defined progressive cut rather than the reference's broken walk (C-4)."""
    _variant = _lib.VARIANT_RECOVER

    def __init__(self, inData=None, x=0, y=0, z=0):
        self.tolerance = 6      # R.h:92-93
        self.maxEpochs = 5
        self.X, self.Y, self.Z = int(x), int(y), int(z)
        self._data = _as_dev_u8(inData).reshape(-1) if inData is not None else None
        self._bs = None
        self._output = None

    def setErrorTolerance(self, errorTolerance):   # R.cpp:9-11
        self.tolerance = int(errorTolerance)

    def setMaxEpochs(self, epochs):                # R.cpp:13-15
        self.maxEpochs = int(epochs)

    def build(self, useThreads=True):              # R.cpp:17-140 (useThreads is meaningless on the GPU)
        if self._data is None:
            raise VrError(-5, "build() without data")
        self._bs = BrickSet(1, (self.X, self.Y, self.Z), self.tolerance, self.maxEpochs, self._variant)
        self._bs.build(self._data)
        return self

    def _need(self):
        if self._bs is None:
            raise VrError(-5, "no tree")
        return self._bs

    # public data members of the reference class
    @property
    def tree(self): return self._need().tree(0)
    @property
    def distanceMap(self): return self._need().distance_map(0)
    @property
    def numActiveNodes(self): return self._need().info(0)["num_active_nodes"]
    @property
    def origTreeDepth(self): return self._need().info(0)["orig_tree_depth"]
    @property
    def maxTreeDepth(self): return self._need().info(0)["max_tree_depth"]

    def levelCut(self, cutDepth=None, outData=None):   # R.cpp:726-835
        bs = self._need()
        self._output = bs.decode(outData, -1 if cutDepth is None else int(cutDepth))
        return self._output

    def measureMaxError(self):                     # R.cpp:386-392
        return measure_error(self._output, self._data)[0]

    def measureMeanError(self):                    # R.cpp:394-401
        return measure_error(self._output, self._data)[1]

    def queryError(self):                          # R.cpp:404-411
        return query_error(self._output, self._data)

    def save(self, filename):                      # R.cpp:521-552
        self._need().save(filename, 0)

    def open(self, filename):                      # R.cpp:554-594 / M.cpp:787-833
        self._bs = BrickSet.open(filename, self._variant)
        self.X, self.Y, self.Z = self._bs.dims
        return self


class MidRangeTree(VolumeKdtree):
    """volume_renderer/MidRangeTree.h:51-271: VolumeKdtree + half-range stream."""
    _variant = _lib.VARIANT_MIDRANGE

    @property
    def tree_range(self): return self._need().tree_range(0)
    @property
    def distanceMap_range(self): return self._need().distance_map_range(0)

    def convertToByteArray(self):                  # M.cpp:1095-1128
        return self._need().packed4(0)

    def levelCutRange(self, cutDepth=None, outData=None):
        """New (the reference never decodes tree_range): the half range per voxel at a cut depth."""
        return self._need().decode_range(outData, -1 if cutDepth is None else int(cutDepth))


class HashedKdtree(VolumeKdtree):
    """volume_renderer/HashedKdtree.h:26-227 -- the interface only; PARITY UNPINNED.

    The reference class cannot be run (heap overflow on the first level, HashedKdtree.cpp:138; collision handling
    seeded from std::random_device, :473), so there is nothing to compare with.  Callers get the same methods and
    members over the VolumeKdtree path with the class's own tolerance (4, HashedKdtree.h:79); the hash-table members
    exist and stay empty."""

    def __init__(self, inData=None, x=0, y=0, z=0):
        super().__init__(inData, x, y, z)
        self.tolerance = 4
        self.numCollisions = 0
        self.hashMask = 0
        self.queryDepth = 0

    @property
    def treeData(self): return self.tree
    @property
    def treeDepth(self): return self.maxTreeDepth

    def levelCut(self, cutDepth=None, outData=None):   # HashedKdtree.h:113
        self.queryDepth = self.treeDepth if cutDepth is None else int(cutDepth)
        return super().levelCut(cutDepth, outData)
