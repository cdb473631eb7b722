"""ctypes front-end for oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/kdtree_oracle.c header).  Nothing under
volumerenderer_amd/ imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def _source_hash():
    import hashlib
    h = hashlib.sha256()
    for s in ("kdtree_oracle.c", "raymarch_oracle.c", "Makefile"):
        h.update(open(os.path.join(_HERE, s), "rb").read())
    return h.hexdigest()


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile).  Staleness is a content hash (mtimes do
    not survive the copy to a GPU box)."""
    stamp = _LIB_PATH + ".srchash"
    fresh = os.path.exists(_LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == _source_hash()
    if force or not fresh:
        import fcntl
        with open(_LIB_PATH + ".lock", "w") as lk:      # ranks of one node: one builds, the others wait
            fcntl.flock(lk, fcntl.LOCK_EX)
            fresh = os.path.exists(_LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == _source_hash()
            if force or not fresh:
                subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
                with open(stamp, "w") as f:
                    f.write(_source_hash())
    return _LIB_PATH


class TraceRec(C.Structure):
    _fields_ = [("depth", C.c_int32), ("epoch", C.c_int32), ("kind", C.c_int32),
                ("distance", C.c_double), ("error", C.c_double), ("df", C.c_double), ("step", C.c_double)]


class Camera(C.Structure):
    """Mirrors vr_camera in include/vrhip.h (main.cpp:33-40,396-397)."""
    _fields_ = [("pos", C.c_float * 3), ("front", C.c_float * 3), ("up", C.c_float * 3),
                ("fov_deg", C.c_float), ("z_near", C.c_float), ("z_far", C.c_float)]


class RenderParams(C.Structure):
    """Mirrors vr_render_params in include/vrhip.h (main.cpp:330-334, raycaster.frag:14)."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("step_size", C.c_float * 3),
                ("iso_value", C.c_float), ("max_samples", C.c_int32), ("mode", C.c_int32),
                ("box_min", C.c_float * 3), ("box_max", C.c_float * 3),
                ("global_dims", C.c_int64 * 3), ("vol_origin", C.c_int64 * 3),
                ("no_early_exit", C.c_int32), ("reserved", C.c_int32)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    p, u8p, i64, i32 = C.c_void_p, C.POINTER(C.c_uint8), C.c_int64, C.c_int32
    L.vko_create.restype = p
    L.vko_create.argtypes = [p, i64, i64, i64]
    L.vko_destroy.argtypes = [p]
    for name in ("vko_set_error_tolerance", "vko_set_max_epochs", "vko_set_guarded", "vko_set_midrange"):
        getattr(L, name).argtypes = [p, C.c_int]
    L.vko_build_to_stage.argtypes = [p, C.c_int]
    L.vko_build_to_stage.restype = C.c_int
    L.vko_build.argtypes = [p]
    L.vko_level_cut.argtypes = [p, C.c_int, p]
    L.vko_level_cut.restype = C.c_int
    L.vko_level_cut_progressive.argtypes = [p, C.c_int, p]
    L.vko_level_cut_progressive.restype = C.c_int
    L.vko_level_cut_range.argtypes = [p, C.c_int, p]
    L.vko_level_cut_range.restype = C.c_int
    L.vko_save.argtypes = [p, C.c_char_p]
    L.vko_open.argtypes = [C.c_char_p]
    L.vko_open.restype = p
    L.vko_open_midrange.argtypes = [C.c_char_p]
    L.vko_open_midrange.restype = p
    for name in ("vko_orig_tree_depth", "vko_max_tree_depth", "vko_num_reverts", "vko_zero_run_rewrites",
                 "vko_trace_len"):
        getattr(L, name).argtypes = [p]
        getattr(L, name).restype = i32
    for name in ("vko_num_active_nodes", "vko_num_orig_nodes", "vko_first_orig_leaf", "vko_tree_bytes",
                 "vko_temp_len", "vko_recon_len", "vko_tree_range_bytes"):
        getattr(L, name).argtypes = [p]
        getattr(L, name).restype = i64
    for name in ("vko_tree_ptr", "vko_distance_map_ptr", "vko_temp_ptr", "vko_recon_ptr", "vko_recon_all_ptr", "vko_tree_range_ptr",
                 "vko_distance_map_range_ptr", "vko_temp_range_ptr", "vko_recon_range_ptr"):
        getattr(L, name).argtypes = [p]
        getattr(L, name).restype = u8p
    L.vko_trace_ptr.argtypes = [p]
    L.vko_trace_ptr.restype = C.POINTER(TraceRec)
    L.vko_dims.argtypes = [p, C.POINTER(i64)]
    L.vko_fnv1a64.argtypes = [p, i64]
    L.vko_fnv1a64.restype = C.c_uint64
    L.vko_gen_sphere.argtypes = [p, C.c_int, C.c_int, C.c_uint32]
    L.vko_measure_max_error.argtypes = [p, p, i64]
    L.vko_measure_max_error.restype = C.c_int
    L.vko_measure_mean_error.argtypes = [p, p, i64]
    L.vko_measure_mean_error.restype = C.c_double
    L.vko_query_error.argtypes = [p, p, i64, p]
    L.vko_mid_convert_to_byte_array.argtypes = [p, p, i64]
    L.vko_mid_convert_to_byte_array.restype = i64
    L.vko_leaf_stats.argtypes = [p, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if hasattr(L, "vro_render"):
        L.vro_render.argtypes = [p, i64, i64, i64, C.POINTER(Camera), C.POINTER(RenderParams), p]
        L.vro_render.restype = C.c_int
        L.vro_composite_over.argtypes = [p, p, i64]
        L.vro_composite_finish.argtypes = [p, p, i64]
        L.vro_composite_slabs.argtypes = [p, C.c_int, i64, i64, C.c_int, C.POINTER(Camera), C.c_int, C.c_int, p]
    _lib = L
    return L


def _u8(ptr, n):
    if n <= 0:
        return np.zeros(0, np.uint8)
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).copy()


def fnv1a64(arr) -> int:
    a = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1)
    return int(lib().vko_fnv1a64(a.ctypes.data, a.size))


def gen_sphere(n, noise_mask=7, seed=12345):
    """SURVEY.md section 8d generator (sphere_n3: mask 7, sphere_n0: mask 0, random: mask>=256)."""
    v = np.empty(n * n * n, np.uint8)
    lib().vko_gen_sphere(v.ctypes.data, n, noise_mask, seed)
    return v.reshape(n, n, n)  # [z][y][x]


class OracleTree:
    """Mirror of the reference's class VolumeKdtree (VolumeKdtree_recover.h:51-264).

    `voxels` is a uint8 array in x-fastest order (shape [Z][Y][X] or flat)."""

    def __init__(self, voxels=None, dims=None, tolerance=6, max_epochs=5, guarded=False, midrange=False,
                 _handle=None):
        self._L = lib()
        if _handle is not None:
            self._h = _handle
            self._vox = None
            return
        v = np.ascontiguousarray(voxels, dtype=np.uint8)
        if dims is None:
            z, y, x = v.shape
            dims = (x, y, z)
        assert v.size == dims[0] * dims[1] * dims[2]
        self._vox = v.reshape(-1)
        self._h = self._L.vko_create(self._vox.ctypes.data, dims[0], dims[1], dims[2])
        self._L.vko_set_error_tolerance(self._h, tolerance)
        self._L.vko_set_max_epochs(self._h, max_epochs)
        self._L.vko_set_guarded(self._h, int(guarded))
        self._L.vko_set_midrange(self._h, int(midrange))

    def __del__(self):
        try:
            if self._h:
                self._L.vko_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @classmethod
    def open(cls, path):
        h = lib().vko_open(os.fsencode(path))
        if not h:
            raise FileNotFoundError(path)
        return cls(_handle=h)

    @classmethod
    def open_midrange(cls, path):
        """MidRangeTree::open restated literally (M.cpp:787-833), defect included: see kdtree_oracle.c."""
        h = lib().vko_open_midrange(os.fsencode(path))
        if not h:
            raise FileNotFoundError(path)
        return cls(_handle=h)

    def save(self, path):
        rc = self._L.vko_save(self._h, os.fsencode(path))
        if rc != 0:
            raise RuntimeError("vko_save failed: %d" % rc)

    def build(self, stage=4):
        rc = self._L.vko_build_to_stage(self._h, stage)
        if rc != 0:
            raise RuntimeError("vko_build failed: %d" % rc)
        return self

    # -- public members of the reference class
    @property
    def origTreeDepth(self): return self._L.vko_orig_tree_depth(self._h)
    @property
    def maxTreeDepth(self): return self._L.vko_max_tree_depth(self._h)
    @property
    def numActiveNodes(self): return self._L.vko_num_active_nodes(self._h)
    @property
    def numOrigNodes(self): return self._L.vko_num_orig_nodes(self._h)
    @property
    def firstOrigLeaf(self): return self._L.vko_first_orig_leaf(self._h)
    @property
    def dims(self):
        d = (C.c_int64 * 3)()
        self._L.vko_dims(self._h, d)
        return tuple(d)
    @property
    def tree(self): return _u8(self._L.vko_tree_ptr(self._h), self._L.vko_tree_bytes(self._h))
    @property
    def distanceMap(self): return _u8(self._L.vko_distance_map_ptr(self._h), self.maxTreeDepth + 1)
    @property
    def temp(self): return _u8(self._L.vko_temp_ptr(self._h), self._L.vko_temp_len(self._h))
    @property
    def recon(self): return _u8(self._L.vko_recon_ptr(self._h), self._L.vko_recon_len(self._h))
    @property
    def recon_all(self): return _u8(self._L.vko_recon_all_ptr(self._h), self.numOrigNodes)
    @property
    def tree_range(self): return _u8(self._L.vko_tree_range_ptr(self._h), self._L.vko_tree_range_bytes(self._h))
    @property
    def distanceMap_range(self): return _u8(self._L.vko_distance_map_range_ptr(self._h), self.maxTreeDepth + 1)
    @property
    def temp_range(self): return _u8(self._L.vko_temp_range_ptr(self._h), self._L.vko_temp_len(self._h))
    @property
    def recon_range(self): return _u8(self._L.vko_recon_range_ptr(self._h), self._L.vko_recon_len(self._h))
    @property
    def numReverts(self): return self._L.vko_num_reverts(self._h)
    @property
    def zeroRunRewrites(self): return self._L.vko_zero_run_rewrites(self._h)

    def trace(self):
        n = self._L.vko_trace_len(self._h)
        ptr = self._L.vko_trace_ptr(self._h)
        return [dict(depth=ptr[i].depth, epoch=ptr[i].epoch, kind=ptr[i].kind, distance=ptr[i].distance,
                     error=ptr[i].error, df=ptr[i].df, step=ptr[i].step) for i in range(n)]

    def leaf_stats(self):
        a, b = C.c_int32(), C.c_int32()
        c, d = C.c_double(), C.c_double()
        self._L.vko_leaf_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return dict(max_before=a.value, max_after=b.value, l1_before=c.value, l1_after=d.value)

    def levelCut(self, cutDepth=None, out=None):
        X, Y, Z = self.dims
        if cutDepth is None:
            cutDepth = self.maxTreeDepth
        if out is None:
            out = np.zeros(X * Y * Z, np.uint8)
        rc = self._L.vko_level_cut(self._h, cutDepth, out.ctypes.data)
        if rc != 0:
            raise RuntimeError("vko_level_cut: malformed stream (%d)" % rc)
        return out.reshape(Z, Y, X)

    def levelCutProgressive(self, cutDepth):
        """Well-defined progressive cut (not in the reference, see kdtree_oracle.c)."""
        X, Y, Z = self.dims
        out = np.zeros(X * Y * Z, np.uint8)
        rc = self._L.vko_level_cut_progressive(self._h, int(cutDepth), out.ctypes.data)
        if rc != 0:
            raise RuntimeError("vko_level_cut_progressive: malformed stream (%d)" % rc)
        return out.reshape(Z, Y, X)

    def levelCutRange(self, cutDepth=None):
        """Progressive cut of MidRangeTree's range stream (not in the reference, see kdtree_oracle.c)."""
        X, Y, Z = self.dims
        out = np.zeros(X * Y * Z, np.uint8)
        rc = self._L.vko_level_cut_range(self._h, int(self.maxTreeDepth if cutDepth is None else cutDepth), out.ctypes.data)
        if rc != 0:
            raise RuntimeError("vko_level_cut_range (%d)" % rc)
        return out.reshape(Z, Y, X)

    def convertToByteArray(self):
        n = self._L.vko_mid_convert_to_byte_array(self._h, None, 0)
        if n < 0:
            raise RuntimeError("not a MidRangeTree")
        out = np.zeros(n, np.uint8)
        self._L.vko_mid_convert_to_byte_array(self._h, out.ctypes.data, n)
        return out


def measure_max_error(decoded, original):
    a = np.ascontiguousarray(decoded, np.uint8).reshape(-1)
    b = np.ascontiguousarray(original, np.uint8).reshape(-1)
    return int(lib().vko_measure_max_error(a.ctypes.data, b.ctypes.data, a.size))


def measure_mean_error(decoded, original):
    a = np.ascontiguousarray(decoded, np.uint8).reshape(-1)
    b = np.ascontiguousarray(original, np.uint8).reshape(-1)
    return float(lib().vko_measure_mean_error(a.ctypes.data, b.ctypes.data, a.size))


def query_error(decoded, original):
    a = np.ascontiguousarray(decoded, np.uint8).reshape(-1)
    b = np.ascontiguousarray(original, np.uint8).reshape(-1)
    out = np.empty_like(a)
    lib().vko_query_error(a.ctypes.data, b.ctypes.data, a.size, out.ctypes.data)
    return out.reshape(np.shape(decoded))


def default_camera():
    """main.cpp:33-40: start camera."""
    cam = Camera()
    cam.pos[:] = (0.0, 0.0, -0.75)
    cam.front[:] = (0.0, 0.0, 1.0)
    cam.up[:] = (0.0, 1.0, 0.0)
    cam.fov_deg, cam.z_near, cam.z_far = 50.0, 0.1, 100.0
    return cam


def default_params(width, height, brick_dims=(256, 256, 128), mode=0, iso=40.0 / 255.0):
    """main.cpp:330-334 uniforms; raycaster.frag:14 MAX_SAMPLES."""
    P = RenderParams()
    P.width, P.height = width, height
    P.step_size[:] = tuple(1.0 / d for d in brick_dims)
    P.iso_value = iso
    P.max_samples = 300
    P.mode = mode
    P.box_min[:] = (0.0, 0.0, 0.0)
    P.box_max[:] = (1.0, 1.0, 1.0)
    P.global_dims[:] = (0, 0, 0)
    P.vol_origin[:] = (0, 0, 0)
    P.no_early_exit = 0
    return P


def render(volume, cam, params):
    """volume: uint8 [Z][Y][X]; returns float32 [H][W][4], row 0 = top."""
    v = np.ascontiguousarray(volume, np.uint8)
    z, y, x = v.shape
    out = np.empty((params.height, params.width, 4), np.float32)
    rc = lib().vro_render(v.ctypes.data, x, y, z, C.byref(cam), C.byref(params), out.ctypes.data)
    assert rc == 0
    return out


def composite_over(front, back):
    f = np.ascontiguousarray(front, np.float32).copy()
    b = np.ascontiguousarray(back, np.float32)
    lib().vro_composite_over(f.ctypes.data, b.ctypes.data, f.size // 4)
    return f


def composite_finish(partial):
    p_ = np.ascontiguousarray(partial, np.float32)
    out = np.empty_like(p_)
    lib().vro_composite_finish(p_.ctypes.data, out.ctypes.data, p_.size // 4)
    return out


def composite_slabs(partials, first_pixel, axis, cam, width, height):
    """partials: float32 [num_slabs][npix][4]; returns [npix][4] (reference of vr_composite_slabs)."""
    a = np.ascontiguousarray(partials, np.float32)
    out = np.empty(a.shape[1:], np.float32)
    lib().vro_composite_slabs(a.ctypes.data, a.shape[0], a.shape[1], int(first_pixel), int(axis), C.byref(cam),
                              int(width), int(height), out.ctypes.data)
    return out
