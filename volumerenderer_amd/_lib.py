"""ctypes binding of the C ABI in include/vrhip.h (libvrhip.so).

The library is the product: if it is missing this module raises -- there is no
Python / CPU fallback for any compute entry point."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvrhip.so")

VR_OK = 0
STATUS = {0: "VR_OK", -1: "VR_ERR_INVALID", -2: "VR_ERR_NO_DEVICE", -3: "VR_ERR_OOM", -4: "VR_ERR_IO",
          -5: "VR_ERR_STATE", -6: "VR_ERR_FORMAT", -7: "VR_ERR_UNSUPPORTED"}
VARIANT_RECOVER, VARIANT_GUARDED, VARIANT_MIDRANGE = 0, 1, 2
RENDER_COMPOSITE, RENDER_ISOSURFACE, RENDER_PARTIAL = 0, 1, 2


class VrError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        super().__init__("%s: %s (%d)" % (what, STATUS.get(status, "?"), status))


class TreeInfo(C.Structure):
    _fields_ = [("X", C.c_int64), ("Y", C.c_int64), ("Z", C.c_int64),
                ("orig_tree_depth", C.c_int32), ("max_tree_depth", C.c_int32),
                ("num_active_nodes", C.c_int64), ("tree_bytes", C.c_int64),
                ("tolerance", C.c_int32), ("max_epochs", C.c_int32), ("variant", C.c_int32),
                ("num_reverts", C.c_int32), ("max_error_before", C.c_int32), ("max_error_after", C.c_int32),
                ("mean_l1_after", C.c_double), ("zero_run_rewrites", C.c_int32), ("est_exact_segments", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("front", C.c_float * 3), ("up", C.c_float * 3),
                ("fov_deg", C.c_float), ("z_near", C.c_float), ("z_far", C.c_float)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("step_size", C.c_float * 3),
                ("iso_value", C.c_float), ("max_samples", C.c_int32), ("mode", C.c_int32),
                ("box_min", C.c_float * 3), ("box_max", C.c_float * 3),
                ("global_dims", C.c_int64 * 3), ("vol_origin", C.c_int64 * 3),
                ("no_early_exit", C.c_int32), ("skip_cell", C.c_int32), ("skip_grid_dev", C.c_void_p)]


# every symbol include/vrhip.h declares, with its signature
_P, _I32, _I64 = C.c_void_p, C.c_int32, C.c_int64
SIGNATURES = {
    "vr_device_count": (_I32, [C.POINTER(_I32)]),
    "vr_set_device": (_I32, [_I32]),
    "vr_status_string": (C.c_char_p, [_I32]),
    "vr_version": (C.c_char_p, []),
    "vr_malloc": (_I32, [C.POINTER(_P), _I64]),
    "vr_free": (_I32, [_P]),
    "vr_upload": (_I32, [_P, _P, _I64, _P]),
    "vr_download": (_I32, [_P, _P, _I64, _P]),
    "vr_brickset_create": (_I32, [C.POINTER(_P), _I32, C.POINTER(_I64), _I32, _I32, _I32]),
    "vr_brickset_destroy": (_I32, [_P]),
    "vr_brickset_set_error_tolerance": (_I32, [_P, _I32]),
    "vr_brickset_set_max_epochs": (_I32, [_P, _I32]),
    "vr_brickset_build": (_I32, [_P, _P, _P]),
    "vr_brickset_info": (_I32, [_P, _I32, C.POINTER(TreeInfo)]),
    "vr_brickset_get_tree": (_I32, [_P, _I32, _P, _I64]),
    "vr_brickset_get_distance_map": (_I32, [_P, _I32, _P, _I32]),
    "vr_brickset_get_tree_range": (_I32, [_P, _I32, _P, _I64]),
    "vr_brickset_get_distance_map_range": (_I32, [_P, _I32, _P, _I32]),
    "vr_brickset_get_packed4": (_I32, [_P, _I32, _P, _I64, C.POINTER(_I64)]),
    "vr_brickset_decode": (_I32, [_P, _I32, _P, _P]),
    "vr_brickset_decode_range": (_I32, [_P, _I32, _P, _P]),
    "vr_brickset_set_tree": (_I32, [_P, _I32, _P, _I64, _I64, _P, _I32]),
    "vr_brickset_save": (_I32, [_P, _I32, C.c_char_p]),
    "vr_brickset_open": (_I32, [C.POINTER(_P), C.c_char_p]),
    "vr_brickset_open_variant": (_I32, [C.POINTER(_P), C.c_char_p, _I32]),
    "vr_measure_error": (_I32, [_P, _P, _I64, C.POINTER(_I32), C.POINTER(C.c_double), _P]),
    "vr_query_error": (_I32, [_P, _P, _I64, _P, _P]),
    "vr_assemble_bricks": (_I32, [_P, _I32, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64), _P, _P]),
    "vr_disassemble_bricks": (_I32, [_P, _I32, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64), _P, _P]),
    "vr_raycast": (_I32, [_P, C.POINTER(_I64), C.POINTER(Camera), C.POINTER(RenderParams), _P, _P]),
    "vr_skip_grid_build": (_I32, [_P, C.POINTER(_I64), _I32, _P, _P]),
    "vr_composite_over": (_I32, [_P, _P, _I64, _P]),
    "vr_composite_finish": (_I32, [_P, _P, _I64, _P]),
    "vr_composite_slabs": (_I32, [_P, _I32, _I64, _I64, _I32, C.POINTER(Camera), C.POINTER(RenderParams), _P, _P]),
    "vr_rccl_unique_id": (_I32, [_P]),
    "vr_compositor_create": (_I32, [C.POINTER(_P), _P, _I32, _I32, _I32, _I32]),
    "vr_compositor_create_from_comm": (_I32, [C.POINTER(_P), _P, _I32, _I32, _I32, _I32]),
    "vr_compositor_composite": (_I32, [_P, _P, _I32, C.POINTER(Camera), C.POINTER(RenderParams), _P, _P]),
    "vr_compositor_destroy": (_I32, [_P]),
    "vr_stream_create": (_I32, [C.POINTER(_P)]),
    "vr_stream_destroy": (_I32, [_P]),
    "vr_stream_synchronize": (_I32, [_P]),
    "vr_stream_wait_event": (_I32, [_P, _P]),
    "vr_event_create": (_I32, [C.POINTER(_P)]),
    "vr_event_destroy": (_I32, [_P]),
    "vr_event_record": (_I32, [_P, _P]),
    "vr_event_synchronize": (_I32, [_P]),
    "vr_event_elapsed_ms": (_I32, [_P, _P, C.POINTER(C.c_float)]),
    "vr_malloc_host": (_I32, [C.POINTER(_P), _I64]),
    "vr_free_host": (_I32, [_P]),
    "vr_upload_async": (_I32, [_P, _P, _I64, _P]),
    "vr_download_async": (_I32, [_P, _P, _I64, _P]),
    "vr_brickset_last_timings": (_I32, [_P, C.POINTER(C.c_float)]),
    "vr_brickset_set_concurrency": (_I32, [_P, C.c_int32]),
    "vr_brickset_set_switch": (_I32, [_P, C.c_char_p, C.c_int32]),
    "vr_brickset_set_compaction": (_I32, [_P, C.c_int32]),
    "vr_debug_set": (_I32, [C.c_char_p, C.c_int32]),
}

_lib = None


def lib():
    """Loads libvrhip.so.  Raises ImportError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libvrhip.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # PyTorch ships its own libamdhip64; it must be in the process BEFORE libvrhip.so is
        # loaded so that both bind to ONE HIP runtime (device pointers and streams are then
        # interchangeable).  Loading libvrhip.so first would start a second runtime that cannot
        # see the device torch holds.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, what=""):
    if status != VR_OK:
        raise VrError(status, what)
