"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol
include/vrhip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as g
    g.build()
    from volumerenderer_amd import _lib
    return _lib.lib()


def test_header_symbols_all_exported(L):
    from volumerenderer_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "vrhip.h")).read()
    declared = set(re.findall(r"\b(vr_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), "libvrhip.so does not export %s" % name
    # and the binding covers exactly the header
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_struct_layouts_match_header():
    from volumerenderer_amd import _lib
    assert C.sizeof(_lib.TreeInfo) == 88
    assert C.sizeof(_lib.Camera) == 48
    assert C.sizeof(_lib.RenderParams) == 32 + 24 + 48 + 8 + 8


def test_status_strings(L):
    assert L.vr_status_string(0) == b"ok"
    assert b"no CPU fallback" in L.vr_status_string(-2)
    assert L.vr_version().startswith(b"vrhip")


def test_argument_validation_and_no_cpu_fallback(L):
    n = C.c_int32(-1)
    assert L.vr_device_count(C.byref(n)) == 0
    h = C.c_void_p()
    dims = (C.c_int64 * 3)(16, 16, 16)
    bad = (C.c_int64 * 3)(1 << 21, 16, 16)       # an axis beyond 2^20 (or 2^31 voxels and more): unsupported
    assert L.vr_brickset_create(None, 1, dims, 1, 2, 0) == -1           # VR_ERR_INVALID
    assert L.vr_brickset_create(C.byref(h), 1, dims, -1, 2, 0) == -1    # negative tolerance
    assert L.vr_brickset_create(C.byref(h), 1, bad, 1, 2, 0) == -7      # VR_ERR_UNSUPPORTED
    if n.value == 0:
        # CPU-only box: every compute entry point must fail loudly
        assert L.vr_brickset_create(C.byref(h), 1, dims, 1, 2, 0) == -2  # VR_ERR_NO_DEVICE
        assert L.vr_set_device(0) == -2
        buf = (C.c_uint8 * 16)()
        assert L.vr_query_error(buf, buf, 16, buf, None) == -2
        assert L.vr_measure_error(buf, buf, 16, None, None, None) == -2
        assert L.vr_composite_over(buf, buf, 1, None) == -2


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under volumerenderer_amd/ or include/ may reference it."""
    for base in ("volumerenderer_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    for needle in ("import oracle", "from oracle", "liboracle", "oracle/"):
                        assert needle not in txt, "%s references the oracle (%s)" % (f, needle)
