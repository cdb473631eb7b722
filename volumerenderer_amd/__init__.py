"""volumerenderer_amd -- MI355X-native kd-tree volume codec + ray-march compositor.

The product is volumerenderer_amd/libvrhip.so (hand-written HIP for gfx950 behind the
C ABI of include/vrhip.h).  This package is the Python host-side mirror of the
reference's class interface, used by the parity tests and the bench."""
from ._lib import (RENDER_COMPOSITE, RENDER_ISOSURFACE, RENDER_PARTIAL, VARIANT_GUARDED, VARIANT_MIDRANGE,
                   VARIANT_RECOVER, Camera, RenderParams, VrError)

__all__ = ["BrickSet", "VolumeKdtree", "MidRangeTree", "HashedKdtree", "VolumeReader", "UnitBrick", "VrError", "Camera",
           "RenderParams"]


def __getattr__(name):  # torch is imported lazily so that `import volumerenderer_amd` stays cheap
    if name in ("BrickSet", "VolumeKdtree", "MidRangeTree", "HashedKdtree", "measure_error", "query_error"):
        from . import codec
        return getattr(codec, name)
    if name in ("VolumeReader", "UnitBrick", "raycast", "default_camera", "default_params", "composite_over",
                "composite_finish", "assemble_bricks", "disassemble_bricks", "fill_volume_brick_map", "build_skip_grid",
                "use_skip_grid"):
        from . import render
        return getattr(render, name)
    raise AttributeError(name)
