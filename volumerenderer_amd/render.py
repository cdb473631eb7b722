"""Host-side mirror of the reference's ingest + render interface over the C ABI.

  VolumeReader   volume_renderer/VolumeReader.h:28-290 (LoadBrickToTexture /
                 LoadBricksToTexture / transferToGPU; the "texture" is a device buffer)
  UnitBrick      volume_renderer/UnitBrick.h:17-119 (Setup/Bind/Draw/Unbind/Delete;
                 Draw() launches the ray-march kernel on the cube's pixel footprint)
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import Camera, RenderParams, VrError, check
from .codec import _as_dev_u8, _stream_ptr


def default_camera():
    """Start camera of main.cpp:33-40."""
    cam = Camera()
    cam.pos[:] = (0.0, 0.0, -0.75)
    cam.front[:] = (0.0, 0.0, 1.0)
    cam.up[:] = (0.0, 1.0, 0.0)
    cam.fov_deg, cam.z_near, cam.z_far = 50.0, 0.1, 100.0
    return cam


def default_params(width=1600, height=1200, brick_dims=(256, 256, 128), mode=_lib.RENDER_COMPOSITE, iso=40.0 / 255.0):
    """Uniforms set at main.cpp:330-334; MAX_SAMPLES raycaster.frag:14; window main.cpp:27."""
    P = RenderParams()
    P.width, P.height = int(width), int(height)
    P.step_size[:] = tuple(1.0 / d for d in brick_dims)
    P.iso_value = iso
    P.max_samples = 300
    P.mode = mode
    P.box_min[:] = (0.0, 0.0, 0.0)
    P.box_max[:] = (1.0, 1.0, 1.0)
    P.global_dims[:] = (0, 0, 0)
    P.vol_origin[:] = (0, 0, 0)
    P.no_early_exit = 0
    P.skip_cell = 0
    P.skip_grid_dev = None
    return P


def build_skip_grid(volume, dims, cell=8, out=None, stream=None):
    """(min, max) per cell^3 voxels (+1 voxel reach of a trilinear fetch) of a device volume: 2 bytes per cell.
    Attach to render params with use_skip_grid(); the frame stays bit-identical, the marcher just does not fetch
    samples the grid proves irrelevant."""
    v = _as_dev_u8(volume)
    d = (C.c_int64 * 3)(*[int(q) for q in dims])
    n = [(int(q) + cell - 1) // cell for q in dims]
    if out is None:
        out = torch.empty(2 * n[0] * n[1] * n[2], dtype=torch.uint8, device="cuda")
    check(_lib.lib().vr_skip_grid_build(C.c_void_p(v.data_ptr()), d, int(cell), C.c_void_p(out.data_ptr()), _stream_ptr(stream)),
          "vr_skip_grid_build")
    return out


def use_skip_grid(params, grid, cell=8):
    params.skip_cell = int(cell) if grid is not None else 0
    params.skip_grid_dev = grid.data_ptr() if grid is not None else None
    params._keep_grid = grid
    return params


def raycast(volume, dims, cam, params, out=None, stream=None):
    """volume: CUDA uint8 (X*Y*Z, x fastest). Returns float32 CUDA [H][W][4], row 0 = top."""
    v = _as_dev_u8(volume)
    d = (C.c_int64 * 3)(*[int(q) for q in dims])
    if v.numel() != d[0] * d[1] * d[2]:
        raise ValueError("volume size does not match dims")
    if out is None:
        out = torch.empty((params.height, params.width, 4), dtype=torch.float32, device="cuda")
    check(_lib.lib().vr_raycast(C.c_void_p(v.data_ptr()), d, C.byref(cam), C.byref(params),
                                C.c_void_p(out.data_ptr()), _stream_ptr(stream)), "vr_raycast")
    return out


def composite_over(front, back, stream=None):
    check(_lib.lib().vr_composite_over(C.c_void_p(front.data_ptr()), C.c_void_p(back.data_ptr()),
                                       front.numel() // 4, _stream_ptr(stream)), "vr_composite_over")
    return front


def composite_finish(partial, out=None, stream=None):
    if out is None:
        out = torch.empty_like(partial)
    check(_lib.lib().vr_composite_finish(C.c_void_p(partial.data_ptr()), C.c_void_p(out.data_ptr()),
                                         partial.numel() // 4, _stream_ptr(stream)), "vr_composite_finish")
    return out


def assemble_bricks(bricks, brick_dims, brick_ijk, grid, out=None, stream=None):
    b = _as_dev_u8(bricks)
    bd = (C.c_int64 * 3)(*[int(q) for q in brick_dims])
    g = (C.c_int64 * 3)(*[int(q) for q in grid])
    ijk = np.ascontiguousarray(brick_ijk, np.int64).reshape(-1, 3)
    nb = ijk.shape[0]
    # the volume is the whole I x J x K grid (VolumeReader.h:184-198 writes at grid coordinates); cells with no
    # brick stay zero.  (The reference sizes it by numBricks, VolumeReader.h:163-168, and overruns for sparse lists.)
    vol_bytes = g[0] * g[1] * g[2] * bd[0] * bd[1] * bd[2]
    if b.numel() < nb * bd[0] * bd[1] * bd[2]:
        raise ValueError("assemble_bricks: %d bricks need %d bytes, got %d" % (nb, nb * bd[0] * bd[1] * bd[2], b.numel()))
    if out is None:
        out = torch.zeros(vol_bytes, dtype=torch.uint8, device="cuda")
    elif out.numel() < vol_bytes:
        raise ValueError("assemble_bricks: the volume needs I*J*K*X*Y*Z = %d bytes, out has %d" % (vol_bytes, out.numel()))
    check(_lib.lib().vr_assemble_bricks(C.c_void_p(b.data_ptr()), nb, bd, ijk.ctypes.data_as(C.POINTER(C.c_int64)), g,
                                        C.c_void_p(out.data_ptr()), _stream_ptr(stream)), "vr_assemble_bricks")
    return out


def disassemble_bricks(volume, brick_dims, brick_ijk, grid, out=None, stream=None):
    v = _as_dev_u8(volume)
    bd = (C.c_int64 * 3)(*[int(q) for q in brick_dims])
    g = (C.c_int64 * 3)(*[int(q) for q in grid])
    ijk = np.ascontiguousarray(brick_ijk, np.int64).reshape(-1, 3)
    nb = ijk.shape[0]
    vol_bytes = g[0] * g[1] * g[2] * bd[0] * bd[1] * bd[2]
    if v.numel() < vol_bytes:
        raise ValueError("disassemble_bricks: the volume must hold I*J*K*X*Y*Z = %d bytes, got %d" % (vol_bytes, v.numel()))
    if out is None:
        out = torch.empty(nb * bd[0] * bd[1] * bd[2], dtype=torch.uint8, device="cuda")
    elif out.numel() < nb * bd[0] * bd[1] * bd[2]:
        raise ValueError("disassemble_bricks: out is too small")
    check(_lib.lib().vr_disassemble_bricks(C.c_void_p(v.data_ptr()), nb, bd, ijk.ctypes.data_as(C.POINTER(C.c_int64)), g,
                                           C.c_void_p(out.data_ptr()), _stream_ptr(stream)), "vr_disassemble_bricks")
    return out


def fill_volume_brick_map(ni=8, nj=8, nk=15):
    """fillVolumeBrickMap (main.cpp:599-619): brick b -> (i, j, k), i fastest."""
    m = {}
    for b in range(ni * nj * nk):
        m[b] = (b % ni, (b // ni) % nj, b // (ni * nj))
    return m


class VolumeReader:
    """volume_renderer/VolumeReader.h:28-290.  `findSourceFile(brick, timestep)` returns a
    raw brick path; `brickMap` maps brick number -> (i, j, k).  `data` is the loaded volume
    (host numpy, like the reference's std::vector), `texture` the device copy that
    transferToGPU() creates in place of the GL 3-D texture."""

    def __init__(self, brick=None, grid=None, findFileFunct=None, bMap=None):
        self.brickDims = tuple(int(v) for v in brick) if brick is not None else (0, 0, 0)
        self.findSourceFile = findFileFunct or self._undefined
        self.brickMap = bMap
        self.data = None
        self.dataDims = (0, 0, 0)
        self.texture = None     # stands in for textureId
        self._tempBrick = None

    @staticmethod
    def _undefined(brick, time):
        raise RuntimeError("\n\nERROR! findSourceFile function not defined.\n")   # VolumeReader.h:62-64

    def _load_volume_from_binary_file(self, filename):                             # VolumeReader.h:244-289
        if not os.path.exists(filename):
            return False
        expected = self.brickDims[0] * self.brickDims[1] * self.brickDims[2]
        if os.path.getsize(filename) != expected:
            raise RuntimeError("File size does not match expected dataset size!")  # :258-260
        self._tempBrick = np.fromfile(filename, dtype=np.uint8)
        return self._tempBrick.size == expected

    def LoadBrickToTexture(self, brick, timestep, dealloc, toGPU=True):            # VolumeReader.h:91-107
        ok = self._load_volume_from_binary_file(self.findSourceFile(brick, timestep))
        self.data, self._tempBrick = self._tempBrick, self.data
        self.dataDims = self.brickDims
        if ok:
            if toGPU:
                self.transferToGPU(dealloc)
        else:
            print("ERROR! Texture load failure!")
        return ok

    def transferToGPU(self, dealloc=True):                                         # VolumeReader.h:114-138
        self.texture = torch.from_numpy(self.data).cuda()
        if dealloc:
            self.data = None
            self._tempBrick = None

    def LoadBricksToTexture(self, numBricks, I, J, K, timestep, dealloc, toGPU=True):  # VolumeReader.h:151-223
        X, Y, Z = self.brickDims
        bricks = np.zeros((numBricks, Z, Y, X), np.uint8)
        ijk = np.zeros((numBricks, 3), np.int64)
        for b in range(numBricks):
            ijk[b] = self.brickMap[b]
            if not self._load_volume_from_binary_file(self.findSourceFile(b, timestep)):
                print("Load error. Brick loading terminated.")
                return False
            bricks[b] = self._tempBrick.reshape(Z, Y, X)
        # placement (VolumeReader.h:184-205) runs on the device, 64-bit indices
        vol = assemble_bricks(torch.from_numpy(bricks).cuda(), self.brickDims, ijk, (I, J, K))
        self.dataDims = (I * X, J * Y, K * Z)
        self.texture = vol
        self.data = vol.cpu().numpy()
        if toGPU and dealloc:
            self.data = None
        return True


class UnitBrick:
    """volume_renderer/UnitBrick.h:17-119.  The proxy cube is implicit in the kernel's
    ray/box set-up; Draw() is one kernel launch over the frame."""

    VERTICES = np.array([[-.5, -.5, -.5], [.5, -.5, -.5], [.5, .5, -.5], [-.5, .5, -.5],
                         [-.5, -.5, .5], [.5, -.5, .5], [.5, .5, .5], [-.5, .5, .5]], np.float32)  # UnitBrick.h:54-61

    def __init__(self):
        self._bound = False
        self.volume = None
        self.dims = None

    def Setup(self):
        self._ready = True

    def Bind(self, volume=None, dims=None):
        self._bound = True
        if volume is not None:
            self.volume, self.dims = volume, dims

    def Unbind(self):
        self._bound = False

    def Delete(self):
        self.volume = None

    def Draw(self, cam, params, out=None, stream=None):
        if not self._bound or self.volume is None:
            raise VrError(-5, "UnitBrick.Draw without Bind(volume, dims)")
        return raycast(self.volume, self.dims, cam, params, out, stream)
