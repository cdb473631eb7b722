"""Headless mirror of the reference viewer's camera / key / mouse state machine
(volume_renderer/main.cpp:30-57 start values, :462-578 do_movement / key_callback / scroll_callback /
mouse_callback / reset): events come from the caller, frames from vr_raycast.  Same fields and
arithmetic (float32) as include/vrhip/Viewer.hpp."""
import math

import numpy as np

from . import _lib
from .render import default_camera, default_params, raycast

KEYS = ("UP", "DOWN", "LEFT", "RIGHT", "ENTER", "0", "1", "ESCAPE")
_f = np.float32


class HeadlessViewer:
    def __init__(self, width=1600, height=1200):
        self.width, self.height = int(width), int(height)
        self.currIsoVal = _f(40.0)            # main.cpp:52
        self.deltaTime = _f(0.0)
        self.keys = {k: False for k in KEYS}
        self.firstMouse = True
        self.shouldClose = False
        self.reset()

    def reset(self):                          # main.cpp:568-578
        self.cameraPos = np.array([0.0, 0.0, -0.75], _f)
        self.cameraFront = np.array([0.0, 0.0, 1.0], _f)
        self.cameraUp = np.array([0.0, 1.0, 0.0], _f)
        self.yaw, self.pitch, self.fov = _f(0.0), _f(0.0), _f(50.0)
        self.lastX, self.lastY = self.width / 2.0, self.height / 2.0

    def key(self, k, press):                  # main.cpp:481-506
        if k == "ESCAPE" and press: self.shouldClose = True
        if k == "ENTER" and press: self.reset()
        if k == "0" and press: self.currIsoVal = max(_f(0.0), _f(self.currIsoVal - _f(5.0)))
        if k == "1" and press: self.currIsoVal = min(_f(255.0), _f(self.currIsoVal + _f(5.0)))
        self.keys[k] = bool(press)

    def scroll(self, yoffset):                # main.cpp:508-518
        if 1.0 <= self.fov <= 50.0: self.fov = _f(self.fov - _f(yoffset))
        if self.fov <= 1.0: self.fov = _f(1.0)
        if self.fov >= 50.0: self.fov = _f(50.0)

    def mouse(self, xpos, ypos, button1):     # main.cpp:525-566
        if not button1:
            self.firstMouse = True
            return
        if self.firstMouse:
            self.lastX, self.lastY, self.firstMouse = xpos, ypos, False
        xoffset, yoffset = xpos - self.lastX, self.lastY - ypos
        self.lastX, self.lastY = xpos, ypos
        self.pitch = _f(self.pitch + _f(yoffset)); self.yaw = _f(self.yaw + _f(xoffset))
        self.pitch = min(_f(89.0), max(_f(-89.0), self.pitch))
        d2r = _f(0.01745329251994329576923690768489)
        p, y = _f(self.pitch * d2r), _f(self.yaw * d2r)
        f = np.array([_f(math.cos(p)) * _f(math.cos(y)), _f(math.sin(p)), _f(math.sin(y))], _f)   # (sic) z without cos(pitch)
        n = _f(np.sqrt(_f(f[0] * f[0] + f[1] * f[1] + f[2] * f[2])))
        self.cameraFront = (f / n).astype(_f) if n > 0 else f

    def advance(self, dt):                    # main.cpp:382, :462-478
        self.deltaTime = _f(dt)
        sp = _f(_f(2.5) * self.deltaTime)
        r = np.cross(self.cameraFront, self.cameraUp).astype(_f)
        n = _f(np.sqrt(_f(np.dot(r, r))))
        if n > 0: r = (r / n).astype(_f)
        if self.keys["UP"]: self.cameraPos = (self.cameraPos + sp * self.cameraFront).astype(_f)
        if self.keys["DOWN"]: self.cameraPos = (self.cameraPos - sp * self.cameraFront).astype(_f)
        if self.keys["LEFT"]: self.cameraPos = (self.cameraPos - r * sp).astype(_f)
        if self.keys["RIGHT"]: self.cameraPos = (self.cameraPos + r * sp).astype(_f)

    def camera(self):
        cam = default_camera()
        cam.pos[:] = tuple(float(v) for v in self.cameraPos)
        cam.front[:] = tuple(float(v) for v in self.cameraFront)
        cam.up[:] = tuple(float(v) for v in self.cameraUp)
        cam.fov_deg = float(self.fov)
        return cam

    def draw(self, volume, dims, brick_dims=(256, 256, 128), mode=_lib.RENDER_COMPOSITE, out=None):
        P = default_params(self.width, self.height, brick_dims, mode, float(self.currIsoVal) / 255.0)
        return raycast(volume, dims, self.camera(), P, out)

    @staticmethod
    def dump_ppm(path, rgba):
        """rgba: (H, W, 4) float array/tensor in [0,1] -> binary PPM (8-bit, like the framebuffer)."""
        a = rgba.detach().cpu().numpy() if hasattr(rgba, "detach") else np.asarray(rgba)
        h, w = a.shape[:2]
        px = np.rint(255.0 * np.clip(a[..., :3], 0.0, 1.0)).astype(np.uint8)
        with open(path, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (w, h))
            f.write(px.tobytes())
