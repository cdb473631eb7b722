// vrhip/Viewer.hpp -- the reference viewer's camera / key / mouse state machine without a window
// (volume_renderer/main.cpp:30-57 start values, :462-578 do_movement, key_callback, scroll_callback,
// mouse_callback, reset), driving vr_raycast and dumping frames.  No GL, no GLFW: events are fed by
// the caller (a script, a test, a remote session).  Plain C++14 over include/vrhip.h.
#pragma once
#include "../vrhip.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

namespace vrhip {

enum ViewerKey { KEY_UP, KEY_DOWN, KEY_LEFT, KEY_RIGHT, KEY_ENTER, KEY_0, KEY_1, KEY_ESCAPE, KEY_COUNT };

class HeadlessViewer {
public:
    // main.cpp:27,33-40,52
    int width = 1600, height = 1200;
    float cameraPos[3], cameraFront[3], cameraUp[3];
    float yaw = 0.0f, pitch = 0.0f, fov = 50.0f, currIsoVal = 40.0f, deltaTime = 0.0f;
    double lastX, lastY;
    bool keys[KEY_COUNT] = {false};
    bool firstMouse = true, shouldClose = false;

    HeadlessViewer(int w = 1600, int h = 1200) : width(w), height(h) { reset(); }

    void reset()                                        // main.cpp:568-578
    {
        set3(cameraPos, 0.0f, 0.0f, -0.75f); set3(cameraFront, 0.0f, 0.0f, 1.0f); set3(cameraUp, 0.0f, 1.0f, 0.0f);
        yaw = 0.0f; pitch = 0.0f; lastX = width / 2.0; lastY = height / 2.0; fov = 50.0f;
    }
    void key(ViewerKey k, bool press)                   // main.cpp:481-506
    {
        if (k == KEY_ESCAPE && press) shouldClose = true;
        if (k == KEY_ENTER && press) reset();
        if (k == KEY_0 && press) currIsoVal = std::max(0.0f, currIsoVal - 5.0f);
        if (k == KEY_1 && press) currIsoVal = std::min(255.0f, currIsoVal + 5.0f);
        keys[k] = press;
    }
    void scroll(double yoffset)                         // main.cpp:508-518
    {
        if (fov >= 1.0f && fov <= 50.0f) fov -= (float)yoffset;
        if (fov <= 1.0f) fov = 1.0f;
        if (fov >= 50.0f) fov = 50.0f;
    }
    void mouse(double xpos, double ypos, bool button1)  // main.cpp:525-566
    {
        if (!button1) { firstMouse = true; return; }
        if (firstMouse) { lastX = xpos; lastY = ypos; firstMouse = false; }
        const double xoffset = xpos - lastX, yoffset = lastY - ypos;
        lastX = xpos; lastY = ypos;
        pitch += (float)yoffset; yaw += (float)xoffset;
        pitch = std::min(89.0f, std::max(-89.0f, pitch));
        const float d2r = 0.01745329251994329576923690768489f;   // glm::radians
        float f[3] = {std::cos(pitch * d2r) * std::cos(yaw * d2r), std::sin(pitch * d2r), std::sin(yaw * d2r)};  // (sic: no cos(pitch) on z)
        normalize(f);
        set3(cameraFront, f[0], f[1], f[2]);
    }
    void advance(float dt)                              // one frame: main.cpp:382 + do_movement :462-478
    {
        deltaTime = dt;
        const float sp = 2.5f * deltaTime;
        float r[3] = {cameraFront[1] * cameraUp[2] - cameraFront[2] * cameraUp[1], cameraFront[2] * cameraUp[0] - cameraFront[0] * cameraUp[2],
                      cameraFront[0] * cameraUp[1] - cameraFront[1] * cameraUp[0]};
        normalize(r);
        for (int i = 0; i < 3; ++i) {
            if (keys[KEY_UP]) cameraPos[i] += sp * cameraFront[i];
            if (keys[KEY_DOWN]) cameraPos[i] -= sp * cameraFront[i];
            if (keys[KEY_LEFT]) cameraPos[i] -= r[i] * sp;
            if (keys[KEY_RIGHT]) cameraPos[i] += r[i] * sp;
        }
    }
    vr_camera camera() const                            // uniforms of main.cpp:396-402
    {
        vr_camera c;
        for (int i = 0; i < 3; ++i) { c.pos[i] = cameraPos[i]; c.front[i] = cameraFront[i]; c.up[i] = cameraUp[i]; }
        c.fov_deg = fov; c.z_near = 0.1f; c.z_far = 100.0f;
        return c;
    }
    // one frame of device volume `vol` (dims x,y,z) into `rgba_dev` (width*height float4), vr_raycast
    vr_status draw(const uint8_t *vol, const int64_t dims[3], vr_render_params P, float *rgba_dev, void *stream = nullptr) const
    {
        P.width = width; P.height = height; P.iso_value = currIsoVal / 255.0f;
        const vr_camera c = camera();
        return vr_raycast(vol, dims, &c, &P, rgba_dev, stream);
    }
    // binary PPM of a host float RGBA frame (what glReadPixels of the 8-bit framebuffer would hold)
    static bool dumpPPM(const std::string &path, const std::vector<float> &rgba, int w, int h)
    {
        FILE *f = std::fopen(path.c_str(), "wb");
        if (!f) return false;
        std::fprintf(f, "P6\n%d %d\n255\n", w, h);
        for (size_t p = 0; p < (size_t)w * h; ++p) {
            unsigned char px[3];
            for (int c = 0; c < 3; ++c) px[c] = (unsigned char)std::lround(255.0f * std::min(1.0f, std::max(0.0f, rgba[4 * p + c])));
            std::fwrite(px, 1, 3, f);
        }
        return std::fclose(f) == 0;
    }

private:
    static void set3(float *v, float a, float b, float c) { v[0] = a; v[1] = b; v[2] = c; }
    static void normalize(float *v)
    {
        const float l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (l > 0.0f) { v[0] /= l; v[1] /= l; v[2] /= l; }
    }
};

} // namespace vrhip
