// Replays a fixed event script through vrhip::HeadlessViewer and prints the camera state after every
// event (tests/test_viewer.py compares it with the Python mirror).  Host logic only: no GPU needed.
#include "vrhip/Viewer.hpp"
#include <cstdio>

static void show(const char *what, const vrhip::HeadlessViewer &v)
{
    std::printf("%s %.9g %.9g %.9g | %.9g %.9g %.9g | %.9g %.9g %.9g %.9g %d\n", what, v.cameraPos[0], v.cameraPos[1], v.cameraPos[2],
                v.cameraFront[0], v.cameraFront[1], v.cameraFront[2], v.yaw, v.pitch, v.fov, v.currIsoVal, (int)v.shouldClose);
}

int main()
{
    using namespace vrhip;
    HeadlessViewer v(1600, 1200);
    show("start", v);
    v.key(KEY_UP, true); v.advance(0.016f); show("up", v);
    v.key(KEY_UP, false); v.key(KEY_LEFT, true); v.advance(0.033f); show("left", v);
    v.key(KEY_LEFT, false);
    v.mouse(800.0, 600.0, true); v.mouse(830.0, 570.0, true); show("drag", v);
    v.mouse(830.0, 400.0, true); show("drag2", v);               // pitch clamps at 89
    v.mouse(0.0, 0.0, false); v.mouse(100.0, 100.0, true); show("regrab", v);
    v.key(KEY_DOWN, true); v.key(KEY_RIGHT, true); v.advance(0.02f); show("downright", v);
    v.key(KEY_DOWN, false); v.key(KEY_RIGHT, false);
    v.scroll(3.0); show("scroll", v);
    for (int i = 0; i < 60; ++i) v.scroll(1.0);
    show("scrollmin", v);
    v.scroll(-100.0); show("scrollmax", v);
    for (int i = 0; i < 10; ++i) v.key(KEY_0, true);
    show("isomin", v);
    for (int i = 0; i < 60; ++i) v.key(KEY_1, true);
    show("isomax", v);
    v.key(KEY_ENTER, true); show("reset", v);
    v.key(KEY_ESCAPE, true); show("escape", v);
    return 0;
}
