/*
 * raymarch_oracle.c -- TEST INFRASTRUCTURE ONLY (see kdtree_oracle.c header).
 *
 * CPU float32 restatement of the reference's two fragment shaders and of the GL
 * state they run under:
 *   /root/reference/volume_renderer/raycaster.vert:10-21, raycaster.frag:18-86
 *   /root/reference/volume_renderer/isosurface.vert:10-21, isosurface.frag:23-159
 *   proxy geometry  UnitBrick.h:54-75 (unit cube [-0.5,0.5]^3, 12 triangles)
 *   uniforms        main.cpp:319-341,396-402 ; GL state main.cpp:367-369,392
 *   texture state   VolumeReader.h:114-127 (R8, GL_LINEAR, GL_CLAMP)
 *
 * PARITY UNPINNED against a real OpenGL driver: the GLSL cannot run here (no GL,
 * no GPU), the reference holds no golden images, and the GLM version is not
 * pinned.  Defined choices: clamp-to-edge sampling with
 * float32 weights, accumulator starts at 0, blue = 255 clamps to 1.0, identity
 * TransformationMatrix, RH lookAt / perspectiveFov with -1..1 depth.  What the
 * tests pin is (a) analytic cases and (b) GPU kernel == this restatement within
 * 2e-3 per channel.
 *
 * The rasteriser is replaced by its per-pixel equivalent: the fragment that
 * survives GL_LESS depth testing without culling is the nearest point of the cube
 * surface along the pixel's view ray inside the [near, far] range; the
 * perspective-correct interpolant vUV there is exactly (hit point + 0.5).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float pos[3], front[3], up[3]; float fov_deg, z_near, z_far; } vro_camera;
typedef struct {
    int32_t width, height;
    float step_size[3];
    float iso_value;
    int32_t max_samples;
    int32_t mode; /* 0 composite, 1 isosurface, 2 partial (c, tau, covered, 0) */
    float box_min[3], box_max[3];
    int64_t global_dims[3];
    int64_t vol_origin[3];
    int32_t no_early_exit;
    int32_t reserved;
} vro_params;

typedef struct { const uint8_t *v; int64_t X, Y, Z; int64_t GX, GY, GZ; int64_t ox, oy, oz; } vro_tex;

static inline int64_t clampi(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* texture(volume, p).r : GL_LINEAR, texel centres at (i+0.5)/N, clamp to edge */
static float tex3d(const vro_tex *t, float px, float py, float pz)
{
    float x = px * (float)t->GX - 0.5f, y = py * (float)t->GY - 0.5f, z = pz * (float)t->GZ - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y), fz0 = floorf(z);
    float fx = x - fx0, fy = y - fy0, fz = z - fz0;
    int64_t x0 = (int64_t)fx0, y0 = (int64_t)fy0, z0 = (int64_t)fz0;
    int64_t xa = clampi(clampi(x0, 0, t->GX - 1) - t->ox, 0, t->X - 1), xb = clampi(clampi(x0 + 1, 0, t->GX - 1) - t->ox, 0, t->X - 1);
    int64_t ya = clampi(clampi(y0, 0, t->GY - 1) - t->oy, 0, t->Y - 1), yb = clampi(clampi(y0 + 1, 0, t->GY - 1) - t->oy, 0, t->Y - 1);
    int64_t za = clampi(clampi(z0, 0, t->GZ - 1) - t->oz, 0, t->Z - 1), zb = clampi(clampi(z0 + 1, 0, t->GZ - 1) - t->oz, 0, t->Z - 1);
    const float k = 1.0f / 255.0f;
#define VX(a, b, c) ((float)t->v[(a) + t->X * ((b) + t->Y * (c))] * k)
    float c000 = VX(xa, ya, za), c100 = VX(xb, ya, za), c010 = VX(xa, yb, za), c110 = VX(xb, yb, za);
    float c001 = VX(xa, ya, zb), c101 = VX(xb, ya, zb), c011 = VX(xa, yb, zb), c111 = VX(xb, yb, zb);
#undef VX
    float c00 = c000 + fx * (c100 - c000), c10 = c010 + fx * (c110 - c010);
    float c01 = c001 + fx * (c101 - c001), c11 = c011 + fx * (c111 - c011);
    float c0 = c00 + fy * (c10 - c00), c1 = c01 + fy * (c11 - c01);
    return c0 + fz * (c1 - c0);
}

static void norm3(float *v)
{
    float l = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (l > 0.0f) { v[0] /= l; v[1] /= l; v[2] /= l; }
    else { v[0] = v[1] = v[2] = 0.0f; }
}
static void cross3(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static inline float sgn(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }
static inline int inside(const float *p)
{   /* dot(sign(p - 0), sign(1 - p)) < 3 -> stop  (raycaster.frag:51) */
    float d = sgn(p[0]) * sgn(1.0f - p[0]) + sgn(p[1]) * sgn(1.0f - p[1]) + sgn(p[2]) * sgn(1.0f - p[2]);
    return !(d < 3.0f);
}

/* out: height*width*4 float, row 0 = top of the image. */
int vro_render(const uint8_t *vol, int64_t X, int64_t Y, int64_t Z, const vro_camera *cam, const vro_params *P, float *out)
{
    vro_tex t;
    t.v = vol; t.X = X; t.Y = Y; t.Z = Z;
    t.GX = P->global_dims[0] > 0 ? P->global_dims[0] : X;
    t.GY = P->global_dims[1] > 0 ? P->global_dims[1] : Y;
    t.GZ = P->global_dims[2] > 0 ? P->global_dims[2] : Z;
    t.ox = P->vol_origin[0]; t.oy = P->vol_origin[1]; t.oz = P->vol_origin[2];
    const int W = P->width, H = P->height;
    /* glm::lookAt(pos, pos+front, up) basis (main.cpp:396) */
    float f[3] = { cam->front[0], cam->front[1], cam->front[2] }, s[3], u[3];
    norm3(f);
    cross3(f, cam->up, s);
    norm3(s);
    cross3(s, f, u);
    /* glm::perspectiveFov(radians(fov), w, h, near, far) (main.cpp:397) */
    const float rad = cam->fov_deg * 0.01745329251994329576923690768489f;
    const float tanY = tanf(0.5f * rad);
    const float tanX = tanY * (float)W / (float)H;
    for (int py = 0; py < H; ++py) {
        for (int px = 0; px < W; ++px) {
            float *o = out + 4 * ((size_t)py * W + px);
            float nx = 2.0f * ((float)px + 0.5f) / (float)W - 1.0f;
            float ny = 1.0f - 2.0f * ((float)py + 0.5f) / (float)H;
            float dir[3];
            for (int k = 0; k < 3; ++k) dir[k] = f[k] + nx * tanX * s[k] + ny * tanY * u[k];
            /* ray / unit cube slabs */
            float t0 = -INFINITY, t1 = INFINITY;
            int miss = 0;
            for (int k = 0; k < 3; ++k) {
                if (dir[k] != 0.0f) {
                    float a = (-0.5f - cam->pos[k]) / dir[k], b = (0.5f - cam->pos[k]) / dir[k];
                    if (a > b) { float q = a; a = b; b = q; }
                    if (a > t0) t0 = a;
                    if (b < t1) t1 = b;
                } else if (cam->pos[k] < -0.5f || cam->pos[k] > 0.5f) miss = 1;
            }
            float th = t0 >= cam->z_near ? t0 : t1; /* front face, else (camera inside / clipped) back face */
            if (miss || t0 > t1 || th < cam->z_near || th > cam->z_far) {
                if (P->mode == 2) { o[0] = 0.0f; o[1] = 1.0f; o[2] = 0.0f; o[3] = 0.0f; }
                else { o[0] = o[1] = o[2] = o[3] = 1.0f; } /* glClearColor(255,255,255,1) clamps to white */
                continue;
            }
            float vuv[3], gd[3], step[3], pos[3];
            for (int k = 0; k < 3; ++k) vuv[k] = (cam->pos[k] + th * dir[k]) + 0.5f;
            for (int k = 0; k < 3; ++k) gd[k] = (vuv[k] - 0.5f) - cam->pos[k];
            norm3(gd);
            for (int k = 0; k < 3; ++k) { step[k] = gd[k] * P->step_size[k]; pos[k] = vuv[k]; }
            if (P->mode == 0) { /* raycaster.frag:37-85 */
                float rgb = 0.0f, A = 0.0f;
                for (int i = 0; i < P->max_samples; ++i) {
                    for (int k = 0; k < 3; ++k) pos[k] = pos[k] + step[k];
                    if (!inside(pos)) break;
                    float smp = tex3d(&t, pos[0], pos[1], pos[2]);
                    float pa = smp - (smp * A);
                    rgb = pa * smp + rgb;
                    A += pa * 0.6f;
                    if (!P->no_early_exit && A > 0.99f) break;
                }
                o[0] = 1.0f - rgb; o[1] = 1.0f - rgb; o[2] = 1.0f; o[3] = A;
            } else if (P->mode == 2) { /* same accumulation as an associative (c, tau) pair, own sub-box only */
                float c = 0.0f, tau = 1.0f;
                for (int i = 0; i < P->max_samples; ++i) {
                    for (int k = 0; k < 3; ++k) pos[k] = pos[k] + step[k];
                    if (!inside(pos)) break;
                    int own = 1;
                    for (int k = 0; k < 3; ++k) if (!(pos[k] >= P->box_min[k] && pos[k] < P->box_max[k])) own = 0;
                    if (!own) continue;
                    float smp = tex3d(&t, pos[0], pos[1], pos[2]);
                    c = c + tau * (smp * smp);
                    tau = tau * (1.0f - 0.6f * smp);
                }
                o[0] = c; o[1] = tau; o[2] = 1.0f; o[3] = 0.0f;
            } else { /* isosurface.frag:77-159 */
                float col[4] = { 1.0f, 1.0f, 1.0f, 1.0f }; /* vec4(255,255,255,1) clamped */
                for (int i = 0; i < P->max_samples; ++i) {
                    for (int k = 0; k < 3; ++k) pos[k] = pos[k] + step[k];
                    if (!inside(pos)) break;
                    float s1 = tex3d(&t, pos[0], pos[1], pos[2]);
                    float s2 = tex3d(&t, pos[0] + step[0], pos[1] + step[1], pos[2] + step[2]);
                    if ((s1 - P->iso_value) < 0.0f && (s2 - P->iso_value) >= 0.0f) {
                        float l[3] = { pos[0], pos[1], pos[2] }, r[3] = { pos[0] + step[0], pos[1] + step[1], pos[2] + step[2] };
                        for (int b = 0; b < 4; ++b) { /* Bisection :23-42 */
                            float m[3] = { (r[0] + l[0]) * 0.5f, (r[1] + l[1]) * 0.5f, (r[2] + l[2]) * 0.5f };
                            float cm = tex3d(&t, m[0], m[1], m[2]);
                            if (cm < P->iso_value) { l[0] = m[0]; l[1] = m[1]; l[2] = m[2]; }
                            else { r[0] = m[0]; r[1] = m[1]; r[2] = m[2]; }
                        }
                        float tc[3] = { (r[0] + l[0]) * 0.5f, (r[1] + l[1]) * 0.5f, (r[2] + l[2]) * 0.5f };
                        const float DELTA = 0.01f; /* GetGradient :47-62 */
                        float N[3];
                        N[0] = (tex3d(&t, tc[0] - DELTA, tc[1], tc[2]) - tex3d(&t, tc[0] + DELTA, tc[1], tc[2])) / 2.0f;
                        N[1] = (tex3d(&t, tc[0], tc[1] - DELTA, tc[2]) - tex3d(&t, tc[0], tc[1] + DELTA, tc[2])) / 2.0f;
                        N[2] = (tex3d(&t, tc[0], tc[1], tc[2] - DELTA) - tex3d(&t, tc[0], tc[1], tc[2] + DELTA)) / 2.0f;
                        norm3(N); /* zero gradient: defined as N = 0 (GLSL leaves it undefined) */
                        float Vv[3] = { -gd[0], -gd[1], -gd[2] };
                        float diffuse = fmaxf(Vv[0] * N[0] + Vv[1] * N[1] + Vv[2] * N[2], 0.0f);
                        float hv[3] = { Vv[0] + Vv[0], Vv[1] + Vv[1], Vv[2] + Vv[2] };
                        norm3(hv);
                        float spec = powf(fmaxf(0.00001f, hv[0] * N[0] + hv[1] * N[1] + hv[2] * N[2]), 250.0f);
                        col[0] = fminf(1.0f, diffuse * 0.39f + spec);
                        col[1] = fminf(1.0f, diffuse * 0.58f + spec);
                        col[2] = fminf(1.0f, diffuse * 0.93f + spec);
                        col[3] = 1.0f;
                        break;
                    }
                }
                o[0] = col[0]; o[1] = col[1]; o[2] = col[2]; o[3] = col[3];
            }
        }
    }
    return 0;
}

/* front OVER back on (c, tau, covered) partial pixels: (c1 + t1*c2, t1*t2) */
void vro_composite_over(float *front, const float *back, int64_t npix)
{
    for (int64_t i = 0; i < npix; ++i) {
        float c1 = front[4 * i], t1 = front[4 * i + 1], c2 = back[4 * i], t2 = back[4 * i + 1];
        front[4 * i] = c1 + t1 * c2;
        front[4 * i + 1] = t1 * t2;
        front[4 * i + 2] = fmaxf(front[4 * i + 2], back[4 * i + 2]);
    }
}

/* colour transfer of raycaster.frag:82-85 on a composited partial image */
void vro_composite_finish(const float *partial, float *rgba, int64_t npix)
{
    for (int64_t i = 0; i < npix; ++i) {
        if (partial[4 * i + 2] > 0.0f) {
            rgba[4 * i] = 1.0f - partial[4 * i]; rgba[4 * i + 1] = 1.0f - partial[4 * i];
            rgba[4 * i + 2] = 1.0f; rgba[4 * i + 3] = 1.0f - partial[4 * i + 1];
        } else { rgba[4 * i] = rgba[4 * i + 1] = rgba[4 * i + 2] = rgba[4 * i + 3] = 1.0f; }
    }
}

/* per-pixel view-ordered composite of num_slabs partial tiles (test reference of vr_composite_slabs) */
void vro_composite_slabs(const float *partials, int num_slabs, int64_t npix, int64_t first, int axis,
                         const vro_camera *cam, int W, int H, float *rgba)
{
    float f[3] = { cam->front[0], cam->front[1], cam->front[2] }, s[3], u[3];
    norm3(f);
    cross3(f, cam->up, s);
    norm3(s);
    cross3(s, f, u);
    const float rad = cam->fov_deg * 0.01745329251994329576923690768489f;
    const float tanY = tanf(0.5f * rad), tanX = tanY * (float)W / (float)H;
    for (int64_t i = 0; i < npix; ++i) {
        int64_t gp = first + i;
        int px = (int)(gp % W), py = (int)(gp / W);
        float nx = 2.0f * ((float)px + 0.5f) / (float)W - 1.0f, ny = 1.0f - 2.0f * ((float)py + 0.5f) / (float)H;
        float d = f[axis] + nx * tanX * s[axis] + ny * tanY * u[axis];
        float c = 0.0f, tau = 1.0f, cov = 0.0f;
        for (int k = 0; k < num_slabs; ++k) {
            int sidx = d >= 0.0f ? k : num_slabs - 1 - k;
            const float *p = partials + 4 * ((int64_t)sidx * npix + i);
            c = c + tau * p[0]; tau = tau * p[1]; cov = fmaxf(cov, p[2]);
        }
        float *o = rgba + 4 * i;
        if (cov > 0.0f) { o[0] = 1.0f - c; o[1] = 1.0f - c; o[2] = 1.0f; o[3] = 1.0f - tau; }
        else { o[0] = o[1] = o[2] = o[3] = 1.0f; }
    }
}
