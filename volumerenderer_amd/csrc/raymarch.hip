// raymarch.hip -- gfx950 kernels for the front-to-back ray-marching compositor
// (reference volume_renderer/raycaster.frag:18-86), the iso-surface marcher
// (volume_renderer/isosurface.frag:23-159), the vertex stage / proxy cube they run on
// (raycaster.vert:10-21, UnitBrick.h:54-100, main.cpp:396-402), plus the small
// data-parallel helpers of the path: brick assembly (VolumeReader.h:151-223), error
// metrics (VolumeKdTree_recover.cpp:386-411) and sort-last compositing.
//
// One thread per pixel; a 64-lane wave covers an 8x8 pixel tile so neighbouring rays
// touch neighbouring voxels (L1/L2 locality of the 8 trilinear taps).  The cube
// rasteriser is replaced by its per-pixel equivalent: nearest cube-surface point along
// the view ray inside [near, far] (GL_LESS, no culling: main.cpp:367-369).
#include "../../include/vrhip.h"
#include "kd_common.h"
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <atomic>

namespace vr {

struct Tex {
    const uint8_t *v;
    int X, Y, Z;       // local extents
    int GX, GY, GZ;    // global extents (== local on the single-GPU path)
    int ox, oy, oz;    // global index of local voxel 0
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// texture(volume, p).r : R8 normalised, GL_LINEAR, clamp-to-edge, float32 weights
__device__ __forceinline__ float tex3d(const Tex &t, float px, float py, float pz)
{
    float x = px * (float)t.GX - 0.5f, y = py * (float)t.GY - 0.5f, z = pz * (float)t.GZ - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y), fz0 = floorf(z);
    float fx = x - fx0, fy = y - fy0, fz = z - fz0;
    int x0 = (int)fx0, y0 = (int)fy0, z0 = (int)fz0;
    int xa = clampi(clampi(x0, 0, t.GX - 1) - t.ox, 0, t.X - 1), xb = clampi(clampi(x0 + 1, 0, t.GX - 1) - t.ox, 0, t.X - 1);
    int ya = clampi(clampi(y0, 0, t.GY - 1) - t.oy, 0, t.Y - 1), yb = clampi(clampi(y0 + 1, 0, t.GY - 1) - t.oy, 0, t.Y - 1);
    int za = clampi(clampi(z0, 0, t.GZ - 1) - t.oz, 0, t.Z - 1), zb = clampi(clampi(z0 + 1, 0, t.GZ - 1) - t.oz, 0, t.Z - 1);
    const float k = 1.0f / 255.0f;
    const int64_t sy = t.X, sz = (int64_t)t.X * t.Y;
    const uint8_t *r0 = t.v + sy * ya + sz * za, *r1 = t.v + sy * yb + sz * za;
    const uint8_t *r2 = t.v + sy * ya + sz * zb, *r3 = t.v + sy * yb + sz * zb;
    // the two x taps of a row are neighbouring bytes except at a clamped edge: one (unaligned) 16-bit load
    // per row instead of two byte loads; at an edge both taps are the same voxel
    typedef uint16_t __attribute__((aligned(1))) u16u;
    const bool pairx = xb == xa + 1;
    const int xl = pairx ? xa : (xa < xb ? xa : xb);
    uint32_t q0, q1, q2, q3;
    if (pairx) { q0 = *(const u16u *)(r0 + xl); q1 = *(const u16u *)(r1 + xl); q2 = *(const u16u *)(r2 + xl); q3 = *(const u16u *)(r3 + xl); }
    else { q0 = r0[xa] | (r0[xb] << 8); q1 = r1[xa] | (r1[xb] << 8); q2 = r2[xa] | (r2[xb] << 8); q3 = r3[xa] | (r3[xb] << 8); }
    float c000 = (float)(q0 & 255u) * k, c100 = (float)(q0 >> 8) * k, c010 = (float)(q1 & 255u) * k, c110 = (float)(q1 >> 8) * k;
    float c001 = (float)(q2 & 255u) * k, c101 = (float)(q2 >> 8) * k, c011 = (float)(q3 & 255u) * k, c111 = (float)(q3 >> 8) * k;
    float c00 = c000 + fx * (c100 - c000), c10 = c010 + fx * (c110 - c010);
    float c01 = c001 + fx * (c101 - c001), c11 = c011 + fx * (c111 - c011);
    float c0 = c00 + fy * (c10 - c00), c1 = c01 + fy * (c11 - c01);
    return c0 + fz * (c1 - c0);
}

__device__ __forceinline__ void norm3(float &a, float &b, float &c)
{
    float l = sqrtf(a * a + b * b + c * c);
    if (l > 0.0f) { a /= l; b /= l; c /= l; } else { a = b = c = 0.0f; }
}
__device__ __forceinline__ float sgn(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }
__device__ __forceinline__ bool inside(float x, float y, float z)
{   // stop = dot(sign(p - texMin), sign(texMax - p)) < 3.0  (raycaster.frag:51)
    float d = sgn(x) * sgn(1.0f - x) + sgn(y) * sgn(1.0f - y) + sgn(z) * sgn(1.0f - z);
    return !(d < 3.0f);
}

// ---- empty-space skipping (new; isosurface_compressed.frag:23-29 only declares the intent) ---------------------
// grid[2 * cell] = min, [2 * cell + 1] = max over the voxels [c * S, c * S + S] per axis (S = cell size; the + 1 is the
// second tap of a trilinear fetch whose base voxel lies in the cell), clamped to the volume.  A fetch at texture position
// p has its base voxel at floor(p * G - 0.5), clamped like the taps themselves, so its eight taps lie inside that cell's
// bounds.
struct SkipGrid { const uint8_t *g; int S, nx, ny, nz; };

__global__ void __launch_bounds__(256)
k_skip_grid(const uint8_t *__restrict__ vol, int X, int Y, int Z, int S, int nx, int ny, int nz, uint8_t *__restrict__ grid)
{
    // one wave per cell: a lane takes whole x-rows of the cell's (S+1)^2 (y, z) columns -- S + 1 consecutive bytes,
    // fetched as aligned 8-byte pieces where the row allows it -- then a wave min / max
    const int64_t cell = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (cell >= (int64_t)nx * ny * nz) return;
    const int cx = (int)(cell % nx), cy = (int)((cell / nx) % ny), cz = (int)(cell / ((int64_t)nx * ny));
    const int x0 = cx * S, y0 = cy * S, z0 = cz * S;
    const int ex = min(S + 1, X - x0), ey = min(S + 1, Y - y0), ez = min(S + 1, Z - z0);
    uint32_t mn = 255, mx = 0;
    const bool wide = (S & 7) == 0 && (X & 7) == 0;          // rows start 8-byte aligned
    for (int r = lane; r < ey * ez; r += 64) {
        const uint8_t *row = vol + (int64_t)x0 + (int64_t)X * ((y0 + r % ey) + (int64_t)Y * (z0 + r / ey));
        int i = 0;
        if (wide)
            for (; i + 8 <= ex; i += 8) {
                const uint2 q = *(const uint2 *)(row + i);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t a = (q.x >> (8 * k)) & 255u, b = (q.y >> (8 * k)) & 255u;
                    mn = min(mn, min(a, b)); mx = max(mx, max(a, b));
                }
            }
        for (; i < ex; ++i) { const uint32_t v = row[i]; mn = min(mn, v); mx = max(mx, v); }
    }
    for (int o = 32; o > 0; o >>= 1) { mn = min(mn, (uint32_t)__shfl_xor((int)mn, o)); mx = max(mx, (uint32_t)__shfl_xor((int)mx, o)); }
    if (lane == 0) { grid[2 * cell] = (uint8_t)mn; grid[2 * cell + 1] = (uint8_t)mx; }
}

// The same grid for 8-voxel cells of a volume whose rows are multiples of 128 voxels (the assembled bench volume):
// one wave per strip of 16 cells along x.  A lane owns a 16-byte piece of a row (two cells) and every eighth of the
// strip's 9 x 9 (y, z) rows; bytes are reduced as packed 16-bit pairs, the + 1 voxel in x comes from the neighbour
// lane's first byte (the last piece reads one byte of the next strip).  Reads each voxel 1.27 times in whole
// 128-byte lines instead of 9-byte row ends by one wave per cell: 17 ms -> 3 ms for the 8 GB volume.
__global__ void __launch_bounds__(256)
k_skip_grid8(const uint8_t *__restrict__ vol, int X, int Y, int Z, int nx, int ny, int nz, uint8_t *__restrict__ grid)
{
    const int nsx = X >> 7;
    const int64_t strip = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (strip >= (int64_t)nsx * ny * nz) return;
    const int lane = threadIdx.x & 63, c = lane & 7, rr = lane >> 3;
    const int sx = (int)(strip % nsx), cy = (int)((strip / nsx) % ny), cz = (int)(strip / ((int64_t)nsx * ny));
    const int x = sx * 128 + c * 16, y0 = cy * 8, z0 = cz * 8;
    const int ey = min(9, Y - y0), ez = min(9, Z - z0), nrows = ey * ez;
    const bool more = c == 7 && x + 16 < X;            // my second cell's last voxel lies in the next strip
    vr_s16x2 mnA = pk_s(0x00FF00FFu), mxA = pk_s(0u), mnB = pk_s(0x00FF00FFu), mxB = pk_s(0u);
    uint32_t fmnA = 255, fmxA = 0, fmnB = 255, fmxB = 0, nmn = 255, nmx = 0;      // first bytes of my cells; the next strip's
    for (int r = rr; r < nrows; r += 8) {
        const uint8_t *row = vol + (int64_t)x + (int64_t)X * ((y0 + r % ey) + (int64_t)Y * (z0 + r / ey));
        const uint4 q = *(const uint4 *)row;
        const vr_s16x2 a0 = pk_s(q.x & 0x00FF00FFu), a1 = pk_s((q.x >> 8) & 0x00FF00FFu), a2 = pk_s(q.y & 0x00FF00FFu),
                       a3 = pk_s((q.y >> 8) & 0x00FF00FFu);
        const vr_s16x2 b0 = pk_s(q.z & 0x00FF00FFu), b1 = pk_s((q.z >> 8) & 0x00FF00FFu), b2 = pk_s(q.w & 0x00FF00FFu),
                       b3 = pk_s((q.w >> 8) & 0x00FF00FFu);
        mnA = __builtin_elementwise_min(mnA, __builtin_elementwise_min(__builtin_elementwise_min(a0, a1), __builtin_elementwise_min(a2, a3)));
        mxA = __builtin_elementwise_max(mxA, __builtin_elementwise_max(__builtin_elementwise_max(a0, a1), __builtin_elementwise_max(a2, a3)));
        mnB = __builtin_elementwise_min(mnB, __builtin_elementwise_min(__builtin_elementwise_min(b0, b1), __builtin_elementwise_min(b2, b3)));
        mxB = __builtin_elementwise_max(mxB, __builtin_elementwise_max(__builtin_elementwise_max(b0, b1), __builtin_elementwise_max(b2, b3)));
        const uint32_t fa = q.x & 255u, fb = q.z & 255u;
        fmnA = min(fmnA, fa); fmxA = max(fmxA, fa);
        fmnB = min(fmnB, fb); fmxB = max(fmxB, fb);
        if (more) { const uint32_t v = row[16]; nmn = min(nmn, v); nmx = max(nmx, v); }
    }
    // cell A = bytes 0..7 plus the first byte of B; cell B = bytes 8..15 plus the first byte of the next piece
    uint32_t cmnA = min(min((uint32_t)(uint16_t)mnA.x, (uint32_t)(uint16_t)mnA.y), fmnB);
    uint32_t cmxA = max(max((uint32_t)(uint16_t)mxA.x, (uint32_t)(uint16_t)mxA.y), fmxB);
    // (shuffles outside the select: a lane that sits out of a cross-lane read hands zeros to the lanes that read it)
    const uint32_t dnMn = (uint32_t)__shfl_down((int)fmnA, 1), dnMx = (uint32_t)__shfl_down((int)fmxA, 1);
    const uint32_t nbMn = c == 7 ? nmn : dnMn, nbMx = c == 7 ? nmx : dnMx;
    uint32_t cmnB = min(min((uint32_t)(uint16_t)mnB.x, (uint32_t)(uint16_t)mnB.y), nbMn);
    uint32_t cmxB = max(max((uint32_t)(uint16_t)mxB.x, (uint32_t)(uint16_t)mxB.y), nbMx);
    for (int o = 8; o < 64; o <<= 1) {
        cmnA = min(cmnA, (uint32_t)__shfl_xor((int)cmnA, o)); cmxA = max(cmxA, (uint32_t)__shfl_xor((int)cmxA, o));
        cmnB = min(cmnB, (uint32_t)__shfl_xor((int)cmnB, o)); cmxB = max(cmxB, (uint32_t)__shfl_xor((int)cmxB, o));
    }
    if (rr == 0) {
        const int64_t cell = (int64_t)(sx * 16 + 2 * c) + (int64_t)nx * (cy + (int64_t)ny * cz);
        *(uint32_t *)(grid + 2 * cell) = cmnA | (cmxA << 8) | (cmnB << 16) | (cmxB << 24);
    }
}

// bounds of the eight taps of tex3d(t, px, py, pz): (min | max << 8)
__device__ __forceinline__ uint32_t skip_bounds(const SkipGrid &sg, const Tex &t, float px, float py, float pz)
{
    const int x0 = clampi((int)floorf(px * (float)t.GX - 0.5f), 0, t.GX - 1), y0 = clampi((int)floorf(py * (float)t.GY - 0.5f), 0, t.GY - 1),
              z0 = clampi((int)floorf(pz * (float)t.GZ - 0.5f), 0, t.GZ - 1);
    const int64_t c = (x0 / sg.S) + (int64_t)sg.nx * ((y0 / sg.S) + (int64_t)sg.ny * (z0 / sg.S));
    return *(const uint16_t *)(sg.g + 2 * c);
}

struct RayArgs {
    Tex t;
    SkipGrid sg;
    vr_camera cam;
    vr_render_params P;
    float f[3], s[3], u[3];
    float tanX, tanY;
    float *out;
};

__global__ void __launch_bounds__(64)
k_raycast(RayArgs a)
{
    // 8x8 pixel tile per wave
    const int px = blockIdx.x * 8 + (threadIdx.x & 7), py = blockIdx.y * 8 + (threadIdx.x >> 3);
    const int W = a.P.width, H = a.P.height;
    if (px >= W || py >= H) return;
    float *o = a.out + 4 * ((size_t)py * W + px);
    const float nx = 2.0f * ((float)px + 0.5f) / (float)W - 1.0f;
    const float ny = 1.0f - 2.0f * ((float)py + 0.5f) / (float)H;
    float dir[3], cp[3] = {a.cam.pos[0], a.cam.pos[1], a.cam.pos[2]};
#pragma unroll
    for (int k = 0; k < 3; ++k) dir[k] = a.f[k] + nx * a.tanX * a.s[k] + ny * a.tanY * a.u[k];
    float t0 = -INFINITY, t1 = INFINITY;
    bool miss = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (dir[k] != 0.0f) {
            float lo = (-0.5f - cp[k]) / dir[k], hi = (0.5f - cp[k]) / dir[k];
            if (lo > hi) { float q = lo; lo = hi; hi = q; }
            if (lo > t0) t0 = lo;
            if (hi < t1) t1 = hi;
        } else if (cp[k] < -0.5f || cp[k] > 0.5f) miss = true;
    }
    const float th = t0 >= a.cam.z_near ? t0 : t1;
    const int mode = a.P.mode;
    if (miss || t0 > t1 || th < a.cam.z_near || th > a.cam.z_far) {
        if (mode == VR_RENDER_PARTIAL) { o[0] = 0.0f; o[1] = 1.0f; o[2] = 0.0f; o[3] = 0.0f; }
        else { o[0] = o[1] = o[2] = o[3] = 1.0f; }   // clear colour (main.cpp:392)
        return;
    }
    float vuv[3], gd[3], st[3], pos[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) vuv[k] = (cp[k] + th * dir[k]) + 0.5f;      // vUV = vVertex + 0.5
#pragma unroll
    for (int k = 0; k < 3; ++k) gd[k] = (vuv[k] - 0.5f) - cp[k];            // raycaster.frag:27
    norm3(gd[0], gd[1], gd[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) { st[k] = gd[k] * a.P.step_size[k]; pos[k] = vuv[k]; }
    const int ns = a.P.max_samples;
    if (mode == VR_RENDER_COMPOSITE) {
        float rgb = 0.0f, A = 0.0f;
        bool probe = true;      // ask the grid only while the ray is in empty space (the last fetch, if any, was 0)
        for (int i = 0; i < ns; ++i) {
            pos[0] = pos[0] + st[0]; pos[1] = pos[1] + st[1]; pos[2] = pos[2] + st[2];
            if (!inside(pos[0], pos[1], pos[2])) break;
            // all eight taps zero: the sample is exactly 0 and the three updates below are exact no-ops
            if (a.sg.g && probe && (skip_bounds(a.sg, a.t, pos[0], pos[1], pos[2]) >> 8) == 0u) continue;
            float smp = tex3d(a.t, pos[0], pos[1], pos[2]);
            probe = smp == 0.0f;
            float pa = smp - (smp * A);          // raycaster.frag:69
            rgb = pa * smp + rgb;                // :70
            A += pa * 0.6f;                      // :72
            if (!a.P.no_early_exit && A > 0.99f) break; // :77
        }
        o[0] = 1.0f - rgb; o[1] = 1.0f - rgb; o[2] = 1.0f; o[3] = A;   // :82-85 (b = 255 clamps)
    } else if (mode == VR_RENDER_PARTIAL) {
        float c = 0.0f, tau = 1.0f;
        for (int i = 0; i < ns; ++i) {
            pos[0] = pos[0] + st[0]; pos[1] = pos[1] + st[1]; pos[2] = pos[2] + st[2];
            if (!inside(pos[0], pos[1], pos[2])) break;
            bool own = true;
#pragma unroll
            for (int k = 0; k < 3; ++k) own = own && (pos[k] >= a.P.box_min[k] && pos[k] < a.P.box_max[k]);
            if (!own) continue;
            float smp = tex3d(a.t, pos[0], pos[1], pos[2]);
            c = c + tau * (smp * smp);
            tau = tau * (1.0f - 0.6f * smp);
        }
        o[0] = c; o[1] = tau; o[2] = 1.0f; o[3] = 0.0f;
    } else {
        float col[4] = {1.0f, 1.0f, 1.0f, 1.0f};         // vec4(255,255,255,1) clamped (isosurface.frag:79)
        const float iso = a.P.iso_value;
        // A step's second fetch is at dataPos + dirStep, which IS the next step's dataPos (the same float addition of
        // the same operands): the shader fetches it twice (isosurface.frag:120-121), here the value is carried over.
        float carried = 0.0f;
        bool haveCarried = false, haveBounds = false;
        uint32_t carriedBounds = 0;
        for (int i = 0; i < ns; ++i) {
            pos[0] = pos[0] + st[0]; pos[1] = pos[1] + st[1]; pos[2] = pos[2] + st[2];
            if (!inside(pos[0], pos[1], pos[2])) break;
            if (a.sg.g) {
                // the test below needs s1 < iso <= s2.  Interpolation in float can leave the taps' range by rounding only,
                // so a whole grey level of margin decides safely: every tap of s1 above iso, or every tap of s2 below it
                // (the second position's bounds are the next step's first, like the fetch itself)
                const uint32_t b1 = haveBounds ? carriedBounds : skip_bounds(a.sg, a.t, pos[0], pos[1], pos[2]);
                haveBounds = false;
                if ((float)((int)(b1 & 255u) - 1) * (1.0f / 255.0f) >= iso) { haveCarried = false; continue; }
                const uint32_t b2 = skip_bounds(a.sg, a.t, pos[0] + st[0], pos[1] + st[1], pos[2] + st[2]);
                carriedBounds = b2; haveBounds = true;
                if ((float)((int)(b2 >> 8) + 1) * (1.0f / 255.0f) < iso) { haveCarried = false; continue; }
            }
            float s1 = haveCarried ? carried : tex3d(a.t, pos[0], pos[1], pos[2]);
            float s2 = tex3d(a.t, pos[0] + st[0], pos[1] + st[1], pos[2] + st[2]);
            carried = s2; haveCarried = true;
            if ((s1 - iso) < 0.0f && (s2 - iso) >= 0.0f) {                     // :126
                float l[3] = {pos[0], pos[1], pos[2]}, r[3] = {pos[0] + st[0], pos[1] + st[1], pos[2] + st[2]};
                for (int b = 0; b < 4; ++b) {                                  // Bisection :23-42
                    float m0 = (r[0] + l[0]) * 0.5f, m1 = (r[1] + l[1]) * 0.5f, m2 = (r[2] + l[2]) * 0.5f;
                    float cm = tex3d(a.t, m0, m1, m2);
                    if (cm < iso) { l[0] = m0; l[1] = m1; l[2] = m2; } else { r[0] = m0; r[1] = m1; r[2] = m2; }
                }
                float tc0 = (r[0] + l[0]) * 0.5f, tc1 = (r[1] + l[1]) * 0.5f, tc2 = (r[2] + l[2]) * 0.5f;
                const float DELTA = 0.01f;                                     // GetGradient :47-62
                float N0 = (tex3d(a.t, tc0 - DELTA, tc1, tc2) - tex3d(a.t, tc0 + DELTA, tc1, tc2)) / 2.0f;
                float N1 = (tex3d(a.t, tc0, tc1 - DELTA, tc2) - tex3d(a.t, tc0, tc1 + DELTA, tc2)) / 2.0f;
                float N2 = (tex3d(a.t, tc0, tc1, tc2 - DELTA) - tex3d(a.t, tc0, tc1, tc2 + DELTA)) / 2.0f;
                norm3(N0, N1, N2);
                float V0 = -gd[0], V1 = -gd[1], V2 = -gd[2];                   // head light: L = V (:142-146)
                float diffuse = fmaxf(V0 * N0 + V1 * N1 + V2 * N2, 0.0f);
                float h0 = V0 + V0, h1 = V1 + V1, h2 = V2 + V2;
                norm3(h0, h1, h2);
                float spec = powf(fmaxf(0.00001f, h0 * N0 + h1 * N1 + h2 * N2), 250.0f); // :73
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

__global__ void k_composite_over(float4 *front, const float4 *back, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 f = front[i], b = back[i];
    f.x = f.x + f.y * b.x;       // (c1 + t1*c2, t1*t2)
    f.y = f.y * b.y;
    f.z = fmaxf(f.z, b.z);
    front[i] = f;
}

__global__ void k_composite_finish(const float4 *partial, float4 *rgba, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = partial[i], o;
    if (p.z > 0.0f) { o.x = 1.0f - p.x; o.y = 1.0f - p.x; o.z = 1.0f; o.w = 1.0f - p.y; }
    else { o.x = o.y = o.z = o.w = 1.0f; }
    rgba[i] = o;
}

static void cross3(const float *a, const float *b, float *o);
static void hnorm3(float *v);

struct SlabArgs {
    const float4 *partials;
    int num_slabs;
    int64_t npix, first;
    int axis, W, H;
    float f[3], s[3], u[3], tanX, tanY;
    float4 *out;
};

__global__ void __launch_bounds__(256)
k_composite_slabs(SlabArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.npix) return;
    const int64_t gp = a.first + i;
    const int px = (int)(gp % a.W), py = (int)(gp / a.W);
    const float nx = 2.0f * ((float)px + 0.5f) / (float)a.W - 1.0f;
    const float ny = 1.0f - 2.0f * ((float)py + 0.5f) / (float)a.H;
    const float d = a.f[a.axis] + nx * a.tanX * a.s[a.axis] + ny * a.tanY * a.u[a.axis];
    float c = 0.0f, tau = 1.0f, cov = 0.0f;
    for (int k = 0; k < a.num_slabs; ++k) {
        const int sidx = d >= 0.0f ? k : a.num_slabs - 1 - k;
        const float4 p = a.partials[(int64_t)sidx * a.npix + i];
        c = c + tau * p.x;          // (c1 + t1*c2, t1*t2)
        tau = tau * p.y;
        cov = fmaxf(cov, p.z);
    }
    float4 o;
    if (cov > 0.0f) { o.x = 1.0f - c; o.y = 1.0f - c; o.z = 1.0f; o.w = 1.0f - tau; }   // raycaster.frag:82-85
    else { o.x = o.y = o.z = o.w = 1.0f; }
    a.out[i] = o;
}

// Brick <-> global volume placement (VolumeReader.h:172-211), 16-byte rows segments.
template <bool TO_VOLUME>
__global__ void __launch_bounds__(256)
k_assemble(const uint8_t *src, uint8_t *dst, int nbricks, int64_t X, int64_t Y, int64_t Z, const int64_t *ijk,
           int64_t I, int64_t J)
{
    const int b = blockIdx.y;
    const int64_t xv = X / 16;                     // 16-byte vectors per row
    const int64_t total = xv * Y * Z;
    const int64_t i = ijk[3 * b], j = ijk[3 * b + 1], k = ijk[3 * b + 2];
    const int64_t GX = X * I, GY = Y * J;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        int64_t xq = q % xv, y = (q / xv) % Y, z = q / (xv * Y);
        int64_t bo = (int64_t)b * X * Y * Z + xq * 16 + X * (y + Y * z);
        int64_t go = (i * X + xq * 16) + GX * ((j * Y + y) + GY * (k * Z + z));
        if (TO_VOLUME) *(uint4 *)(dst + go) = *(const uint4 *)(src + bo);
        else *(uint4 *)(dst + bo) = *(const uint4 *)(src + go);
    }
}

__global__ void __launch_bounds__(256)
k_measure_error(const uint8_t *a, const uint8_t *b, int64_t n, int *maxErr, unsigned long long *sumErr)
{
    int m = 0;
    unsigned long long s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int e = (int)a[i] - (int)b[i];
        e = e < 0 ? -e : e;
        m = e > m ? e : m;
        s += (unsigned)e;
    }
    for (int o = 32; o > 0; o >>= 1) { int u = __shfl_xor(m, o); m = u > m ? u : m; s += __shfl_xor(s, o); }
    if ((threadIdx.x & 63) == 0) { if (m) atomicMax(maxErr, m); if (s) atomicAdd(sumErr, s); }
}

__global__ void __launch_bounds__(256)
k_query_error(const uint8_t *a, const uint8_t *b, int64_t n, uint8_t *out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int e = (int)a[i] - (int)b[i];
        out[i] = (uint8_t)(e < 0 ? -e : e);
    }
}

static void cross3(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static void hnorm3(float *v)
{
    float l = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (l > 0.0f) { v[0] /= l; v[1] /= l; v[2] /= l; } else { v[0] = v[1] = v[2] = 0.0f; }
}

int raycast_launch(const uint8_t *vol, const int64_t dims[3], const vr_camera *cam, const vr_render_params *P,
                   float *rgba, hipStream_t st)
{
    RayArgs a;
    a.t.v = vol;
    a.t.X = (int)dims[0]; a.t.Y = (int)dims[1]; a.t.Z = (int)dims[2];
    a.t.GX = P->global_dims[0] > 0 ? (int)P->global_dims[0] : a.t.X;
    a.t.GY = P->global_dims[1] > 0 ? (int)P->global_dims[1] : a.t.Y;
    a.t.GZ = P->global_dims[2] > 0 ? (int)P->global_dims[2] : a.t.Z;
    a.t.ox = (int)P->vol_origin[0]; a.t.oy = (int)P->vol_origin[1]; a.t.oz = (int)P->vol_origin[2];
    a.cam = *cam;
    a.P = *P;
    a.sg.g = nullptr; a.sg.S = 1; a.sg.nx = a.sg.ny = a.sg.nz = 0;
    // the grid describes volume_dev as a whole: only used where the local volume IS the texture (single-GPU path)
    if (P->skip_grid_dev && P->skip_cell > 0 && a.t.GX == a.t.X && a.t.GY == a.t.Y && a.t.GZ == a.t.Z && a.t.ox == 0 && a.t.oy == 0 && a.t.oz == 0) {
        a.sg.g = P->skip_grid_dev; a.sg.S = P->skip_cell;
        a.sg.nx = (a.t.X + a.sg.S - 1) / a.sg.S; a.sg.ny = (a.t.Y + a.sg.S - 1) / a.sg.S; a.sg.nz = (a.t.Z + a.sg.S - 1) / a.sg.S;
    }
    // glm::lookAt basis and glm::perspectiveFov half-angle tangents (main.cpp:396-397)
    for (int k = 0; k < 3; ++k) a.f[k] = cam->front[k];
    hnorm3(a.f);
    cross3(a.f, cam->up, a.s);
    hnorm3(a.s);
    cross3(a.s, a.f, a.u);
    const float rad = cam->fov_deg * 0.01745329251994329576923690768489f;
    a.tanY = tanf(0.5f * rad);
    a.tanX = a.tanY * (float)P->width / (float)P->height;
    a.out = rgba;
    dim3 grid((P->width + 7) / 8, (P->height + 7) / 8);
    hipLaunchKernelGGL(k_raycast, grid, dim3(64), 0, st, a);
    return launch_status("raymarch");
}

// process-wide debugging switch (vr_debug_set("skip_grid_v1", 1), or VRHIP_SKIP_GRID_V1 in the environment when the
// library is first used): the one-wave-per-cell kernel for every grid
std::atomic<int> g_skipGridV1{-1};

int skip_grid_launch(const uint8_t *vol, const int64_t dims[3], int S, uint8_t *grid, hipStream_t st)
{
    int v1 = g_skipGridV1.load(std::memory_order_relaxed);
    if (v1 < 0) { v1 = getenv("VRHIP_SKIP_GRID_V1") ? 1 : 0; g_skipGridV1.store(v1, std::memory_order_relaxed); }
    // k_skip_grid8 loads 16 bytes per lane and stores (min, max) pairs as 32-bit words: an offset sub-buffer of a
    // caller's allocation goes through the byte-wise kernel
    const int nx = (int)((dims[0] + S - 1) / S), ny = (int)((dims[1] + S - 1) / S), nz = (int)((dims[2] + S - 1) / S);
    const int64_t cells = (int64_t)nx * ny * nz;
    const bool aligned = (((uintptr_t)vol & 15u) == 0u) && (((uintptr_t)grid & 3u) == 0u);
    if (S == 8 && (dims[0] & 127) == 0 && !v1 && aligned) {
        const int64_t strips = (dims[0] >> 7) * (int64_t)ny * nz;
        hipLaunchKernelGGL(k_skip_grid8, dim3((unsigned)((strips + 3) / 4)), dim3(256), 0, st, vol, (int)dims[0], (int)dims[1],
                           (int)dims[2], nx, ny, nz, grid);
        return launch_status("skip_grid");
    }
    hipLaunchKernelGGL(k_skip_grid, dim3((unsigned)((cells + 3) / 4)), dim3(256), 0, st, vol, (int)dims[0], (int)dims[1], (int)dims[2], S,
                       nx, ny, nz, grid);
    return launch_status("skip_grid");
}

int composite_slabs_launch(const float *partials, int nslabs, int64_t npix, int64_t first, int axis, const vr_camera *cam,
                           const vr_render_params *P, float *rgba, hipStream_t st)
{
    SlabArgs a;
    a.partials = (const float4 *)partials; a.num_slabs = nslabs; a.npix = npix; a.first = first;
    a.axis = axis; a.W = P->width; a.H = P->height;
    for (int k = 0; k < 3; ++k) a.f[k] = cam->front[k];
    hnorm3(a.f);
    cross3(a.f, cam->up, a.s);
    hnorm3(a.s);
    cross3(a.s, a.f, a.u);
    const float rad = cam->fov_deg * 0.01745329251994329576923690768489f;
    a.tanY = tanf(0.5f * rad);
    a.tanX = a.tanY * (float)P->width / (float)P->height;
    a.out = (float4 *)rgba;
    hipLaunchKernelGGL(k_composite_slabs, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, a);
    return launch_status("composite_slabs");
}

int composite_over_launch(float *front, const float *back, int64_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_composite_over, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (float4 *)front,
                       (const float4 *)back, n);
    return launch_status("raymarch");
}
int composite_finish_launch(const float *partial, float *rgba, int64_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_composite_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       (const float4 *)partial, (float4 *)rgba, n);
    return launch_status("raymarch");
}
int assemble_launch(bool toVolume, const uint8_t *src, uint8_t *dst, int nb, const int64_t bd[3], const int64_t *ijkDev,
                    const int64_t grid[3], hipStream_t st)
{
    int64_t total = bd[0] / 16 * bd[1] * bd[2];
    unsigned gx = (unsigned)((total + 255) / 256);
    if (gx > 4096) gx = 4096;
    if (toVolume)
        hipLaunchKernelGGL(k_assemble<true>, dim3(gx, nb), dim3(256), 0, st, src, dst, nb, bd[0], bd[1], bd[2], ijkDev,
                           grid[0], grid[1]);
    else
        hipLaunchKernelGGL(k_assemble<false>, dim3(gx, nb), dim3(256), 0, st, src, dst, nb, bd[0], bd[1], bd[2], ijkDev,
                           grid[0], grid[1]);
    return launch_status("raymarch");
}
int measure_error_launch(const uint8_t *a, const uint8_t *b, int64_t n, int *maxErrDev, unsigned long long *sumDev,
                         hipStream_t st)
{
    unsigned g = (unsigned)((n + 255) / 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_measure_error, dim3(g), dim3(256), 0, st, a, b, n, maxErrDev, sumDev);
    return launch_status("raymarch");
}
int query_error_launch(const uint8_t *a, const uint8_t *b, int64_t n, uint8_t *out, hipStream_t st)
{
    unsigned g = (unsigned)((n + 255) / 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_query_error, dim3(g), dim3(256), 0, st, a, b, n, out);
    return launch_status("raymarch");
}

} // namespace vr
