// kd_decode.hip -- gfx950 kernels for VolumeKdtree::levelCut at full depth
// (reference volume_renderer/VolumeKdTree_recover.cpp:726-835).
//
// The reference walks the preorder 2-bit stream with one serial stack machine.  Here
// the stream is cut at depth Ds = D-K by a side-car index (token offset + decoded
// scalar of every depth-Ds subtree root, emitted for free by the encoder's scan or
// rebuilt from the bytes alone by build_index_from_stream) and the 2^Ds subtrees are
// decoded independently.
#include "brickset.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace vr {

int launch_status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    if (getenv("VRHIP_DEBUG")) fprintf(stderr, "[vrhip] %s: %s\n", what, hipGetErrorString(e));
    return -1;
}

void make_geom(Geom &g, const int64_t dims[3])
{
    memset(&g, 0, sizeof(g));
    g.X = (int32_t)dims[0]; g.Y = (int32_t)dims[1]; g.Z = (int32_t)dims[2];
    g.voxels = dims[0] * dims[1] * dims[2];
    int64_t ext[3] = {dims[0], dims[1], dims[2]};
    for (int k = 0; k < 3; ++k) { int n = 0; while (((int64_t)1 << (n + 1)) <= dims[k]) ++n; g.nb[k] = n; }
    g.D = g.nb[0] + g.nb[1] + g.nb[2];
    for (int d = 0; d < g.D; ++d) {            // split-axis rule, R.cpp:151-159
        int sd = d % 3, i = 0;
        while (ext[0] * ext[1] * ext[2] > 1 && ext[sd] == 1) sd = (d + ++i) % 3;
        ext[sd] /= 2;
        int b = 0; while (((int64_t)1 << (b + 1)) <= ext[sd]) ++b;
        g.axis[d] = (uint8_t)sd;
        g.bit[d] = (uint8_t)b;                 // the coordinate bit this split decides
    }
}

// local rank inside a depth-(D-K) subtree -> packed voxel offset (dx | dy<<10 | dz<<20)
void make_lut(const Geom &g, int K, std::vector<uint32_t> &lut)
{
    lut.assign((size_t)1 << K, 0);
    for (uint32_t lr = 0; lr < (1u << K); ++lr) {
        uint32_t c[3] = {0, 0, 0};
        for (int q = 0; q < K; ++q) {
            int d = g.D - K + q;
            uint32_t b = (lr >> (K - 1 - q)) & 1u;
            c[g.axis[d]] |= b << g.bit[d];
        }
        lut[lr] = c[0] | (c[1] << 10) | (c[2] << 20);
    }
}

struct DecodeArgs {
    const uint8_t *tree;
    int64_t treeCap;
    const uint32_t *idxOff;
    const uint8_t *idxVal;
    int64_t nIdx;
    const Ctrl *ctrls;
    const uint32_t *lut;
    uint8_t *out;
    Geom g;
    int D, K, Ds;
};

// v1: one lane per subtree, direct byte stores.
__global__ void __launch_bounds__(64)
k_decode_lane(DecodeArgs a)
{
    const int brick = blockIdx.y;
    const int64_t s = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (s >= a.nIdx) return;
    const uint32_t off = a.idxOff[(int64_t)brick * a.nIdx + s];
    const int val0 = a.idxVal[(int64_t)brick * a.nIdx + s];
    const uint8_t *dmap = a.ctrls[brick].distanceMap;
    int ox, oy, oz;
    rank_to_xyz(a.g, (uint32_t)(s << a.K), ox, oy, oz);
    uint8_t *O = a.out + (int64_t)brick * a.g.voxels + ox + (int64_t)a.g.X * (oy + (int64_t)a.g.Y * oz);
    const int64_t sy = a.g.X, sz = (int64_t)a.g.X * a.g.Y;
    const int K = a.K;
    auto fill = [&](uint32_t lo, uint32_t cnt, int v) {
        for (uint32_t lr = lo; lr < lo + cnt; ++lr) {
            uint32_t p = a.lut[lr];
            O[(p & 1023u) + sy * ((p >> 10) & 1023u) + sz * (p >> 20)] = (uint8_t)v;
        }
    };
    if (off == VR_IDX_DEAD) { fill(0, 1u << K, val0); return; }
    const uint32_t *W = (const uint32_t *)(a.tree + (int64_t)brick * a.treeCap);
    uint32_t pos = off;
    int vals[16];
    int j = 0;
    uint32_t path = 0;
    while (true) {
        int tok = (W[pos >> 4] >> ((pos & 15u) * 2u)) & 3u;
        ++pos;
        int v = j == 0 ? val0 : apply_code(vals[j - 1], tok, dmap[a.Ds + j]);
        vals[j] = v;
        bool terminal = false;
        if (tok == 3) { fill(path << (K - j), 1u << (K - j), v); terminal = true; }
        else if (j == K) {
            for (int c = 1; c <= VR_CHAIN_LEVELS; ++c) {   // grown branch: same voxel, distances 64..1
                int t2 = (W[pos >> 4] >> ((pos & 15u) * 2u)) & 3u;
                ++pos;
                if (t2 == 3) break;
                v = apply_code(v, t2, dmap[a.D + c]);
            }
            fill(path, 1, v);
            terminal = true;
        }
        if (terminal) {
            while (j > 0 && (path & 1u)) { path >>= 1; --j; }
            if (j == 0) break;
            path |= 1u;
        } else { ++j; path <<= 1; }
    }
}

int decode_launch(BrickSet *bs, uint8_t *out, hipStream_t st)
{
    DecodeArgs a;
    a.tree = bs->mid.tree; a.treeCap = bs->treeCap;
    a.idxOff = bs->idxOff; a.idxVal = bs->idxVal; a.nIdx = bs->nIdx;
    a.ctrls = bs->mid.ctrl; a.lut = bs->lut; a.out = out; a.g = bs->g;
    a.D = bs->D; a.K = bs->K; a.Ds = bs->Ds;
    hipEventRecord(bs->ev[5], st);
    hipLaunchKernelGGL(k_decode_lane, dim3((unsigned)((bs->nIdx + 63) / 64), bs->B), dim3(64), 0, st, a);
    hipEventRecord(bs->ev[6], st);
    return launch_status("decode");
}

// Serial pass over a foreign stream (host): the side-car index from the bytes alone.
// Also validates the grammar.  Returns 0 or a negative code.
int build_index_from_stream(BrickSet *bs, int brick, const uint8_t *tree, int64_t numActive, const uint8_t *dmap,
                            std::vector<uint32_t> &offs, std::vector<uint8_t> &vals)
{
    const int D = bs->D, Ds = bs->Ds;
    offs.assign((size_t)bs->nIdx, VR_IDX_DEAD);
    vals.assign((size_t)bs->nIdx, 0);
    auto get = [&](int64_t p) { return (tree[p >> 2] >> ((p & 3) * 2)) & 3; };
    int v[VR_MAX_DEPTH];
    int64_t pos = 0;
    int j = 0;
    uint32_t path = 0;
    while (true) {
        if (pos >= numActive) return -1;
        const int64_t here = pos;
        int tok = get(pos++);
        int val = j == 0 ? dmap[0] : apply_code(v[j - 1], tok, dmap[j]);
        v[j] = val;
        if (j == Ds) { offs[path] = (uint32_t)here; vals[path] = (uint8_t)val; }
        bool terminal = false;
        if (tok == 3) {
            if (j < Ds) {
                uint32_t lo = path << (Ds - j), hi = (path + 1) << (Ds - j);
                for (uint32_t q = lo; q < hi; ++q) { offs[q] = VR_IDX_DEAD; vals[q] = (uint8_t)val; }
            }
            terminal = true;
        } else if (j == D) {
            for (int c = 1; c <= VR_CHAIN_LEVELS; ++c) {
                if (pos >= numActive) return -2;
                if (get(pos++) == 3) break;
            }
            terminal = true;
        }
        if (terminal) {
            while (j > 0 && (path & 1u)) { path >>= 1; --j; }
            if (j == 0) break;
            path |= 1u;
        } else { ++j; path <<= 1; }
    }
    return pos == numActive ? 0 : -3;
}

} // namespace vr
