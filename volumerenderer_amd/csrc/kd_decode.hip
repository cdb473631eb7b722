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

// ---- v2: tile decode --------------------------------------------------------------
// One lane decodes one depth-(D-6) subtree = a 4x4x4 voxel block (64 leaves, up to
// ~600 tokens).  A wave takes 32x2x1 such blocks = a 128x8x4 voxel tile whose rows are
// whole 128-byte lines of the output volume.  Lanes emit their leaves in stream (Morton)
// order, four per dword, into an LDS tile laid out [dword q][lane] (bank = lane, so
// the divergent walk writes conflict-free); the wave then gathers 16-byte row pieces
// from it and stores 8 full lines per instruction.  Requirements: the six deepest
// split levels cycle through x,y,z twice (any order) and X>=128, Y>=8, Z>=4;
// everything else takes k_decode_lane.
struct TileArgs {
    const uint8_t *tree;
    int64_t treeCap;
    const uint32_t *idxOff;
    const uint8_t *idxVal;
    int64_t nIdx;
    const Ctrl *ctrls;
    uint8_t *out;
    Geom g;
    int D, Ds;
    int jx, jy, jz;          // position of each axis among the three deepest split levels (0 = deepest)
    int tilesX, tilesY, tilesZ;
};

#define DEC_WAVES 4

// rotation of the lane index inside dword row q of the LDS tile (see the gather stage): (q0+q1+q2+2*q3)&3, as a packed table
__device__ __forceinline__ int lds_rot(int q) { return (int)((0x433EE994u >> (2 * q)) & 3u); }

__global__ void __launch_bounds__(64 * DEC_WAVES)
k_decode_tile(TileArgs a)
{
    __shared__ uint32_t tileS[DEC_WAVES][16 * 64];
    __shared__ uint8_t stkS[DEC_WAVES][8 * 64];
    __shared__ uint8_t dmS[16];      // [1..6] tree levels Ds+1..D, [9..15] grown-branch levels D+1..D+7
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int brick = blockIdx.y;
    const int tileId = blockIdx.x * DEC_WAVES + wave;
    const int tx = tileId % a.tilesX, ty = (tileId / a.tilesX) % a.tilesY, tz = tileId / (a.tilesX * a.tilesY);
    uint32_t *tile = tileS[wave];
    uint8_t *stk = stkS[wave];
    const bool tileValid = tileId < a.tilesX * a.tilesY * a.tilesZ;
    if (threadIdx.x < 16) {
        const uint8_t *dmap = a.ctrls[brick].distanceMap;
        const int t = threadIdx.x;
        dmS[t] = t == 0 ? 0 : (t < 8 ? dmap[a.Ds + (t < 7 ? t : 0)] : dmap[a.D + (t - 8)]);   // [0] = 0: the subtree root keeps the index value
    }
    __syncthreads();
    if (!tileValid) return;

    // ---- which subtree is mine
    int sc[3] = {tx * 32 + (lane & 31), ty * 2 + (lane >> 5), tz};   // subtree coords (units of 4 voxels)
    uint32_t s = 0;
    for (int d = 0; d < a.Ds; ++d) s = (s << 1) | ((uint32_t)(sc[a.g.axis[d]] >> (a.g.bit[d] - 2)) & 1u);
    const uint32_t off = a.idxOff[(int64_t)brick * a.nIdx + s];
    const int val0 = a.idxVal[(int64_t)brick * a.nIdx + s];

    // ---- walk my token run, leaves out in Morton order.
    // One action per iteration and lane (consume one token, or write one dword of a
    // pruned node's fill), so the wave runs a short uniform loop body instead of nested
    // divergent loops.  Tokens come from a 64-bit buffer topped up with one 32-bit word
    // every 16 iterations: a lane consumes at most 16 tokens per block, so >= 32 valid
    // bits at block start can never underflow, and the word loaded at block start is only
    // merged at block end -- its global-memory latency hides behind the 16 iterations.
    if (off == VR_IDX_DEAD) {
        const uint32_t vv = (uint32_t)val0 * 0x01010101u;
#pragma unroll
        for (int q = 0; q < 16; ++q) tile[q * 64 + ((lane + lds_rot(q)) & 63)] = vv;
    }
    {
        const uint32_t *W = (const uint32_t *)(a.tree + (int64_t)brick * a.treeCap);
        bool done = off == VR_IDX_DEAD;
        const uint32_t o0 = done ? 0u : off;
        uint32_t wi = o0 >> 4;
        const int sh0 = (int)(o0 & 15u) * 2;
        unsigned long long buf = ((unsigned long long)W[wi] | ((unsigned long long)W[wi + 1] << 32)) >> sh0;
        int nb = 64 - sh0;
        wi += 2;
        uint32_t p = 1;                 // path with a leading sentinel bit: depth = bitlen(p) - 1
        int chain = 0;                  // 0: tree token expected, 1..7: next grown-branch step
        int v = val0;
        int fill = 0, q = 0, na = 0;
        uint32_t fillv = 0, acc = 0;
        // value stack rows 0..5 = pushed ancestors, row 6 = scratch for predicated-off pushes,
        // row 7 = root's "parent" (the index value itself; dmS[0] = 0 leaves it unchanged)
        stk[7 * 64 + lane] = (uint8_t)val0;
        while (__ballot(!done) != 0ull) {
            const uint32_t wn = W[wi];  // consumed (maybe) at the end of this block
#pragma unroll 1
            for (int it = 0; it < 16; ++it) {
                if (done) continue;
                if (fill > 0) {
                    tile[q * 64 + ((lane + lds_rot(q)) & 63)] = fillv;
                    ++q;
                    if (--fill == 0 && p == 0x80000000u) done = true;
                    continue;
                }
                // ---- one token, straight-line (selects, no branches)
                const int tok = (int)(buf & 3ull);
                buf >>= 2; nb -= 2;
                const int j = 31 - __clz((int)p);
                const bool tree = chain == 0;
                const int sv = stk[(tree ? ((j + 7) & 7) : 7) * 64 + lane];
                const int dist = dmS[tree ? j : 8 + chain];
                const int pv = tree ? sv : v;
                const int delta = tok == 1 ? dist : (tok == 2 ? -dist : 0);
                int nv = pv + delta;
                nv = nv < 0 ? 0 : (nv > 255 ? 255 : nv);       // decoder step R.cpp:783-787
                v = nv;
                const bool is3 = tok == 3;
                const bool term = is3 || chain == VR_CHAIN_LEVELS;
                const bool desc = tree && !is3 && j < 6;
                stk[(desc ? j : 6) * 64 + lane] = (uint8_t)v;
                const int count = tree ? (64 >> j) : 1;
                p = desc ? (p << 1) : p;
                chain = term ? 0 : (tree ? (j == 6 ? 1 : 0) : chain + 1);
                if (term) {
                    if (count >= 4) { fill = count >> 2; fillv = (uint32_t)v * 0x01010101u; }
                    else {
                        acc |= ((count == 2 ? 0x0101u : 1u) * (uint32_t)v) << (8 * na);
                        na += count;
                        if (na == 4) { tile[q * 64 + ((lane + lds_rot(q)) & 63)] = acc; ++q; acc = 0; na = 0; }
                    }
                    uint32_t np = p + 1u;
                    np >>= (__ffs((int)np) - 1);
                    p = np;
                    if (np == 1u) { p = 0x80000000u; done = fill == 0; }   // parked: no further tokens are mine
                }
            }
            if (nb <= 32) { buf |= (unsigned long long)wn << nb; nb += 32; ++wi; }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): my LDS writes have landed
    __builtin_amdgcn_wave_barrier();

    // ---- gather 16-byte row pieces and store whole 128-byte lines
    const int jx = a.jx, jy = a.jy, jz = a.jz;
    const int c = lane & 7;
    uint8_t *O = a.out + (int64_t)brick * a.g.voxels + (int64_t)tx * 128 + c * 16;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int R = st * 8 + (lane >> 3);
        const int y = R & 7, z = R >> 3;
        const int dy = y & 3, dz = z & 3;
        const int rb = ((dy & 1) << jy) | ((dy >> 1) << (3 + jy)) | ((dz & 1) << jz) | ((dz >> 1) << (3 + jz));
        const int q0 = rb >> 2, b0 = rb & 3;
        const int lsBase = 4 * c + 32 * (y >> 2);
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int Ls = lsBase + k;
            if (jx == 0) {          // dx0 -> byte +1, dx1 -> dword +2
                uint32_t A = tile[q0 * 64 + ((Ls + lds_rot(q0)) & 63)], B = tile[(q0 + 2) * 64 + ((Ls + lds_rot(q0 + 2)) & 63)];
                uint32_t lo = (A >> (8 * b0)) & 0xFFFFu, hi = (B >> (8 * b0)) & 0xFFFFu;
                o[k] = lo | (hi << 16);
            } else if (jx == 1) {   // dx0 -> byte +2, dx1 -> dword +4
                uint32_t A = tile[q0 * 64 + ((Ls + lds_rot(q0)) & 63)], B = tile[(q0 + 4) * 64 + ((Ls + lds_rot(q0 + 4)) & 63)];
                uint32_t a0 = (A >> (8 * b0)) & 0xFFu, a1 = (A >> (8 * b0 + 16)) & 0xFFu;
                uint32_t c0 = (B >> (8 * b0)) & 0xFFu, c1 = (B >> (8 * b0 + 16)) & 0xFFu;
                o[k] = a0 | (a1 << 8) | (c0 << 16) | (c1 << 24);
            } else {                // dx0 -> dword +1, dx1 -> dword +8
                uint32_t A = tile[q0 * 64 + ((Ls + lds_rot(q0)) & 63)], B = tile[(q0 + 1) * 64 + ((Ls + lds_rot(q0 + 1)) & 63)];
                uint32_t C2 = tile[(q0 + 8) * 64 + ((Ls + lds_rot(q0 + 8)) & 63)], D2 = tile[(q0 + 9) * 64 + ((Ls + lds_rot(q0 + 9)) & 63)];
                o[k] = ((A >> (8 * b0)) & 0xFFu) | (((B >> (8 * b0)) & 0xFFu) << 8) | (((C2 >> (8 * b0)) & 0xFFu) << 16) |
                       (((D2 >> (8 * b0)) & 0xFFu) << 24);
            }
        }
        const int64_t gy = (int64_t)ty * 8 + y, gz = (int64_t)tz * 4 + z;
        *(uint4 *)(O + (int64_t)a.g.X * (gy + (int64_t)a.g.Y * gz)) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

static bool tile_geometry(const BrickSet *bs, TileArgs &a)
{
    const Geom &g = bs->g;
    if (bs->K != 6 || g.D < 6 || g.X < 128 || g.Y < 8 || g.Z < 4) return false;
    const int D = g.D;
    int seen = 0;
    for (int q = 0; q < 3; ++q) {
        if (g.axis[D - 6 + q] != g.axis[D - 3 + q]) return false;
        if (g.bit[D - 6 + q] != 1 || g.bit[D - 3 + q] != 0) return false;
        seen |= 1 << g.axis[D - 3 + q];
    }
    if (seen != 7) return false;
    int pos[3];
    for (int q = 0; q < 3; ++q) pos[g.axis[D - 3 + q]] = 2 - q;   // deepest level -> rank bit 0
    a.jx = pos[0]; a.jy = pos[1]; a.jz = pos[2];
    a.tilesX = g.X / 128; a.tilesY = g.Y / 8; a.tilesZ = g.Z / 4;
    return true;
}

int decode_launch(BrickSet *bs, uint8_t *out, hipStream_t st)
{
    hipEventRecord(bs->ev[5], st);
    TileArgs t;
    if (!getenv("VRHIP_DECODE_V1") && tile_geometry(bs, t)) {
        t.tree = bs->mid.tree; t.treeCap = bs->treeCap;
        t.idxOff = bs->idxOff; t.idxVal = bs->idxVal; t.nIdx = bs->nIdx;
        t.ctrls = bs->mid.ctrl; t.out = out; t.g = bs->g; t.D = bs->D; t.Ds = bs->Ds;
        const int ntiles = t.tilesX * t.tilesY * t.tilesZ;
        hipLaunchKernelGGL(k_decode_tile, dim3((unsigned)((ntiles + DEC_WAVES - 1) / DEC_WAVES), bs->B),
                           dim3(64 * DEC_WAVES), 0, st, t);
    } else {
        DecodeArgs a;
        a.tree = bs->mid.tree; a.treeCap = bs->treeCap;
        a.idxOff = bs->idxOff; a.idxVal = bs->idxVal; a.nIdx = bs->nIdx;
        a.ctrls = bs->mid.ctrl; a.lut = bs->lut; a.out = out; a.g = bs->g;
        a.D = bs->D; a.K = bs->K; a.Ds = bs->Ds;
        hipLaunchKernelGGL(k_decode_lane, dim3((unsigned)((bs->nIdx + 63) / 64), bs->B), dim3(64), 0, st, a);
    }
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
