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
#include <algorithm>
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
    g.D = g.nb[0] + g.nb[1] + g.nb[2];         // R.cpp:26-29 (floor of the logarithms)
    // axis[] / bit[]: the per-depth split axis.  Only meaningful for power-of-two extents, where every node of a
    // depth splits the same axis; general extents go through BrickSet::srcIdx / ownerRank instead.
    for (int d = 0; d < g.D && d < 32; ++d) {  // split-axis rule, R.cpp:151-159
        int sd = d % 3, i = 0;
        while (ext[0] * ext[1] * ext[2] > 1 && ext[sd] == 1) sd = (d + ++i) % 3;
        ext[sd] /= 2;
        int b = 0; while (((int64_t)1 << (b + 1)) <= ext[sd]) ++b;
        g.axis[d] = (uint8_t)sd;
        g.bit[d] = (uint8_t)b;                 // the coordinate bit this split decides
    }
}

// ---- general extents: the two geometry tables (BrickSet::srcIdx, ownerRank) ------------------------------------
// Walks of the reference's own box arithmetic, one thread per leaf / per voxel.  Split sizes do not depend on the
// box position ((2 min + e) / 2 = min + e / 2), but the axis does depend on the node: "while the box has more than
// one cell and extent 1 on the axis, take the next axis" (R.cpp:151-159).
struct Box3 { int lo[3], hi[3]; };
__device__ __forceinline__ int split_axis(const Box3 &b, int depth, bool &more)
{
    const long long e0 = b.hi[0] - b.lo[0], e1 = b.hi[1] - b.lo[1], e2 = b.hi[2] - b.lo[2];
    more = e0 * e1 * e2 > 1;
    int sd = depth % 3, i = 0;
    while (more && (b.hi[sd] - b.lo[sd]) == 1) sd = (depth + ++i) % 3;
    return sd;
}

// buildRecursive's boxes (R.cpp:143-201): a node of one cell still "splits" -- its left child gets an empty box, its
// right child the cell -- and a leaf reads the voxel at its box's min corner whatever the box holds (R.cpp:194-195)
__global__ void __launch_bounds__(256)
k_geom_src(int X, int Y, int Z, int D, uint32_t *__restrict__ src)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if ((unsigned long long)r >= (1ull << D)) return;
    Box3 b = {{0, 0, 0}, {X, Y, Z}};
    for (int d = 0; d < D; ++d) {
        bool more;
        const int sd = split_axis(b, d, more);
        const int mid = (b.lo[sd] + b.hi[sd]) / 2;
        if ((r >> (D - 1 - d)) & 1u) b.lo[sd] = mid; else b.hi[sd] = mid;
    }
    src[r] = (uint32_t)b.lo[0] + (uint32_t)X * ((uint32_t)b.lo[1] + (uint32_t)Y * (uint32_t)b.lo[2]);
}

// levelCut's boxes (R.cpp:790-799, 821-830): a box of one cell is handed to BOTH children unsplit, so below it the
// same cell is written once per leaf and the last leaf in preorder -- all the way right -- wins; a box of more cells
// is split like the encoder's.  Pruned nodes write their whole box: the same value for every leaf rank below them.
// The walk does not stop at the leaves: a grown branch's nodes are "left children" too (R.cpp:806-833), so a leaf box
// of several cells is halved again at every branch level and only what is left at the branch's LAST node is written;
// the other cells of the box keep the zero of the freshly sized output vector (R.cpp:733, SURVEY C-10).
// surv[v] = how many of those halvings (from depth D on) voxel v survives (255: all of them).
__global__ void __launch_bounds__(256)
k_geom_owner(int X, int Y, int Z, int D, uint32_t *__restrict__ owner, uint8_t *__restrict__ surv)
{
    const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
    if (v >= (long long)X * Y * Z) return;
    const int c[3] = {(int)(v % X), (int)((v / X) % Y), (int)(v / ((long long)X * Y))};
    Box3 b = {{0, 0, 0}, {X, Y, Z}};
    uint32_t r = 0;
    for (int d = 0; d < D; ++d) {
        bool more;
        const int sd = split_axis(b, d, more);
        uint32_t bit = 1u;
        if (more) {
            const int mid = (b.lo[sd] + b.hi[sd]) / 2;
            bit = c[sd] >= mid ? 1u : 0u;
            if (bit) b.lo[sd] = mid; else b.hi[sd] = mid;
        }
        r = (r << 1) | bit;
    }
    owner[v] = r;
    int sv = 255;
    for (int j = 0; j < VR_CHAIN_LEVELS; ++j) {
        bool more;
        const int sd = split_axis(b, D + j, more);
        if (!more) break;                       // one cell left: mine, for every further level
        const int mid = (b.lo[sd] + b.hi[sd]) / 2;
        if (c[sd] >= mid) { sv = j; break; }    // the (j+1)-th halving drops me
        b.hi[sd] = mid;
    }
    surv[v] = (uint8_t)sv;
}

int build_general_geometry(BrickSet *bs)
{
    const Geom &g = bs->g;
    const size_t nLeaf = (size_t)1 << g.D;
    if (hipMalloc(&bs->srcIdx, nLeaf * sizeof(uint32_t)) != hipSuccess) return -3;
    if (hipMalloc(&bs->ownerRank, (size_t)g.voxels * sizeof(uint32_t)) != hipSuccess) return -3;
    if (hipMalloc(&bs->ownerSurv, (size_t)g.voxels) != hipSuccess) return -3;
    hipLaunchKernelGGL(k_geom_src, dim3((unsigned)((nLeaf + 255) / 256)), dim3(256), 0, 0, g.X, g.Y, g.Z, g.D, bs->srcIdx);
    hipLaunchKernelGGL(k_geom_owner, dim3((unsigned)((g.voxels + 255) / 256)), dim3(256), 0, 0, g.X, g.Y, g.Z, g.D, bs->ownerRank, bs->ownerSurv);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return launch_status("geometry");
}

// out[voxel] = value of the leaf rank that owns it, if the box its terminal node writes still holds the voxel
// (rankVals: value | branch nodes below the leaf << 8; 0 branch nodes = the leaf or an ancestor is pruned: whole box)
__global__ void __launch_bounds__(256)
k_owner_gather(const uint16_t *__restrict__ rankVals, int64_t leafStride, const uint32_t *__restrict__ owner,
               const uint8_t *__restrict__ surv, int64_t voxels, uint8_t *__restrict__ out)
{
    const int brick = blockIdx.y;
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= voxels) return;
    const uint32_t e = rankVals[(int64_t)brick * leafStride + owner[v]];
    out[(int64_t)brick * voxels + v] = (e >> 8) <= (uint32_t)surv[v] ? (uint8_t)e : (uint8_t)0;
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
    const unsigned long long *idxBase;   // 64-bit trees: stream offset of every 4096-leaf block (idxOff is then block-relative)
    int64_t nBase;
    const uint32_t *idxOff;
    const uint8_t *idxVal;
    int64_t nIdx;
    const Ctrl *ctrls;
    const uint32_t *lut;
    uint8_t *out;
    Geom g;
    int D, K, Ds;
    int cut;                    // progressive cut depth (maxTreeDepth = the reference's levelCut)
    const uint8_t *idxValCut;   // cut < Ds: scalar of every subtree's ancestor at depth `cut`
};

// v1: one lane per subtree, direct byte stores.  RANK_OUT: the value of every leaf RANK goes to a.out (B * 2^D bytes,
// general extents: k_owner_gather then hands the voxels their owners' values) instead of the voxels themselves.
template <bool RANK_OUT>
__global__ void __launch_bounds__(64)
k_decode_lane(DecodeArgs a)
{
    const int brick = blockIdx.y;
    const int64_t s = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (s >= a.nIdx) return;
    const uint32_t off = a.idxOff[(int64_t)brick * a.nIdx + s];
    const int val0 = a.cut < a.Ds ? a.idxValCut[(int64_t)brick * a.nIdx + s] : a.idxVal[(int64_t)brick * a.nIdx + s];
    const uint8_t *dmapG = a.ctrls[brick].distanceMap;
    uint8_t dmap[VR_MAX_DEPTH + 8];
    for (int q = 0; q <= a.D + VR_CHAIN_LEVELS; ++q) dmap[q] = q > a.cut ? 0 : dmapG[q];   // no refinement below the cut
    int ox = 0, oy = 0, oz = 0;
    if (!RANK_OUT) rank_to_xyz(a.g, (uint32_t)(s << a.K), ox, oy, oz);
    uint8_t *O = RANK_OUT ? a.out + 2 * ((int64_t)brick * ((int64_t)1 << a.D) + (s << a.K))
                          : a.out + (int64_t)brick * a.g.voxels + ox + (int64_t)a.g.X * (oy + (int64_t)a.g.Y * oz);
    const int64_t sy = a.g.X, sz = (int64_t)a.g.X * a.g.Y;
    const int K = a.K;
    int branchNodes = 0;      // RANK_OUT: nodes of the grown branch below the leaf being written (0: pruned at or above it)
    auto fill = [&](uint32_t lo, uint32_t cnt, int v) {
        for (uint32_t lr = lo; lr < lo + cnt; ++lr) {
            if (RANK_OUT) { ((uint16_t *)O)[lr] = (uint16_t)(v | (branchNodes << 8)); continue; }
            uint32_t p = a.lut[lr];
            O[(p & 1023u) + sy * ((p >> 10) & 1023u) + sz * (p >> 20)] = (uint8_t)v;
        }
    };
    if (off == VR_IDX_DEAD) { fill(0, 1u << K, val0); return; }
    const uint32_t *W = (const uint32_t *)(a.tree + (int64_t)brick * a.treeCap);
    unsigned long long pos = off;
    if (a.idxBase) pos += a.idxBase[(int64_t)brick * a.nBase + (s >> 6)];
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
            branchNodes = VR_CHAIN_LEVELS;
            for (int c = 1; c <= VR_CHAIN_LEVELS; ++c) {   // grown branch: same voxel, distances 64..1
                int t2 = (W[pos >> 4] >> ((pos & 15u) * 2u)) & 3u;
                ++pos;
                if (t2 == 3) { branchNodes = c; break; }
                v = apply_code(v, t2, dmap[a.D + c]);
            }
            fill(path, 1, v);
            branchNodes = 0;
            terminal = true;
        }
        if (terminal) {
            while (j > 0 && (path & 1u)) { path >>= 1; --j; }
            if (j == 0) break;
            path |= 1u;
        } else { ++j; path <<= 1; }
    }
}

// ---- tile decode --------------------------------------------------------------------
// One lane decodes one depth-(D-6) subtree = a 4x4x4 voxel block (64 leaves, up to 639
// tokens).  A wave takes 32x2x1 such blocks = a 128x8x4 voxel tile whose rows are whole
// 128-byte lines of the output volume.
//   stage   every lane copies the next DEC_SW words of ITS token run into LDS
//           ([word][lane], bank = lane), so the walk can look 16 tokens ahead at any
//           bit position with one ds_read2st64_b32 + v_alignbit;
//   walk    a uniform loop, one action per lane and iteration: a tree token, a
//           grown-branch step, or 4 voxels of a pruned node's fill.  A branch step skips
//           a run of "keep" codes with one ctz and swallows a following terminator, and
//           a leaf swallows an immediate terminator, so a voxel costs ~3 iterations
//           instead of one per token.  Leaves are emitted in stream (Morton) order into an
//           LDS tile of words [leaf / 4][lane] (bank = lane); a pruned pair or quad of voxels
//           costs no extra iteration, larger pruned nodes are filled 8 voxels at a time;
//   gather  the wave re-reads the tile as 16-byte row pieces and stores 8 full 128-byte
//           lines per instruction.
// Requirements: the six deepest split levels cycle through x,y,z twice (any order) and
// X>=128, Y>=8, Z>=4; everything else takes k_decode_lane.
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
    int ltx, lty;               // log2 of tilesX, tilesY (brick extents are powers of two): tile coordinates by shifts
    int cut;                    // progressive cut depth (maxTreeDepth = the reference's levelCut)
    const uint8_t *idxValCut;   // cut < Ds: scalar of every subtree's ancestor at depth `cut`
    const uint32_t *spread;     // BrickSet::spread (coordinate -> Morton rank bits)
    const uint8_t *fine;        // BrickSet::fineIdx (k_decode_fine only)
    const uint32_t *tables;     // BrickSet::decTables (k_decode_fine) / BrickSet::chainTab (k_decode_quad)
    const uint8_t *val3;        // BrickSet::idxVal3 (k_decode_quad only)
    uint8_t kqBit[8];           // k_decode_quad: bit i of a workgroup's ticket number = this bit of the tile's number
};

#define DEC_WAVES 4
#define DEC_SW 19            // staged words per lane (usable lookahead: DEC_SW-2 words per stage); 19 = four blocks per CU in LDS (40 480 B each)

__device__ __forceinline__ int med3i(int x, int lo, int hi)
{   // min(max(x, lo), hi) for lo <= hi in one instruction
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
    return r;
}

__device__ __forceinline__ int mad24i(int x, int y, int z)
{   // x * y + z for operands within 24 bits, one full-rate instruction
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}

__device__ __forceinline__ int clamp_add(int pv, int tok, int dist)
{   // decoder step R.cpp:783-787 / 814-818, branch-free
    const int delta = tok == 1 ? dist : (tok == 2 ? -dist : 0);
    int nv = pv + delta;
    return nv < 0 ? 0 : (nv > 255 ? 255 : nv);
}

// grown-branch tables (256 threads): compose v -> min(max(v + A, LO), HI) over steps first..first+n-1 (distances
// dmS[8 + step]), stop at a terminator (code 3).  Table 1: branch tokens 1-4 (256 entries), table 2: 5-7 (64).
// Entry: [0:10) A + 256, [10:18) LO, [18:26) HI, [26:29) tokens consumed, 29 a terminator ended it.
__device__ __forceinline__ void chain_tables(const uint8_t *dmS, uint32_t *lutC1, uint32_t *lutC2)
{
    const int idx = threadIdx.x;
    for (int tb = lutC1 ? 0 : 1; tb < 2; ++tb) {
        const int first = tb == 0 ? 1 : 5, n = tb == 0 ? 4 : 3;
        if (tb == 1 && idx >= 64) break;
        int A = 0, LO = 0, HI = 255, len = 0, term = 0;
        for (int q = 0; q < n; ++q) {
            const int tok = (idx >> (2 * q)) & 3;
            ++len;
            if (tok == 3) { term = 1; break; }
            const int dist = dmS[8 + first + q];
            const int dl = tok == 1 ? dist : (tok == 2 ? -dist : 0);
            A += dl;
            LO += dl; LO = LO < 0 ? 0 : (LO > 255 ? 255 : LO);
            HI += dl; HI = HI < 0 ? 0 : (HI > 255 ? 255 : HI);
        }
        // k_decode_fine (no table 1) takes LO / HI as whole bytes: min / max read them without a separate extract
        const uint32_t ent = lutC1 ? (uint32_t)(A + 256) | ((uint32_t)LO << 10) | ((uint32_t)HI << 18) | ((uint32_t)len << 26) |
                                         ((uint32_t)term << 29)
                                   : (uint32_t)(A + 256) | ((uint32_t)len << 10) | ((uint32_t)term << 13) | ((uint32_t)LO << 16) |
                                         ((uint32_t)HI << 24);
        if (tb == 0) lutC1[idx] = ent; else lutC2[idx] = ent;
    }
}

// the wave re-reads its LDS tile (words [leaf / 4][subtree], `stride` words per row) as 16-byte row pieces
// and stores whole 128-byte lines of the output volume
__device__ __forceinline__ void tile_gather(const TileArgs &a, const uint32_t *tile, int stride, int live, int brick,
                                            int tx, int ty, int tz, int lane)
{
    const int jx = a.jx, jy = a.jy, jz = a.jz;
    const int c = lane & 7;
    uint8_t *O = a.out + (int64_t)brick * a.g.voxels + (int64_t)tx * 128 + c * 16;
    const auto tile_at = [stride](int rank) { return (rank >> 2) * (4 * stride) + (rank & 3); };   // byte of a leaf within a column
    const int r1 = live * tile_at(1 << jx), r2 = live * tile_at(8 << jx);             // steps of dx bit 0 / bit 1
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int R = st * 8 + (lane >> 3);
        const int y = R & 7, z = R >> 3;
        const int dy = y & 3, dz = z & 3;
        const int rb = ((dy & 1) << jy) | ((dy >> 1) << (3 + jy)) | ((dz & 1) << jz) | ((dz >> 1) << (3 + jz));
        const uint8_t *src = (const uint8_t *)tile + tile_at(rb) * live + 4 * (4 * c + 32 * (y >> 2));
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint8_t *sk = src + 4 * k;
            o[k] = (uint32_t)sk[0] | ((uint32_t)sk[r1] << 8) | ((uint32_t)sk[r2] << 16) | ((uint32_t)sk[r1 + r2] << 24);
        }
        const int64_t gy = (int64_t)ty * 8 + y, gz = (int64_t)tz * 4 + z;
        *(uint4 *)(O + (int64_t)a.g.X * (gy + (int64_t)a.g.Y * gz)) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}


#ifndef QD_NT
#define QD_NT 1
#endif
// decoded voxels are written once and not read again by this launch: streaming stores keep them from displacing the
// stream and side-car lines the neighbouring tiles are about to read
__device__ __forceinline__ void store_out16(uint8_t *p, uint4 v)
{
#if QD_NT
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, (u32x4 *)p);
#else
    *(uint4 *)p = v;
#endif
}

// the same gather with wide LDS reads (k_decode_quad).  A 16-byte row piece = the voxels x = 16c .. 16c+15 of one (y, z)
// = the same leaf positions of four neighbouring blocks 4c .. 4c+3, whose tile words are consecutive: one ds_read_b128
// per (dx bit 1 [, dx bit 0]) brings them all, v_perm picks the bytes.  y = lane >> 3 and z = the store's index.
__device__ __forceinline__ void tile_gather_wide(const TileArgs &a, const uint32_t *tile, int stride, int brick,
                                                 int tx, int ty, int tz, int lane)
{
    const int jx = a.jx, jy = a.jy, jz = a.jz;
    const int c = lane & 7, y = lane >> 3, dy = y & 3;
    uint8_t *O = a.out + (int64_t)brick * a.g.voxels + (int64_t)tx * 128 + c * 16 +
                 (int64_t)a.g.X * ((int64_t)ty * 8 + y + (int64_t)a.g.Y * ((int64_t)tz * 4));
    const int64_t zs = (int64_t)a.g.X * a.g.Y;
    const uint32_t rby = ((uint32_t)(dy & 1) << jy) | ((uint32_t)(dy >> 1) << (3 + jy));
    const uint32_t *col = tile + 4 * c + 32 * (y >> 2);              // my four blocks' column of the tile
    if (jx < 2) {                                                     // dx bit 0 lives in the byte index
#pragma unroll
        for (int z = 0; z < 4; ++z) {
            const uint32_t rb = rby | ((uint32_t)(z & 1) << jz) | ((uint32_t)(z >> 1) << (3 + jz));
            const uint32_t b0 = rb & 3u, b1 = b0 | (1u << jx);
            const uint32_t sel = b0 | (b1 << 8) | ((4u + b0) << 16) | ((4u + b1) << 24);
            const uint32_t g0 = rb >> 2, g1 = g0 | (2u << jx);
            const uint4 A = *(const uint4 *)(col + g0 * stride), B = *(const uint4 *)(col + g1 * stride);
            store_out16(O + zs * z, make_uint4(__builtin_amdgcn_perm(B.x, A.x, sel), __builtin_amdgcn_perm(B.y, A.y, sel),
                                               __builtin_amdgcn_perm(B.z, A.z, sel), __builtin_amdgcn_perm(B.w, A.w, sel)));
        }
    } else {                                                          // jx == 2: dx bit 0 is bit 0 of the word's row
#pragma unroll
        for (int z = 0; z < 4; ++z) {
            const uint32_t rb = rby | ((uint32_t)(z & 1) << jz) | ((uint32_t)(z >> 1) << (3 + jz));
            const uint32_t b = rb & 3u, sel = b | ((4u + b) << 8) | 0x0c0c0000u;
            const uint32_t g0 = rb >> 2;
            const uint4 A = *(const uint4 *)(col + g0 * stride), B = *(const uint4 *)(col + (g0 | 1u) * stride),
                        C = *(const uint4 *)(col + (g0 | 8u) * stride), E = *(const uint4 *)(col + (g0 | 9u) * stride);
            const auto mk = [sel](uint32_t w00, uint32_t w01, uint32_t w10, uint32_t w11) {
                return __builtin_amdgcn_perm(__builtin_amdgcn_perm(w11, w10, sel), __builtin_amdgcn_perm(w01, w00, sel), 0x05040100u);
            };
            store_out16(O + zs * z, make_uint4(mk(A.x, B.x, C.x, E.x), mk(A.y, B.y, C.y, E.y), mk(A.z, B.z, C.z, E.z), mk(A.w, B.w, C.w, E.w)));
        }
    }
}

// a tile whose 64 blocks all lie under pruned nodes: row 0 of the tile holds one replicated value per block, a row
// piece is four consecutive words of it, the same for the tile's four z
__device__ __forceinline__ void tile_fill_dead(const TileArgs &a, const uint32_t *tile, int brick, int tx, int ty, int tz, int lane)
{
    const int c = lane & 7, y = lane >> 3;
    uint8_t *O = a.out + (int64_t)brick * a.g.voxels + (int64_t)tx * 128 + c * 16 +
                 (int64_t)a.g.X * ((int64_t)ty * 8 + y + (int64_t)a.g.Y * ((int64_t)tz * 4));
    const int64_t zs = (int64_t)a.g.X * a.g.Y;
    const uint4 v = *(const uint4 *)(tile + 4 * c + 32 * (y >> 2));
#pragma unroll
    for (int z = 0; z < 4; ++z) store_out16(O + zs * z, v);
}

__global__ void __launch_bounds__(64 * DEC_WAVES)
k_decode_tile(TileArgs a)
{
    __shared__ uint32_t tileS[DEC_WAVES][16 * 64];     // [leaf / 4][lane], four consecutive leaves (Morton ranks) per word
    __shared__ uint32_t strS[DEC_WAVES][DEC_SW * 64];   // [word][lane]
    __shared__ uint8_t stkS[DEC_WAVES][8 * 64];         // [level][lane]
    __shared__ uint8_t dmS[16];      // [1..6] tree levels Ds+1..D, [9..15] grown-branch levels D+1..D+7
    __shared__ uint32_t lutP[260];   // tree-token action table [level][code][next code] (built below from dmS);
                                     // [129 + key]: second word (one bank further: both come with one ds_read2_b32)
    __shared__ uint32_t lutC1[256], lutC2[64];   // grown-branch tables: branch tokens 1-4 and 5-7
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int brick = blockIdx.y;
    const int tileId = blockIdx.x * DEC_WAVES + wave;
    uint32_t *tile = tileS[wave];
    uint32_t *str = strS[wave];
    uint8_t *stk = stkS[wave];
    const bool tileValid = tileId < a.tilesX * a.tilesY * a.tilesZ;
    if (threadIdx.x < 16) {
        const uint8_t *dmap = a.ctrls[brick].distanceMap;
        const int t = threadIdx.x;
        // [0] = 0: the subtree root keeps the value stored in the index; levels below a progressive cut
        // refine nothing (distance 0), so every voxel gets the scalar of its ancestor at the cut depth
        const int depth = t < 8 ? a.Ds + (t < 7 ? t : 0) : a.D + (t - 8);
        dmS[t] = (t == 0 || depth > a.cut) ? 0 : dmap[depth];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        // tree-token action table, keyed by (level j, the node's code, the next token): a node that descends
        // is handled together with its left child, whose token follows it immediately.  Packed:
        //   [0:9) delta of the node + 256   [9:18) delta of the left child + 256 (256 = none)
        //   18 second (left child handled)  [19:21) descents (0: node terminal, 1: child terminal -> continue at the
        //   node's right child, 2: continue at the child's left child)  21 lf (a voxel leaf with a branch behind it)
        //   [22:25) voxels the step emits at once: a voxel leaf (1) or a pruned pair / quad (2, 4)
        //   [25:29) size / 8 of a larger pruned node (filled 8 voxels per iteration)   30 terminal   31 valid
        // second word, what the step would otherwise derive with compares: [0:10) byte offset of the value-stack
        // row the node's value goes to (scratch row 6 when it does not descend), [10:20) same for the child,
        // [20:23) bits of tree tokens consumed, [23:25) path shift, 25 path "or 1" (continue at the right child)
        const int key = threadIdx.x, j = key >> 4, t0 = key & 3, t1 = (key >> 2) & 3;
        uint32_t ent = 256u | (256u << 9);              // row 7: the no-op entry idle lanes read
        uint32_t ent2 = (6u * 64u) | ((6u * 64u) << 10);
        if (j < 7) {
            const auto delta = [&](int tok, int lvl) { const int d = dmS[lvl]; return tok == 1 ? d : (tok == 2 ? -d : 0); };
            const bool is30 = t0 == 3, desc0 = !is30 && j < 6;
            int d1 = 0, second = 0, ndesc = 0, lf, single, fillc;
            if (!desc0) {
                lf = (j == 6 && !is30) ? 1 : 0;
                fillc = (is30 && j < 6) ? (64 >> j) : 0;
                single = j == 6 ? 1 : 0;
            } else {
                const bool is31 = t1 == 3, leaf1 = j + 1 == 6;
                second = 1;
                d1 = delta(t1, j + 1);
                ndesc = (!is31 && !leaf1) ? 2 : 1;
                lf = (leaf1 && !is31) ? 1 : 0;
                fillc = (is31 && !leaf1) ? (64 >> (j + 1)) : 0;
                single = leaf1 ? 1 : 0;
            }
            const int size = single ? 1 : fillc;
            ent = (uint32_t)(delta(t0, j) + 256) | ((uint32_t)(d1 + 256) << 9) | ((uint32_t)second << 18) | ((uint32_t)ndesc << 19) |
                  ((uint32_t)lf << 21) | ((uint32_t)(size <= 4 ? size : 0) << 22) | ((uint32_t)(size >= 8 ? size >> 3 : 0) << 25) |
                  ((ndesc < 2 ? 1u : 0u) << 30) | (1u << 31);
            ent2 = (uint32_t)((ndesc >= 1 ? j : 6) * 64) | ((uint32_t)((ndesc == 2 ? j + 1 : 6) * 64) << 10) |
                   ((uint32_t)(2 + 2 * second) << 20) | ((uint32_t)ndesc << 23) | ((ndesc == 1 ? 1u : 0u) << 25);
        }
        lutP[key] = ent;
        lutP[129 + key] = ent2;
    }
    chain_tables(dmS, lutC1, lutC2);
    __syncthreads();
    if (!tileValid) return;
    const int tx = tileId & (a.tilesX - 1), ty = (tileId >> a.ltx) & (a.tilesY - 1), tz = tileId >> (a.ltx + a.lty);

    // ---- which subtree is mine
    const int sc[3] = {tx * 32 + (lane & 31), ty * 2 + (lane >> 5), tz};   // subtree coords (units of 4 voxels)
    const uint32_t s = (a.spread[4 * sc[0]] | a.spread[a.g.X + 4 * sc[1]] | a.spread[a.g.X + a.g.Y + 4 * sc[2]]) >> 6;
    const uint32_t off = a.idxOff[(int64_t)brick * a.nIdx + s];
    const int val0 = a.cut < a.Ds ? a.idxValCut[(int64_t)brick * a.nIdx + s] : a.idxVal[(int64_t)brick * a.nIdx + s];

    // a tile whose 64 subtrees all lie under pruned nodes (constant regions) needs no walk: one value per
    // subtree goes to tile row 0 and the gather below reads every leaf from there
    const bool waveDead = __ballot(off != VR_IDX_DEAD) == 0ull;
    if (waveDead) tile[lane] = (uint32_t)val0 * 0x01010101u;
    else {
        const uint32_t *W = (const uint32_t *)(a.tree + (int64_t)brick * a.treeCap);
        const bool dead = off == VR_IDX_DEAD;
        bool done = false;
        uint32_t wbase = dead ? 0u : (off >> 4);
        uint32_t bitpos = dead ? 0u : (off & 15u) * 2u;     // relative to the staged window
        uint32_t p = 1;                 // path with a leading sentinel bit: depth = bitlen(p) - 1
        int v = val0;
        int fill = dead ? 64 : 0;       // voxels of a pruned node still to write
        int leaf = 0;                   // next leaf (Morton rank) to emit
        uint32_t cur = 0;               // the tile word under construction (leaves 4 * (leaf / 4) ..)
        if (dead) p = 0x80000000u;
        // value stack rows 0..5 = pushed ancestors, row 6 = scratch for predicated-off pushes,
        // row 7 = the root's "parent" (the index value itself; dmS[0] = 0 leaves it unchanged)
        stk[7 * 64 + lane] = (uint8_t)val0;
        while (__ballot(!done) != 0ull) {
            // ---- stage the next DEC_SW words of my run
            wbase += bitpos >> 5;
            bitpos &= 31u;
#pragma unroll
            for (int k = 0; k < DEC_SW; ++k) str[k * 64 + lane] = W[wbase + k];
            __builtin_amdgcn_s_waitcnt(0xC07F);
            // ---- walk while somebody can still look ahead.  The body is branch-free per lane:
            // what a tree token decides (signed value delta, terminal?, descend?, fill size) comes
            // packed from an LDS table keyed by (level, code); a voxel leaf's whole grown branch
            // (up to 7 more tokens incl. terminator) is folded into the same step through two more
            // tables that hold, for 4 + 3 branch tokens, the composed clamp-add v -> min(max(v+A,LO),HI),
            // the tokens consumed and whether a terminator ended it.  So a lane spends one iteration
            // per tree node and none on branch tokens, and its booleans never become exec-mask algebra.
            while (true) {
                const bool act = !done && (bitpos >> 5) < DEC_SW - 2;
                if (__ballot(act) == 0ull) break;
                const bool filling = act && fill > 0;
                if (__ballot(filling) != 0ull) {            // wave-uniform: only pruned nodes of 8+ voxels come here
                    if (filling) {
                        uint32_t *t8 = tile + (leaf >> 2) * 64 + lane;
                        cur = (uint32_t)v * 0x01010101u;
                        t8[0] = cur; t8[64] = cur;
                        leaf += 8; fill -= 8;
                        done = done || (fill == 0 && p == 0x80000000u);
                    }
                }
                const bool tk = act && !filling;                    // this lane consumes tokens now
                const uint32_t k = bitpos >> 5;
                const uint32_t w0 = str[k * 64 + lane], w1 = str[(k + 1) * 64 + lane];
                const uint32_t x = __builtin_amdgcn_alignbit(w1, w0, bitpos & 31u);   // 16 tokens ahead
                const uint32_t j = (31u - (uint32_t)__clz((int)p)) & 7u;
                // idle lanes read the no-op row 7: nothing is consumed, written or changed
                const uint32_t ei = (tk ? j : 7u) * 16u + (x & 15u);
                const uint32_t e = lutP[ei], f = lutP[ei + 129u];
                // the one voxel leaf of this step, if any, is the node itself (level 6) or its left child (level 5)
                const uint32_t xc = x >> (j == 6u ? 2u : 4u);
                const uint32_t e1 = lutC1[xc & 255u], e2 = lutC2[(xc >> 8) & 63u];
                const int sv = stk[((j + 7u) & 7u) * 64 + lane];
                const int nv = med3i(sv + (int)(e & 511u) - 256, 0, 255);        // decoder step R.cpp:783-787
                const int nv1 = med3i(nv + (int)((e >> 9) & 511u) - 256, 0, 255);  // ... and the left child's
                const uint32_t ndesc = (e >> 19) & 3u;
                const int vb = nv1;                                              // == nv when there is no second node (delta 0)
                // grown branch of a voxel leaf: first 4 tokens, then (unless terminated) 3 more
                const int b1 = med3i(vb + (int)(e1 & 1023u) - 256, (int)((e1 >> 10) & 255u), (int)((e1 >> 18) & 255u));
                const int b2 = med3i(b1 + (int)(e2 & 1023u) - 256, (int)((e2 >> 10) & 255u), (int)((e2 >> 18) & 255u));
                const uint32_t lf = (e >> 21) & 1u;                              // voxel leaf with a branch behind it
                const uint32_t more = lf & ~(e1 >> 29);                          // first four were no terminator
                const uint32_t clen = ((e1 >> 26) & 7u) + (more ? ((e2 >> 26) & 7u) : 0u);
                const int vo = lf ? (more ? b2 : b1) : vb;
                const bool t = ((e >> 30) & 1u) != 0u;
                const uint32_t ne = (e >> 22) & 7u;                              // 0, 1, 2 or 4 voxels emitted here
                const uint32_t fillc = (e >> 22) & 0x78u;                         // 8 * bits [25:29)
                v = tk ? vo : v;
                stk[(f & 1023u) + lane] = (uint8_t)nv;
                stk[((f >> 10) & 1023u) + lane] = (uint8_t)nv1;
                bitpos += ((f >> 20) & 7u) + (lf ? 2u * clen : 0u);
                // the voxels go into the word under construction, which is (re)written every step: an unfinished or
                // stale word is harmless, the step that completes a word writes all four of its bytes
                const uint32_t em = ((1u << ((ne * 8u) & 31u)) - 1u) << (((uint32_t)leaf & 3u) * 8u) | (uint32_t)((int32_t)(e << 7) >> 31);
                cur = (((uint32_t)vo * 0x01010101u) & em) | (cur & ~em);
                tile[min(leaf >> 2, 15) * 64 + lane] = cur;
                leaf += (int)ne;
                fill = fillc ? (int)fillc : fill;
                uint32_t np = p + 1u;
                np >>= (__ffs((int)np) - 1);
                const bool parked = t && ndesc == 0u && np == 1u;   // no further tokens are mine
                const uint32_t pd = (p << ((f >> 23) & 3u)) | ((f >> 25) & 1u);   // descents: continue below
                p = (t && ndesc == 0u) ? (parked ? 0x80000000u : np) : pd;
                done = done || (parked && fill == 0);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): my LDS writes have landed
    __builtin_amdgcn_wave_barrier();

    tile_gather(a, tile, 64, waveDead ? 0 : 1, brick, tx, ty, tz, lane);
}

// ---- fine tile decode -----------------------------------------------------------------
// Same tile, same gather, but ONE LANE PER FOUR VOXELS and no walk: next to the depth-Ds index there is, for every
// 4-leaf subtree, the number of tokens it owns in preorder (its 7 nodes, their grown branches, and the ancestors down
// from depth Ds whose first leaf is its first leaf): from the fused encoder, or from the host parse of a foreign stream.
//   per tile   one lane per depth-Ds subtree decodes its 15 nodes of depths Ds .. Ds+3 (their tokens head the runs of
//              the even 4-leaf subtrees: prefix sums of the 16 counts, eight unaligned dword loads) and parks, in the
//              tile word of each 4-leaf subtree, the scalar it hangs from (replicated: already final where an
//              ancestor is pruned) with its count XORed into byte 1;
//   16 steps   a DPP row (16 lanes) takes one depth-Ds subtree, four of them per step: a row prefix sum of the counts
//              gives every lane its token offset, its four stream words arrive one step ahead, and it decodes its own
//              7 nodes as straight-line predicated code (at most 39 + 4 tokens, staged through LDS for the
//              bit-addressed lookahead; a voxel leaf with its grown branch is two table lookups);
//   gather     as in k_decode_tile.
#define FD_WAVES 4
#define FD_TS 68          // tile row stride in words: lanes (g, S) -> bank 4g + S, conflict-free

// voxel-leaf table (256 threads, 1024 entries): a leaf's code and the first four tokens of its grown branch,
// composed like chain_tables: v -> min(max(v + A, LO), HI).  [0:10) A + 512, [10:13) tokens consumed (code included),
// 13 ended (pruned leaf or terminator; otherwise branch tokens 5-7 follow: lutC2), byte 2 LO, byte 3 HI.
__device__ __forceinline__ void leaf_table(const uint8_t *dmS, uint32_t *lutL)
{
    for (int idx = threadIdx.x; idx < 1024; idx += 256) {
        int A = 0, LO = 0, HI = 255, len = 0, term = 0;
        for (int q = 0; q < 5; ++q) {
            const int tok = (idx >> (2 * q)) & 3;
            ++len;
            if (tok == 3) { term = 1; break; }
            const int dist = q == 0 ? dmS[6] : dmS[8 + q];
            const int dl = tok == 1 ? dist : (tok == 2 ? -dist : 0);
            A += dl;
            LO += dl; LO = LO < 0 ? 0 : (LO > 255 ? 255 : LO);
            HI += dl; HI = HI < 0 ? 0 : (HI > 255 ? 255 : HI);
        }
        lutL[idx] = (uint32_t)(A + 512) | ((uint32_t)len << 10) | ((uint32_t)term << 13) | ((uint32_t)LO << 16) | ((uint32_t)HI << 24);
    }
}

// k_decode_fine's tables depend on the brick (its distanceMap) and the cut only: built once per decode, one block
// per brick, and copied into LDS by every block of the brick (building them in place cost each block as many
// instructions as a fifth of its tiles).  Layout: leaf table (1024 words), branch table 2 (64), distances (16 bytes).
#define FD_TABLE_WORDS (1024 + 64 + 4)

__global__ void __launch_bounds__(256)
k_fine_tables(const Ctrl *ctrls, int D, int Ds, int cut, uint32_t *tables)
{
    __shared__ uint8_t dmS[16];
    __shared__ uint32_t lutL[1024], lutC2[64];
    const int brick = blockIdx.x;
    if (threadIdx.x < 16) {
        const uint8_t *dmap = ctrls[brick].distanceMap;
        const int t = threadIdx.x;
        const int depth = t < 8 ? Ds + (t < 7 ? t : 0) : D + (t - 8);
        dmS[t] = (t == 0 || depth > cut) ? 0 : dmap[depth];      // as in k_decode_tile
    }
    __syncthreads();
    chain_tables(dmS, nullptr, lutC2);
    leaf_table(dmS, lutL);
    __syncthreads();
    uint32_t *o = tables + (int64_t)brick * FD_TABLE_WORDS;
    for (int i = threadIdx.x; i < 1024; i += 256) o[i] = lutL[i];
    if (threadIdx.x < 64) o[1024 + threadIdx.x] = lutC2[threadIdx.x];
    if (threadIdx.x < 4) o[1088 + threadIdx.x] = ((const uint32_t *)dmS)[threadIdx.x];
}

__global__ void __launch_bounds__(64 * FD_WAVES, 6)
k_decode_fine(TileArgs a)
{
    __shared__ uint32_t tileS[FD_WAVES][16 * FD_TS];    // [leaf / 4][subtree]
    __shared__ uint32_t strS[FD_WAVES][4 * 64];         // [word][lane]: my 4 stream words
    __shared__ uint32_t offS[FD_WAVES][64];             // per subtree of the tile: token offset of its root
    __shared__ uint32_t tabS[FD_TABLE_WORDS];
    uint32_t *lutL = tabS, *lutC2 = tabS + 1024;
    const uint8_t *dmS = (const uint8_t *)(tabS + 1088);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int brick = blockIdx.y;
    const int tileId = blockIdx.x * FD_WAVES + wave;
    uint32_t *tile = tileS[wave];
    uint32_t *str = strS[wave];
    const bool tileValid = tileId < a.tilesX * a.tilesY * a.tilesZ;
    const int tx = tileId & (a.tilesX - 1), ty = (tileId >> a.ltx) & (a.tilesY - 1), tz = tileId >> (a.ltx + a.lty);
    uint32_t off = VR_IDX_DEAD;
    int val0 = 0;
    uint4 cv = make_uint4(0, 0, 0, 0);
    if (tileValid) {
        const int sc[3] = {tx * 32 + (lane & 31), ty * 2 + (lane >> 5), tz};
        const uint32_t s = (a.spread[4 * sc[0]] | a.spread[a.g.X + 4 * sc[1]] | a.spread[a.g.X + a.g.Y + 4 * sc[2]]) >> 6;
        off = a.idxOff[(int64_t)brick * a.nIdx + s];
        val0 = a.cut < a.Ds ? a.idxValCut[(int64_t)brick * a.nIdx + s] : a.idxVal[(int64_t)brick * a.nIdx + s];
        if (off != VR_IDX_DEAD) cv = *(const uint4 *)(a.fine + ((int64_t)brick * a.nIdx + s) * 16);
    }
    const unsigned long long liveMask = __ballot(off != VR_IDX_DEAD);
    // blocks whose tiles all lie in pruned regions (constant bricks, pure fluid) need no tables
    if (__syncthreads_or(liveMask != 0ull ? 1 : 0)) {
        const uint32_t *tg = a.tables + (int64_t)brick * FD_TABLE_WORDS;
        for (int i = threadIdx.x; i < FD_TABLE_WORDS; i += 64 * FD_WAVES) tabS[i] = tg[i];
        __syncthreads();
    }
    if (!tileValid) return;
    if (liveMask == 0ull) tile[lane] = (uint32_t)val0 * 0x01010101u;      // as in k_decode_tile
    else {
        offS[wave][lane] = off;
        const int d1 = dmS[1], d2 = dmS[2], d3 = dmS[3], d4 = dmS[4], d5 = dmS[5];
        const uint8_t *TB = a.tree + (int64_t)brick * a.treeCap;
        const uint32_t *W = (const uint32_t *)TB;
        {
            // ---- once per tile, one lane per block: the 15 nodes of depths Ds .. Ds+3.  Their tokens head the runs of
            // the even 4-leaf subtrees (prefix sums of the counts); every subtree's tile word starts out as the scalar
            // its root hangs from (the final word where an ancestor is pruned), replicated.
            const bool deadB = off == VR_IDX_DEAD;
            const uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w};
            uint32_t P[8], x[8];
            P[0] = deadB ? 0u : off;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t pr = (cw[i] & 0x00FF00FFu) + ((cw[i] >> 8) & 0x00FF00FFu);     // two pair sums
                P[2 * i + 1] = P[2 * i] + (pr & 0xFFFFu);
                if (i < 3) P[2 * i + 2] = P[2 * i + 1] + (pr >> 16);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {           // >= 13 tokens from token P[i] on (one unaligned load)
                struct __attribute__((packed, aligned(1))) U32u { uint32_t v; };
                x[i] = ((const U32u *)(TB + (P[i] >> 2)))->v >> ((P[i] & 3u) * 2u);
            }
            const auto step = [](int &v, bool &al, uint32_t tok, int dist) {    // R.cpp:783-787 under "not pruned yet"
                tok = al ? (tok & 3u) : 0u;
                v = clamp_add(v, (int)tok, dist);
                al = al && tok != 3u;
            };
            int v0 = val0;
            bool a0 = !deadB;
            step(v0, a0, x[0], 0);                                   // depth Ds keeps the index value
            int v1[2], v2[4], v3[8];
            bool a1[2], a2[4], a3[8];
#pragma unroll
            for (int i = 0; i < 2; ++i) {           // owners: subtrees 0 (after 1 token), 8
                v1[i] = v0; a1[i] = a0;
                step(v1[i], a1[i], i == 0 ? x[0] >> 2 : x[4], d1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {           // owners: subtrees 0 (after 2), 4, 8 (after 1), 12
                v2[i] = v1[i >> 1]; a2[i] = a1[i >> 1];
                step(v2[i], a2[i], x[2 * i] >> (i == 0 ? 4 : (i == 2 ? 2 : 0)), d2);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {           // owners: subtrees 0 (after 3), 2, 4 (after 1), 6, 8 (after 2), 10, 12 (after 1), 14
                v3[i] = v2[i >> 1]; a3[i] = a2[i >> 1];
                step(v3[i], a3[i], x[i] >> (i == 0 ? 6 : (i == 4 ? 4 : ((i & 1) ? 0 : 2))), d3);
            }
#pragma unroll
            // (+ the subtree's count, XORed into byte 1: zero for every subtree whose word is already final)
            for (int gg = 0; gg < 16; ++gg)
                tile[gg * FD_TS + lane] = ((uint32_t)v3[gg >> 1] * 0x01010101u) ^ (((cw[gg >> 2] >> ((gg & 3) * 8)) & 255u) << 8);
        }
        const int g = lane & 15;
        const uint32_t ownN = g == 0 ? 4u : (uint32_t)(__ffs(g) - 1);     // ancestors (depth >= Ds) whose tokens head my run
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        // the stream words of step `it` are requested one step ahead
        uint32_t nw0 = 0, nw1 = 0, nw2 = 0, nw3 = 0, nbit0 = 0, nM = 0, nV = 0;
        const auto request = [&](int it) {
            if ((((uint32_t)(liveMask >> (4 * it))) & 15u) == 0u) return;       // wave-uniform
            const int S = 4 * it + (lane >> 4);
            const uint32_t so = offS[wave][S];
            const uint32_t tw = tile[g * FD_TS + S];            // parked by the per-tile pass: scalar and count
            const uint32_t c = ((tw >> 8) ^ tw) & 255u;
            nV = tw & 255u;
            uint32_t inc = c;                                   // prefix sum within the row of 16 lanes
            inc += dpp_u32<0x111, 0xf>(0, inc);
            inc += dpp_u32<0x112, 0xf>(0, inc);
            inc += dpp_u32<0x114, 0xf>(0, inc);
            inc += dpp_u32<0x118, 0xf>(0, inc);
            const bool deadRow = so == VR_IDX_DEAD;
            const uint32_t tokpos = (deadRow ? 0u : so) + inc - c;
            const uint32_t *Wp = W + (tokpos >> 4);
            nbit0 = (tokpos & 15u) * 2u;
            // my root exists <=> I own more tokens than the ancestors heading my run (a pruned ancestor ends the run)
            nM = c > ownN ? ~0u : 0u;
            nw0 = Wp[0]; nw1 = Wp[1]; nw2 = Wp[2]; nw3 = Wp[3];
        };
        request(0);
        for (int it = 0; it < 16; ++it) {
            const bool liveStep = (((uint32_t)(liveMask >> (4 * it))) & 15u) != 0u;   // wave-uniform
            const uint32_t w0 = nw0, w1 = nw1, w2 = nw2, w3 = nw3, bit0 = nbit0;
            uint32_t M = nM;                    // all ones while no ancestor is pruned (else V is final for my voxels)
            int V = (int)nV;
            if (it + 1 < 16) request(it + 1);
            if (liveStep) {
                const int S = 4 * it + (lane >> 4);
                uint32_t word;
                str[lane] = w0; str[64 + lane] = w1; str[128 + lane] = w2; str[192 + lane] = w3;
                const uint32_t x0 = __builtin_amdgcn_alignbit(w1, w0, bit0);        // my first 16 tokens
                uint32_t cb = M & (2u * ownN);      // bits of my tokens consumed: the ancestors' are done
                // one tree token (R.cpp:783-787) where mask m is set: as a signed 2-bit field, code 1 -> +1, code 2 -> -2,
                // code 3 -> -1, so (s + 1) >> 1 is the sign of the step and s == -1 the pruned node
                const auto node = [&](uint32_t x, uint32_t m, int dist) {
                    const int s2 = __builtin_amdgcn_sbfe((int)x, cb, 2) & (int)m;
                    V = med3i(mad24i((s2 + 1) >> 1, dist, V), 0, 255);
                    M = s2 == -1 ? 0u : M;
                    cb += m & 2u;
                };
                node(x0, M, d4);                                              // my own 4-leaf subtree's root
                const int V4 = V;
                const uint32_t M4 = M;
                const auto window = [&](uint32_t bp) {      // 16 tokens from bit bp of my words
                    const uint32_t k = bp >> 5;
                    // (bits past my fourth word are never part of a token of mine: any word will do there)
                    return __builtin_amdgcn_alignbit(str[min(k + 1u, 3u) * 64 + lane], str[k * 64 + lane], bp & 31u);
                };
                // one voxel leaf: its code + grown branch at the low end of y (R.cpp:655-704 as the decoder sees it)
                const auto leaf = [&](uint32_t y) {
                    const uint32_t e1 = lutL[y & 1023u], e2 = lutC2[(y >> 10) & 63u];
                    const int b1 = min(max(V + (int)(e1 & 1023u) - 512, (int)((e1 >> 16) & 255u)), (int)(e1 >> 24));
                    const int b2 = min(max(b1 + (int)(e2 & 1023u) - 256, (int)((e2 >> 16) & 255u)), (int)(e2 >> 24));
                    const bool ended = ((e1 >> 13) & 1u) != 0u;
                    const uint32_t len = ((e1 >> 10) & 7u) + (ended ? 0u : ((e2 >> 10) & 7u));
                    cb += (2u * len) & M;
                    const int v = ended ? b1 : b2;
                    return (uint32_t)(M ? v : V);
                };
                // first pair: everything up to the end of the first leaf's branch is within x0
                node(x0, M, d5);
                word = leaf(x0 >> cb);
                word |= leaf(window(bit0 + cb)) << 8;
                // second pair
                V = V4; M = M4;
                {
                    const uint32_t yb = window(bit0 + cb), before = cb;
                    cb = 0;
                    node(yb, M, d5);
                    const uint32_t used = cb;
                    cb = before + used;
                    word |= leaf(yb >> used) << 16;
                    word |= leaf(window(bit0 + cb)) << 24;
                }
                tile[g * FD_TS + S] = word;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    tile_gather(a, tile, FD_TS, liveMask == 0ull ? 0 : 1, brick, tx, ty, tz, lane);
}


// ---- quad decode (the roofline kernel) -----------------------------------------------------------
// Same tile, same side-car counts as k_decode_fine, one lane per four voxels, but nothing on the step's critical
// path waits for LDS and a voxel leaf is ONE table lookup:
//   * the scalars of the depth-(D-3) nodes come from a third side-car (8 bytes per depth-Ds node, written by
//     k_index12 / the host parse), so the per-tile pass that decoded the 15 upper nodes of every block -- and
//     re-read their tokens -- is gone: a tile starts by parking (scalar, count) per 4-leaf subtree;
//   * a lane's tokens (at most 4 + 3 + 4 * 8 = 39) sit in three 32-bit windows cut from its four stream words with
//     v_alignbit; where a leaf ends follows from the positions of the '3' tokens in its window
//     (x & x >> 1 & 0x5555, v_ffbl): no table, no LDS round trip between a leaf and its successor;
//   * the grown branch (R.cpp:655-704 as levelCut sees it, R.cpp:783-787) is one lookup in a 16384-entry table keyed
//     by its seven tokens: the composed clamp-add v -> min(max(v + A, LO), HI).  Branch distances are 64..1 for every
//     brick (R.cpp:94-97), so ONE table serves the launch (64 KiB of LDS per 16-wave workgroup); tokens behind the
//     first '3' are forced to '3' before the lookup, so a leaf's key never depends on its successor's tokens;
//   * a pruned node is a token that reads as '3' (also forced where an ancestor is pruned): its subtree's voxels take
//     its parent's scalar through selects, never through exec-mask branches;
//   * the workgroup's front end (index pre-pass into LDS, two tickets per wave, ticket order by emit block, per-level
//     delta tables) and what bounds the launch are described at the kernel and in DESIGN.md 3.3.
#ifndef QD_WAVES
#define QD_WAVES 16
#endif
#ifndef QD_KEYMASK
#define QD_KEYMASK 0xFFFCu      // (timing experiments only: a smaller mask fakes a smaller table)
#endif
#ifndef QD_TPW
#define QD_TPW 16           // tiles per wave: amortises the table copy
#endif
#define QD_TS 68            // tile row stride in words (as FD_TS)
#ifndef QD_PF
#define QD_PF 4             // steps the stream-word requests run ahead
#endif
#ifndef QD_CHAIN_ENTRIES
#define QD_CHAIN_ENTRIES 16384
#endif

// entry: byte 0 = A (int8), byte 2 = LO, byte 3 = HI; `levels` = branch levels at or above the cut (7: levelCut at
// full depth; fewer: progressive cut inside the branch, deeper levels refine nothing)
__global__ void __launch_bounds__(256)
k_chain_table(int levels, uint32_t *__restrict__ out)
{
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    int A = 0, LO = 0, HI = 255;
    for (int q = 0; q < VR_CHAIN_LEVELS; ++q) {
        const int tok = (int)((idx >> (2 * q)) & 3u);
        if (tok == 3) break;
        const int dist = q < levels ? (64 >> q) : 0;
        const int dl = tok == 1 ? dist : (tok == 2 ? -dist : 0);
        A += dl;
        LO += dl; LO = LO < 0 ? 0 : (LO > 255 ? 255 : LO);
        HI += dl; HI = HI < 0 ? 0 : (HI > 255 ? 255 : HI);
    }
    out[idx] = (uint32_t)(A & 255) | ((uint32_t)LO << 16) | ((uint32_t)HI << 24);
}

__device__ __forceinline__ uint32_t ffbl_u32(uint32_t x)
{   // index of the lowest set bit, 0xFFFFFFFF for 0
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t ones_from(uint32_t f, uint32_t x)
{   // x | (~0 << (f & 31)) in one instruction
    uint32_t r;
    asm("v_lshl_or_b32 %0, -1, %1, %2" : "=v"(r) : "v"(f), "v"(x));
    return r;
}

// one voxel leaf: yl = the 16 bits that start at its code.  Returns the voxel; e = bits before its last token
// (the leaf takes e + 2 bits: code + branch tokens up to and including the terminator, at most 8 tokens)
// The three tree levels of a quad move a scalar by 0 / +distance / -distance / 0 for the token 0 / 1 / 2 / 3
// (R.cpp:783-787): four words per level in LDS (QuadDelta), read with the token as the address -- on gfx950 the
// arithmetic form (v_bfe_i32, add, shift, v_mad_i32_i24) costs 14 cycles of the SIMD per node, two plain
// instructions and an LDS read that nothing waits for cost 5.
struct QuadDelta { int d4[4], d5[4], d6[4]; };

__device__ __forceinline__ int qd_leaf(uint32_t yl, int V5, const int *delta6, const uint32_t *chainS, uint32_t &e)
{
    const uint32_t T = yl & (yl >> 1) & 0x5555u;            // bit 2i <=> token i is '3' (i = 0: the code, a pruned leaf)
    const uint32_t f = ffbl_u32(T);
    e = min(f, 14u);
    const int dl = *(const int *)((const char *)delta6 + ((yl & 3u) << 2));
    const uint32_t ym = ones_from(f, yl);                   // the tokens from the first '3' on read as '3'
    // branch tokens = bits 2..15.  (Measured on gfx950: a two-operand VALU instruction takes 2 cycles of the SIMD,
    // a three-operand or byte-select (SDWA) one 4 -- scratch/mb/valu_rate.hip -- so the three byte selects below cost
    // what six plain instructions would; reading the entry's bytes with three LDS reads instead was 27 % slower.)
    const uint32_t ent = *(const uint32_t *)((const char *)chainS + (ym & QD_KEYMASK));
    // The leaf's own step clamps to [0, 255] (R.cpp:783-787) before the branch's composed clamp-add f runs; f is
    // monotone with f(0) = LO and f(255) = HI, and f(v) = min(max(v + A, LO), HI) on [0, 255], so
    // f(clamp(x, 0, 255)) = min(max(x + A, LO), HI) for every x: the inner clamp needs no instruction.
    int v = V5 + dl + (int)(int8_t)(ent & 255u);
    v = min(max(v, (int)((ent >> 16) & 255u)), (int)(ent >> 24));
    return v;
}

// a depth-(D-1) node and its two leaves: y = the 32 bits that start at the node's token, yh the 32 after them.
// dead3 = 3 where an ancestor is pruned (the node then reads as pruned), else 0.  Returns the two voxels (bytes 0, 1);
// used = bits the pair takes.
__device__ __forceinline__ uint32_t qd_pair(uint32_t y, uint32_t yh, uint32_t dead3, int Vp, const QuadDelta *qd,
                                            const uint32_t *chainS, uint32_t &used)
{
    const uint32_t c5 = (y & 3u) | dead3;
    const int V5 = med3i(Vp + *(const int *)((const char *)qd->d5 + (c5 << 2)), 0, 255);
    const bool pr = c5 == 3u;                               // pruned (or under a pruned node): both voxels = Vp
    uint32_t e1, e2;
    int v1 = qd_leaf(y >> 2, V5, qd->d6, chainS, e1);
    int v2 = qd_leaf(__builtin_amdgcn_alignbit(yh, y, e1 + 4u), V5, qd->d6, chainS, e2);
    v1 = pr ? Vp : v1;
    v2 = pr ? Vp : v2;
    used = pr ? 2u : e1 + e2 + 6u;
    return (uint32_t)v1 | ((uint32_t)v2 << 8);
}

#ifndef QD_MINW
#define QD_MINW 1
#endif
__global__ void __launch_bounds__(64 * QD_WAVES, QD_MINW)
k_decode_quad(TileArgs a)
{
    // one struct: the table sits at LDS address 0, so a lookup's address is the masked key itself
    struct Shared {
        uint32_t chain[QD_CHAIN_ENTRIES];
        uint32_t tile[QD_WAVES][16 * QD_TS];    // [leaf / 4][block of the tile]
        uint32_t off[QD_WAVES][64];             // per block of the tile: token offset of its root
        unsigned long long live[QD_WAVES * QD_TPW];   // per tile of the workgroup: which of its 64 blocks have a live root
        uint8_t val[QD_WAVES * QD_TPW][64];     // ... and the scalars of their roots (all a dead tile needs)
        QuadDelta delta;                        // 0 / +d / -d / 0 per token for depths D-2, D-1, D
        int next;                               // next tile of the workgroup nobody has taken yet
    };
    __shared__ __attribute__((aligned(16))) Shared sm;
    uint32_t *chainS = sm.chain;
    uint32_t (*offS)[64] = sm.off;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int brick = blockIdx.y;
    const int ntiles = a.tilesX * a.tilesY * a.tilesZ;
    const int tile0 = blockIdx.x * (QD_WAVES * QD_TPW);               // the workgroup's tiles: tile0 .. tile0 + 16 * QD_TPW - 1
    uint32_t *tile = sm.tile[wave];
    // ---- which tile a ticket stands for.  A tile is 128 x 8 x 4 voxels (whole 128-byte lines per row); the strings it
    // reads belong to eight 16 x 16 x 16 emit blocks, each shared with the seven other tiles of the same 16 x 16 (y, z)
    // cell.  Tickets are numbered so that those eight tiles are consecutive: the workgroup's waves read an emit block's
    // string while its lines are still in the CU's cache, instead of fetching every line from HBM up to eight times
    // (tile numbers in address order did: 11.3 GB fetched for 4.5 GB of stream and 2 GB of side-cars).
    static_assert(QD_WAVES * QD_TPW == 256, "ticket numbers are permuted as 8-bit numbers");
    const auto tile_of = [&](int kq) {
        int t = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t |= ((kq >> i) & 1) << a.kqBit[i];
        return t;
    };
    // ---- index pre-pass: the root offsets and scalars of all the workgroup's blocks, QD_TPW tiles per wave, as
    // independent loads (one memory round trip for the lot).  A tile under pruned nodes then costs its wave no global
    // load at all -- its per-tile chain spread table -> index -> stores was latency, not bandwidth: a volume of
    // constant blocks decoded at 2 TB/s -- and the table is copied only if some tile needs it.
    bool any = false;
    {
        uint32_t offP[QD_TPW], valP[QD_TPW];
#pragma unroll
        for (int k = 0; k < QD_TPW; ++k) {
            const int tileId = tile0 + tile_of(wave * QD_TPW + k);
            offP[k] = VR_IDX_DEAD; valP[k] = 0;
            if (tileId < ntiles) {
                const int tx = tileId & (a.tilesX - 1), ty = (tileId >> a.ltx) & (a.tilesY - 1), tz = tileId >> (a.ltx + a.lty);
                const int sc[3] = {tx * 32 + (lane & 31), ty * 2 + (lane >> 5), tz};
                const uint32_t s = (a.spread[4 * sc[0]] | a.spread[a.g.X + 4 * sc[1]] | a.spread[a.g.X + a.g.Y + 4 * sc[2]]) >> 6;
                offP[k] = a.idxOff[(int64_t)brick * a.nIdx + s];
                valP[k] = a.idxVal[(int64_t)brick * a.nIdx + s];
            }
        }
#pragma unroll
        for (int k = 0; k < QD_TPW; ++k) {
            const unsigned long long m = __ballot(offP[k] != VR_IDX_DEAD);
            if (lane == 0) sm.live[wave * QD_TPW + k] = m;
            sm.val[wave * QD_TPW + k][lane] = (uint8_t)valP[k];
            any = any || m != 0ull;
        }
    }
    if (threadIdx.x == 0) sm.next = 0;
    if (threadIdx.x < 12) {         // distances of depths D-2, D-1, D (0 below a progressive cut)
        const uint8_t *dmap = a.ctrls[brick].distanceMap;
        const int lv = threadIdx.x >> 2, tok = threadIdx.x & 3, depth = a.D - 2 + lv;
        const int dist = depth <= a.cut ? dmap[depth] : 0;
        (&sm.delta.d4[0])[threadIdx.x] = tok == 1 ? dist : (tok == 2 ? -dist : 0);
    }
    if (__syncthreads_or(any ? 1 : 0)) {
        const uint4 *src = (const uint4 *)a.tables;
        uint4 *dst = (uint4 *)chainS;
#pragma unroll
        for (int i = 0; i < QD_CHAIN_ENTRIES / 4 / (64 * QD_WAVES); ++i) dst[i * 64 * QD_WAVES + threadIdx.x] = src[i * 64 * QD_WAVES + threadIdx.x];
        __syncthreads();
    }
    const QuadDelta *qd = &sm.delta;
    const int g = lane & 15;
    const uint32_t ownN = g == 0 ? 4u : (uint32_t)(__ffs(g) - 1);     // ancestors (depth >= Ds) whose tokens head my run
    const uint32_t p0 = 2u * ownN, p1 = p0 + 2u;                       // bit of my root's token / of my first pair's
    const uint32_t *W = (const uint32_t *)(a.tree + (int64_t)brick * a.treeCap);
    // ---- the waves take the workgroup's tiles from a counter: tiles in pruned regions cost a fraction of a detailed
    // one, and with the table only one workgroup fits a CU, so a fixed split would leave SIMDs idle behind the slowest wave.
    // A wave holds TWO tickets: while it decodes one tile, the index data of its next one (root offsets, side-car counts,
    // depth-(D-3) scalars) is already on its way -- with four waves per SIMD nothing else hides that round trip, and a
    // tile's loads (index -> side-cars -> stream words) were twice its arithmetic.
    // rank of my block inside a tile, and the tile's own part, which is wave-uniform (the spread table is bitwise linear)
    const uint32_t spreadLane = a.spread[4 * (lane & 31)] | a.spread[a.g.X + 4 * (lane >> 5)];
    struct Ticket { int kq; unsigned long long liveMask; uint32_t off, val0; uint4 cv; uint2 sv; };
    const auto take = [&](Ticket &t) {
        int kq = 0;
        if (lane == 0) kq = atomicAdd(&sm.next, 1);
        kq = __builtin_amdgcn_readfirstlane(kq);
        t.kq = kq >= QD_WAVES * QD_TPW ? -1 : kq;       // wave-uniform
        t.liveMask = 0ull; t.off = VR_IDX_DEAD; t.val0 = 0;
        t.cv = make_uint4(0, 0, 0, 0); t.sv = make_uint2(0, 0);
        if (t.kq < 0) return;
        t.liveMask = sm.live[kq];
        t.val0 = sm.val[kq][lane];
        if (t.liveMask != 0ull && ((t.liveMask >> lane) & 1ull)) {
            const int tileId = tile0 + tile_of(kq);
            const int tx = tileId & (a.tilesX - 1), ty = (tileId >> a.ltx) & (a.tilesY - 1), tz = tileId >> (a.ltx + a.lty);
            const uint32_t spreadTile = a.spread[128 * tx] | a.spread[a.g.X + 8 * ty] | a.spread[a.g.X + a.g.Y + 4 * tz];
            const int64_t io = (int64_t)brick * a.nIdx + ((spreadLane | spreadTile) >> 6);
            t.off = a.idxOff[io];          // (cached: the pre-pass has just read it; live <=> not VR_IDX_DEAD)
            t.cv = *(const uint4 *)(a.fine + io * 16);
            t.sv = *(const uint2 *)(a.val3 + io * 8);
        }
    };
    Ticket nxt;
    take(nxt);
    while (nxt.kq >= 0) {
        const Ticket cur = nxt;
        take(nxt);
        const int tileId = tile0 + tile_of(cur.kq);
        if (tileId >= ntiles) continue;       // (a last, partial group: wave-uniform)
        const int tx = tileId & (a.tilesX - 1), ty = (tileId >> a.ltx) & (a.tilesY - 1), tz = tileId >> (a.ltx + a.lty);
        const unsigned long long liveMask = cur.liveMask;
        const uint32_t off = cur.off, val0 = cur.val0;
        if (liveMask == 0ull) tile[lane] = val0 * 0x01010101u;          // as in k_decode_tile: one value per block
        else {
            // ---- park, for each 4-leaf subtree of my block: the scalar of its depth-(D-3) parent (bits 0-7), the token
            // offset of its run inside the block's (8-17: prefix sum of the side-car counts, done once here instead of a
            // DPP scan per step) and whether its root exists (18: it owns more tokens than the ancestors heading its
            // run -- a pruned ancestor ends the run).  A block under a pruned node parks "no root" with offset 0 at the
            // stream's first word, or, where its whole step is skipped, the final words.
            const bool deadB = off == VR_IDX_DEAD;
            const uint4 cv = cur.cv;
            const uint2 sv = cur.sv;
            offS[wave][lane] = deadB ? 0u : off;
            const uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w}, sw[2] = {sv.x, sv.y};
            const uint32_t rep = val0 * 0x01010101u;
            const bool stepDead = ((uint32_t)(liveMask >> (lane & 60)) & 15u) == 0u;     // my step's four blocks are all dead
            const uint32_t deadW = stepDead ? rep : val0;
            uint32_t run = 0;
#ifdef RG_KO_PARK           // (timing experiments)
#pragma unroll
            for (int gg = 0; gg < 1; ++gg) {
#else
#pragma unroll
            for (int gg = 0; gg < 16; ++gg) {
#endif
                const uint32_t cgg = (cw[gg >> 2] >> (8 * (gg & 3))) & 255u;
                const int own = gg == 0 ? 4 : (gg & 1 ? 0 : (gg & 2 ? 1 : (gg & 4 ? 2 : 3)));
                const uint32_t vgg = (sw[gg >> 3] >> (8 * ((gg >> 1) & 3))) & 255u;
                const uint32_t w = vgg | (run << 8) | ((uint32_t)(own - (int)cgg) & 0x40000u);
                tile[gg * QD_TS + lane] = deadB ? deadW : w;
                run += cgg;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            // ---- 16 steps of 4 blocks x 16 lanes.  The stream words of a step are requested QD_PF steps ahead
            // (a slot holds the four words only: scalar, bit offset and root flag are re-read from the park word when the
            // step runs -- two LDS reads per step buy twice the request depth in the same registers)
            uint32_t qw[QD_PF][4];
#pragma unroll
            for (int i = 0; i < QD_PF; ++i) qw[i][0] = qw[i][1] = qw[i][2] = qw[i][3] = 0;
            const auto request = [&](int it, int slot) {
                if ((((uint32_t)(liveMask >> (4 * it))) & 15u) == 0u) return;       // wave-uniform
                const int S = 4 * it + (lane >> 4);
                const uint32_t tokpos = offS[wave][S] + ((tile[g * QD_TS + S] >> 8) & 1023u);
                const uint32_t *Wp = W + (tokpos >> 4);
                qw[slot][0] = Wp[0]; qw[slot][1] = Wp[1]; qw[slot][2] = Wp[2]; qw[slot][3] = Wp[3];
            };
#pragma unroll
            for (int i = 0; i < QD_PF; ++i) request(i, i);
            // one step's decode: my four voxels from my four stream words
            const auto compute = [&](uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t b, int V3, bool live) -> uint32_t {
                // my tokens: bits [0, 96) from my first token on
                const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, b), hi = __builtin_amdgcn_alignbit(w2, w1, b),
                               hh = __builtin_amdgcn_alignbit(w3, w2, b);
                // my depth-(D-2) root (behind the ancestors' tokens)
                const uint32_t c4 = live ? __builtin_amdgcn_ubfe(lo, p0, 2) : 3u;
                const int V4 = med3i(V3 + *(const int *)((const char *)qd->d4 + (c4 << 2)), 0, 255);
                const uint32_t dead3 = c4 == 3u ? 3u : 0u;
                uint32_t used1, used2;
                const uint32_t b01 = qd_pair(__builtin_amdgcn_alignbit(hi, lo, p1), __builtin_amdgcn_alignbit(hh, hi, p1), dead3, V4,
                                             qd, chainS, used1);
                const uint32_t p2 = p1 + used1;                   // <= 10 + 34
                const bool q = p2 >= 32u;
                const uint32_t A_ = q ? hi : lo, B_ = q ? hh : hi, C_ = q ? 0u : hh;
                const uint32_t b23 = qd_pair(__builtin_amdgcn_alignbit(B_, A_, p2), __builtin_amdgcn_alignbit(C_, B_, p2), dead3, V4,
                                             qd, chainS, used2);
                return b01 | (b23 << 16);
            };
            // two steps per trip: their dependency chains (parse -> table lookups -> clamp-adds) are independent, so the
            // scheduler can fill one's LDS waits with the other's arithmetic (a CU holds only 4 such waves per SIMD)
#pragma unroll
            for (int ip = 0; ip < 8; ++ip) {
                const int i0 = 2 * ip, i1 = 2 * ip + 1, s0 = i0 % QD_PF, s1 = i1 % QD_PF;
                const bool live0 = (((uint32_t)(liveMask >> (4 * i0))) & 15u) != 0u, live1 = (((uint32_t)(liveMask >> (4 * i1))) & 15u) != 0u;
                const int S0 = 4 * i0 + (lane >> 4), S1 = 4 * i1 + (lane >> 4);
                const uint32_t a0 = qw[s0][0], a1 = qw[s0][1], a2 = qw[s0][2], a3 = qw[s0][3];
                const uint32_t c0 = qw[s1][0], c1 = qw[s1][1], c2 = qw[s1][2], c3 = qw[s1][3];
                const uint32_t twA = tile[g * QD_TS + S0], twC = tile[g * QD_TS + S1];
                const uint32_t ab = ((offS[wave][S0] + (twA >> 8)) & 15u) * 2u, cb = ((offS[wave][S1] + (twC >> 8)) & 15u) * 2u;
                const int aV = (int)(twA & 255u), cV = (int)(twC & 255u);
                const bool aL = (twA & 0x40000u) != 0u, cL = (twC & 0x40000u) != 0u;
                if (i0 + QD_PF < 16) request(i0 + QD_PF, s0);
                if (i1 + QD_PF < 16) request(i1 + QD_PF, s1);
                if (live0 && live1) {
                    const uint32_t r0 = compute(a0, a1, a2, a3, ab, aV, aL), r1 = compute(c0, c1, c2, c3, cb, cV, cL);
                    tile[g * QD_TS + S0] = r0;
                    tile[g * QD_TS + S1] = r1;
                } else if (live0) tile[g * QD_TS + S0] = compute(a0, a1, a2, a3, ab, aV, aL);
                else if (live1) tile[g * QD_TS + S1] = compute(c0, c1, c2, c3, cb, cV, cL);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        if (liveMask == 0ull) tile_fill_dead(a, tile, brick, tx, ty, tz, lane);
        else tile_gather_wide(a, tile, QD_TS, brick, tx, ty, tz, lane);
        __builtin_amdgcn_s_waitcnt(0xC07F);           // the next tile parks into the words the gather has just read
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- region decode (the roofline kernel since round 3) ---------------------------------------------------------
// k_decode_quad's arithmetic on a different decomposition of the volume.  There a wave owned a 128 x 8 x 4 voxel tile
// (whole output lines) whose tokens lie in EIGHT 4096-leaf block strings, each shared with seven other tiles: every
// step's stream words were a gather of 16-byte requests from four strings, the side-cars were strided, and the
// 64 KiB branch table allowed one 16-wave workgroup per CU.  Here
//   * a WAVE owns one emit block (a depth-(D-12) subtree = 16 x 16 x 16 voxels): ONE contiguous string (read exactly
//     once by the launch, by this wave) and contiguous side-cars (256 B offsets, 64 B scalars, 1 KiB counts, 512 B
//     depth-(D-3) scalars);
//   * the string is streamed into a small per-wave LDS ring by LDS-DMA (global_load_lds_dwordx4, 1 KiB pieces, a few
//     pieces ahead, refilled as steps retire): a step's four stream words are LDS reads, no register is spent on
//     request slots and no step waits for a memory round trip;
//   * a WORKGROUP of eight waves owns the eight emit blocks of a 128 x 16 x 16 region: results go to a shared LDS
//     image and leave, after one barrier, as whole 128-byte lines (eight lines per store instruction);
//   * the grown branch is two lookups (tokens 1-4: 256 entries, 5-7: 64 entries, 1.25 KiB instead of 64 KiB) whose
//     clamp-adds are applied to both leaves of a sibling pair at once in packed 16-bit lanes.
// LDS per workgroup: 1.4 KiB tables + 32.1 KiB image + 8 rings => two or three workgroups per CU.
// Word address of quad q (= leaf rank >> 2, 10 bits) inside its block's image: bits 0-5 = the low six bits of q
// permuted so that the two lowest x bits come first (a 16-byte read then holds four x-neighbours), bits 6-9 = the
// step q >> 6, and bits 2-4 XORed with the step's low three bits, so that the step accesses (64 consecutive words),
// the park stores (one 64-leaf block per lane) and the gather's 16-byte reads (with the block images 4 words apart in
// the banks) are all conflict-free or two-way.
#ifndef RG_NP
#define RG_NP 4             // ring pieces of 1 KiB per wave
#endif
#ifndef RG_MINW
#define RG_MINW 4           // waves per SIMD asked of the register allocator (4: two workgroups per CU, 6: three)
#endif
#ifndef RG_PER
#define RG_PER 16           // regions per workgroup
#endif
#ifndef RG_WAVES
#define RG_WAVES 8          // emit blocks (waves) of a region along x: 8 = whole 128-byte lines, 4 = 64-byte half lines
#endif
#define RG_LW (RG_WAVES == 8 ? 3 : 2)
#define RG_REGX (16 * RG_WAVES)
#define RG_BLK_WORDS 1028
#define RG_RING_MASK (RG_NP * 256 - 1)
static_assert(RG_NP >= 4 && (RG_NP & (RG_NP - 1)) == 0, "the side-cars of the next region are staged in ring slots 2 and 3: at least four pieces, a power of two");

struct RegionArgs {
    const uint8_t *tree;
    int64_t treeCap;
    const uint32_t *idxOff;
    const uint8_t *idxVal;
    const uint8_t *fine;
    const uint8_t *val3;
    int64_t nIdx;
    const Ctrl *ctrls;
    uint8_t *out;
    const uint32_t *spread;
    int X, Y;
    int64_t voxels;
    int D, cut;
    int lrx, lry;               // log2 of the regions along x and y
    int jx;                     // position of x among the three deepest split levels (0 = deepest)
    uint32_t lanePos;           // 6 nibbles: lane bit i handles bit lanePos[i] of (quad & 63)
    uint32_t parkP[4];          // 16 bytes: image word (0..63) of quad g of a 64-leaf block ...
    uint32_t parkS;             // 4 bytes:  ... plus that of the block's two low rank bits
    uint32_t gAddr[4];          // 8 halfwords: image-word contribution of gather bit i (a y or z bit of the 16 x 16 plane)
    uint32_t gByte;             // 8 nibbles:  byte-in-word contribution of gather bit i
    uint32_t gOut[8];           // output byte offset contribution of gather bit i
    uint32_t xRead[2];          // 4 halfwords: image-word XOR of the gather's read k (the x bits above the two lowest)
    uint32_t blkX, blkY, blkZ;  // 6 x 5 bits each: bit k of x >> 4 (y >> 4, z >> 4) is this bit of the emit block's number (= leaf rank >> 12)
    int nreg;                   // regions of a brick
    unsigned long long *dbg;    // RG_STAMP builds only: cycle sums (total, park, steps, barrier 1, gather, barrier 2)
};

__device__ __forceinline__ uint32_t wave_incl_scan_max_dpp(uint32_t v)
{
    v = max(v, dpp_u32<0x111, 0xf>(0, v));
    v = max(v, dpp_u32<0x112, 0xf>(0, v));
    v = max(v, dpp_u32<0x114, 0xf>(0, v));
    v = max(v, dpp_u32<0x118, 0xf>(0, v));
    v = max(v, dpp_u32<0x142, 0xa>(0, v));
    v = max(v, dpp_u32<0x143, 0xc>(0, v));
    return v;
}

// A workgroup barrier that orders LDS only.  __syncthreads() also drains the vector-memory counter while an LDS-DMA
// is pending (its fence cannot tell the DMA's LDS write from a store): every barrier of the pipelined loop would
// wait for the next region's loads and string pieces, which exist to be in flight across it.
__device__ __forceinline__ void rg_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct RegionShared {
    uint32_t leafA[1024];           // LDS address 0: a leaf's code + branch tokens 1-4 -> A (16 bits) | LO << 16 | HI << 24
    uint32_t chainB[64];            // branch tokens 5-7 -> A (16 bits) | LO << 16 | HI << 24
    int d4[4], d5[4];               // 0 / +d / -d / 0 per token at depths D-2, D-1
    uint32_t buf[RG_WAVES * RG_BLK_WORDS];
    uint32_t ring[RG_WAVES][RG_NP * 256 + 4];      // + 4: the ring's first words again (a lane reads 4 consecutive words)
};

// a depth-(D-1) node and its two leaves.  y = the 32 bits that start at the node's token, yh the 32 after them;
// dead3 = 3 where an ancestor is pruned.  Returns the two voxels (bytes 0, 1); used = bits the pair takes.
// A pruned pair is data, not control: its leaves' tokens are forced to read '3' (code step 0, both branch tables the
// identity) and its own step is 0, so the table path returns the parent's scalar for both voxels.
__device__ __forceinline__ uint32_t rg_pair(uint32_t y, uint32_t yh, uint32_t dead3, int Vp, const RegionShared &sm, uint32_t &used)
{
    const uint32_t c5 = (y & 3u) | dead3;
    const int V5 = med3i(Vp + *(const int *)((const char *)sm.d5 + (c5 << 2)), 0, 255);
    const bool pr = c5 == 3u;                               // pruned (or under a pruned node): both voxels = Vp
    const uint32_t keep = pr ? 0u : ~0u;
    // where the two leaves end: a leaf = its code + the branch up to and including the first '3', at most 8 tokens.
    // In a pruned pair both "end at once" (f = 0): their tokens all read '3' below.
    const uint32_t yl1 = y >> 2;
    const uint32_t f1 = ffbl_u32(yl1 & (yl1 >> 1) & 0x5555u) & keep;
    const uint32_t e1 = min(f1, 14u);
    const uint32_t ym1 = ones_from(f1, yl1);                // the tokens from the first '3' on read as '3'
    const uint32_t yl2 = __builtin_amdgcn_alignbit(yh, y, e1 + 4u);
    const uint32_t f2 = ffbl_u32(yl2 & (yl2 >> 1) & 0x5555u) & keep;
    const uint32_t e2 = min(f2, 14u);
    const uint32_t ym2 = ones_from(f2, yl2);
    // both leaves in packed 16-bit lanes: the two composed clamp-adds v -> min(max(v + A, LO), HI).  The first table
    // takes the leaf's code with branch tokens 1-4: the code's own step clamps to [0, 255] before the branch runs, and
    // clamp-adds compose to clamp-adds, so (code, tokens 1-4) is one entry keyed by ten bits of the window
    const uint32_t a1 = *(const uint32_t *)((const char *)sm.leafA + ((ym1 << 2) & 0xFFCu));
    const uint32_t a2 = *(const uint32_t *)((const char *)sm.leafA + ((ym2 << 2) & 0xFFCu));
    const uint32_t b1 = *(const uint32_t *)((const char *)sm.chainB + ((ym1 >> 8) & 0xFCu));
    const uint32_t b2 = *(const uint32_t *)((const char *)sm.chainB + ((ym2 >> 8) & 0xFCu));
    vr_s16x2 v = pk_s((uint32_t)V5 * 0x10001u);
    v = v + pk_s(__builtin_amdgcn_perm(a2, a1, 0x05040100u));                               // A: low halves
    v = __builtin_elementwise_max(v, pk_s(__builtin_amdgcn_perm(a2, a1, 0x0C060C02u)));     // LO: byte 2
    v = __builtin_elementwise_min(v, pk_s(__builtin_amdgcn_perm(a2, a1, 0x0C070C03u)));     // HI: byte 3
    v = v + pk_s(__builtin_amdgcn_perm(b2, b1, 0x05040100u));
    v = __builtin_elementwise_max(v, pk_s(__builtin_amdgcn_perm(b2, b1, 0x0C060C02u)));
    v = __builtin_elementwise_min(v, pk_s(__builtin_amdgcn_perm(b2, b1, 0x0C070C03u)));
    used = pr ? 2u : e1 + e2 + 6u;
    return __builtin_amdgcn_perm(0, pk_u(v), 0x0C0C0200u);
}

// one step's decode: my four voxels from my park word and the four stream words behind it
__device__ __forceinline__ uint32_t rg_quad(uint32_t pw, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t p0, uint32_t p1,
                                            const RegionShared &sm)
{
#ifdef RG_KO_COMPUTE        // (timing experiments: the step's arithmetic replaced by an XOR of its inputs)
    return pw ^ w0 ^ w1 ^ w2 ^ w3 ^ p0;
#endif
    const uint32_t b = (pw >> 7) & 30u;                    // bit of my first token in w0
    const int V3 = (int)(pw & 255u);
    const uint32_t dead = ((pw >> 24) & 1u) - 1u;           // all ones <=> my root does not exist
    // my tokens: bits [0, 96) from my first token on
    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, b), hi = __builtin_amdgcn_alignbit(w2, w1, b),
                   hh = __builtin_amdgcn_alignbit(w3, w2, b);
    const uint32_t c4 = __builtin_amdgcn_ubfe(lo, p0, 2) | (dead & 3u);      // my depth-(D-2) root (behind the ancestors' tokens)
    const int V4 = med3i(V3 + *(const int *)((const char *)sm.d4 + (c4 << 2)), 0, 255);
    const uint32_t dead3 = c4 == 3u ? 3u : 0u;
    uint32_t used1, used2;
    const uint32_t b01 = rg_pair(__builtin_amdgcn_alignbit(hi, lo, p1), __builtin_amdgcn_alignbit(hh, hi, p1), dead3, V4, sm, used1);
    const uint32_t p2 = p1 + used1;                   // <= 10 + 34
    const bool q = p2 >= 32u;
    const uint32_t A_ = q ? hi : lo, B_ = q ? hh : hi, C_ = q ? 0u : hh;
    const uint32_t b23 = rg_pair(__builtin_amdgcn_alignbit(B_, A_, p2), __builtin_amdgcn_alignbit(C_, B_, p2), dead3, V4, sm, used2);
    return b01 | (b23 << 16);
}

// "all but the n youngest vector-memory operations of this wave are done", n known only at run time
__device__ __forceinline__ void rg_vm_wait(uint32_t n)
{
    switch (n < 24u ? n : 24u) {
#define RG_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    RG_W(0) RG_W(1) RG_W(2) RG_W(3) RG_W(4) RG_W(5) RG_W(6) RG_W(7) RG_W(8) RG_W(9) RG_W(10) RG_W(11) RG_W(12)
    RG_W(13) RG_W(14) RG_W(15) RG_W(16) RG_W(17) RG_W(18) RG_W(19) RG_W(20) RG_W(21) RG_W(22) RG_W(23) RG_W(24)
#undef RG_W
    }
}

// LDS-DMA from inline asm: hipcc then neither counts the load nor orders LDS accesses behind it.  (Told about an
// LDS-DMA -- the builtin -- it waits vmcnt(0) before EVERY later LDS access, because it cannot tell the DMA's LDS
// destination from the image; and any register load it does know of it waits for with vmcnt(0) inside a loop, which
// also drains every store and string piece in flight.)  So every load of the pipelined loop is one of these, the
// data lands in LDS, and the waits are counted by hand: rg_vm_wait(operations issued since).
#define RG_DMA(SUFFIX, gptr, ldsByte) do { uint32_t keep_; \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_" SUFFIX " %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(gptr), "s"(ldsByte) : "memory"); } while (0)

#ifndef RG_IDXD
#define RG_IDXD 3           // regions the index entries run ahead
#endif

__global__ void __launch_bounds__(64 * RG_WAVES, RG_MINW)
k_decode_region(RegionArgs a)
{
    // Workgroup i of a brick takes the brick's regions i, i + gridDim.x, ...: tables once, and the regions software-
    // pipelined.  While region k is decoded and stored, the index entries of regions k+1 .. k+RG_IDXD, and the counts,
    // depth-(D-3) scalars and first string pieces of region k+1 are in flight or landed, so in steady state no wave
    // waits for a memory round trip (one workgroup per region spent 2.5 of its 5.4 ms in those round trips alone).
    struct Shared {
        RegionShared t;
        uint32_t idxOff[RG_WAVES][RG_IDXD][64];     // landed index entries: token offset of every 64-leaf block's root ...
        uint32_t idxVal[RG_WAVES][RG_IDXD][16];     // ... and its scalar (64 bytes)
        uint32_t blkTab[3][64];                     // emit block number = blkTab[0][x >> 4] | blkTab[1][y >> 4] | blkTab[2][z >> 4]
    };
    __shared__ __attribute__((aligned(16))) Shared smw;
    RegionShared &sm = smw.t;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int brick = blockIdx.y;
    const int G = (int)gridDim.x;
    const int nreg = a.nreg;
    int rid = (int)blockIdx.x;
    if (rid >= nreg) return;
    // ---- tables
    {
        const uint8_t *dmap = a.ctrls[brick].distanceMap;
        const int t = threadIdx.x;
        if (t < 192) {       // rank bits of the coordinates above the emit block (bit deposit, once per workgroup)
            const int ax = t >> 6, v = t & 63;
            const uint32_t pos = ax == 0 ? a.blkX : (ax == 1 ? a.blkY : a.blkZ);
            uint32_t r = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k) r |= (((uint32_t)v >> k) & 1u) << ((pos >> (5 * k)) & 31u);
            smw.blkTab[ax][v] = r;
        }
        // a clamp-add f(v) = min(max(v + A, LO), HI) with LO = f(0), HI = f(255); steps: [0, 255]-clamped adds
        const auto compose = [](int tok0, bool withCode, int d0, const int *dist, int n, uint32_t key) {
            int A = 0, LO = 0, HI = 255;
            bool go = true;
            if (withCode) {
                go = tok0 != 3;                 // a pruned leaf: nothing follows
                const int dl = tok0 == 1 ? d0 : (tok0 == 2 ? -d0 : 0);
                A += dl; LO = min(max(LO + dl, 0), 255); HI = min(max(HI + dl, 0), 255);
            }
            for (int q = 0; q < n; ++q) {
                const int tok = (int)((key >> (2 * q)) & 3u);
                go = go && tok != 3;
                const int dl = !go ? 0 : (tok == 1 ? dist[q] : (tok == 2 ? -dist[q] : 0));
                A += dl; LO = min(max(LO + dl, 0), 255); HI = min(max(HI + dl, 0), 255);
            }
            return ((uint32_t)A & 0xFFFFu) | ((uint32_t)LO << 16) | ((uint32_t)HI << 24);
        };
        {
            int distA[4], distB[3];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int depth = a.D + 1 + q; distA[q] = depth <= a.cut ? dmap[depth] : 0; }
#pragma unroll
            for (int q = 0; q < 3; ++q) { const int depth = a.D + 5 + q; distB[q] = depth <= a.cut ? dmap[depth] : 0; }
            const int d6 = a.D <= a.cut ? dmap[a.D] : 0;
            for (int e = t; e < 1024; e += 64 * RG_WAVES) sm.leafA[e] = compose(e & 3, true, d6, distA, 4, (uint32_t)e >> 2);
            if (t < 64) sm.chainB[t] = compose(0, false, 0, distB, 3, (uint32_t)t);
        }
        if (t < 8) {
            const int lv = t >> 2, tok = t & 3, depth = a.D - 2 + lv;
            const int dist = depth <= a.cut ? dmap[depth] : 0;
            (lv ? sm.d5 : sm.d4)[tok] = tok == 1 ? dist : (tok == 2 ? -dist : 0);
        }
    }
    rg_barrier();           // the tables
    // ---- per-lane constants
    uint32_t *buf = sm.buf + wave * RG_BLK_WORDS;
    uint32_t *ringW = sm.ring[wave];
    const auto lds_byte = [](const void *p) { return (uint32_t)(size_t)(__attribute__((address_space(3))) const void *)p; };
    const uint32_t ringLds = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_byte(ringW));
    const uint32_t idxOffLds = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_byte(smw.idxOff[wave][0]));
    const uint32_t idxValLds = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_byte(smw.idxVal[wave][0]));
    const uint32_t laneTerm = (uint32_t)(64 * (lane >> 2)) + (((a.parkS >> (8 * (lane & 3))) & 255u) ^ ((uint32_t)((lane >> 2) & 7) << 2));
    const uint32_t *W = (const uint32_t *)(a.tree + (int64_t)brick * a.treeCap);
    const uint32_t capWords = (uint32_t)(a.treeCap >> 2);
    // my role in a step: lane l <-> image word l of the step's 64
    uint32_t qlow = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) qlow |= (((uint32_t)lane >> i) & 1u) << ((a.lanePos >> (4 * i)) & 15u);
    const uint32_t g = qlow & 15u;
    const uint32_t ownN = g == 0u ? 4u : (uint32_t)(__ffs((int)g) - 1);     // ancestors (depth >= D-6) whose tokens head my run
    const uint32_t p0 = 2u * ownN, p1 = p0 + 2u;                           // bit of my root's token / of my first pair's
    // my rows of the gather: the 8 - RG_LW low gather bits <- lane >> RG_LW and the store's index (two bits), the RG_LW
    // high ones <- wave.  (Gather bits 0 .. 5-RG_LW from the lane, then 2 of the store, then the wave's.)
    uint32_t addrL = 0, byteL = 0, outL = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i >= 6 - RG_LW && i < 8 - RG_LW) continue;
        const bool on = i < 6 - RG_LW ? ((lane >> (RG_LW + i)) & 1) != 0 : ((wave >> (i - (8 - RG_LW))) & 1) != 0;
        addrL ^= on ? (a.gAddr[i >> 1] >> (16 * (i & 1))) & 0xFFFFu : 0u;
        byteL |= on ? (a.gByte >> (4 * i)) & 15u : 0u;
        outL += on ? a.gOut[i] : 0u;
    }
    // ---- the pipeline's bookkeeping (all wave-uniform)
    uint32_t ops = 0;                       // vector-memory operations this wave has issued in the loop (DMAs and stores)
    uint32_t markIdx[RG_IDXD];              // ... as of just after the index entries of a slot were requested,
    uint32_t markP[RG_NP];                  // ... and just after the piece in a ring slot was.  "Done" = all but the (ops - mark)
                                            // operations issued since are: loads return in issue order.  (Arrays indexed by
                                            // unrolled compare-selects only: they stay in scalar registers.)
    const auto mark_of_idx = [&](int slot) { uint32_t m = 0;
#pragma unroll
        for (int i = 0; i < RG_IDXD; ++i) m = slot == i ? markIdx[i] : m;
        return m; };
    const auto mark_of_piece = [&](uint32_t p) { uint32_t m = 0;
#pragma unroll
        for (int i = 0; i < RG_NP; ++i) m = (p & (RG_NP - 1)) == (uint32_t)i ? markP[i] : m;
        return m; };
#pragma unroll
    for (int i = 0; i < RG_NP; ++i) markP[i] = 0;
    const auto index_of = [&](int r) -> int64_t {       // first index entry of my emit block (the wave-th along x) of region r
        const uint32_t rx = (uint32_t)r & ((1u << a.lrx) - 1u), ry = ((uint32_t)r >> a.lrx) & ((1u << a.lry) - 1u), rz = (uint32_t)r >> (a.lrx + a.lry);
        const uint32_t blk = smw.blkTab[0][rx * (uint32_t)RG_WAVES + (uint32_t)wave] | smw.blkTab[1][ry] | smw.blkTab[2][rz];
        return (int64_t)brick * a.nIdx + ((int64_t)blk << 6);
    };
    const auto request_index = [&](int r, int slot) {    // 256 + 64 bytes into index slot `slot`
        const int64_t io = index_of(r);
        RG_DMA("dword", a.idxOff + io + lane, idxOffLds + (uint32_t)slot * 256u);
        if (lane < 16) RG_DMA("dword", (const uint32_t *)(a.idxVal + io) + lane, idxValLds + (uint32_t)slot * 64u);
        ops += 2;
#pragma unroll
        for (int i = 0; i < RG_IDXD; ++i) markIdx[i] = slot == i ? ops : markIdx[i];
    };
    const auto issue_piece = [&](uint32_t wbase, uint32_t p) {      // piece p of a string: words [wbase + 256 p, + 256) -> ring slot p mod RG_NP
        const uint32_t w = wbase + 256u * p + 4u * (uint32_t)lane;
        if (w + 4u <= capWords) RG_DMA("dwordx4", W + w, ringLds + (p & (RG_NP - 1)) * 1024u);
        ++ops;
#pragma unroll
        for (int i = 0; i < RG_NP; ++i) markP[i] = (p & (RG_NP - 1)) == (uint32_t)i ? ops : markP[i];
    };
    // the staged state of a region
    unsigned long long liveMask = 0ull;
    uint32_t wbase = 0, totalPieces = 0, issued = 0, markSide = 0, offC = VR_IDX_DEAD, valC = 0;
    // From the offsets alone: the string runs from the first live 64-leaf block's root to (a bound on) the last one's
    // end.  Its first two pieces go to ring slots 0 and 1; the counts (1 KiB) and depth-(D-3) scalars (512 B) of the
    // region land in slots 3 and 2, which the park reads before the string's pieces 2 and 3 are requested: a trip of
    // two steps takes fewer than 512 words, so the first one never needs piece 2.
    const auto stage = [&](int r, int slot) {
        rg_vm_wait(ops - mark_of_idx(slot));
        offC = smw.idxOff[wave][slot][lane];
        valC = (smw.idxVal[wave][slot][lane >> 2] >> (8 * (lane & 3))) & 255u;
        const bool liveL = offC != VR_IDX_DEAD;
        liveMask = __ballot(liveL);
        wbase = 0; totalPieces = 0; issued = 0;
        if (liveMask == 0ull) return;
        const int firstL = __ffsll((long long)liveMask) - 1, lastL = 63 - __clzll((long long)liveMask);
        const uint32_t firstOff = (uint32_t)__builtin_amdgcn_readlane((int)offC, firstL), lastOff = (uint32_t)__builtin_amdgcn_readlane((int)offC, lastL);
        wbase = (firstOff >> 4) & ~3u;                                  // first word, 16-byte aligned
        const uint32_t tp = ((((lastOff + 575u + 15u) >> 4) - wbase) + 255u) >> 8;    // (a 64-leaf subtree is at most 63 + 64 * 8 tokens)
        totalPieces = min(tp, (capWords - wbase + 255u) >> 8);                        // (every piece has a lane inside the buffer)
        for (uint32_t p = 0; p < 2u && p < totalPieces; ++p) { issue_piece(wbase, p); ++issued; }
        const int64_t io = index_of(r);
        RG_DMA("dwordx4", a.fine + (io + lane) * 16, ringLds + 3u * 1024u);
        if (lane < 32) RG_DMA("dwordx4", a.val3 + io * 8 + lane * 16, ringLds + 2u * 1024u);
        ops += 2;
        markSide = ops;
    };
    // ---- prologue: index entries of my first RG_IDXD regions, the first one staged
#pragma unroll
    for (int d = 0; d < RG_IDXD; ++d) markIdx[d] = 0;
#pragma unroll
    for (int d = 0; d < RG_IDXD; ++d) if (rid + d * G < nreg) request_index(rid + d * G, d);
    stage(rid, 0);
    int slot = 0;           // index slot of the current region
#ifdef RG_STAMP             // (diagnostic build: where a wave's cycles go; never timed)
    unsigned long long tAcc[6] = {0, 0, 0, 0, 0, 0}, tPiece = 0;
#define RG_T(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tAcc[i] += now_ - tLast; tLast = now_; } while (0)
    unsigned long long tLast = __builtin_amdgcn_s_memtime();
    const unsigned long long tStart = tLast;
#else
#define RG_T(i) do { } while (0)
#endif
    for (; rid < nreg; rid += G) {
        RG_T(5);
        if (liveMask == 0ull) {
            // an emit block under pruned nodes: one value per 64-leaf block, no stream, no further side-car
            const uint32_t rep = valC * 0x01010101u;
#pragma unroll
            for (int gg = 0; gg < 16; ++gg) buf[laneTerm ^ ((a.parkP[gg >> 2] >> (8 * (gg & 3))) & 255u)] = rep;
            if (rid + RG_IDXD * G < nreg) request_index(rid + RG_IDXD * G, slot);
        } else {
            // ---- park, for each 4-leaf subtree of my 64-leaf block: the scalar of its depth-(D-3) parent (bits 0-7), the
            // token position of its run relative to the ring's first word (8-23: prefix sum of the side-car counts) and
            // whether its root exists (24: it owns more tokens than the ancestors heading its run -- a pruned ancestor
            // ends the run).  A 64-leaf block under a pruned node parks its scalar with "no root", or, where its whole
            // step of four blocks is dead (the step is skipped), its final words.
            const bool liveL = offC != VR_IDX_DEAD;
            rg_vm_wait(ops - markSide);
            const uint4 cv = *(const uint4 *)(ringW + 768 + 4 * lane);
            const uint2 sv = *(const uint2 *)(ringW + 512 + 2 * lane);
            const uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w}, sw2[2] = {sv.x, sv.y};
            const bool stepDead = ((uint32_t)(liveMask >> (lane & 60)) & 15u) == 0u;
            const uint32_t deadW = stepDead ? valC * 0x01010101u : valC;
            uint32_t run = liveL ? offC - wbase * 16u : 0u;
#ifdef RG_KO_PARK           // (timing experiments)
#pragma unroll
            for (int gg = 0; gg < 1; ++gg) {
#else
#pragma unroll
            for (int gg = 0; gg < 16; ++gg) {
#endif
                const uint32_t cgg = (cw[gg >> 2] >> (8 * (gg & 3))) & 255u;
                const int own = gg == 0 ? 4 : (gg & 1 ? 0 : (gg & 2 ? 1 : (gg & 4 ? 2 : 3)));
                const uint32_t vgg = (sw2[gg >> 3] >> (8 * ((gg >> 1) & 3))) & 255u;
                const uint32_t w = vgg | (run << 8) | ((uint32_t)(own - (int)cgg) & 0x01000000u);
                buf[laneTerm ^ ((a.parkP[gg >> 2] >> (8 * (gg & 3))) & 255u)] = liveL ? w : deadW;
                run += cgg;
            }
            const uint32_t mEnd = wave_incl_scan_max_dpp(liveL ? run : 0u);       // end of the last live block up to mine
            RG_T(1);
            // the side-cars are in registers: their ring slots take the string's pieces 2 and 3
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (uint32_t p = 2; p < (uint32_t)RG_NP && p < totalPieces; ++p) { issue_piece(wbase, p); ++issued; }
            if (rid + RG_IDXD * G < nreg) request_index(rid + RG_IDXD * G, slot);
            uint32_t guardNext = RG_NP;          // next piece that lands on the ring's first slot (its first words feed the guard)
            uint32_t landed = 0;                 // pieces known to have landed
#pragma unroll 1
            for (int t = 0; t < 8; ++t) {       // two steps per trip: their chains are independent, the scheduler interleaves them
                const uint32_t nib = ((uint32_t)(liveMask >> (8 * t))) & 255u;
                if (nib == 0u) continue;       // (dead blocks parked their final words)
                const uint32_t endTok = (uint32_t)__builtin_amdgcn_readlane((int)mEnd, 8 * t + 7);
                // the pieces this trip reads must have landed
                const uint32_t need = (((endTok + 15u) >> 4) + 255u) >> 8;
                if (need > landed) {
#ifdef RG_STAMP
                    const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
                    rg_vm_wait(ops - mark_of_piece(need - 1u));
#ifdef RG_STAMP
                    tPiece += __builtin_amdgcn_s_memtime() - w0_;
#endif
                    landed = need;
                    while (guardNext < need) {     // a piece has landed on the ring's first slot: its first four words again behind the last slot
                        if (lane < 4) ringW[RG_NP * 256 + lane] = ringW[lane];
                        guardNext += RG_NP;
                    }
                }
                const uint32_t sw = (uint32_t)((2 * t) & 7) << 2;
                const uint32_t addr0 = (uint32_t)(128 * t) + ((uint32_t)lane ^ sw), addr1 = (uint32_t)(128 * t + 64) + ((uint32_t)lane ^ (sw | 4u));
                if ((nib & 15u) != 0u && (nib >> 4) != 0u) {
                    const uint32_t pw0 = buf[addr0], pw1 = buf[addr1];
                    const uint32_t *r0 = ringW + ((pw0 >> 12) & RG_RING_MASK), *r1 = ringW + ((pw1 >> 12) & RG_RING_MASK);
                    const uint32_t x0 = r0[0], x1 = r0[1], x2 = r0[2], x3 = r0[3];
                    const uint32_t y0 = r1[0], y1 = r1[1], y2 = r1[2], y3 = r1[3];
                    const uint32_t q0 = rg_quad(pw0, x0, x1, x2, x3, p0, p1, sm), q1 = rg_quad(pw1, y0, y1, y2, y3, p0, p1, sm);
                    buf[addr0] = q0;
                    buf[addr1] = q1;
                } else {
                    const uint32_t addr = (nib & 15u) != 0u ? addr0 : addr1;
                    const uint32_t pw0 = buf[addr];
                    const uint32_t *r0 = ringW + ((pw0 >> 12) & RG_RING_MASK);
                    const uint32_t x0 = r0[0], x1 = r0[1], x2 = r0[2], x3 = r0[3];
                    buf[addr] = rg_quad(pw0, x0, x1, x2, x3, p0, p1, sm);
                }
                // refill: the slot of piece p is free once every word of piece p - RG_NP lies before the next trip's first
                // (a trip takes fewer than 512 words: three refills keep up with two to spare)
                const uint32_t consumedWord = endTok >> 4;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (this trip's ring reads are back)
                while (issued < totalPieces && 256u * (issued - (RG_NP - 1)) <= consumedWord) { issue_piece(wbase, issued); ++issued; }
            }
        }
        RG_T(2);
        // ---- where this region's voxels go
        const uint32_t rxC = (uint32_t)rid & ((1u << a.lrx) - 1u), ryC = ((uint32_t)rid >> a.lrx) & ((1u << a.lry) - 1u), rzC = (uint32_t)rid >> (a.lrx + a.lry);
        uint8_t *O = a.out + (int64_t)brick * a.voxels + ((int64_t)rzC * 16 * a.Y + (int64_t)ryC * 16) * a.X + (int64_t)rxC * RG_REGX + (lane & (RG_WAVES - 1)) * 16;
        // ---- the pipeline: the next region staged (the ring is free now; a stale piece of this region still in flight
        // lands before the next region's piece for the same slot: loads return in issue order)
        slot = slot + 1 == RG_IDXD ? 0 : slot + 1;
        if (rid + G < nreg) stage(rid + G, slot);
        RG_T(2);
        rg_barrier();           // every block of the region is decoded
        RG_T(3);
        // ---- gather: a 16-byte row piece of emit block c = lane & 7 per lane, eight whole 128-byte lines per store.
        // All reads of the image first (they are independent), then the byte picks and the stores.
#ifdef RG_KO_GATHER         // (timing experiments)
        if (false)
#endif
        {
            const uint32_t *img = sm.buf + (lane & (RG_WAVES - 1)) * RG_BLK_WORDS;
            const uint32_t xr1 = a.xRead[0] >> 16, xr2 = a.xRead[1] & 0xFFFFu, xr3 = a.xRead[1] >> 16;
            uint32_t addrI[4], bselI[4], ooI[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                uint32_t addr = addrL, bsel = byteL, oo = outL;
#pragma unroll
                for (int i = 6 - RG_LW; i < 8 - RG_LW; ++i) {
                    const bool on = ((it >> (i - (6 - RG_LW))) & 1) != 0;
                    addr ^= on ? (a.gAddr[i >> 1] >> (16 * (i & 1))) & 0xFFFFu : 0u;
                    bsel |= on ? (a.gByte >> (4 * i)) & 15u : 0u;
                    oo += on ? a.gOut[i] : 0u;
                }
                addrI[it] = addr; bselI[it] = bsel; ooI[it] = oo;
            }
            if (a.jx < 2) {          // x bit 0 lives in the byte index: a word holds two x-neighbours
                uint4 P[4], Q[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) { P[it] = *(const uint4 *)(img + addrI[it]); Q[it] = *(const uint4 *)(img + (addrI[it] ^ xr1)); }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const uint32_t b0 = bselI[it], b1 = b0 | (1u << a.jx);
                    const uint32_t sel = b0 | (b1 << 8) | ((4u + b0) << 16) | ((4u + b1) << 24);
                    const uint4 o4 = make_uint4(__builtin_amdgcn_perm(P[it].y, P[it].x, sel), __builtin_amdgcn_perm(P[it].w, P[it].z, sel),
                                                __builtin_amdgcn_perm(Q[it].y, Q[it].x, sel), __builtin_amdgcn_perm(Q[it].w, Q[it].z, sel));
#ifdef RG_KO_STORE          // (timing experiments: no store; the count of operations must stay right)
                    asm volatile("" :: "v"(o4.x), "v"(o4.y), "v"(o4.z), "v"(o4.w));
#else
                    store_out16(O + ooI[it], o4);
                    ++ops;
#endif
                }
            } else {                 // jx == 2: four x-neighbours in four words
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    uint4 P[2], Q[2], R[2], T[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const uint32_t ad = addrI[2 * h + k];
                        P[k] = *(const uint4 *)(img + ad); Q[k] = *(const uint4 *)(img + (ad ^ xr1));
                        R[k] = *(const uint4 *)(img + (ad ^ xr2)); T[k] = *(const uint4 *)(img + (ad ^ xr3));
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const uint32_t bsel = bselI[2 * h + k];
                        const uint32_t sel = bsel | ((4u + bsel) << 8) | 0x0c0c0000u;
                        const auto mk = [sel](const uint4 &u) {
                            return __builtin_amdgcn_perm(__builtin_amdgcn_perm(u.w, u.z, sel), __builtin_amdgcn_perm(u.y, u.x, sel), 0x05040100u);
                        };
                        const uint4 o4 = make_uint4(mk(P[k]), mk(Q[k]), mk(R[k]), mk(T[k]));
#ifdef RG_KO_STORE
                        asm volatile("" :: "v"(o4.x), "v"(o4.y), "v"(o4.z), "v"(o4.w));
#else
                        store_out16(O + ooI[2 * h + k], o4);
                        ++ops;
#endif
                    }
                }
            }
        }
        RG_T(4);
        rg_barrier();           // the image is free for the next region's park
    }
#ifdef RG_STAMP
    RG_T(5);
    if (lane == 0 && a.dbg) {
        atomicAdd(&a.dbg[0], __builtin_amdgcn_s_memtime() - tStart);
        for (int i = 1; i < 6; ++i) atomicAdd(&a.dbg[i], tAcc[i]);
        atomicAdd(&a.dbg[6], 1ull);
        atomicAdd(&a.dbg[7], tPiece);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (no LDS-DMA may outlive the workgroup's LDS)
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
    a.ltx = 0; while ((1 << a.ltx) < a.tilesX) ++a.ltx;
    a.lty = 0; while ((1 << a.lty) < a.tilesY) ++a.lty;
    if ((1 << a.ltx) != a.tilesX || (1 << a.lty) != a.tilesY) return false;
    // ticket order inside a group of 256 tiles (k_decode_quad): first the bits that stay inside a 16 x 16 (y, z) cell --
    // z bits 0-1 and y bit 0 of the tile coordinate -- then the others in address order
    {
        const int first[3] = {a.ltx + a.lty, a.ltx + a.lty + 1, a.ltx};
        int n = 0;
        bool used[8] = {false, false, false, false, false, false, false, false};
        for (int i = 0; i < 3; ++i) {
            const bool exists = i < 2 ? (1 << (i + 1)) <= a.tilesZ : a.tilesY >= 2;
            if (exists && first[i] < 8 && !used[first[i]]) { a.kqBit[n++] = (uint8_t)first[i]; used[first[i]] = true; }
        }
        for (int b = 0; b < 8; ++b) if (!used[b]) a.kqBit[n++] = (uint8_t)b;
    }
    return true;
}

// k_decode_region's geometry: the twelve deepest levels must be four (a, b, c) triples in one axis order, deciding
// coordinate bits 3, 2, 1, 0 (a 4096-leaf emit block is then a 16 x 16 x 16 box), with whole 128 x 16 x 16 regions
static bool region_geometry(const BrickSet *bs, RegionArgs &a)
{
    const Geom &g = bs->g;
    const int D = g.D;
    if (bs->K != 6 || D < 12 || bs->generalGeom || bs->idx64 || (bs->treeCap & 15)) return false;
    if (g.X < RG_REGX || g.X < 128 || g.Y < 16 || g.Z < 16) return false;
    if ((g.X & (g.X - 1)) || (g.Y & (g.Y - 1)) || (g.Z & (g.Z - 1))) return false;
    int pos[3] = {-1, -1, -1};
    for (int q = 0; q < 3; ++q) pos[g.axis[D - 3 + q]] = 2 - q;           // deepest level -> rank bit 0
    if (pos[0] < 0 || pos[1] < 0 || pos[2] < 0) return false;
    for (int k = 0; k < 4; ++k)
        for (int q = 0; q < 3; ++q) {
            const int d = D - 3 * (k + 1) + q;
            if (g.axis[d] != g.axis[D - 3 + q] || g.bit[d] != k) return false;
        }
    const int jx = pos[0], jy = pos[1], jz = pos[2];
    // quad index q = leaf rank >> 2 (10 bits).  Bits 0-5 go to lane / image-word bits: the two lowest x bits first
    int Pmap[6], nx = 0, nxt = 2;
    bool isx[6] = {false, false, false, false, false, false};
    for (int k = 0; k < 4; ++k) {
        const int qb = 3 * k + jx - 2;
        if (qb >= 0 && qb < 6) { isx[qb] = true; Pmap[qb] = nx++; }
    }
    if (nx != 2) return false;
    for (int qb = 0; qb < 6; ++qb) if (!isx[qb]) Pmap[qb] = nxt++;
    a.lanePos = 0;
    for (int qb = 0; qb < 6; ++qb) a.lanePos |= (uint32_t)qb << (4 * Pmap[qb]);
    for (int i = 0; i < 4; ++i) a.parkP[i] = 0;
    for (int gg = 0; gg < 16; ++gg) {
        uint32_t w = 0;
        for (int i = 0; i < 4; ++i) if ((gg >> i) & 1) w |= 1u << Pmap[i];
        a.parkP[gg >> 2] |= w << (8 * (gg & 3));
    }
    a.parkS = 0;
    for (int sv = 0; sv < 4; ++sv) {
        uint32_t w = 0;
        if (sv & 1) w |= 1u << Pmap[4];
        if (sv & 2) w |= 1u << Pmap[5];
        a.parkS |= w << (8 * sv);
    }
    // image-word contribution of quad-index bit qb (XOR-linear: a permutation plus the step swizzle)
    const auto contrib = [&](int qb) -> uint32_t {
        if (qb < 6) return 1u << Pmap[qb];
        const int i = qb - 6;
        return (1u << (6 + i)) | (i < 3 ? 1u << (2 + i) : 0u);
    };
    // the eight (y, z) bits of a region's 16 x 16 plane
    struct GB { uint32_t addr, byte, out; } bits[8], ord[8];
    for (int ax = 1; ax <= 2; ++ax)
        for (int k = 0; k < 4; ++k) {
            const int rb = 3 * k + (ax == 1 ? jy : jz);
            GB &b = bits[(ax - 1) * 4 + k];
            b.addr = rb < 2 ? 0u : contrib(rb - 2);
            b.byte = rb < 2 ? 1u << rb : 0u;
            b.out = ax == 1 ? (uint32_t)((1 << k) * g.X) : (uint32_t)((int64_t)(1 << k) * g.X * g.Y);
        }
    // gather bits 0 .. 5-RG_LW come from lane >> RG_LW.  Eight emit blocks per region: bit 1 the plane bit at image bit 5
    // (the 16-byte bank slot's top bit), bits 0 and 2 plane bits that do not move the slot at all (byte index, image bit
    // 9): the eight rows of a store instruction then read conflict-free (the lanes of one row are the eight emit blocks,
    // 4 words = one slot apart).  Four blocks per region: the same three first, any fourth (a two-way conflict at worst).
    bool used[8] = {false, false, false, false, false, false, false, false};
    int n = 0;
    const auto take = [&](int i) { ord[n++] = bits[i]; used[i] = true; };
    int free0 = -1, free1 = -1, top = -1;
    for (int i = 0; i < 8; ++i) {
        if (bits[i].addr == 32u && top < 0) top = i;
        else if ((bits[i].addr & 0x3Cu) == 0u) { if (free0 < 0) free0 = i; else if (free1 < 0) free1 = i; }
    }
    if (free0 >= 0) take(free0); else { for (int i = 0; i < 8; ++i) if (!used[i] && i != top && i != free1) { take(i); break; } }
    if (top >= 0) take(top); else { for (int i = 0; i < 8; ++i) if (!used[i] && i != free1) { take(i); break; } }
    if (free1 >= 0) take(free1); else { for (int i = 0; i < 8; ++i) if (!used[i]) { take(i); break; } }
    for (int i = 0; i < 8; ++i) if (!used[i]) take(i);
    for (int i = 0; i < 4; ++i) a.gAddr[i] = 0;
    a.gByte = 0;
    for (int i = 0; i < 8; ++i) {
        a.gAddr[i >> 1] |= ord[i].addr << (16 * (i & 1));
        a.gByte |= ord[i].byte << (4 * i);
        a.gOut[i] = ord[i].out;
    }
    // x bits above the two lowest: the gather's reads
    uint32_t xr[4] = {0, 0, 0, 0};
    if (jx < 2) xr[1] = contrib(3 * 3 + jx - 2);
    else { xr[1] = contrib(6); xr[2] = contrib(9); xr[3] = xr[1] ^ xr[2]; }
    a.xRead[0] = xr[0] | (xr[1] << 16);
    a.xRead[1] = xr[2] | (xr[3] << 16);
    a.jx = jx;
    a.X = g.X; a.Y = g.Y; a.voxels = g.voxels;
    a.lrx = 0; while ((RG_REGX << a.lrx) < g.X) ++a.lrx;
    a.lry = 0; while ((16 << a.lry) < g.Y) ++a.lry;
    a.nreg = (g.X / RG_REGX) * (g.Y / 16) * (g.Z / 16);
    // the emit block of a 16^3 box: the rank bits above the twelve lowest
    uint32_t bpos[3] = {0, 0, 0};
    for (int d = 0; d < D - 12; ++d) {
        const int ax = g.axis[d], kb = g.bit[d] - 4;       // coordinate bit kb + 4 of axis ax sits at rank bit D - 1 - d
        if (kb < 0 || kb >= 6) return false;
        bpos[ax] |= (uint32_t)(D - 1 - d - 12) << (5 * kb);
    }
    a.blkX = bpos[0]; a.blkY = bpos[1]; a.blkZ = bpos[2];
    return true;
}

// scalar of every depth-Ds subtree's ancestor at depth `cut` (< Ds), from the encoder's BFS codes
__global__ void __launch_bounds__(256)
k_cut_values(const uint8_t *__restrict__ codes, int64_t codeStride, const Ctrl *ctrls, int Ds, int cut, int64_t nIdx,
             uint8_t *__restrict__ out)
{
    const int brick = blockIdx.y;
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= nIdx) return;
    const uint8_t *Cb = codes + (int64_t)brick * codeStride;
    const uint8_t *dmap = ctrls[brick].distanceMap;
    int val = dmap[0];
    // a pruned node ends the path: what the level loop left in the array below it is not part of the tree.  (In the mid
    // stream those codes are all "keep" or 3 anyway; in a MidRangeTree's half-range stream, pruned where the mid stream
    // is, M.cpp:864-865, they are whatever its own level loop chose.)
    for (int j = 1; j <= cut; ++j) {
        const int code = cget(Cb, ((int64_t)1 << j) + (s >> (Ds - j)));
        if (code == 3) break;
        val = apply_code(val, code, dmap[j]);
    }
    out[(int64_t)brick * nIdx + s] = (uint8_t)val;
}

int decode_launch(BrickSet *bs, uint8_t *out, int cut, hipStream_t st, bool rangeStream)
{
    hipEventRecord(bs->ev[5], st);
    // MidRangeTree's second stream is emitted in lock step with the first (M.cpp:871-982): same token positions,
    // so the same side-car offsets serve it; only the scalars (its own codes, its own distanceMap) differ
    const Stream2 &sm = rangeStream ? bs->rng : bs->mid;
    const uint8_t *cutVals = nullptr, *idxVals = bs->idxVal;
    if (cut < bs->Ds || rangeStream) {
        if (!bs->idxValCut) return -2;
        if (!bs->foreign)
            hipLaunchKernelGGL(k_cut_values, dim3((unsigned)((bs->nIdx + 255) / 256), bs->B), dim3(256), 0, st,
                               sm.codes, bs->codeStride, sm.ctrl, bs->Ds, cut < bs->Ds ? cut : bs->Ds, bs->nIdx, bs->idxValCut);
        cutVals = bs->idxValCut;   // foreign streams: filled by the host from the bytes (capi)
        if (rangeStream) idxVals = bs->idxValCut;
    }
    TileArgs t;
    if (bs->generalGeom) {
        // general extents: rank-domain decode, then every voxel takes its owner leaf's value
        if (!bs->rankVals && hipMalloc(&bs->rankVals, (size_t)bs->B * bs->leafStride * 2) != hipSuccess) return -3;
        DecodeArgs a;
        a.tree = sm.tree; a.treeCap = bs->treeCap;
        a.idxBase = bs->idx64 ? bs->idxBase : nullptr; a.nBase = bs->nEmitBlk;
        a.idxOff = bs->idxOff; a.idxVal = idxVals; a.nIdx = bs->nIdx;
        a.ctrls = sm.ctrl; a.lut = nullptr; a.out = bs->rankVals; a.g = bs->g;
        a.D = bs->D; a.K = bs->K; a.Ds = bs->Ds;
        a.cut = cut; a.idxValCut = cutVals;
        hipLaunchKernelGGL(k_decode_lane<true>, dim3((unsigned)((bs->nIdx + 63) / 64), bs->B), dim3(64), 0, st, a);
        hipLaunchKernelGGL(k_owner_gather, dim3((unsigned)((bs->g.voxels + 255) / 256), bs->B), dim3(256), 0, st,
                           (const uint16_t *)bs->rankVals, bs->leafStride, bs->ownerRank, bs->ownerSurv, bs->g.voxels, out);
    } else if (!bs->sw.decodeV1 && tile_geometry(bs, t)) {
        t.tree = sm.tree; t.treeCap = bs->treeCap;
        t.idxOff = bs->idxOff; t.idxVal = idxVals; t.nIdx = bs->nIdx;
        t.ctrls = sm.ctrl; t.out = out; t.g = bs->g; t.D = bs->D; t.Ds = bs->Ds;
        t.cut = cut; t.idxValCut = cutVals; t.spread = bs->spread;
        const int ntiles = t.tilesX * t.tilesY * t.tilesZ;
        t.fine = bs->fineIdx;
        t.val3 = bs->idxVal3;
        bool useFine = bs->fineIdx && (int)bs->fineHas.size() == bs->B && !rangeStream && !bs->sw.decodeWalk;
        for (int i = 0; useFine && i < bs->B; ++i) useFine = bs->fineHas[(size_t)i] != 0;
        // k_decode_region / k_decode_quad: cuts at or below depth D-3 (the third side-car holds the depth-(D-3) scalars
        // at full precision); shallower progressive cuts keep k_decode_fine, which decodes the upper nodes itself
        const bool useQuad = useFine && bs->idxVal3 && cut >= bs->D - 3 && !bs->sw.decodeFineV1;
        RegionArgs r;
        if (useQuad && !bs->sw.decodeQuad && region_geometry(bs, r)) {
            r.tree = sm.tree; r.treeCap = bs->treeCap;
            r.idxOff = bs->idxOff; r.idxVal = idxVals; r.fine = bs->fineIdx; r.val3 = bs->idxVal3; r.nIdx = bs->nIdx;
            r.ctrls = sm.ctrl; r.out = out; r.spread = bs->spread; r.D = bs->D; r.cut = cut;
            // a workgroup decodes every RG_PER-th region of its brick: enough regions to amortise its tables and the
            // pipeline's fill, enough workgroups (a few thousand for the bench volume) to balance the chip
            r.dbg = nullptr;
#ifdef RG_STAMP
            {
                static unsigned long long *dbgDev = nullptr;
                if (!dbgDev) { hipMalloc(&dbgDev, 64); hipMemset(dbgDev, 0, 64); }
                unsigned long long h[8];
                hipMemcpy(h, dbgDev, 64, hipMemcpyDeviceToHost);
                if (h[6]) fprintf(stderr, "[rg stamp] waves %llu  cycles/wave: total %.0f park %.0f steps+stage %.0f (of it waiting for string pieces %.0f) barrier1 %.0f gather %.0f barrier2+top %.0f\n", h[6],
                                  (double)h[0] / h[6], (double)h[1] / h[6], (double)h[2] / h[6], (double)h[7] / h[6], (double)h[3] / h[6], (double)h[4] / h[6], (double)h[5] / h[6]);
                hipMemset(dbgDev, 0, 64);
                r.dbg = dbgDev;
            }
#endif
            unsigned wgs = (unsigned)((r.nreg + RG_PER - 1) / RG_PER);
            if ((int64_t)wgs * bs->B < 2048) wgs = (unsigned)std::min<int64_t>(r.nreg, (2048 + bs->B - 1) / bs->B);
            hipLaunchKernelGGL(k_decode_region, dim3(wgs, bs->B), dim3(64 * RG_WAVES), 0, st, r);
        } else if (useQuad) {
            const int levels = cut - bs->D < 0 ? 0 : (cut - bs->D > VR_CHAIN_LEVELS ? VR_CHAIN_LEVELS : cut - bs->D);
            // one table per number of refining levels, all written once: two decodes of one set on different streams at
            // different cuts never rewrite a table the other is reading
            if (!bs->chainTab && hipMalloc(&bs->chainTab, (size_t)(VR_CHAIN_LEVELS + 1) * QD_CHAIN_ENTRIES * 4) != hipSuccess) return -3;
            if (!bs->chainTabReady) {
                for (int lv = 0; lv <= VR_CHAIN_LEVELS; ++lv)
                    hipLaunchKernelGGL(k_chain_table, dim3(QD_CHAIN_ENTRIES / 256), dim3(256), 0, st, lv, bs->chainTab + (size_t)lv * QD_CHAIN_ENTRIES);
                // (the first use may come from any stream: make the tables visible to all of them before going on)
                if (hipStreamSynchronize(st) != hipSuccess) return -1;
                bs->chainTabReady = true;
            }
            t.tables = bs->chainTab + (size_t)levels * QD_CHAIN_ENTRIES;
            const int per = QD_WAVES * QD_TPW;
            hipLaunchKernelGGL(k_decode_quad, dim3((unsigned)((ntiles + per - 1) / per), bs->B), dim3(64 * QD_WAVES), 0, st, t);
        } else {
        if (useFine && !bs->decTables && hipMalloc(&bs->decTables, (size_t)bs->B * FD_TABLE_WORDS * 4) != hipSuccess) return -3;
        t.tables = bs->decTables;
        if (useFine)
            hipLaunchKernelGGL(k_fine_tables, dim3(bs->B), dim3(256), 0, st, bs->mid.ctrl, bs->D, bs->Ds, cut, bs->decTables);
        if (useFine)
            hipLaunchKernelGGL(k_decode_fine, dim3((unsigned)((ntiles + FD_WAVES - 1) / FD_WAVES), bs->B),
                               dim3(64 * FD_WAVES), 0, st, t);
        else
            hipLaunchKernelGGL(k_decode_tile, dim3((unsigned)((ntiles + DEC_WAVES - 1) / DEC_WAVES), bs->B),
                               dim3(64 * DEC_WAVES), 0, st, t);
        }
    } else {
        DecodeArgs a;
        a.tree = sm.tree; a.treeCap = bs->treeCap;
        a.idxBase = nullptr; a.nBase = 0;
        a.idxOff = bs->idxOff; a.idxVal = idxVals; a.nIdx = bs->nIdx;
        a.ctrls = sm.ctrl; a.lut = bs->lut; a.out = out; a.g = bs->g;
        a.D = bs->D; a.K = bs->K; a.Ds = bs->Ds;
        a.cut = cut; a.idxValCut = cutVals;
        hipLaunchKernelGGL(k_decode_lane<false>, dim3((unsigned)((bs->nIdx + 63) / 64), bs->B), dim3(64), 0, st, a);
    }
    hipEventRecord(bs->ev[6], st);
    return launch_status("decode");
}

// Foreign stream, progressive cut above the index level: scalar of every depth-Ds subtree's ancestor at
// depth `cut` (< Ds), from the bytes alone.
int cut_values_from_stream(BrickSet *bs, const uint8_t *tree, int64_t numActive, const uint8_t *dmap, int cut,
                           std::vector<uint8_t> &vals)
{
    const int D = bs->D, Ds = bs->Ds;
    vals.assign((size_t)bs->nIdx, 0);
    auto get = [&](int64_t p) { return (tree[p >> 2] >> ((p & 3) * 2)) & 3; };
    int v[VR_MAX_DEPTH];
    int64_t pos = 0;
    int j = 0;
    uint32_t path = 0;
    while (true) {
        if (pos >= numActive) return -1;
        int tok = get(pos++);
        int val = j == 0 ? dmap[0] : (j <= cut ? apply_code(v[j - 1], tok, dmap[j]) : v[j - 1]);
        v[j] = val;
        if (j == Ds) vals[path] = (uint8_t)val;
        bool terminal = false;
        if (tok == 3) {
            if (j < Ds) {
                uint32_t lo = path << (Ds - j), hi = (path + 1) << (Ds - j);
                for (uint32_t q = lo; q < hi; ++q) vals[q] = (uint8_t)val;
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

// Serial pass over a foreign stream (host): the side-car index from the bytes alone.
// Also validates the grammar.  Returns 0 or a negative code.
int build_index_from_stream(BrickSet *bs, int brick, const uint8_t *tree, int64_t numActive, const uint8_t *dmap,
                            std::vector<uint32_t> &offs, std::vector<uint8_t> &vals, std::vector<uint8_t> &fine,
                            std::vector<uint8_t> &val3)
{
    const int D = bs->D, Ds = bs->Ds;
    offs.assign((size_t)bs->nIdx, VR_IDX_DEAD);
    vals.assign((size_t)bs->nIdx, 0);
    // K == 6: tokens owned by each 4-leaf subtree of a depth-Ds node, what k_prune_emit12 leaves for k_decode_fine.
    // A token at depth >= Ds belongs to the 4-leaf subtree that holds its node's first leaf.
    const bool wantFine = bs->K == 6 && D >= 6;
    fine.assign(wantFine ? (size_t)bs->nIdx * 16 : 0, 0);
    val3.assign(wantFine ? (size_t)bs->nIdx * 8 : 0, 0);      // decoded scalar of every depth-(D-3) node (k_decode_quad)
    auto own = [&](uint32_t path, int j) {
        if (!wantFine || j < Ds) return;
        const uint32_t first = path << (D - j);     // first leaf (rank) below the node
        fine[(size_t)(first >> 6) * 16 + ((first >> 2) & 15u)] += 1;
    };
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
        if (wantFine && j == D - 3) val3[path] = (uint8_t)val;
        own(path, j);
        bool terminal = false;
        if (tok == 3) {
            if (j < Ds) {
                uint32_t lo = path << (Ds - j), hi = (path + 1) << (Ds - j);
                for (uint32_t q = lo; q < hi; ++q) { offs[q] = VR_IDX_DEAD; vals[q] = (uint8_t)val; }
            }
            if (wantFine && j < D - 3) {
                uint32_t lo = path << (D - 3 - j), hi = (path + 1) << (D - 3 - j);
                for (uint32_t q = lo; q < hi; ++q) val3[q] = (uint8_t)val;
            }
            terminal = true;
        } else if (j == D) {
            for (int c = 1; c <= VR_CHAIN_LEVELS; ++c) {
                if (pos >= numActive) return -2;
                own(path, j);
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
