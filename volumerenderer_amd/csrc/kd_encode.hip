// kd_encode.hip -- gfx950 kernels for VolumeKdtree::build (reference
// volume_renderer/VolumeKdTree_recover.cpp:17-140, "R.cpp").  Batched over the
// bricks of a brickset: blockIdx.y (or blockIdx.x for per-brick control kernels)
// selects the brick.  No MFMA: this is byte/integer work bound by HBM and, for the
// running-mean estimator, by a true sequential dependency.
//
// Stages
//   pyramid   R.cpp:143-201  bottom-up min/max -> midrange heap `temp`
//   compress  R.cpp:206-384  per level: serial running-mean start distance,
//                            <= maxEpochs clamped gradient-descent epochs; the GD
//                            control flow runs on the device (k_control) so the
//                            whole build is one launch sequence without host syncs
//   prune     R.cpp:596-629  bottom-up, level synchronous (race free, == serial)
//   convert   R.cpp:631-724  preorder emission by a prefix sum over leaf ranks:
//                            leaf rank r owns the tokens of the live internal
//                            nodes whose first leaf is r, then its own leaf token
//                            and grown chain
#include "brickset.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

namespace vr {

#define FILL_NODES_PER_BLOCK 1024
#define EMIT_RANKS_PER_BLOCK 256

// ---------------------------------------------------------------- pyramid ----
// One block reduces 2^L inputs (L <= 10) at depth dLeaf and writes the midranges of
// depths dLeaf-1 .. dLeaf-L; the block root's (min,max) goes to outMin/outMax so the
// next round can continue upwards.  FROM_VOXELS gathers voxels in Morton order and
// also writes the leaf level.
template <bool FROM_VOXELS>
__global__ void __launch_bounds__(256)
k_pyramid(Geom g, int dLeaf, int L, const uint8_t *__restrict__ vox, const uint8_t *__restrict__ inMin,
          const uint8_t *__restrict__ inMax, int64_t inStride, uint8_t *__restrict__ temp, int64_t heapStride,
          uint8_t *__restrict__ tempRange, uint8_t *__restrict__ outMin, uint8_t *__restrict__ outMax,
          int64_t outStride, const uint32_t *__restrict__ srcIdx)
{
    __shared__ uint8_t smn[2][1024], smx[2][1024];
    const int brick = blockIdx.y;
    const uint32_t n = 1u << L;
    const uint32_t base = blockIdx.x << L;
    uint8_t *T = temp + (int64_t)brick * heapStride;
    uint8_t *TR = tempRange ? tempRange + (int64_t)brick * heapStride : nullptr;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        uint32_t r = base + i;
        uint8_t mn, mx;
        if (FROM_VOXELS) {
            int64_t vi;
            if (srcIdx) vi = srcIdx[r];            // general extents: the min corner of the leaf's box (BrickSet::srcIdx)
            else {
                int x, y, z;
                rank_to_xyz(g, r, x, y, z);
                vi = x + (int64_t)g.X * (y + (int64_t)g.Y * z);
            }
            uint8_t v = vox[(int64_t)brick * g.voxels + vi];
            mn = mx = v;
            T[((int64_t)1 << dLeaf) + r] = v;       // leaf: (v+v)/2 = v  (R.cpp:194-198)
            if (TR) TR[((int64_t)1 << dLeaf) + r] = 0; // half range of a single voxel (M.cpp:235)
        } else {
            mn = inMin[(int64_t)brick * inStride + r];
            mx = inMax[(int64_t)brick * inStride + r];
        }
        smn[0][i] = mn;
        smx[0][i] = mx;
    }
    __syncthreads();
    for (int l = 1; l <= L; ++l) {
        const uint32_t m = n >> l;
        const int src = (l - 1) & 1, dst = l & 1;
        const int da = dLeaf - l;
        for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
            uint8_t a = smn[src][2 * i], b = smn[src][2 * i + 1];
            uint8_t c = smx[src][2 * i], d = smx[src][2 * i + 1];
            uint8_t mn = a < b ? a : b, mx = c > d ? c : d;
            smn[dst][i] = mn;
            smx[dst][i] = mx;
            int64_t idx = ((int64_t)1 << da) + (base >> l) + i;
            T[idx] = (uint8_t)(((int)mx + (int)mn) >> 1);      // (byte)((max+min)/2.0) R.cpp:198
            if (TR) TR[idx] = (uint8_t)(((int)mx - (int)mn) >> 1); // M.cpp:235
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        outMin[(int64_t)brick * outStride + blockIdx.x] = smn[L & 1][0];
        outMax[(int64_t)brick * outStride + blockIdx.x] = smx[L & 1][0];
    }
}

#ifndef ENC_NT
#define ENC_NT 1
#endif
// level arrays are written by one whole-volume pass and read by the next: nothing of them survives in a cache until
// then, so the large stores stream past it
__device__ __forceinline__ void st16(void *p, uint4 v)
{
#if ENC_NT
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, (u32x4 *)p);
#else
    *(uint4 *)p = v;
#endif
}
__device__ __forceinline__ void st8(void *p, uint2 v)
{
#if ENC_NT
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 t; t.x = v.x; t.y = v.y;
    __builtin_nontemporal_store(t, (u32x2 *)p);
#else
    *(uint2 *)p = v;
#endif
}
__device__ __forceinline__ void st4(void *p, uint32_t v)
{
#if ENC_NT
    __builtin_nontemporal_store(v, (uint32_t *)p);
#else
    *(uint32_t *)p = v;
#endif
}

#ifndef ENC_NTL
#define ENC_NTL 1
#endif
__device__ __forceinline__ uint4 ld16(const void *p)
{
#if ENC_NTL
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 t = __builtin_nontemporal_load((const u32x4 *)p);
    return make_uint4(t.x, t.y, t.z, t.w);
#else
    return *(const uint4 *)p;
#endif
}
__device__ __forceinline__ uint2 ld8(const void *p)
{
#if ENC_NTL
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 t = __builtin_nontemporal_load((const u32x2 *)p);
    return make_uint2(t.x, t.y);
#else
    return *(const uint2 *)p;
#endif
}

// Bottom 12 levels in one block: the 4096 leaves under one depth-(D-12) node form a
// 2^ax x 2^ay x 2^az box (ax+ay+az = 12).  With ax >= 4 every thread fetches one aligned
// 16-byte x-run of voxels (coalesced), scatters it to Morton order in LDS, and the block
// reduces min/max twelve times there; leaves and the dense levels leave as 16/8/4-byte
// stores.  Replaces the per-voxel gather of k_pyramid<true>.
struct Pyr12Geom {
    int ax, ay, az; uint16_t sx[16];   // sx[i]: Morton rank bits of x = i (low 4 x bits)
    // XCD-aware block order (swz != 0): the eight 16-voxel-wide boxes that share every 128-byte line of an
    // x-run go to the same XCD (workgroup id mod 8) back to back, so that XCD's L2 fetches each line once
    int swz, nbx, nby, lnbx, lnby;     // lnbx, lnby = log2
    const uint32_t *spread;            // BrickSet::spread
};

// (min,max) of sibling pairs held in packed 16-bit lanes: A = (a0,a1), B = (b0,b1) -> (f(a0,a1), f(b0,b1))
__device__ __forceinline__ vr_s16x2 pk_pair_min(uint32_t A, uint32_t B)
{
    return __builtin_elementwise_min(pk_s(__builtin_amdgcn_perm(B, A, 0x05040100u)), pk_s(__builtin_amdgcn_perm(B, A, 0x07060302u)));
}
__device__ __forceinline__ vr_s16x2 pk_pair_max(uint32_t A, uint32_t B)
{
    return __builtin_elementwise_max(pk_s(__builtin_amdgcn_perm(B, A, 0x05040100u)), pk_s(__builtin_amdgcn_perm(B, A, 0x07060302u)));
}
__device__ __forceinline__ uint32_t pk_mid(vr_s16x2 mn, vr_s16x2 mx) { return pk_u((mn + mx) >> 1); }   // R.cpp:198
__device__ __forceinline__ uint32_t pk_half_range(vr_s16x2 mn, vr_s16x2 mx) { return pk_u((mx - mn) >> 1); } // M.cpp:235

__global__ void __launch_bounds__(256)
k_pyramid12(Geom g, Pyr12Geom pg, const uint8_t *__restrict__ vox, uint8_t *__restrict__ temp, int64_t heapStride,
            uint8_t *__restrict__ tempRange, uint8_t *__restrict__ outMin, uint8_t *__restrict__ outMax,
            int64_t outStride, uint8_t *__restrict__ blockFlag, uint8_t *__restrict__ blockFlagR)
{
    __shared__ __attribute__((aligned(16))) uint8_t leaf[4096];
    __shared__ uint32_t waveMM[4];
    const int brick = blockIdx.y, D = g.D;
    // which 2^ax x 2^ay x 2^az box: enumerated by box coordinates (x fastest), not by Morton index
    // (nbx, nby are powers of two -- the extents are -- so the box coordinates come from shifts and masks: the three
    // 32-bit divisions they replaced were ~200 scalar instructions in front of every wave's first load)
    uint32_t bid = blockIdx.x;
    if (pg.swz) {
        const uint32_t slot = bid >> 3;
        const uint32_t grp = (slot >> 3) * 8u + (bid & 7u), m = slot & 7u;       // group -> XCD grp % 8, members in a row
        const int lgx = pg.lnbx - 3;                                              // log2(nbx / 8)
        bid = ((grp >> lgx) << pg.lnbx) + ((grp & ((1u << lgx) - 1u)) << 3) + m;
    }
    const int bx = (int)((bid & ((1u << pg.lnbx) - 1u)) << pg.ax), by = (int)(((bid >> pg.lnbx) & ((1u << pg.lnby) - 1u)) << pg.ay),
              bz = (int)((bid >> (pg.lnbx + pg.lnby)) << pg.az);
    const uint32_t *sp = pg.spread;
    const uint32_t base = sp[bx] | sp[g.X + by] | sp[g.X + g.Y + bz];           // Morton rank of the box origin
    uint8_t *T = temp + (int64_t)brick * heapStride;
    uint8_t *TR = tempRange ? tempRange + (int64_t)brick * heapStride : nullptr;
    const int t = threadIdx.x, lane = t & 63;
    const int nxs = 1 << (pg.ax - 4);
    const int xs = t & (nxs - 1), y = (t >> (pg.ax - 4)) & ((1 << pg.ay) - 1), z = t >> (pg.ax - 4 + pg.ay);
    // Morton rank of (xs*16, y, z) inside the box (the 12 deepest split levels)
    const uint32_t r0 = sp[xs * 16] | sp[g.X + y] | sp[g.X + g.Y + z];
    const uint4 v4 = *(const uint4 *)(vox + (int64_t)brick * g.voxels + (bx + xs * 16) +
                                      (int64_t)g.X * ((by + y) + (int64_t)g.Y * (bz + z)));
    const uint32_t vw[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int i = 0; i < 16; ++i) leaf[r0 | pg.sx[i]] = (uint8_t)(vw[i >> 2] >> ((i & 3) * 8));
    __syncthreads();
    // 16 Morton-consecutive leaves per thread: they and the four levels above them never leave registers
    const uint4 lv = *(const uint4 *)(&leaf[t * 16]);
    st16(T + ((int64_t)1 << D) + base + t * 16, lv);                                // leaf: (v+v)/2 = v
    if (TR) *(uint4 *)(TR + ((int64_t)1 << D) + base + t * 16) = make_uint4(0, 0, 0, 0);
    const uint32_t w[4] = {lv.x, lv.y, lv.z, lv.w};
    vr_s16x2 mn1[4], mx1[4];
    uint32_t md[4], hr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                       // level D-1: lanes = nodes 2q, 2q+1
        const vr_s16x2 lo = pk_s(w[q] & 0x00FF00FFu), hi = pk_s((w[q] >> 8) & 0x00FF00FFu);
        mn1[q] = __builtin_elementwise_min(lo, hi);
        mx1[q] = __builtin_elementwise_max(lo, hi);
        md[q] = pk_mid(mn1[q], mx1[q]);
        hr[q] = pk_half_range(mn1[q], mx1[q]);
    }
    {
        const int64_t o = ((int64_t)1 << (D - 1)) + (base >> 1) + t * 8;
        st8(T + o, make_uint2(__builtin_amdgcn_perm(md[1], md[0], 0x06040200u), __builtin_amdgcn_perm(md[3], md[2], 0x06040200u)));
        if (TR) *(uint2 *)(TR + o) = make_uint2(__builtin_amdgcn_perm(hr[1], hr[0], 0x06040200u), __builtin_amdgcn_perm(hr[3], hr[2], 0x06040200u));
    }
    vr_s16x2 mn2[2], mx2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {                       // level D-2
        mn2[q] = pk_pair_min(pk_u(mn1[2 * q]), pk_u(mn1[2 * q + 1]));
        mx2[q] = pk_pair_max(pk_u(mx1[2 * q]), pk_u(mx1[2 * q + 1]));
    }
    {
        const int64_t o = ((int64_t)1 << (D - 2)) + (base >> 2) + t * 4;
        st4(T + o, __builtin_amdgcn_perm(pk_mid(mn2[1], mx2[1]), pk_mid(mn2[0], mx2[0]), 0x06040200u));
        if (TR) *(uint32_t *)(TR + o) = __builtin_amdgcn_perm(pk_half_range(mn2[1], mx2[1]), pk_half_range(mn2[0], mx2[0]), 0x06040200u);
    }
    const vr_s16x2 mn3 = pk_pair_min(pk_u(mn2[0]), pk_u(mn2[1])), mx3 = pk_pair_max(pk_u(mx2[0]), pk_u(mx2[1]));   // level D-3
    {
        const int64_t o = ((int64_t)1 << (D - 3)) + (base >> 3) + t * 2;
        const uint32_t m3 = pk_mid(mn3, mx3), h3 = pk_half_range(mn3, mx3);
        *(uint16_t *)(T + o) = (uint16_t)((m3 & 0xFFu) | ((m3 >> 8) & 0xFF00u));
        if (TR) *(uint16_t *)(TR + o) = (uint16_t)((h3 & 0xFFu) | ((h3 >> 8) & 0xFF00u));
    }
    int mn = min((int)mn3.x, (int)mn3.y), mx = max((int)mx3.x, (int)mx3.y);        // level D-4: one node per thread
    {
        const int64_t o = ((int64_t)1 << (D - 4)) + (base >> 4) + t;
        T[o] = (uint8_t)((mn + mx) >> 1);
        if (TR) TR[o] = (uint8_t)((mx - mn) >> 1);
    }
    // levels D-5 .. D-10 inside the wave: (min, 255 - max) packed so that one packed min reduces both
    uint32_t p = (uint32_t)mn | ((uint32_t)(255 - mx) << 16);
#pragma unroll
    for (int k = 1; k <= 6; ++k) {
        uint32_t q;
        if (k == 1) q = dpp_u32<0x101, 0xf>(p, p);          // row_shl:1 (lane i <- lane i+1)
        else if (k == 2) q = dpp_u32<0x102, 0xf>(p, p);
        else if (k == 3) q = dpp_u32<0x104, 0xf>(p, p);
        else if (k == 4) q = dpp_u32<0x108, 0xf>(p, p);
        else q = (uint32_t)__shfl_down((int)p, 1 << (k - 1));
        p = pk_u(__builtin_elementwise_min(pk_s(p), pk_s(q)));
        if ((lane & ((1 << k) - 1)) == 0) {
            const int a = (int)(p & 0xFFFFu), b = 255 - (int)(p >> 16);
            const int64_t o = ((int64_t)1 << (D - 4 - k)) + (base >> (4 + k)) + (t >> k);
            T[o] = (uint8_t)((a + b) >> 1);
            if (TR) TR[o] = (uint8_t)((b - a) >> 1);
        }
    }
    if (lane == 0) waveMM[t >> 6] = p;
    __syncthreads();
    if (t == 0) {                                       // levels D-11 (two nodes) and D-12 (the block's root)
        int a[2], b[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t u = pk_u(__builtin_elementwise_min(pk_s(waveMM[2 * h]), pk_s(waveMM[2 * h + 1])));
            a[h] = (int)(u & 0xFFFFu); b[h] = 255 - (int)(u >> 16);
            const int64_t o = ((int64_t)1 << (D - 11)) + (base >> 11) + h;
            T[o] = (uint8_t)((a[h] + b[h]) >> 1);
            if (TR) TR[o] = (uint8_t)((b[h] - a[h]) >> 1);
        }
        const int ra = min(a[0], a[1]), rbm = max(b[0], b[1]);
        const int64_t o = ((int64_t)1 << (D - 12)) + (base >> 12);
        T[o] = (uint8_t)((ra + rbm) >> 1);
        if (TR) TR[o] = (uint8_t)((rbm - ra) >> 1);
        outMin[(int64_t)brick * outStride + (base >> 12)] = (uint8_t)ra;
        outMax[(int64_t)brick * outStride + (base >> 12)] = (uint8_t)rbm;
        // SkipBlocks: bit 0 = every voxel of the block has one value (the skip bit is set, or not, by the level loop)
        if (blockFlag) blockFlag[(int64_t)brick * ((int64_t)1 << (D - 12)) + (base >> 12)] = ra == rbm ? 1u : 0u;
        if (blockFlagR) blockFlagR[(int64_t)brick * ((int64_t)1 << (D - 12)) + (base >> 12)] = ra == rbm ? 1u : 0u;   // half ranges all 0
    }
}

// --------------------------------------------------------------- compress ----
struct ReconBufs { uint8_t *b[3]; };

// Constant 4096-leaf blocks (kd_encode.hip k_pyramid12 sees min == max at their depth-(D-12) root) whose depth-(D-3)
// nodes are already reconstructed exactly: every node below has truth == parent's reconstruction, so it keeps
// (code 0), reproduces the value and adds no error at ANY distance (encodeNode R.cpp:457-502 with pd = 0), and the
// running-mean estimator never counts it (R.cpp:415-455).  The level loop does not touch such a block at levels D-1 and
// D at all -- no loads, no stores; its codes and reconstruction there stay unwritten and nothing reads them: the fused
// prune/emit kernel takes the block as one pruned subtree from the flag.  flag[block]: bit 0 constant (k_pyramid12),
// bit 1 skip (set by the level D-2 fill, whose wave is exactly one block and sees truth == parents there).
struct SkipBlocks {
    uint8_t *flag;        // B * nBlk, or null: feature off (MidRangeTree, tolerance 0, small trees, general extents)
    int64_t nBlk;
    int Dm2;              // D - 2
};
__device__ __forceinline__ bool skip_block(const SkipBlocks &sk, int brick, int d, uint32_t node)
{   // node of level d (d == D-1 or D) inside a skipped block?
    return sk.flag && d > sk.Dm2 && (sk.flag[(int64_t)brick * sk.nBlk + (node >> (10 + d - sk.Dm2))] & 2u) != 0u;
}

__device__ inline int phys_buf(const Ctrl &c, int role) { return role == 0 ? c.ra : c.rb; }

// rootMin/rootMax: the pyramid's (min, max) of each brick's root, or null.  A brick whose voxels are
// all equal has a closed-form encoding (k_const_finish); every later kernel skips it.
__global__ void k_ctrl_init(Ctrl *ctrls, const uint8_t *rootMin, const uint8_t *rootMax, int64_t mmStride)
{
    Ctrl &c = ctrls[blockIdx.x];
    if (threadIdx.x) return;
    c.constBrick = 0; c.constVal = 0; c.zeroRun = 0;
    if (rootMin) {
        const int mn = rootMin[(int64_t)blockIdx.x * mmStride], mx = rootMax[(int64_t)blockIdx.x * mmStride];
        c.constBrick = mn == mx ? 1 : 0;
        c.constVal = mn;
    }
    c.currentDistance = c.currentError = c.currentDF = c.currentStepSize = 0.0; // defect C-1 pinned to zero
    c.previousDistance = c.previousError = c.previousDF = c.previousStepSize = 0.0;
    c.errMinus = c.errPlus = 0;
    c.statL1 = 0;
    c.numActive = 0;
    c.epoch = 0; c.active = 0; c.fillThisEpoch = 0;
    c.cur = 0; c.prev = 1; c.pendingEqual = 0;
    c.par = 0; c.ra = 1; c.rb = 2;
    c.altValid = 0; c.altSel = 0; c.altDist = 0;
    c.numReverts = 0; c.maxErrBefore = 0; c.maxErrAfter = 0;
    c.estS = 0; c.estC = 0; c.estTbase = 0; c.estFallbacks = 0; c.estSeg = 0; c.estDone = 0; c.emitOverflow = 0;
    for (int i = 0; i < VR_MAX_DEPTH + 8; ++i) c.distanceMap[i] = 0;
}

// ---- running-mean start distance (R.cpp:254-266, encodeNodeEstimate R.cpp:415-455) ----
// The filter state (S,C) feeds back into every decision:
//     counted <=> pd>0 && ( S < pd*(2C+1) || (t>p && 2t-p>255) || (t<p && 2t<p) )      (*)
// (exact integer form of the reference's double arithmetic; S = sum of counted pd,
// C = their count, pd = |parent - truth|).  Plain fixed-point iteration over the whole
// level does not converge quickly (a flipped decision perturbs every later one), so
// the level is processed in three steps that are exact by construction:
//   k_est_head  one wave walks the first EST_HEAD nodes in order (64 nodes per step,
//               ballot iterated to its fixed point = the serial result by induction
//               on lanes) and fixes a window of EST_CAND candidate thresholds.
//   k_est_summ  fully parallel: for every later segment of EST_SEG nodes and every
//               candidate T, the sums the segment would add IF floor(S/(2C+1)) == T
//               held at every node of it (then (*) reduces to pd > T), plus two
//               bounds A,B such that the hypothesis is true for a start state (S0,C0)
//               whenever  S0 - T(2C0+1) >= A  and  S0 - (T+1)(2C0+1) < B.
//   k_est_walk  one wave carries the exact (S,C) through the segments, 64 per step:
//               prefix-sums the hypothesised sums, checks A/B for every segment with
//               its exact start state, commits up to the first failure and walks only
//               that segment node by node.
#define EST_SEG 1024
#define EST_CAND 8          // widest candidate window (records hold EST_CAND entries)
#define EST_ROUNDS 5        // rounds 0,1 use a 4-wide window, later rounds EST_CAND-wide recentred ones; the last one walks exactly if it must
#define EST_HEAD 4096

// Segment summaries of one brick: one plane per candidate, 16 bytes (sumS, sumC, A, B) per segment in it.
// k_est_summ writes a candidate's record with one 16-byte transaction (four lanes); the walking wave reads 64
// consecutive segments of its candidate as one contiguous kilobyte.
#define est_at(ci, seg) (((int64_t)(ci) * summStride + (int64_t)(seg)) * 4)

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane)
{
    (void)lane;
    return wave_incl_scan_add_dpp(v);      // DPP network: no LDS-crossbar round trips in the serial walks
}

// value of a wave-uniform lane: v_readlane (a few cycles) instead of an LDS-crossbar shuffle
__device__ __forceinline__ uint32_t lane_u32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ unsigned long long lane_u64(unsigned long long v, int l)
{
    return (unsigned long long)lane_u32((uint32_t)v, l) | ((unsigned long long)lane_u32((uint32_t)(v >> 32), l) << 32);
}

// exact in-order walk of nodes [lo, hi) by one wave; (S,C) are wave-uniform
#define EST_STAGE_NODES 4096        // est_exact_chain's LDS staging area holds this many truths and half as many parents
__device__ inline void est_exact_chain(const uint8_t *__restrict__ T, const uint8_t *__restrict__ P, int d, uint32_t lo,
                                       uint32_t hi, unsigned long long &S, uint32_t &C, int lane, const SkipBlocks &sk, int brick,
                                       uint8_t *stage = nullptr)
{
    // A walk is a chain of 64-node steps, each waiting for its bytes: with the next step's loads in flight it still ran at
    // one memory round trip per step (50 us for the 4096 nodes k_est_head walks on every level).  Ranges that fit are
    // staged in LDS by ONE batch of 16-byte loads (this wave is the workgroup: its LDS accesses are ordered).
    const uint32_t len = hi - lo;
    const bool staged = stage && d > 0 && len >= 64u && len <= (uint32_t)EST_STAGE_NODES && (len & 63u) == 0u && (lo & 31u) == 0u;
    const uint8_t *sT = stage, *sP = stage + EST_STAGE_NODES;
    if (staged) {
        for (uint32_t i = (uint32_t)lane * 16u; i < len; i += 1024u) *(uint4 *)(stage + i) = *(const uint4 *)(T + lo + i);
        for (uint32_t i = (uint32_t)lane * 16u; i < len / 2u; i += 1024u) *(uint4 *)(stage + EST_STAGE_NODES + i) = *(const uint4 *)(P + (lo >> 1) + i);
    }
    // the next 64 nodes are in flight while these are decided (the walk is latency-bound otherwise); chunks inside a
    // skipped block (SkipBlocks: nothing there counts, and its parents' reconstruction is not in memory) are passed over
    const bool sk0 = skip_block(sk, brick, d, lo);
    int tn = 0, pn = 0;
    if (staged) { tn = sT[lane]; pn = sP[lane >> 1]; }
    else { tn = (!sk0 && lo + lane < hi) ? T[lo + lane] : 0; pn = (!sk0 && lo + lane < hi && d > 0) ? P[(lo + lane) >> 1] : 0; }
    bool skipThis = sk0;
    for (uint32_t base = lo; base < hi; base += 64) {
        uint32_t i = base + lane;
        bool valid = i < hi;
        const int t = tn, p = pn;
        const bool skipped = skipThis;
        {
            const uint32_t i2 = i + 64;
            skipThis = base + 64 < hi && skip_block(sk, brick, d, base + 64);
            if (staged) {
                const uint32_t j2 = i2 - lo;
                tn = j2 < len ? sT[j2] : 0;
                pn = j2 < len ? sP[j2 >> 1] : 0;
            } else {
                tn = (!skipThis && i2 < hi) ? T[i2] : 0;
                pn = (!skipThis && i2 < hi && d > 0) ? P[i2 >> 1] : 0;
            }
        }
        if (skipped) continue;
        int pd = p > t ? p - t : t - p;
        bool forced = (t > p && 2 * t - p > 255) || (t < p && 2 * t < p);
        bool cand = valid && pd > 0;
        // first guess: the threshold at the chunk start
        bool dec = cand && (forced || S < (unsigned long long)pd * (2ull * C + 1ull));
        unsigned long long mask = __ballot(dec);
        for (int it = 0; it < 65; ++it) {
            uint32_t cpre = __popcll(mask & ((1ull << lane) - 1ull));
            uint32_t v = dec ? (uint32_t)pd : 0u;
            uint32_t spre = wave_incl_scan_u32(v, lane) - v;
            bool nd = cand && (forced || (S + spre) < (unsigned long long)pd * (2ull * (C + cpre) + 1ull));
            unsigned long long nmask = __ballot(nd);
            dec = nd;
            if (nmask == mask) break;
            mask = nmask;
        }
        // (a DPP scan + readlane: six dependent LDS-crossbar shuffles would cost more than the rest of the step)
        S += (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_add_dpp(dec ? (uint32_t)pd : 0u), 63);
        C += __popcll(mask);
    }
}

// candidate window [base, base+nc) around the current threshold
__device__ inline int est_window_base(long long T, int nc)
{
    long long b = T - (nc <= 4 ? 1 : 2);
    if (b < 0) b = 0;
    if (b > 256 - nc) b = 256 - nc;
    return (int)b;
}

__device__ inline void est_finish(Ctrl &c, unsigned long long S, uint32_t C, int maxEpochs)
{
    c.currentDistance = C > 0 ? round((double)S / (double)C) : 0.0; // R.cpp:263-266
    c.previousDistance = 0.0;  // R.cpp:272-274
    c.previousStepSize = 255.0;
    c.previousError = 65025.0;
    c.epoch = 0;
    c.active = maxEpochs > 0 ? 1 : 0;
    c.fillThisEpoch = c.active;
    c.errMinus = c.errPlus = 0;
    c.cur = 0; c.prev = 1; c.pendingEqual = 0;
}

__global__ void __launch_bounds__(64)
k_est_head(int d, int maxEpochs, Ctrl *ctrls, const uint8_t *__restrict__ temp, int64_t heapStride, ReconBufs rb,
           int64_t leafStride, SkipBlocks sk)
{
    const int brick = blockIdx.x, lane = threadIdx.x;
    Ctrl &c = ctrls[brick];
    if (c.constBrick) return;
    const uint8_t *T = temp + (int64_t)brick * heapStride + ((int64_t)1 << d);
    const uint8_t *P = rb.b[c.par] + (int64_t)brick * leafStride;
    const uint32_t n = 1u << d;
    unsigned long long S = 0;
    uint32_t C = 0;
    __shared__ __attribute__((aligned(16))) uint8_t stage[EST_STAGE_NODES + EST_STAGE_NODES / 2];
    est_exact_chain(T, P, d, 0, n < EST_HEAD ? n : EST_HEAD, S, C, lane, sk, brick, stage);
    if (lane == 0) {
        if (n <= EST_HEAD) est_finish(c, S, C, maxEpochs);
        else {
            c.estS = S; c.estC = C;
            c.estSeg = EST_HEAD / EST_SEG;
            c.estDone = 0;
            c.estTbase = est_window_base((long long)(S / (2ull * C + 1ull)), 4);
        }
    }
}

// One wave per segment, 16 consecutive nodes per lane as two chains of 8 (nodes k and k+8 share a
// packed 16-bit register pair).  With h = (t>p ? 255-t : t) the reference's "forced" cases are
// pd > h, so under the hypothesis floor(S/(2C+1)) == Th node k counts  <=>  pd_k > min(Th, h_k).
__global__ void __launch_bounds__(256)
k_est_summ(int d, int nc, Ctrl *ctrls, const uint8_t *__restrict__ temp, int64_t heapStride, ReconBufs rb,
           int64_t leafStride, uint32_t *__restrict__ summ, int64_t summStride, SkipBlocks sk)
{
    const int brick = blockIdx.y, lane = threadIdx.x & 63;
    const Ctrl &c = ctrls[brick];
    const uint32_t n = 1u << d;
    // one scalar round trip for the control-block fields, in front of the first branch
    const int cConst = c.constBrick, cDone = c.estDone, cSeg = c.estSeg, cPar = c.par, cTbase = c.estTbase;
    if (cConst || cDone) return;
    // segments from where the walk stands; later rounds run on a small grid (most bricks are done by then)
    const uint8_t *Tl = temp + (int64_t)brick * heapStride + ((int64_t)1 << d) + lane * 16;
    const uint8_t *Pl = (cPar == 0 ? rb.b[0] : (cPar == 1 ? rb.b[1] : rb.b[2])) + (int64_t)brick * leafStride + lane * 8;
    const uint32_t nseg = n / EST_SEG, seg0 = (uint32_t)cSeg + blockIdx.x * 4 + (threadIdx.x >> 6);
    // the next segment's bytes are in flight while this one is summarised (a wave's single load round trip is
    // what bounds this kernel: twice the bytes in flight per wave)
    uint4 tvN = make_uint4(0, 0, 0, 0);
    uint2 pvN = make_uint2(0, 0);
    bool skipN = seg0 < nseg && skip_block(sk, brick, d, seg0 * EST_SEG);       // a segment inside a skipped block: all zero, unread
    if (seg0 < nseg && !skipN) { tvN = ld16(Tl + (size_t)seg0 * EST_SEG); pvN = ld8(Pl + (size_t)seg0 * (EST_SEG / 2)); }
    for (uint32_t seg = seg0; seg < nseg; seg += gridDim.x * 4) {
    const uint4 tv = tvN;
    const uint2 pv = pvN;
    const bool skipped = skipN;
    {
        const uint32_t sn = seg + gridDim.x * 4;
        skipN = sn < nseg && skip_block(sk, brick, d, sn * EST_SEG);
        if (sn < nseg && !skipN) { tvN = ld16(Tl + (size_t)sn * EST_SEG); pvN = ld8(Pl + (size_t)sn * (EST_SEG / 2)); }
    }
    if (skipped) {
        uint32_t *outz = summ + (int64_t)brick * summStride * (4 * EST_CAND);
        if (lane < 4 * nc) outz[est_at(lane >> 2, seg) + (lane & 3)] = 0;
        continue;
    }
    const uint32_t tw[4] = {tv.x, tv.y, tv.z, tv.w};
    vr_s16x2 pd[8], h[8];
    uint32_t anyPd = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        // lanes: (node k, node k+8); their parents are bytes k>>1 of the two parent words
        const uint32_t bsel = (uint32_t)(k & 3), psel = (uint32_t)(k >> 1);
        const uint32_t T2 = __builtin_amdgcn_perm(tw[(k >> 2) + 2], tw[k >> 2], 0x0c040c00u | bsel | (bsel << 16));
        const uint32_t P2 = __builtin_amdgcn_perm(pv.y, pv.x, 0x0c040c00u | psel | (psel << 16));
        const vr_s16x2 diff = pk_s(T2) - pk_s(P2), nd = (vr_s16x2)(0) - diff;
        pd[k] = __builtin_elementwise_max(diff, nd);
        h[k] = pk_s(T2 ^ (pk_u(nd >> 15) & 0x00FF00FFu));
        anyPd |= pk_u(pd[k]);
    }
    const int Tbase = cTbase;
    uint32_t *out = summ + (int64_t)brick * summStride * (4 * EST_CAND);
    if (__ballot(anyPd != 0) == 0ull) {      // parents reproduce the truths exactly (constant regions): nothing counts
        if (lane < 4 * nc) out[est_at(lane >> 2, seg) + (lane & 3)] = 0;
        continue;
    }
    // every h of the wave at or above the last candidate (the usual case away from 0 and 255): min(Th, h) = Th
    vr_s16x2 hmin = h[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) hmin = __builtin_elementwise_min(hmin, h[k]);
#ifdef EST_NO_HBIG          // (timing experiments)
    const bool hBig = false;
#else
    const bool hBig = __ballot(min((int)hmin.x, (int)hmin.y) < Tbase + nc - 1) == 0ull;
#endif
#pragma unroll 1
    for (int ci = 0; ci < nc; ++ci) {
        const int Th = Tbase + ci;
        const vr_s16x2 Th2 = pk_s((uint32_t)Th * 0x10001u), w2 = pk_s((uint32_t)(2 * Th) * 0x10001u);
        // per chain: low = 2*Th*cc - s, and the extremes of low / low + 2cc over the positions BEFORE each node
        vr_s16x2 low = (vr_s16x2)(0), cc = (vr_s16x2)(0), amax = (vr_s16x2)(0), bmin = (vr_s16x2)(0);
        const vr_s16x2 two2 = pk_s(0x00020002u);
        if (hBig) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const vr_s16x2 dm = (Th2 - pd[k]) >> 15;     // -1 where the node counts
            low = dm * (pd[k] - w2) + low;
            cc -= dm;
            if (k < 7) {
                amax = __builtin_elementwise_max(amax, low);
                bmin = __builtin_elementwise_min(bmin, pk_mad(cc, two2, low));
            }
        }
        } else
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const vr_s16x2 dm = (__builtin_elementwise_min(Th2, h[k]) - pd[k]) >> 15;     // -1 where the node counts
            low = dm * (pd[k] - w2) + low;          // += 2 Th - pd where it counts (v_pk_mad_i16)
            cc -= dm;
            if (k < 7) {
                amax = __builtin_elementwise_max(amax, low);
                bmin = __builtin_elementwise_min(bmin, pk_mad(cc, two2, low));
            }
        }
        const int low1 = low.x, low2 = low.y, cc1 = cc.x, cc2 = cc.y;
        const int lowT = low1 + low2, ccT = cc1 + cc2;
        int a = max((int)amax.x, max((int)amax.y, 0) + low1);          // chain 2 starts at position 8 (its own 0 included)
        int b = min((int)bmin.x, min((int)bmin.y, 0) + low1 + 2 * cc1);
        const uint32_t sT = (uint32_t)(2 * Th * ccT - lowT);
        // one scan for both sums: s < 2^18 per wave, cc <= 1024
        const uint32_t packed = sT | ((uint32_t)ccT << 20);
        const uint32_t incl = wave_incl_scan_add_dpp(packed), excl = incl - packed;
        const int s0 = (int)(excl & 0xFFFFFu), c0 = (int)(excl >> 20);
        a += 2 * Th * c0 - s0;
        b += 2 * (Th + 1) * c0 - s0;
        a = wave_max_i32_dpp(a);
        b = wave_min_i32_dpp(b);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (lane < 4)
            out[est_at(ci, seg) + lane] = lane == 0 ? (tot & 0xFFFFFu) : (lane == 1 ? (tot >> 20) : (lane == 2 ? (uint32_t)a : (uint32_t)b));
    }
    }
}

__global__ void __launch_bounds__(64)
k_est_walk(int d, int maxEpochs, int nc, int ncNext, int lastRound, Ctrl *ctrls, const uint8_t *__restrict__ temp, int64_t heapStride, ReconBufs rb,
           int64_t leafStride, const uint32_t *__restrict__ summ, int64_t summStride, SkipBlocks sk)
{
    const int brick = blockIdx.x, lane = threadIdx.x;
    Ctrl &c = ctrls[brick];
    const uint8_t *T = temp + (int64_t)brick * heapStride + ((int64_t)1 << d);
    const uint8_t *P = rb.b[c.par] + (int64_t)brick * leafStride;
    const uint32_t n = 1u << d, nseg = n / EST_SEG;
    const uint32_t *sm = summ + (int64_t)brick * summStride * (4 * EST_CAND);
    if (c.constBrick || c.estDone) return;
    __shared__ __attribute__((aligned(16))) uint8_t stage[EST_STAGE_NODES + EST_STAGE_NODES / 2];    // est_exact_chain's (a failed segment)
    unsigned long long S = c.estS;
    uint32_t C = c.estC;
    const int Tbase = c.estTbase;
    long long Tc = (long long)(S / (2ull * C + 1ull));
    uint32_t seg = (uint32_t)c.estSeg;
    int fallbacks = 0;
    // 256 segments per step, four consecutive ones per lane (64 contiguous bytes of its candidate's plane), the next
    // step's records in flight while these are checked (same candidate: the common case).  (64 per step, one per lane:
    // 128 steps of one memory round trip each on the leaf level, 220 us.)
    uint32_t preSeg = 0xFFFFFFFFu;
    long long preCi = -1;
    uint4 pre[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pre[j] = make_uint4(0, 0, 0x80000000u, 0x7FFFFFFFu);
    const auto fetch = [&](long long ci_, uint32_t seg_, uint4 (&r)[4]) {
        const uint32_t k0 = seg_ + 4u * (uint32_t)lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[j] = make_uint4(0, 0, 0x80000000u, 0x7FFFFFFFu);          // (sums 0, A = INT_MIN, B = INT_MAX: always true)
            if (k0 + (uint32_t)j < nseg) r[j] = *(const uint4 *)(sm + est_at((int)ci_, k0 + (uint32_t)j));
        }
    };
    while (seg < nseg) {
        const long long ci = Tc - Tbase;
        if (ci < 0 || ci >= nc) {            // threshold left the candidate window
            if (!lastRound) {                // next round: new summaries around the new threshold
                if (lane == 0) {
                    c.estS = S; c.estC = C; c.estSeg = (int)seg;
                    c.estTbase = est_window_base(Tc, ncNext);
                    c.estFallbacks += fallbacks;
                }
                return;
            }
            est_exact_chain(T, P, d, seg * EST_SEG, n, S, C, lane, sk, brick);   // last resort: walk the rest in order
            fallbacks += (int)(nseg - seg);
            break;
        }
        uint4 r[4];
        if (preSeg == seg && preCi == ci) {
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = pre[j];
        } else fetch(ci, seg, r);
        preSeg = seg + 256u; preCi = ci;
        fetch(ci, preSeg, pre);
        const uint32_t k0 = seg + 4u * (uint32_t)lane;
        const uint32_t ssL = r[0].x + r[1].x + r[2].x + r[3].x, scL = r[0].y + r[1].y + r[2].y + r[3].y;
        const uint32_t si = wave_incl_scan_u32(ssL, lane), sci = wave_incl_scan_u32(scL, lane);
        uint32_t eS = si - ssL, eC = sci - scL;              // sums of the segments before mine in this step
        int failJ = 4;
        uint32_t fS = 0, fC = 0;                             // ... and before my first failing segment
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long S0 = (long long)S + (long long)eS;
            const long long q = 2ll * ((long long)C + (long long)eC) + 1ll;
            const bool ok = k0 + (uint32_t)j >= nseg || (S0 - Tc * q >= (long long)(int)r[j].z && S0 - (Tc + 1) * q < (long long)(int)r[j].w);
            if (!ok && failJ == 4) { failJ = j; fS = eS; fC = eC; }
            eS += r[j].x; eC += r[j].y;
        }
        const unsigned long long bad = __ballot(failJ < 4);
        if (bad == 0ull) {
            S += lane_u32(si, 63);
            C += lane_u32(sci, 63);
            seg += 256u;
            continue;
        }
        const int f = __ffsll((long long)bad) - 1;            // the lane of the first segment whose hypothesis fails
        S += lane_u32(fS, f);
        C += lane_u32(fC, f);
        seg += 4u * (uint32_t)f + lane_u32((uint32_t)failJ, f);
        est_exact_chain(T, P, d, seg * EST_SEG, (seg + 1) * EST_SEG, S, C, lane, sk, brick, stage);
        Tc = (long long)(S / (2ull * C + 1ull));
        seg += 1;
        ++fallbacks;
    }
    if (lane == 0) { est_finish(c, S, C, maxEpochs); c.estFallbacks += fallbacks; c.estDone = 1; }
}

__device__ inline unsigned long long block_sum_u64(unsigned long long v, unsigned long long *sh)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    unsigned long long r = 0;
    if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    __syncthreads();
    return r; // valid in thread 0
}

// One gradient-descent evaluation of a level (R.cpp:307-313 fused with :336-359):
// writes the 2-bit code and the reconstruction at `current`, and accumulates err^2 at
// current (per-block partials, summed in node order by k_control to reproduce the
// reference's serial double sum) and at current-1 / current+1 (exact integer atomics).
__global__ void __launch_bounds__(256)
k_fill(int d, Ctrl *ctrls, const uint8_t *__restrict__ temp, uint8_t *__restrict__ codes, int64_t heapStride,
       int64_t codeStride, ReconBufs rb, int64_t leafStride, unsigned long long *__restrict__ blockErr, int64_t nErrBlk)
{
    __shared__ unsigned long long sh[4];
    const int brick = blockIdx.y;
    Ctrl &c = ctrls[brick];
    if (c.constBrick || !c.fillThisEpoch) return;
    const uint32_t n = 1u << d;
    const uint8_t *T = temp + (int64_t)brick * heapStride + ((int64_t)1 << d);
    uint8_t *Cb = codes + (int64_t)brick * codeStride;
    const uint8_t *P = rb.b[c.par] + (int64_t)brick * leafStride;
    uint8_t *R = rb.b[phys_buf(c, c.cur)] + (int64_t)brick * leafStride;
    const int dist = (int)(uint8_t)c.currentDistance;
    const int distM = (int)(uint8_t)fmax(0.0, c.currentDistance - 1.0);   // R.cpp:334
    const int distP = (int)(uint8_t)fmin(255.0, c.currentDistance + 1.0);
    unsigned long long e0 = 0, em = 0, ep = 0;
    const uint32_t i0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    uint32_t pk = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t i = i0 + k;
        if (i < n) {
            int t = T[i];
            int p = d > 0 ? P[i >> 1] : 0;
            Enc e = encode_node(t, p, dist);
            pk |= (uint32_t)e.code << (2 * k);
            R[i] = (uint8_t)e.recon;
            e0 += (unsigned)(e.err * e.err);
            int a = encode_node(t, p, distM).err, b = encode_node(t, p, distP).err;
            em += (unsigned)(a * a);
            ep += (unsigned)(b * b);
        }
    }
    if (i0 < n) {
        if (d >= 2) Cb[(((int64_t)1 << d) + i0) >> 2] = (uint8_t)pk;     // four codes = one whole byte
        else if (d == 0) Cb[0] = (uint8_t)((Cb[0] & ~0x0Cu) | (pk << 2));            // heap node 1 (only writer)
        else Cb[0] = (uint8_t)((Cb[0] & 0x0Fu) | ((pk & 0xFu) << 4));                 // heap nodes 2, 3
    }
    e0 = block_sum_u64(e0, sh);
    em = block_sum_u64(em, sh);
    ep = block_sum_u64(ep, sh);
    if (threadIdx.x == 0) {
        blockErr[(int64_t)brick * nErrBlk + blockIdx.x] = e0;
        if (em) atomicAdd(&c.errMinus, em);
        if (ep) atomicAdd(&c.errPlus, ep);
    }
}

// Vector form of k_fill for levels with >= 4096 nodes: 16 consecutive nodes per thread
// (one 16-byte load of truths, one 8-byte load of parents, 16-byte stores of codes and
// reconstructions); a wave covers exactly one 1024-node partial-sum block of k_control.
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// A sibling pair of leaves as the level loop of a leafless build left it (k_prune_emit12): truth minus reconstruction at
// the distance of the buffer the loop ended on, codes at the distance of the last fill.  The two differ after a reverted
// epoch: the reference restores the reconstruction, not the codes (SURVEY C-2).  Same reduced encodeNode as k_fill16.
__device__ __forceinline__ void leaf_pair_encode(uint32_t tword, int tsel, uint32_t pword, int psel, uint32_t dR2, uint32_t dC2,
                                                 vr_s16x2 &dl, uint32_t &codes2)
{
    const EncPair c = enc_pair(tword, tsel, pword, psel);
    const vr_s16x2 x = enc_pair_x(c, dR2), nx = (vr_s16x2)(0) - x, ax = __builtin_elementwise_max(x, nx);
    uint32_t take = pk_u((ax - c.pd) >> 15);                                 // 0xFFFF per lane where |x| < pd: the step is taken
    // reconstruction = up ? t + x : t - x where taken, the parent's otherwise
    const uint32_t stepD = (c.up & pk_u(nx)) | (~c.up & pk_u(x));
    dl = pk_s((take & stepD) | (~take & pk_u(c.T2 - c.P2)));
    if (dC2 != dR2) {                                                        // (uniform per brick)
        const vr_s16x2 xc = enc_pair_x(c, dC2);
        take = pk_u((pk_abs(xc) - c.pd) >> 15);
    }
    codes2 = take & pk_u(pk_s(c.up) + pk_s(0x00020002u));                   // up ? 1 : 2 (per-lane wrap)
}

#ifndef FILL_ITERS
#define FILL_ITERS 8        // chunks of 4096 nodes per workgroup of k_fill16 (large levels)
#endif
template <bool STORE>     // false (the leaf level of a leafless build): errors only -- k_prune_emit12 recomputes codes and reconstruction
__global__ void __launch_bounds__(256)
k_fill16(int d, int maxEpochs, Ctrl *ctrls, const uint8_t *__restrict__ temp, uint8_t *__restrict__ codes, int64_t heapStride,
         int64_t codeStride, ReconBufs rb, int64_t leafStride, unsigned long long *__restrict__ blockErr, int64_t nErrBlk,
         SkipBlocks sk, int64_t errPlane, int iters)
{
    // A workgroup takes `iters` consecutive chunks of 4096 nodes (a wave: 1024 = one partial-sum block of k_control),
    // the next chunk's loads in flight while this one is evaluated: two million four-wave workgroups at the leaf level
    // spent a third of the launch being dispatched and reading the control block (a level-D fill in which every wave
    // skipped took 0.94 ms).
    __shared__ unsigned long long shm[4], shp[4];
    const int brick = blockIdx.y;
    // everything the wave needs from the brick's control block is fetched in ONE scalar round trip, before the
    // first branch (field by field behind branches it was seven dependent ones in front of the data loads)
    Ctrl &c = ctrls[brick];
    const int cConst = c.constBrick, cFill = c.fillThisEpoch, cPar = c.par, cCur = c.cur, cRa = c.ra, cRb = c.rb, cEpoch = c.epoch;
    const double cDist = c.currentDistance;
    if (cConst || !cFill) return;
    if (!STORE && c.altSel) return;          // this epoch's partials exist already (Ctrl::altSel)
    const uint8_t *T = temp + (int64_t)brick * heapStride + ((int64_t)1 << d);
    uint8_t *Cd = codes + (int64_t)brick * codeStride + ((int64_t)1 << (d - 2));   // packed: 4 codes per byte
    const int rPhys = cCur == 0 ? cRa : cRb;                                         // phys_buf(c, c.cur)
    const uint8_t *P = (cPar == 0 ? rb.b[0] : (cPar == 1 ? rb.b[1] : rb.b[2])) + (int64_t)brick * leafStride;
    uint8_t *R = (rPhys == 0 ? rb.b[0] : (rPhys == 1 ? rb.b[1] : rb.b[2])) + (int64_t)brick * leafStride;
    const int dist = (int)(uint8_t)cDist;
    const int distM = (int)(uint8_t)fmax(0.0, cDist - 1.0);   // R.cpp:334
    const int distP = (int)(uint8_t)fmin(255.0, cDist + 1.0);
    // the central difference (R.cpp:333-362) only steers a FOLLOWING epoch: in the last one its result is
    // never read (VolumeKdtree.cpp:333 skips it outright), so the two extra evaluations are not made
    const bool needDF = cEpoch + 1 < maxEpochs;
    const uint32_t d2 = (uint32_t)dist * 0x10001u, dm2 = (uint32_t)distM * 0x10001u, dp2 = (uint32_t)distP * 0x10001u;
    const int w = threadIdx.x >> 6;
    const uint32_t nchunk = (uint32_t)(((size_t)1 << d) >> 12);
    uint32_t chunk = blockIdx.x * (uint32_t)iters;
    const uint32_t chunkEnd = chunk + (uint32_t)iters < nchunk ? chunk + (uint32_t)iters : nchunk;
    unsigned long long accM = 0, accP = 0;
    // my wave's 1024 nodes inside a skipped block (SkipBlocks): no error, nothing to load, nothing anybody will read
    bool skippedN = skip_block(sk, brick, d, (chunk * 4u + (uint32_t)w) << 10);
    uint4 tvN = make_uint4(0, 0, 0, 0);
    uint2 pvN = make_uint2(0, 0);
    if (!skippedN) { const size_t i0 = ((size_t)chunk * 256u + threadIdx.x) * 16u; tvN = ld16(T + i0); pvN = ld8(P + (i0 >> 1)); }
    for (; chunk < chunkEnd; ++chunk) {
    const size_t i0 = ((size_t)chunk * 256u + threadIdx.x) * 16u;
    const bool skipped = skippedN;
    const uint4 tv = tvN;
    const uint2 pv = pvN;
    if (chunk + 1 < chunkEnd) {
        skippedN = skip_block(sk, brick, d, ((chunk + 1u) * 4u + (uint32_t)w) << 10);
        tvN = make_uint4(0, 0, 0, 0); pvN = make_uint2(0, 0);
        if (!skippedN) { const size_t i1 = i0 + 4096u; tvN = ld16(T + i1); pvN = ld8(P + (i1 >> 1)); }
    }
    const uint32_t tw[4] = {tv.x, tv.y, tv.z, tv.w}, pw[2] = {pv.x, pv.y};
    uint32_t e0 = 0, em = 0, ep = 0, wa = 0, wb = 0, rw[4] = {tw[0], tw[1], tw[2], tw[3]}, rprev = 0;
    // a wave whose 1024 truths all equal their parents' reconstruction (constant regions) has nothing to decide:
    // every code is "keep", the reconstruction is the truth, the error 0 at any distance (pd = 0 in encodeNode)
    const uint32_t differs = (tw[0] ^ __builtin_amdgcn_perm(0, pw[0], 0x01010000u)) | (tw[1] ^ __builtin_amdgcn_perm(0, pw[0], 0x03030202u)) |
                             (tw[2] ^ __builtin_amdgcn_perm(0, pw[1], 0x01010000u)) | (tw[3] ^ __builtin_amdgcn_perm(0, pw[1], 0x03030202u));
    const bool busy = !skipped && __ballot(differs != 0u) != 0ull;
    // level D-2: my wave IS one 4096-leaf block's nodes of this level.  Constant block (k_pyramid12) and every truth
    // equal to its parent's reconstruction: the two levels below need not be visited (SkipBlocks)
    if (sk.flag && d == sk.Dm2 && !busy && (threadIdx.x & 63) == 0) {
        uint8_t *f = sk.flag + (int64_t)brick * sk.nBlk + (size_t)chunk * 4 + w;
        if (*f & 1u) *f = 3u;        // the same answer in every epoch of the level: it depends on truths and parents only
    }
    if (busy)
#pragma unroll
    for (int j = 0; j < 8; ++j) {                        // sibling pair j: nodes 2j, 2j+1, parent byte j
        const EncPair c = enc_pair(tw[j >> 1], j & 1, pw[j >> 2], j & 3);
        const vr_s16x2 x = enc_pair_x(c, d2), ax = pk_abs(x);
        e0 = pk_sumsq(__builtin_elementwise_min(c.pd, ax), e0);
        if (STORE) {
        const uint32_t take = pk_u((ax - c.pd) >> 15);                 // 0xFFFF per lane where |x| < pd
        const uint32_t code2 = take & pk_u(pk_s(c.up) + pk_s(0x00020002u));   // up ? 1 : 2 (per-lane wrap)
        if (j < 4) wa |= code2 << (4 * j); else wb |= code2 << (4 * (j - 4));
        const vr_s16x2 r = pk_mad(x, pk_mad(pk_s(c.up), pk_s(0xFFFEFFFEu), pk_s(0xFFFFFFFFu)), c.T2);   // up ? t + x : t - x  (sign = -2 up - 1)
        const uint32_t rec = (take & pk_u(r)) | (~take & pk_u(c.P2));
        if (j & 1) rw[j >> 1] = __builtin_amdgcn_perm(rec, rprev, 0x06040200u); else rprev = rec;
        }
        if (needDF) {
            em = pk_sumsq(enc_pair_err(c, dm2), em);
            ep = pk_sumsq(enc_pair_err(c, dp2), ep);
        }
    }
    // even nodes sit at bits 4j, odd ones at 16+4j: fold to 2 bits per node
    const uint32_t cpk = ((wa | (wa >> 14)) & 0xFFFFu) | ((wb | (wb >> 14)) << 16);
    if (!skipped && STORE) {
        st4(Cd + (i0 >> 2), cpk);
        st16(R + i0, make_uint4(rw[0], rw[1], rw[2], rw[3]));
    }
    // per-lane sums are < 2^21, a wave's < 2^27: 32-bit DPP scans, the total in lane 63
    unsigned long long s0 = 0, sm = 0, sp = 0;
    if (busy) {
        s0 = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_add_dpp(e0), 63);
        sm = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_add_dpp(em), 63);
        sp = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_add_dpp(ep), 63);
    }
    if ((threadIdx.x & 63) == 0) {
        blockErr[(int64_t)brick * nErrBlk + (size_t)chunk * 4 + w] = s0;   // 1024 nodes per wave
        if (!STORE && needDF) {     // the central difference's sums per block too: a following epoch at distance -1 / +1 is these
            blockErr[errPlane + (int64_t)brick * nErrBlk + (size_t)chunk * 4 + w] = sm;
            blockErr[2 * errPlane + (int64_t)brick * nErrBlk + (size_t)chunk * 4 + w] = sp;
        }
    }
    accM += sm; accP += sp;
    }
    if ((threadIdx.x & 63) == 0) { shm[w] = accM; shp[w] = accP; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long tm = shm[0] + shm[1] + shm[2] + shm[3], tp = shp[0] + shp[1] + shp[2] + shp[3];
        if (tm) atomicAdd(&c.errMinus, tm);
        if (tp) atomicAdd(&c.errPlus, tp);
    }
}

// End of epoch e and head of epoch e+1 of the while loop at R.cpp:275-366, one wave
// per brick.  `currentError += err^2` runs over the level in node order starting from
// a fractional carry (defect C-1), so the double rounds only when the running sum
// crosses a power of two: while the sum provably stays inside its binade every partial
// sum is exactly representable, so whole block partials (64 per step, prefix-summed
// across the wave) are added at once; the block in which the sum crosses is walked
// node by node with the reference's own rounded additions.
__device__ __forceinline__ double next_pow2_above(double s)
{
    int ex;
    frexp(s, &ex);              // s in [2^(ex-1), 2^ex)
    return ldexp(1.0, ex);
}

__device__ inline double ctl_walk_block(double s, const uint8_t *__restrict__ T, const uint8_t *__restrict__ P, int d,
                                        int dist, uint32_t lo, uint32_t hi, int lane)
{
    int tn = lo + lane < hi ? T[lo + lane] : 0, pn = (lo + lane < hi && d > 0) ? P[(lo + lane) >> 1] : 0;   // one step ahead
    for (uint32_t base = lo; base < hi; base += 64) {
        uint32_t i = base + lane;
        uint32_t e = 0;
        const int t = tn, p = pn;
        {
            const uint32_t i2 = i + 64;
            tn = i2 < hi ? T[i2] : 0;
            pn = (i2 < hi && d > 0) ? P[i2 >> 1] : 0;
        }
        if (i < hi) {
            int er = encode_node(t, p, dist).err;
            e = (uint32_t)(er * er);
        }
        const uint32_t incl = wave_incl_scan_u32(e, lane);
        const uint32_t total = lane_u32(incl, 63);
        uint32_t consumed = 0;
        int start = 0;
        while (true) {
            bool ok = true;
            if (lane >= start && s != 0.0) ok = (s + (double)(incl - consumed)) < next_pow2_above(s);
            const unsigned long long bad = ~__ballot(ok);
            if (bad == 0ull) { s = s + (double)(total - consumed); break; }
            const int f = __ffsll((long long)bad) - 1;
            const uint32_t exclF = lane_u32(incl - e, f), eF = lane_u32(e, f);
            s = s + (double)(exclF - consumed);   // exact: still inside the binade
            s = s + (double)eF;                   // the reference's rounded add that crosses it
            consumed = exclF + eF;
            start = f + 1;
        }
    }
    return s;
}

// The same walk for a whole 1024-node block with 16 consecutive nodes per lane: one load round trip, the err^2
// prefix sums built once (local prefix + one wave scan), and every crossing found by comparing each node's
// prefix with the current bound -- instead of sixteen dependent 64-node steps.
__device__ inline double ctl_walk_block16(double s, const uint8_t *__restrict__ T, const uint8_t *__restrict__ P, int dist,
                                          uint32_t lo, int lane)
{
    const uint4 tv = *(const uint4 *)(T + lo + lane * 16);
    const uint2 pv = *(const uint2 *)(P + (lo >> 1) + lane * 8);
    const uint32_t tw[4] = {tv.x, tv.y, tv.z, tv.w}, pw[2] = {pv.x, pv.y};
    const uint32_t d2 = (uint32_t)dist * 0x10001u;
    uint32_t e[16], pre[16];          // err^2 of my nodes, inclusive prefix inside the lane
    uint32_t run = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const EncPair c = enc_pair(tw[j >> 1], j & 1, pw[j >> 2], j & 3);
        const vr_s16x2 er = enc_pair_err(c, d2);
        e[2 * j] = (uint32_t)((int)er.x * (int)er.x);
        e[2 * j + 1] = (uint32_t)((int)er.y * (int)er.y);
        run += e[2 * j]; pre[2 * j] = run;
        run += e[2 * j + 1]; pre[2 * j + 1] = run;
    }
    const uint32_t inclL = wave_incl_scan_add_dpp(run);       // < 2^27
    const uint32_t baseL = inclL - run, total = (uint32_t)__builtin_amdgcn_readlane((int)inclL, 63);
    uint32_t consumed = 0;
    int startPos = 0;                                          // nodes before block-local position startPos are added
    while (true) {
        // first node at or after startPos whose addition (with everything before it) would leave the binade
        int fail = 16;
        if (s != 0.0) {
            const double bound = next_pow2_above(s);
#pragma unroll
            for (int k = 15; k >= 0; --k)
                if (lane * 16 + k >= startPos && !((s + (double)(baseL + pre[k] - consumed)) < bound)) fail = k;
        }
        const unsigned long long bad = __ballot(fail < 16);
        if (bad == 0ull) { s = s + (double)(total - consumed); break; }
        const int f = __ffsll((long long)bad) - 1;
        const int kF = __builtin_amdgcn_readlane(fail, f);
        uint32_t eF = 0, gF = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k == kF) { eF = e[k]; gF = baseL + pre[k]; }
        eF = (uint32_t)__builtin_amdgcn_readlane((int)eF, f);
        gF = (uint32_t)__builtin_amdgcn_readlane((int)gF, f);
        s = s + (double)(gF - eF - consumed);   // exact: still inside the binade
        s = s + (double)eF;                     // the reference's rounded add that crosses it
        consumed = gF;
        startPos = f * 16 + kF + 1;
    }
    return s;
}

// 64 partials (1024 nodes each) starting at block `base`, added in order: whole partials while the sum stays inside
// its binade, the crossing block node by node
#define CTL_STAGE_BLOCKS 8192       // k_control keeps a level's per-block partials in LDS when there are at most this many (32 KiB)
__device__ inline double ctl_partials64(double s, const unsigned long long *__restrict__ be, const uint32_t *stg, uint32_t base, uint32_t nblk,
                                        const uint8_t *__restrict__ T, const uint8_t *__restrict__ P, int d, int dist,
                                        uint32_t n, int lane)
{
    const unsigned long long e2 = base + lane < nblk ? (stg ? (unsigned long long)stg[base + lane] : be[base + lane]) : 0ull;
    // two 32-bit DPP scans (24-bit limbs: 64 partials of < 2^27 each cannot overflow either)
    const unsigned long long incl = (unsigned long long)wave_incl_scan_add_dpp((uint32_t)(e2 & 0xFFFFFFull)) +
                                    ((unsigned long long)wave_incl_scan_add_dpp((uint32_t)(e2 >> 24)) << 24);
    const unsigned long long total = lane_u64(incl, 63);
    unsigned long long consumed = 0;
    int start = 0;
    while (true) {
        bool ok = true;
        // an integer-valued sum below 2^53 adds integers exactly whatever the binade
        const bool sInt = (s == floor(s)) && (s + (double)(total - consumed)) < 9007199254740992.0;
        if (lane >= start && !sInt) ok = (s + (double)(incl - consumed)) < next_pow2_above(s);
        const unsigned long long bad = ~__ballot(ok);
        if (bad == 0ull) { s = s + (double)(total - consumed); break; }
        const int f = __ffsll((long long)bad) - 1;
        const unsigned long long exclF = lane_u64(incl - e2, f), inclF = lane_u64(incl, f);
        s = s + (double)(exclF - consumed);
        uint32_t lo = (base + f) * FILL_NODES_PER_BLOCK, hi = lo + FILL_NODES_PER_BLOCK;
        if (hi > n) hi = n;
        s = (hi - lo == 1024u && d > 0) ? ctl_walk_block16(s, T, P, dist, lo, lane) : ctl_walk_block(s, T, P, d, dist, lo, hi, lane);
        consumed = inclF;
        start = f + 1;
    }
    return s;
}

__global__ void __launch_bounds__(64)
k_control(int d, int maxEpochs, int guarded, Ctrl *ctrls, const uint8_t *__restrict__ temp, int64_t heapStride,
          ReconBufs rb, int64_t leafStride, const unsigned long long *__restrict__ blockErr, int64_t nErrBlk, int64_t errPlane,
          int leaflessLeaf)
{
    const int brick = blockIdx.x, lane = threadIdx.x;
    __shared__ __attribute__((aligned(16))) uint32_t ctlStage[CTL_STAGE_BLOCKS];
    Ctrl &c = ctrls[brick];
    if (c.constBrick) return;
    const uint32_t n = 1u << d;
    const bool ending = c.active && c.fillThisEpoch;
    double s = c.currentError;
    if (ending) {
        const uint8_t *T = temp + (int64_t)brick * heapStride + ((int64_t)1 << d);
        const uint8_t *P = rb.b[c.par] + (int64_t)brick * leafStride;
        const int dist = (int)(uint8_t)c.currentDistance;
        const uint32_t nblk = (n + FILL_NODES_PER_BLOCK - 1) / FILL_NODES_PER_BLOCK;
        const unsigned long long *be = blockErr + (int64_t)c.altSel * errPlane + (int64_t)brick * nErrBlk;
        // Two levels: every lane sums one chunk of 64 partials (64 independent loads, no dependent steps), the wave
        // adds whole chunks in order while the running double provably stays inside its binade, and only a chunk in
        // which it crosses a power of two is opened (ctl_partials64, which opens only the crossing block).  A level
        // has ~log2 crossings, so this touches a handful of chunks instead of stepping through all of them.
        const uint32_t nchunk = (nblk + 63u) >> 6;
        // the partials (each < 2^27) go to LDS in one batch of coalesced loads: the chunk sums and every chunk that is
        // opened below then cost no memory round trip (64 uncoalesced 8-byte loads per lane in dependent batches, and
        // one more round trip per opened chunk, were a third of this kernel's 25-75 us)
        const uint32_t *stg = nullptr;
        if (nblk >= 256u && nblk <= (uint32_t)CTL_STAGE_BLOCKS && (nblk & 255u) == 0u) {
            for (uint32_t i = (uint32_t)lane * 4u; i < nblk; i += 256u) {
                const uint4 a = *(const uint4 *)(be + i), b = *(const uint4 *)(be + i + 2);
                *(uint4 *)(ctlStage + i) = make_uint4(a.x, a.z, b.x, b.z);
            }
            stg = ctlStage;
        }
        for (uint32_t cb = 0; cb < nchunk; cb += 64) {
            const uint32_t ch = cb + (uint32_t)lane;
            unsigned long long tot = 0;
            if (ch < nchunk && stg) {
                const uint32_t *q = stg + (size_t)ch * 64;      // (whole chunks: nblk is a multiple of 256)
#pragma unroll 8
                for (int k = 0; k < 64; ++k) tot += q[(k + lane) & 63];      // rotated: the lanes' rows sit 64 words apart
            } else if (ch < nchunk) {
                const unsigned long long *q = be + (size_t)ch * 64;
                const uint32_t m = nblk - ch * 64 < 64u ? nblk - ch * 64 : 64u;
                if (m == 64u) {
#pragma unroll 8
                    for (int k = 0; k < 64; ++k) tot += q[k];
                } else
                    for (uint32_t k = 0; k < m; ++k) tot += q[k];
            }
            // chunk totals < 2^33: 24-bit low limbs (sum < 2^30) and high limbs < 2^9 (sum < 2^15)
            const unsigned long long incl = (unsigned long long)wave_incl_scan_add_dpp((uint32_t)(tot & 0xFFFFFFull)) +
                                            ((unsigned long long)wave_incl_scan_add_dpp((uint32_t)(tot >> 24)) << 24);
            const unsigned long long total = lane_u64(incl, 63);
            unsigned long long consumed = 0;
            int start = 0;
            while (true) {
                bool ok = true;
                const bool sInt = (s == floor(s)) && (s + (double)(total - consumed)) < 9007199254740992.0;
                if (lane >= start && !sInt) ok = (s + (double)(incl - consumed)) < next_pow2_above(s);
                const unsigned long long bad = ~__ballot(ok);
                if (bad == 0ull) { s = s + (double)(total - consumed); break; }
                const int f = __ffsll((long long)bad) - 1;
                const unsigned long long exclF = lane_u64(incl - tot, f), inclF = lane_u64(incl, f);
                s = s + (double)(exclF - consumed);       // exact: still inside the binade
                s = ctl_partials64(s, be, stg, (cb + (uint32_t)f) * 64u, nblk, T, P, d, dist, n, lane);
                consumed = inclF;
                start = f + 1;
            }
        }
    }
    if (lane != 0) return;
    if (ending) {
        c.roleDist[c.cur] = (int)(uint8_t)c.currentDistance;   // what this epoch's fill wrote its buffer and the codes with
        c.codesDist = (int)(uint8_t)c.currentDistance;
        // (a real fill that made the central difference leaves its minus / plus partials behind)
        c.altValid = leaflessLeaf && !c.altSel && c.epoch + 1 < maxEpochs;
        c.altDist = (int)(uint8_t)c.currentDistance;
        c.altSel = 0;
        c.currentError = s / (double)n;                        // R.cpp:315
        if (c.currentError < 1.0) {                            // R.cpp:319
            c.active = 0;
        } else if (c.epoch != 0 && c.currentError > c.previousError) { // revert R.cpp:323-331
            c.currentError = c.previousError;
            c.currentDistance = c.previousDistance;
            c.currentDF = c.previousDF;
            c.currentStepSize = c.previousStepSize / 2.0;
            int tmp = c.cur; c.cur = c.prev; c.prev = tmp;     // recon.swap(reconPreviousEpoch)
            c.pendingEqual = 0;
            c.numReverts++;
            c.epoch++;
        } else {
            if (!guarded || c.epoch + 1 < maxEpochs) {         // VolumeKdtree.cpp:333 guard
                double e0 = (double)c.errMinus / (double)n;    // R.cpp:346,357
                double e1 = (double)c.errPlus / (double)n;
                c.currentDF = (e1 - e0) / 2.0;                 // R.cpp:361, h = 1
                c.currentStepSize = fmax(-4.0, fmin(4.0, -1.25 * c.currentDF));
                // reconPreviousEpoch = recon (R.cpp:364) without a copy: the buffer just
                // written becomes the snapshot and the next fill goes to the other one.
                c.prev = c.cur;
                c.cur = 1 - c.cur;
                c.pendingEqual = 1;
            }
            c.epoch++;
        }
    }
    if (c.active) {                                            // while-condition + loop head
        if (!(c.epoch < maxEpochs && fabs(c.previousStepSize) >= 0.5)) {
            c.active = 0;
        } else if (c.epoch != 0) {
            c.previousDistance = c.currentDistance;
            c.previousError = c.currentError;
            c.previousDF = c.currentDF;
            c.previousStepSize = c.currentStepSize;
            c.currentDistance = round(fmin(255.0, fmax(0.0, c.previousDistance + c.previousStepSize)));
            if (c.currentDistance == c.previousDistance) c.active = 0; // R.cpp:287-288
        }
    }
    c.fillThisEpoch = c.active;
    c.errMinus = c.errPlus = 0;
    if (c.active) c.pendingEqual = 0; // the coming fill overwrites the non-snapshot buffer
    // the coming epoch is the last that can run (no central difference of its own) and sits one step beside the fill
    // whose partials are in memory: no fill
    if (leaflessLeaf && c.active && c.altValid && c.epoch + 1 >= maxEpochs) {
        const int nd = (int)(uint8_t)c.currentDistance;
        const int dm = (int)(uint8_t)fmax(0.0, (double)c.altDist - 1.0), dp = (int)(uint8_t)fmin(255.0, (double)c.altDist + 1.0);   // k_fill16's distM / distP
        if (nd != c.altDist) c.altSel = nd == dm ? 1 : (nd == dp ? 2 : 0);
    }
}

// R.cpp:369-381: record the level's distance, make its reconstruction the next level's parents.
__global__ void k_level_end(int d, Ctrl *ctrls)
{
    if (threadIdx.x) return;
    Ctrl &c = ctrls[blockIdx.x];
    if (c.constBrick) return;
    c.distanceMap[d] = (uint8_t)c.currentDistance;
    int finalRole = c.pendingEqual ? c.prev : c.cur;
    c.finalReconDist = c.roleDist[finalRole];
    c.finalCodesDist = c.codesDist;
    int finalPhys = phys_buf(c, finalRole);
    int otherPhys = phys_buf(c, 1 - finalRole);
    int oldPar = c.par;
    c.par = finalPhys;
    c.ra = oldPar;
    c.rb = otherPhys;
    c.cur = 0; c.prev = 1; c.pendingEqual = 0;
    c.active = 0; c.fillThisEpoch = 0;
    c.altValid = 0; c.altSel = 0;
}

// The range stream's prune follows the mid stream (M.cpp:864-865); see k_prune_*.
// ------------------------------------------------------------------ prune ----
// per-block statistics record: sum |recon - truth| after growth (40 bits) | max error before << 40 | max after << 48
__device__ __forceinline__ unsigned long long stat_pack(unsigned long long l1, int maxBefore, int maxAfter)
{
    return l1 | ((unsigned long long)maxBefore << 40) | ((unsigned long long)maxAfter << 48);
}
__device__ __forceinline__ unsigned long long stat_merge(unsigned long long a, unsigned long long b)
{
    const unsigned long long M40 = (1ull << 40) - 1ull;
    const unsigned long long mb = max((a >> 40) & 255ull, (b >> 40) & 255ull), ma = max((a >> 48) & 255ull, (b >> 48) & 255ull);
    return ((a & M40) + (b & M40)) | (mb << 40) | (ma << 48);
}
__global__ void __launch_bounds__(256)
k_prune_leaf(int D, int tol, Ctrl *ctrls, const uint8_t *__restrict__ temp, uint8_t *__restrict__ codes,
             uint8_t *__restrict__ codesRange, int64_t heapStride, int64_t codeStride, ReconBufs rb, int64_t leafStride,
             int maxDepth,
             unsigned long long *__restrict__ blockL1, int64_t nEmitBlk)
{
    __shared__ unsigned long long shl[4];
    const int brick = blockIdx.y;
    Ctrl &c = ctrls[brick];
    if (c.constBrick) return;
    const uint32_t n = 1u << D;
    uint32_t r = blockIdx.x * 256u + threadIdx.x;
    int err = 0, fe = 0;
    if (r < n) {
        const int64_t hi = (int64_t)brick * heapStride + ((int64_t)1 << D) + r;
        int t = temp[hi];
        int rec = rb.b[c.par][(int64_t)brick * leafStride + r];
        err = rec > t ? rec - t : t - rec;
        fe = err;
        uint8_t *Cb = codes + (int64_t)brick * codeStride;
        const int64_t ci = ((int64_t)1 << D) + r;
        const int code = cget(Cb, ci);
        if (code == 0 && err < tol) {        // R.cpp:618-626 (leaf: no children)
            cset3(Cb, ci);
            if (codesRange) cset3(codesRange + (int64_t)brick * codeStride, ci);
        } else if (code != 3) {              // an unpruned leaf is always live: its error after branch growth
            int depth = D;
            while (depth < maxDepth) {
                const int e2 = rec > t ? rec - t : t - rec;
                if (e2 > tol) { ++depth; rec = encode_node(t, rec, 64 >> (depth - D - 1)).recon; }
                else break;
            }
            fe = rec > t ? rec - t : t - rec;
        }
    }
    int fm = fe;
    unsigned long long l1 = (unsigned long long)fe;
    for (int o = 32; o > 0; o >>= 1) {
        int u = __shfl_xor(err, o); err = u > err ? u : err;
        int w = __shfl_xor(fm, o); fm = w > fm ? w : fm;
        l1 += __shfl_xor(l1, o);
    }
    // statistics leave as per-block records (no atomics on the brick's control block: every wave of a
    // brick hitting one address serialises on one L2 channel); k_emit_stats reduces them
    if ((threadIdx.x & 63) == 0) shl[threadIdx.x >> 6] = stat_pack(l1, err, fm);    // R.cpp:71-76, 115-129
    __syncthreads();
    if (threadIdx.x == 0)
        blockL1[(int64_t)brick * nEmitBlk + blockIdx.x] = stat_merge(stat_merge(shl[0], shl[1]), stat_merge(shl[2], shl[3]));
}

__global__ void __launch_bounds__(256)
k_prune_level(int d, const Ctrl *ctrls, uint8_t *__restrict__ codes, uint8_t *__restrict__ codesRange, int64_t codeStride)
{
    const int brick = blockIdx.y;
    if (ctrls[brick].constBrick) return;
    const uint32_t n = 1u << d;
    uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    uint8_t *Cb = codes + (int64_t)brick * codeStride;
    const int64_t me = ((int64_t)1 << d) + p, ch = ((int64_t)1 << (d + 1)) + 2 * (int64_t)p;
    if (cget(Cb, ch) == 3 && cget(Cb, ch + 1) == 3 && cget(Cb, me) == 0) {  // R.cpp:624
        cset3(Cb, me);
        if (codesRange) cset3(codesRange + (int64_t)brick * codeStride, me);
    }
}

// The grown branch of a leaf (R.cpp:655-704: encodeNode against the leaf's own reconstruction with
// distances 64, 32, .., 1 until the error is within tolerance) depends only on the initial error
// m0 = |truth - recon| and the side, as long as the clamps at 0 / 255 cannot matter: a clamped step
// lands min(truth, 255 - truth) away from the truth, which is never an improvement while that is
// >= the current error.  lut[m0] = final error | count << 8 | tokens << 16 (2 bits each, "add" = towards the truth
// first; byte-aligned so that two entries fold into packed lanes with one v_perm each); the other side swaps
// add <-> sub.  Leaves with m0 > min(t, 255 - t) take
// the exact step-by-step path.
#define VR_CHAIN_LUT_SIGNED 260     // word offset of the signed-key table inside BrickSet::chainLut (512 entries)
__global__ void k_chain_lut(int tol, int nsteps, uint32_t *__restrict__ lut)
{
    const int m0 = threadIdx.x, t = 128;
    int rec = m0 <= 127 ? t - m0 : 0;
    uint32_t bits = 0, n = 0;
    int lastKeep = 0;                 // the last evaluated branch node kept its parent's value (code 0)
    for (int i = 0; i < nsteps; ++i) {
        const int err = rec > t ? rec - t : t - rec;
        if (err > tol) { const Enc e = encode_node(t, rec, 64 >> i); rec = e.recon; bits |= (uint32_t)e.code << (2 * n); ++n; lastKeep = e.code == 0; }
        else { bits |= 3u << (2 * n); ++n; break; }
    }
    const int fe = rec > t ? rec - t : t - rec;
    lut[m0] = (uint32_t)fe | (n << 8) | (bits << 16);     // byte 0: final error, byte 1: token count, bytes 2-3: tokens
    // the same keyed by the SIGNED error truth - reconstruction (9 bits, at lut + VR_CHAIN_LUT_SIGNED): a negative error
    // takes the mirrored branch (add <-> sub), so k_prune_emit12 needs neither the mirror nor the select
    lut[VR_CHAIN_LUT_SIGNED + m0] = lut[m0];
    if (m0 >= 1) lut[VR_CHAIN_LUT_SIGNED + 512 - m0] = (uint32_t)fe | (n << 8) | ((bits ^ (((bits ^ (bits >> 1)) & 0x1555u) * 3u)) << 16);
    if (m0 == 0) lut[VR_CHAIN_LUT_SIGNED + 256] = 0;
    // entries 0..127 are the ones a leaf can reach through the table (m0 <= min(t, 255 - t)); one that ends on a
    // "keep" would need the reference's zero-run rewrite (R.cpp:662-669,686-688): counted, asserted zero by the tests
    const int zr = __syncthreads_count(m0 <= 127 && lastKeep);
    if (m0 == 0) lut[256] = (uint32_t)zr;
}
__device__ __forceinline__ uint32_t chain_mirror(uint32_t ch) { return ch ^ (((ch ^ (ch >> 1)) & 0x1555u) * 3u); }   // add <-> sub

// Leaf prune + the 12 levels above it in one block (the level-synchronous kernels above
// stay for small bricks and for the levels nearer the root): 16 leaves per thread as
// 16-byte loads, pruned flags carried upwards in LDS, each level's codes read-modified-
// written in place.  Equal to the serial recursion (R.cpp:596-629) because a node only
// depends on its two children.
__global__ void __launch_bounds__(256)
k_prune12(int D, int tol, Ctrl *ctrls, const uint8_t *__restrict__ temp, uint8_t *__restrict__ codes,
          uint8_t *__restrict__ codesRange, int64_t heapStride, int64_t codeStride, ReconBufs rb, int64_t leafStride,
          int maxDepth, uint32_t *__restrict__ subTok, int64_t nEmitBlk, unsigned long long *__restrict__ blockL1,
          const uint32_t *__restrict__ chainLut)
{
    __shared__ uint32_t lutS[256];
    __shared__ uint8_t fl[2][2048];
    __shared__ uint16_t cnt[2][2048];      // tokens a (live) subtree emits, carried upwards with the flags
    __shared__ uint8_t lv[2048], lvOld[2048];     // codes of the block's nodes at depths D-12 .. D-2, heap order
    const int brick = blockIdx.y, t = threadIdx.x;
    Ctrl &c = ctrls[brick];
    if (c.constBrick) return;
    const uint32_t base = blockIdx.x << 12;
    uint8_t *Cb = codes + (int64_t)brick * codeStride;
    uint8_t *CR = codesRange ? codesRange + (int64_t)brick * codeStride : nullptr;
    // every global load of the block is issued before any is looked at (one memory round trip)
    const uint32_t lutV = chainLut[t];
    uint32_t lvB[8];
    int lvSh[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int h = q * 256 + (t ? t : (q ? 0 : 1));           // heap index inside the block (index 0 is unused)
        const int lq = 31 - __clz(h);
        const int64_t ni = ((int64_t)1 << (D - 12 + lq)) + (((int64_t)blockIdx.x) << lq) + (h - (1 << lq));
        lvB[q] = Cb[ni >> 2];
        lvSh[q] = (int)(ni & 3) * 2;
    }
    const int64_t li = ((int64_t)1 << D) + base + t * 16;
    uint32_t cpk = *(const uint32_t *)(Cb + (li >> 2));       // my 16 leaf codes, packed
    const uint4 tv = *(const uint4 *)(temp + (int64_t)brick * heapStride + li);
    const uint4 rv = *(const uint4 *)(rb.b[c.par] + (int64_t)brick * leafStride + base + t * 16);
    lutS[t] = lutV;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int h = q * 256 + t;
        const uint8_t cv0 = (uint8_t)((lvB[q] >> lvSh[q]) & 3u);
        if (h) { lv[h] = cv0; lvOld[h] = cv0; }
    }
    const uint32_t tw[4] = {tv.x, tv.y, tv.z, tv.w}, rw[4] = {rv.x, rv.y, rv.z, rv.w};
    // Sibling leaves 2j, 2j+1 share a packed 16-bit register pair.  State of a leaf: m = |truth - recon|,
    // sg = 0xFFFF where truth < recon.  One step of the grown branch with distance d (encodeNode with the
    // leaf's own reconstruction as parent, kd_common.h): x = min(d - m, sg ? t : 255 - t); taken iff |x| < m;
    // then m = |x| and the side flips iff x > 0.
    const uint32_t tol2 = (uint32_t)tol * 0x10001u;
    uint32_t T2[8], m[8], sg[8], act[8], nt[8];
    uint32_t wa = 0, wb = 0, bothMask = 0, anyAct = 0;
    vr_s16x2 mxB = (vr_s16x2)(0);
    __syncthreads();                      // lutS
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t sel = (j & 1) ? 0x0c030c02u : 0x0c010c00u;
        T2[j] = __builtin_amdgcn_perm(0, tw[j >> 1], sel);
        const vr_s16x2 dl = pk_s(T2[j]) - pk_s(__builtin_amdgcn_perm(0, rw[j >> 1], sel));
        const vr_s16x2 mm = pk_abs(dl);
        m[j] = pk_u(mm);
        sg[j] = pk_u(dl >> 15);
        mxB = __builtin_elementwise_max(mxB, mm);
        const uint32_t lt = pk_u((mm - pk_s(tol2)) >> 15);                                     // err < tol
        const uint32_t cl2 = ((cpk >> (4 * j)) & 3u) | (((cpk >> (4 * j + 2)) & 3u) << 16);     // the pair's codes
        const uint32_t isz = pk_u((pk_s(cl2) - pk_s(0x00010001u)) >> 15), is3 = pk_u((pk_s(0x00020002u) - pk_s(cl2)) >> 15);
        const uint32_t newp = isz & lt;                                                        // R.cpp:618-626
        const uint32_t pruned = newp | is3;
        if (j < 4) wa |= (newp & 0x000C0003u) << (4 * j); else wb |= (newp & 0x000C0003u) << (4 * (j - 4));
        bothMask |= ((pruned & (pruned >> 16)) & 1u) << j;
        // an unpruned leaf is always live (pruning is closed downwards): its code, then the grown branch --
        // from the table unless a clamp could matter (m > min(t, 255 - t)), then step by step below
        const vr_s16x2 lim = __builtin_elementwise_min(pk_s(T2[j]), pk_s(T2[j] ^ 0x00FF00FFu));
        const uint32_t viol = pk_u((lim - mm) >> 15);
        const uint32_t e0 = lutS[m[j] & 255u], e1 = lutS[(m[j] >> 16) & 255u];
        const uint32_t useL = ~pruned & ~viol;
        nt[j] = 0x00010001u + (useL & __builtin_amdgcn_perm(e1, e0, 0x0c050c01u));                // the two counts
        m[j] = (useL & __builtin_amdgcn_perm(e1, e0, 0x0c040c00u)) | (~useL & m[j]);               // the two final errors
        act[j] = ~pruned & viol;
        anyAct |= act[j];
    }
    cpk |= ((wa | (wa >> 16)) & 0xFFFFu) | ((wb | (wb >> 16)) << 16);
    const int nsteps = maxDepth - D;      // distanceMap[D+1..] = 64, 32, .., 1 (R.cpp:23,94-97)
    for (int i = 0; i < nsteps; ++i) {
        if (__ballot(anyAct != 0) == 0ull) break;        // wave-uniform: every branch of the wave has ended
        const uint32_t d2 = (uint32_t)(64 >> i) * 0x10001u;
        anyAct = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const vr_s16x2 mm = pk_s(m[j]);
            const uint32_t gt = pk_u((pk_s(tol2) - mm) >> 15);                                 // err > tol: grow
            nt[j] = pk_u(pk_s(nt[j]) - pk_s(act[j]));                                          // a code or the terminator
            const uint32_t go = act[j] & gt;
            const uint32_t lim = (sg[j] & T2[j]) | (~sg[j] & (T2[j] ^ 0x00FF00FFu));
            const vr_s16x2 x = __builtin_elementwise_min(pk_s(d2) - mm, pk_s(lim));
            const vr_s16x2 nx = (vr_s16x2)(0) - x, ax = __builtin_elementwise_max(x, nx);
            const uint32_t take = go & pk_u((ax - mm) >> 15);
            m[j] = (take & pk_u(ax)) | (~take & m[j]);
            sg[j] ^= take & pk_u(nx >> 15);
            act[j] = go;
            anyAct |= go;
        }
    }
    vr_s16x2 mxA = (vr_s16x2)(0), l1p = (vr_s16x2)(0);   // encoder's own statistics after branch growth,
#pragma unroll
    for (int j = 0; j < 8; ++j) {                         // over every leaf (R.cpp:115-129)
        mxA = __builtin_elementwise_max(mxA, pk_s(m[j]));
        l1p += pk_s(m[j]);
    }
    int maxErr = max((int)mxB.x, (int)mxB.y), maxAfter = max((int)mxA.x, (int)mxA.y);
    const uint32_t l1After = (uint32_t)((int)l1p.x + (int)l1p.y);
    uint32_t pr = 0;            // pruned flags of sibling pairs: bit k <=> both leaves 2k, 2k+1 pruned
    pr = bothMask;
    *(uint32_t *)(Cb + (li >> 2)) = cpk;
    if (CR) {
        uint32_t q = *(const uint32_t *)(CR + (li >> 2));
        q |= (cpk & (cpk >> 1) & 0x55555555u) * 3u;     // every pruned leaf: range code 3 as well (M.cpp:864-865)
        *(uint32_t *)(CR + (li >> 2)) = q;
    }
    unsigned long long l1w = l1After;
    for (int o = 32; o > 0; o >>= 1) {
        int u = __shfl_xor(maxErr, o); maxErr = u > maxErr ? u : maxErr;
        int w = __shfl_xor(maxAfter, o); maxAfter = w > maxAfter ? w : maxAfter;
        l1w += __shfl_xor(l1w, o);
    }
    if ((t & 63) == 0)      // one record per wave = 1024 leaves
        blockL1[(int64_t)brick * nEmitBlk + (size_t)blockIdx.x * 4 + (t >> 6)] = stat_pack(l1w, maxErr, maxAfter);
    // level D-1: 8 nodes per thread, children flags in registers
    {
        const int64_t ni = ((int64_t)1 << (D - 1)) + (base >> 1) + t * 8;
        uint32_t v = *(const uint16_t *)(Cb + (ni >> 2)), vr = CR ? *(const uint16_t *)(CR + (ni >> 2)) : 0u;   // 8 codes
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool both = (pr >> k) & 1u;
            const int code = (int)((v >> (2 * k)) & 3u);
            const bool p = both && code == 0;
            if (p) { v |= 3u << (2 * k); vr |= 3u << (2 * k); }
            const bool f3 = p || code == 3;
            fl[1][t * 8 + k] = (uint8_t)(f3 ? 1 : 0);
            cnt[1][t * 8 + k] = (uint16_t)(f3 ? 1u : 1u + (nt[k] & 0xFFFFu) + (nt[k] >> 16));
        }
        *(uint16_t *)(Cb + (ni >> 2)) = (uint16_t)v;
        if (CR) *(uint16_t *)(CR + (ni >> 2)) = (uint16_t)vr;
    }
    __syncthreads();
    // levels D-2 .. D-12 of this block (2047 codes) were fetched into LDS up front (one memory
    // round trip instead of eleven dependent ones); prune them there and write them back once
    for (int l = 2; l <= 12; ++l) {
        const int m = 4096 >> l, src = (l - 1) & 1, dst = l & 1;
        const int hb = 1 << (12 - l);               // heap base of this level inside the block
        for (int i = t; i < m; i += 256) {
            const bool both = fl[src][2 * i] && fl[src][2 * i + 1];
            const int code = lv[hb + i];
            const bool p = both && code == 0;
            if (p) lv[hb + i] = 3;
            const bool f3 = p || code == 3;
            fl[dst][i] = (uint8_t)(f3 ? 1 : 0);
            const uint32_t cn = f3 ? 1u : 1u + cnt[src][2 * i] + cnt[src][2 * i + 1];
            cnt[dst][i] = (uint16_t)cn;
            if (l == 10 && subTok) subTok[(int64_t)brick * nEmitBlk + (size_t)blockIdx.x * 4 + i] = cn;   // one emit block
        }
        __syncthreads();
    }
    // write the block's levels back packed: levels with >= 4 nodes own whole bytes (plain stores of
    // every byte); the top two levels (1 and 2 nodes) share bytes with neighbouring blocks -> atomic OR
    for (int hb4 = 1 + t; hb4 < 512; hb4 += 256) {     // heap bytes 1..511 <-> heap nodes 4..2047
        const int h = hb4 * 4;
        const int lq = 31 - __clz(h);                   // depth below the block root (>= 2)
        const int64_t gi = ((int64_t)1 << (D - 12 + lq)) + (((int64_t)blockIdx.x) << lq) + (h - (1 << lq));
        const uint8_t pk = (uint8_t)(lv[h] | (lv[h + 1] << 2) | (lv[h + 2] << 4) | (lv[h + 3] << 6));
        const uint8_t po = (uint8_t)(lvOld[h] | (lvOld[h + 1] << 2) | (lvOld[h + 2] << 4) | (lvOld[h + 3] << 6));
        if (pk != po) {
            Cb[gi >> 2] = pk;
            if (CR) CR[gi >> 2] |= (uint8_t)(pk ^ po);   // newly pruned nodes: range code 3 as well (byte owned by me)
        }
    }
    if (t >= 1 && t < 4 && lv[t] != lvOld[t]) {          // heap nodes 1..3: the block root and its children
        const int lq = 31 - __clz(t);
        const int64_t gi = ((int64_t)1 << (D - 12 + lq)) + (((int64_t)blockIdx.x) << lq) + (t - (1 << lq));
        cset3(Cb, gi);
        if (CR) cset3(CR, gi);
    }
}

// ---------------------------------------------------------------- convert ----
// Tokens owned by leaf rank r, in stream order: the live internal nodes whose first
// leaf is r (depth ascending), then the leaf token and its grown chain (R.cpp:655-704).
// A node is live iff it is the root or its parent was not pruned (pruning is closed
// downwards, so "some ancestor pruned" == "parent pruned").
struct Owned {
    unsigned long long spineBits; // 2 bits per token, first token in the low bits
    uint32_t leafBits;
    unsigned long long spineBitsR; // MidRangeTree: the range stream's codes for the same tokens
    uint32_t leafBitsR;
    int nSpine, nLeaf;
    int preDs;       // tokens owned by r that precede the depth-Ds node (index entry)
    int aliveAtDs;
    int finalErr;    // |recon - temp| of the leaf after branch growth (-1: leaf not visited)
    int zeroRun;     // the branch ended on an evaluated "keep" at the last level (Ctrl::zeroRun)
};

__device__ inline Owned owned_tokens(const uint8_t *__restrict__ Cb, const uint8_t *__restrict__ CbR,
                                     const uint8_t *__restrict__ Tb, const uint8_t *__restrict__ TbR,
                                     const uint8_t *__restrict__ Rl, const uint8_t *__restrict__ RlR, int D,
                                     int maxDepth, int tol, const uint8_t *dmap, const uint8_t *dmapR, int Ds,
                                     uint32_t r)
{
    Owned o;
    o.spineBits = 0; o.leafBits = 0; o.spineBitsR = 0; o.leafBitsR = 0;
    o.nSpine = 0; o.nLeaf = 0; o.preDs = 0; o.aliveAtDs = 0; o.finalErr = -1; o.zeroRun = 0;
    const int jmin = r ? D - (__ffs((int)r) - 1) : 0;
    bool alive = true;
    if (jmin > 0) alive = cget(Cb, ((int64_t)1 << (jmin - 1)) + (r >> (D - jmin + 1))) != 3;
    int j = jmin;
    for (; alive && j < D; ++j) {
        if (j == Ds) { o.preDs = o.nSpine; o.aliveAtDs = 1; }
        const int64_t ni = ((int64_t)1 << j) + (r >> (D - j));
        int code = cget(Cb, ni);
        o.spineBits |= (unsigned long long)code << (2 * o.nSpine);
        if (CbR) o.spineBitsR |= (unsigned long long)cget(CbR, ni) << (2 * o.nSpine);
        o.nSpine++;
        if (code == 3) alive = false;
    }
    if (alive) {
        if (Ds == D) { o.preDs = o.nSpine; o.aliveAtDs = 1; }
        const int64_t li = ((int64_t)1 << D) + r;
        int code = cget(Cb, li);
        o.leafBits = (uint32_t)code;
        if (CbR) o.leafBitsR = (uint32_t)cget(CbR, li);
        o.nLeaf = 1;
        int t = Tb[li], rec = Rl[r];
        int tR = 0, recR = 0;
        if (CbR) { tR = TbR[li]; recR = RlR[r]; }
        if (code != 3) {
            int depth = D;
            while (depth < maxDepth) {
                int err = rec > t ? rec - t : t - rec;
                if (err > tol) {                       // grow the branch (R.cpp:695-697, 657-660)
                    depth++;
                    Enc e = encode_node(t, rec, dmap[depth]);
                    rec = e.recon;
                    if (depth == maxDepth && e.code == 0) o.zeroRun = 1;
                    o.leafBits |= (uint32_t)e.code << (2 * o.nLeaf);
                    if (CbR) {                         // M.cpp: range stream re-encoded in lock-step
                        Enc er = encode_node(tR, recR, dmapR[depth]);
                        recR = er.recon;
                        o.leafBitsR |= (uint32_t)er.code << (2 * o.nLeaf);
                    }
                    o.nLeaf++;
                } else {                               // terminator (R.cpp:699-703, 671-674)
                    o.leafBits |= 3u << (2 * o.nLeaf);
                    if (CbR) o.leafBitsR |= 3u << (2 * o.nLeaf);
                    o.nLeaf++;
                    break;
                }
            }
        }
        o.finalErr = rec > t ? rec - t : t - rec;
    }
    return o;
}

struct EmitArgs {
    const uint8_t *codes, *codesR, *temp, *tempR;
    ReconBufs rb, rbR;
    Ctrl *ctrls, *ctrlsR;
    int64_t heapStride, leafStride, codeStride;
    int D, maxDepth, tol, Ds, K;
    uint32_t *blockTot, *blockOff;
    unsigned long long *blockOff64; // trees of more than 2^32 tokens (origTreeDepth > 28): the scan in 64 bits; index entries then
    unsigned long long *idxBase;    // stay relative to their 4096-leaf block and idxBase[block] holds the block's stream offset
    unsigned long long *blockL1;   // per-block sum |recon - temp| after growth (reduced by k_emit_stats)
    uint8_t *blockAlive, *blockVal; // k_block_alive: flags, scalar above the block
    unsigned long long *blockSpine; // k_block_alive: tokens above depth D-10 owned by the block's first rank
    unsigned long long *blockSpineR; // ... the same tokens of MidRangeTree's range stream
    int64_t nEmitBlk;
    uint8_t *tree, *treeR;
    int64_t treeCap;
    const unsigned long long *brickOff;   // compact_launch: brick b's stream starts at tree + brickOff[b] (else at tree + b * treeCap)
    int64_t compactCap;
    uint32_t *idxOff;
    uint8_t *idxVal;
    uint8_t *idxVal3;              // k_concat12: decoded scalars of the depth-(D-3) nodes (k_decode_quad)
    int64_t nIdx;
    const uint32_t *chainLut;
};

__device__ inline uint32_t block_excl_scan_u32(uint32_t v, uint32_t *shWave, uint32_t &total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan_add_dpp(v);
    if (lane == 63) shWave[w] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { if (i < w) woff += shWave[i]; tot += shWave[i]; }
    __syncthreads();
    total = tot;
    return woff + incl - v;
}

// the same for a block of exactly four waves: four LDS reads, no loop over a run-time wave count
__device__ __forceinline__ uint32_t block4_excl_scan_u32(uint32_t v, uint32_t *shWave, uint32_t &total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan_add_dpp(v);
    if (lane == 63) shWave[w] = incl;
    __syncthreads();
    const uint32_t s0 = shWave[0], s1 = shWave[1], s2 = shWave[2], s3 = shWave[3];
    __syncthreads();
    total = s0 + s1 + s2 + s3;
    const uint32_t woff = w == 0 ? 0u : (w == 1 ? s0 : (w == 2 ? s0 + s1 : s0 + s1 + s2));
    return woff + incl - v;
}

// ---- prune + block-local emit (VolumeKdtree streams with D >= 12: K = 6, Ds = D-6) --------------
// One block owns a depth-(D-12) subtree: 4096 leaves, 16 per thread.  It prunes the twelve levels
// bottom-up like k_prune12 (the four above the leaves in registers, the other eight in LDS), and then,
// top-down, writes the subtree's own preorder token string -- assuming its root is live -- into LDS
// and from there into the block's slot of the brick's stream buffer: slot b starts at word b * PE_WORDS
// (a block's string is at most 4095 + 8 * 4096 tokens = 9216 bytes).  That block-gapped buffer IS what the decoders
// read (k_index12 points the decode index at the slots); the reference's contiguous layout (R.cpp:631-724 writes one
// array) is produced from it by k_concat12 when somebody asks for the bytes -- vr_brickset_get_tree, save,
// get_packed4 -- not on the build -> levelCut path, where that copy was 4.5 ms of a 43 ms build.
#define PE_WORDS 2320
#define PE_TOKENS (PE_WORDS * 16)

struct PruneEmitArgs {
    int D, tol, maxDepth;
    Ctrl *ctrls;
    uint8_t *temp;                 // truths (leaf half, read) / staging (internal half, written)
    uint8_t *codes;
    int64_t heapStride, codeStride, leafStride;
    ReconBufs rb;
    uint32_t *subTok;              // tokens of the block's subtree (k_block_alive reads it from blockOff[])
    int64_t nEmitBlk;
    unsigned long long *blockL1;
    const uint32_t *chainLut;
    uint32_t *idxOff;              // block-local token offset of every depth-Ds node (k_concat12 makes it global)
    int64_t nIdx;
    uint32_t *fineIdx;             // 16 bytes per depth-Ds node: tokens owned by each of its 4-leaf subtrees (k_decode_fine)
    // MidRangeTree's second stream (k_prune_emit12<true>): its own truths, codes, reconstruction, control blocks
    Ctrl *ctrlsR;
    uint8_t *tempR, *codesR;
    ReconBufs rbR;
    uint8_t *gap, *gapR;           // the streams' block-gapped buffers (Stream2::tree)
    int64_t treeCap;
    SkipBlocks sk, skR;            // blocks the level loop left alone below depth D-2: all "keep", exactly reproduced (per stream)
};

__device__ __forceinline__ void pe_put(uint32_t *W, uint32_t bitpos, unsigned long long v, int ntok)
{   // v: at most 44 bits of tokens
    if (ntok <= 0) return;
    const uint32_t sh = bitpos & 31u, w = bitpos >> 5;
    const unsigned long long lo = v << sh;
    atomicOr(&W[w], (uint32_t)lo);
    if ((uint32_t)(lo >> 32)) atomicOr(&W[w + 1], (uint32_t)(lo >> 32));
    const uint32_t hi = sh ? (uint32_t)(v >> (64u - sh)) : 0u;
    if (hi) atomicOr(&W[w + 2], hi);
}

// RANGE (MidRangeTree, M.cpp:864-865, 871-982): the half-range stream has the mid stream's STRUCTURE -- a node is
// pruned, a branch grows and ends exactly where the mid stream's does -- and its own token VALUES: its level-loop codes,
// and along a branch encodeNode against its own reconstruction (M.cpp:930-933).  The second launch recomputes the mid
// stream's prune decisions and branch lengths from the mid arrays (same code, so the same result), swaps the range
// stream's values in, and writes the same string shape into the range stream's staging area.  Index, counts and
// statistics belong to the first launch.
template <bool RANGE, bool LEAFLESS>
__global__ void __launch_bounds__(256, RANGE ? 4 : 6)
k_prune_emit12(PruneEmitArgs a)
{
    __shared__ uint32_t lutS[512];                   // grown branch by signed initial error (k_chain_lut)
    __shared__ uint32_t W[PE_WORDS];
    __shared__ uint8_t codeH[256], codeOldH[256];    // the block's nodes of depths D-12 .. D-5, heap order (1 .. 255)
    __shared__ uint8_t flH[512];                     // "subtree is a single pruned token" flags; 256 + t = the depth-(D-4) nodes
    __shared__ uint16_t cntH[512];                   // tokens a live subtree emits
    __shared__ uint32_t shw[4];
    const int brick = blockIdx.y, t = threadIdx.x, D = a.D, tol = a.tol;
    Ctrl &c = a.ctrls[brick];
    const int cConst = c.constBrick, cPar = c.par, cRa = c.ra, cDistR = c.finalReconDist, cDistC = c.finalCodesDist;     // one scalar round trip
    if (cConst) return;
    const uint32_t blk = blockIdx.x, base = blk << 12;
    uint8_t *Cb = a.codes + (int64_t)brick * a.codeStride;
    // ---- every global load of the block, in one batch
    const uint32_t lutV = a.chainLut[VR_CHAIN_LUT_SIGNED + t], lutV2 = a.chainLut[VR_CHAIN_LUT_SIGNED + 256 + t];
    const int hU = t ? t : 1, lqU = 31 - __clz(hU);
    const int64_t niU = ((int64_t)1 << (D - 12 + lqU)) + ((int64_t)blk << lqU) + (hU - (1 << lqU));
    const uint32_t upB = Cb[niU >> 2];
    const int64_t n4 = ((int64_t)1 << (D - 4)) + (base >> 4) + t, n3 = ((int64_t)1 << (D - 3)) + (base >> 3) + 2 * t;
    const int64_t n2 = ((int64_t)1 << (D - 2)) + (base >> 2) + 4 * t, n1 = ((int64_t)1 << (D - 1)) + (base >> 1) + 8 * t;
    const uint32_t c4B = Cb[n4 >> 2], c3B = Cb[n3 >> 2], c2B = Cb[n2 >> 2];
    const int64_t li = ((int64_t)1 << D) + base + t * 16;
    // a block the level loop skipped (SkipBlocks) has no codes and no reconstruction in memory at depths D-1 and D:
    // they are all "keep" and exact, which is what zeros here say
    const bool skipB = a.sk.flag && (a.sk.flag[(int64_t)brick * a.sk.nBlk + blk] & 2u) != 0u;
    uint32_t c1H = 0, cpk = 0;
    uint4 tv = make_uint4(0, 0, 0, 0), rv = make_uint4(0, 0, 0, 0);
    uint2 pv = make_uint2(0, 0), pvR = make_uint2(0, 0);      // leafless: my leaves' eight parents (the level loop's last parents buffer but one)
    if (!skipB) {
        c1H = *(const uint16_t *)(Cb + (n1 >> 2));
        tv = *(const uint4 *)(a.temp + (int64_t)brick * a.heapStride + li);
        if (LEAFLESS)
            pv = *(const uint2 *)((cRa == 0 ? a.rb.b[0] : (cRa == 1 ? a.rb.b[1] : a.rb.b[2])) + (int64_t)brick * a.leafStride + (base >> 1) + t * 8);
        else {
            cpk = *(const uint32_t *)(Cb + (li >> 2));       // my 16 leaf codes, packed
            rv = *(const uint4 *)((cPar == 0 ? a.rb.b[0] : (cPar == 1 ? a.rb.b[1] : a.rb.b[2])) + (int64_t)brick * a.leafStride + base + t * 16);
        }
    }
    // RANGE: the same pieces of the range stream
    uint8_t *CbR = RANGE ? a.codesR + (int64_t)brick * a.codeStride : nullptr;
    uint32_t upBR = 0, c4BR = 0, c3BR = 0, c2BR = 0, c1HR = 0, cpkR = 0;
    uint4 tvR = make_uint4(0, 0, 0, 0), rvR = make_uint4(0, 0, 0, 0);
    int cDistRR = 0, cDistCR = 0;
    bool skipBR = false;
    if (RANGE) {
        const Ctrl &cr = a.ctrlsR[brick];
        const int cParR = cr.par, cRaR = cr.ra;
        cDistRR = cr.finalReconDist; cDistCR = cr.finalCodesDist;
        upBR = CbR[niU >> 2]; c4BR = CbR[n4 >> 2]; c3BR = CbR[n3 >> 2]; c2BR = CbR[n2 >> 2];
        // (a block the range stream's level loop skipped: all "keep" and exact there, like the mid stream's above)
        skipBR = a.skR.flag && (a.skR.flag[(int64_t)brick * a.skR.nBlk + blk] & 2u) != 0u;
        if (!skipBR) {
            c1HR = *(const uint16_t *)(CbR + (n1 >> 2));
            tvR = *(const uint4 *)(a.tempR + (int64_t)brick * a.heapStride + li);
            if (LEAFLESS)
                pvR = *(const uint2 *)((cRaR == 0 ? a.rbR.b[0] : (cRaR == 1 ? a.rbR.b[1] : a.rbR.b[2])) + (int64_t)brick * a.leafStride + (base >> 1) + t * 8);
            else {
                cpkR = *(const uint32_t *)(CbR + (li >> 2));
                rvR = *(const uint4 *)((cParR == 0 ? a.rbR.b[0] : (cParR == 1 ? a.rbR.b[1] : a.rbR.b[2])) + (int64_t)brick * a.leafStride + base + t * 16);
            }
        }
    }
    // leafless: the leaf stage below encodes every sibling pair itself; all that is needed up here is "every truth equals
    // its parent's reconstruction" (then all codes are "keep" and the leaves exact: what cpk == 0, tv == rv say)
    if (LEAFLESS) {
        rv = tv;
        cpk = ((tv.x ^ __builtin_amdgcn_perm(0, pv.x, 0x01010000u)) | (tv.y ^ __builtin_amdgcn_perm(0, pv.x, 0x03030202u)) |
               (tv.z ^ __builtin_amdgcn_perm(0, pv.y, 0x01010000u)) | (tv.w ^ __builtin_amdgcn_perm(0, pv.y, 0x03030202u))) != 0u ? 1u : 0u;
    }
    lutS[t] = lutV; lutS[256 + t] = lutV2;
    // ---- a block of 4096 exactly reproduced leaves under all-"keep" codes (constant regions) is one pruned subtree:
    // its string is the single token 3, its index entries are "pruned", its statistics zero.  Only the root's code
    // is written back: nothing reads the codes below a pruned node for anything but their (equal) scalars.
    const bool plain = tol >= 1 && cpk == 0u && tv.x == rv.x && tv.y == rv.y && tv.z == rv.z && tv.w == rv.w;
    // (RANGE: the first launch has already turned such a block's root into a 3)
    const uint32_t upC = (upB >> ((int)(niU & 3) * 2)) & 3u;
    const uint32_t myCodes = ((c4B >> ((int)(n4 & 3) * 2)) & 3u) | ((c3B >> ((int)(n3 & 3) * 2)) & 15u) | c2B | c1H |
                             ((RANGE && hU == 1 && upC == 3u) ? 0u : upC);
    if (__syncthreads_and(plain && myCodes == 0u)) {
        if (RANGE) {        // the range stream's copy of that one token
            if (t == 0) {
                cset3(CbR, ((int64_t)1 << (D - 12)) + blk);
                *((uint32_t *)(a.gapR + (int64_t)brick * a.treeCap) + (size_t)blk * PE_WORDS) = 3u;
            }
            return;
        }
        if (t == 0) {
            cset3(Cb, ((int64_t)1 << (D - 12)) + blk);
            a.subTok[(int64_t)brick * a.nEmitBlk + blk] = 1;
            *((uint32_t *)(a.gap + (int64_t)brick * a.treeCap) + (size_t)blk * PE_WORDS) = 3u;
        }
        if (t < 4) a.blockL1[(int64_t)brick * a.nEmitBlk + (size_t)blk * 4 + t] = stat_pack(0, 0, 0);
        if (t < 64) a.idxOff[(int64_t)brick * a.nIdx + (base >> 6) + t] = VR_IDX_DEAD;
        return;
    }
    // (the string buffer and the upper codes: only blocks that get here need them -- two thirds leave above -- and
    // barriers lie between these stores and their first readers)
    { const uint8_t cu = (uint8_t)((upB >> ((int)(niU & 3) * 2)) & 3u); codeH[t] = cu; codeOldH[t] = cu; }
    for (int i = t; i < PE_WORDS; i += 256) W[i] = 0;
    // ---- leaves: prune (R.cpp:618-626), grown branches (R.cpp:655-704); sibling leaves share packed 16-bit lanes
    const uint32_t tw[4] = {tv.x, tv.y, tv.z, tv.w}, rw[4] = {rv.x, rv.y, rv.z, rv.w}, pwL[2] = {pv.x, pv.y};
    const uint32_t dR2 = (uint32_t)cDistR * 0x10001u, dC2 = (uint32_t)cDistC * 0x10001u;
    const uint32_t tol2 = (uint32_t)tol * 0x10001u;
    uint32_t nt[8], Lb[8];      // per lane: tokens the leaf emits when live; its code + grown branch << 2
    uint32_t bothMask = 0;
    vr_s16x2 mxB = (vr_s16x2)(0), mxA = (vr_s16x2)(0), l1p = (vr_s16x2)(0);
    const int nsteps = a.maxDepth - D;    // distanceMap[D+1..] = 64, 32, .., 1 (R.cpp:23,94-97)
    // a wave whose 1024 leaves all carry code 0 and are reproduced exactly (constant regions) prunes them all:
    // one '3' per leaf, no branches, no statistics to add
#ifdef PE_KO_LEAF
    const bool busy = false;
#else
    const bool busy = __ballot(!plain) != 0ull;
#endif
    if (!busy) {
        bothMask = 0x5555u;
#pragma unroll
        for (int j = 0; j < 8; ++j) { nt[j] = 0x00010001u; Lb[j] = 0x00030003u; }
    }
    if (busy)
#pragma unroll
    for (int half = 0; half < 2; ++half) {          // two halves of four sibling pairs: halves the transient registers
        uint32_t T2[4], m[4], sg[4], act[4];
        uint32_t anyAct = 0, keepEnd = 0;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = half * 4 + jj;
            const uint32_t sel = (j & 1) ? 0x0c030c02u : 0x0c010c00u;
            T2[jj] = __builtin_amdgcn_perm(0, tw[j >> 1], sel);
            vr_s16x2 dl;
            uint32_t cl2;                                                                           // the pair's codes
            if (LEAFLESS) leaf_pair_encode(tw[j >> 1], j & 1, pwL[j >> 2], j & 3, dR2, dC2, dl, cl2);
            else {
                dl = pk_s(T2[jj]) - pk_s(__builtin_amdgcn_perm(0, rw[j >> 1], sel));
                cl2 = ((cpk >> (4 * j)) & 3u) | (((cpk >> (4 * j + 2)) & 3u) << 16);
            }
            const vr_s16x2 mm = pk_abs(dl);
            m[jj] = pk_u(mm);
            sg[jj] = pk_u(dl >> 15);
            mxB = __builtin_elementwise_max(mxB, mm);
            const uint32_t lt = pk_u((mm - pk_s(tol2)) >> 15);                                     // err < tol
            const uint32_t isz = pk_u((pk_s(cl2) - pk_s(0x00010001u)) >> 15), is3 = pk_u((pk_s(0x00020002u) - pk_s(cl2)) >> 15);
            const uint32_t newp = isz & lt;
            const uint32_t pruned = newp | is3;
            const uint32_t lcode = cl2 | (newp & 0x00030003u);
            bothMask |= ((pruned & (pruned >> 16)) & 1u) << (2 * j);     // at the bit of the pair node's code (below)
            const vr_s16x2 lim = __builtin_elementwise_min(pk_s(T2[jj]), pk_s(T2[jj] ^ 0x00FF00FFu));
            const uint32_t viol = pk_u((lim - mm) >> 15);                                          // a clamp could matter
            const uint32_t e0 = lutS[pk_u(dl) & 511u], e1 = lutS[(pk_u(dl) >> 16) & 511u];         // keyed by the signed error
            const uint32_t useL = ~pruned & ~viol;
            const uint32_t ch2 = __builtin_amdgcn_perm(e1, e0, 0x07060302u);                           // the two token strings
            Lb[j] = lcode | ((useL & ch2) << 2);
            nt[j] = 0x00010001u + (useL & __builtin_amdgcn_perm(e1, e0, 0x0c050c01u));            // the two counts
            m[jj] = (useL & __builtin_amdgcn_perm(e1, e0, 0x0c040c00u)) | (~useL & m[jj]);         // the two final errors
            act[jj] = ~pruned & viol;
            anyAct |= act[jj];
        }
#ifdef PE_KO_STEP
        for (int i = 0; i < 0; ++i) {
#else
        for (int i = 0; i < nsteps; ++i) {    // exact stepping for the leaves the table does not cover
#endif
            if (__ballot(anyAct != 0) == 0ull) break;
            const uint32_t d2 = (uint32_t)(64 >> i) * 0x10001u;
            anyAct = 0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = half * 4 + jj;
                if (__ballot(act[jj] != 0u) == 0ull) continue;     // such leaves are rare: usually one pair slot of the wave at most
                const vr_s16x2 mm = pk_s(m[jj]);
                const uint32_t gt = pk_u((pk_s(tol2) - mm) >> 15);
                nt[j] = pk_u(pk_s(nt[j]) - pk_s(act[jj]));
                const uint32_t go = act[jj] & gt;
                const uint32_t term = (act[jj] ^ go) & 0x00030003u;                                 // R.cpp:699-703
                const uint32_t lim = (sg[jj] & T2[jj]) | (~sg[jj] & (T2[jj] ^ 0x00FF00FFu));
                const vr_s16x2 x = __builtin_elementwise_min(pk_s(d2) - mm, pk_s(lim));
                const vr_s16x2 nx = (vr_s16x2)(0) - x, ax = __builtin_elementwise_max(x, nx);
                const uint32_t take = go & pk_u((ax - mm) >> 15);
                const uint32_t dir = pk_u(pk_s(0x00010001u) - pk_s(sg[jj]));                        // add = 1, sub = 2
                Lb[j] |= ((take & dir) | term) << (2 * i + 2);
                // an evaluated node that keeps leaves its error (> tol) unchanged, so the branch goes on unless this
                // was the last level: only there can a branch end on a "keep" (zero-run rewrite, see Ctrl::zeroRun)
                if (i == nsteps - 1) keepEnd |= go & ~take;
                m[jj] = (take & pk_u(ax)) | (~take & m[jj]);
                sg[jj] ^= take & pk_u(nx >> 15);
                act[jj] = go;
                anyAct |= go;
            }
        }
        if (!RANGE && keepEnd) atomicAdd(&c.zeroRun, (int)((keepEnd & 1u) + ((keepEnd >> 16) & 1u)));
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { mxA = __builtin_elementwise_max(mxA, pk_s(m[jj])); l1p += pk_s(m[jj]); }
    }
    if (!RANGE) {   // statistics (R.cpp:71-76, 115-129): one record per wave = 1024 leaves
        const int maxErr = wave_max_i32_dpp(max((int)mxB.x, (int)mxB.y)), maxAfter = wave_max_i32_dpp(max((int)mxA.x, (int)mxA.y));
        const unsigned long long l1w = (uint32_t)__builtin_amdgcn_readlane(
            (int)wave_incl_scan_add_dpp((uint32_t)((int)l1p.x + (int)l1p.y)), 63);       // < 2^18 per wave
        if ((t & 63) == 0)
            a.blockL1[(int64_t)brick * a.nEmitBlk + (size_t)blk * 4 + (t >> 6)] = stat_pack(l1w, maxErr, maxAfter);
    }
    // ---- RANGE: the leaves' tokens in the range stream.  A pruned leaf is a 3 there too (M.cpp:864-865); a live one
    // carries its own level-loop code, then as many branch nodes as the mid stream's branch has, each encodeNode
    // against the range stream's own reconstruction with the same distances (M.cpp:930-933), then the mid stream's
    // terminator if it has one
    uint32_t LbR[8];
    if (RANGE) {
        // sibling leaves in packed 16-bit lanes, like the mid stream's: per branch level one reduced encodeNode
        // (kd_common.h) for both, applied where the mid stream's branch still has an evaluated node at that level;
        // a wave leaves the level loop as soon as none of its branches is that long
        const uint32_t twR[4] = {tvR.x, tvR.y, tvR.z, tvR.w}, rwR[4] = {rvR.x, rvR.y, rvR.z, rvR.w}, pwR[2] = {pvR.x, pvR.y};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (!busy) { LbR[j] = 0x00030003u; continue; }
            const uint32_t sel = (j & 1) ? 0x0c030c02u : 0x0c010c00u;
            const vr_s16x2 T2 = pk_s(__builtin_amdgcn_perm(0, twR[j >> 1], sel));
            vr_s16x2 rec = pk_s(__builtin_amdgcn_perm(0, rwR[j >> 1], sel));
            uint32_t cR = ((cpkR >> (4 * j)) & 3u) | (((cpkR >> (4 * j + 2)) & 3u) << 16);
            if (LEAFLESS) {       // the range stream's own leaf level, recomputed like the mid stream's
                vr_s16x2 dlR;
                leaf_pair_encode(twR[j >> 1], j & 1, pwR[j >> 2], j & 3, (uint32_t)cDistRR * 0x10001u, (uint32_t)cDistCR * 0x10001u, dlR, cR);
                rec = T2 - dlR;
            }
            const uint32_t mid = Lb[j];
            const vr_s16x2 n2 = pk_s(nt[j]) - pk_s(0x00010001u);                        // branch tokens of the mid stream
            const uint32_t prunedM = ((mid & (mid >> 1)) & 0x00010001u) * 0xFFFFu;          // the leaf's own token is a 3
            vr_u16x2 lastTok = __builtin_bit_cast(vr_u16x2, mid) >> __builtin_bit_cast(vr_u16x2, n2 + n2);   // token n of each string
            const uint32_t lt = __builtin_bit_cast(uint32_t, lastTok);
            const uint32_t hasN = pk_u((vr_s16x2)(0) - n2) >> 15 & 0x00010001u;            // n > 0
            const uint32_t termB = (lt & (lt >> 1)) & hasN;                                 // bit 0 of each lane: has a terminator
            const vr_s16x2 k2 = n2 - pk_s(termB);                                           // evaluated branch nodes
            uint32_t bits = (prunedM & 0x00030003u) | (~prunedM & cR);
            for (int i = 0; i < VR_CHAIN_LEVELS; ++i) {
                const uint32_t am = pk_u((pk_s((uint32_t)i * 0x10001u) - k2) >> 15);       // 0xFFFF where k > i
                if (__ballot(am != 0u) == 0ull) break;
                const uint32_t d2 = (uint32_t)(64 >> i) * 0x10001u;
                const vr_s16x2 diff = T2 - rec, nd = (vr_s16x2)(0) - diff;
                const vr_s16x2 pd = __builtin_elementwise_max(diff, nd);
                const uint32_t up = pk_u(nd >> 15);
                const vr_s16x2 h = pk_s(pk_u(T2) ^ (up & 0x00FF00FFu));
                const vr_s16x2 x = __builtin_elementwise_min(pk_s(d2) - pd, h), ax = pk_abs(x);
                const uint32_t take = pk_u((ax - pd) >> 15) & am;
                const uint32_t code2 = take & pk_u(pk_s(up) + pk_s(0x00020002u)) & 0x00030003u;    // up ? 1 : 2
                const vr_s16x2 r = pk_mad(x, pk_mad(pk_s(up), pk_s(0xFFFEFFFEu), pk_s(0xFFFFFFFFu)), T2);   // up ? t + x : t - x
                rec = pk_s((take & pk_u(r)) | (~take & pk_u(rec)));
                bits |= code2 << (2 * i + 2);
            }
            const vr_u16x2 term = __builtin_bit_cast(vr_u16x2, termB * 3u) << __builtin_bit_cast(vr_u16x2, k2 + k2 + pk_s(0x00020002u));
            LbR[j] = bits | __builtin_bit_cast(uint32_t, term);
        }
    }
    // ---- depths D-1 .. D-4 of my 16 leaves, in registers (R.cpp:596-629: a node only depends on its children), all
    // nodes of a level at once: the eight depth-(D-1) codes sit at bits 2k, the four depth-(D-2) codes at bits 4q, the
    // two depth-(D-3) codes at bits 8r, the depth-(D-4) code at bit 0 -- a node's children are the codes at its own bit
    // position and at the next position of the finer level, so "both children are pruned tokens" is one AND of shifts
    uint32_t a1 = c1H;
    uint32_t a2 = c2B; a2 = (a2 | (a2 << 4)) & 0x0F0Fu; a2 = (a2 | (a2 << 2)) & 0x3333u;
    uint32_t a3 = (c3B >> ((int)(n3 & 3) * 2)) & 15u; a3 = (a3 | (a3 << 6)) & 0x0303u;
    uint32_t a4 = (c4B >> ((int)(n4 & 3) * 2)) & 3u;
    const auto prune_level = [](uint32_t &codes, uint32_t bothChildren, uint32_t at) -> uint32_t {
        const uint32_t keep = ~(codes | (codes >> 1)) & at;          // code == 0
        const uint32_t p = keep & bothChildren;                       // ... and both children pruned: becomes a 3
        codes |= p | (p << 1);
        return codes & (codes >> 1) & at;                             // the level's pruned tokens
    };
    const uint32_t F1 = prune_level(a1, bothMask, 0x5555u);
    const uint32_t F2 = prune_level(a2, F1 & (F1 >> 2), 0x1111u);
    const uint32_t F3 = prune_level(a3, F2 & (F2 >> 4), 0x0101u);
    const uint32_t F4 = prune_level(a4, F3 & (F3 >> 8), 0x0001u);
    // tokens a subtree emits: 1 if its root is a pruned token, else 1 + its children's
    int cnt1[8], cnt2[4], cnt3[2];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        cnt1[k] = 1 + (((int)(nt[k] & 0xFFFFu) + (int)(nt[k] >> 16)) & ~__builtin_amdgcn_sbfe((int)F1, 2 * k, 1));
#pragma unroll
    for (int q = 0; q < 4; ++q) cnt2[q] = 1 + ((cnt1[2 * q] + cnt1[2 * q + 1]) & ~__builtin_amdgcn_sbfe((int)F2, 4 * q, 1));
#pragma unroll
    for (int r = 0; r < 2; ++r) cnt3[r] = 1 + ((cnt2[2 * r] + cnt2[2 * r + 1]) & ~__builtin_amdgcn_sbfe((int)F3, 8 * r, 1));
    const int cnt4 = 1 + ((cnt3[0] + cnt3[1]) & ~__builtin_amdgcn_sbfe((int)F4, 0, 1));
    flH[256 + t] = (uint8_t)F4;
    cntH[256 + t] = (uint16_t)cnt4;
    __syncthreads();
    // ---- depths D-5 .. D-12 in LDS
    for (int lq = 7; lq >= 0; --lq) {
        if (t < (1 << lq)) {
            const int h = (1 << lq) + t;
            const uint32_t code = codeH[h];
            const bool p = flH[2 * h] && flH[2 * h + 1] && code == 0;
            if (p) codeH[h] = 3;
            const bool f = p || code == 3;
            flH[h] = (uint8_t)(f ? 1 : 0);
            cntH[h] = (uint16_t)(f ? 1u : 1u + cntH[2 * h] + cntH[2 * h + 1]);
        }
        __syncthreads();
    }
    // RANGE: from here on the token VALUES are the range stream's: its own codes, 3 wherever the mid stream pruned
    // (M.cpp:864-865); the shape (which tokens exist) is unchanged because a token is 3 in both streams or in neither
    if (RANGE) {
        const auto merge = [](uint32_t mid, uint32_t rng) { const uint32_t m3 = (mid & (mid >> 1) & 0x55555555u) * 3u; return (rng & ~m3) | m3; };
        uint32_t r2 = c2BR; r2 = (r2 | (r2 << 4)) & 0x0F0Fu; r2 = (r2 | (r2 << 2)) & 0x3333u;      // spread like a2, a3
        uint32_t r3 = (c3BR >> ((int)(n3 & 3) * 2)) & 15u; r3 = (r3 | (r3 << 6)) & 0x0303u;
        a1 = merge(a1, c1HR) & 0xFFFFu;
        a2 = merge(a2, r2) & 0x3333u;
        a3 = merge(a3, r3) & 0x0303u;
        a4 = merge(a4, (c4BR >> ((int)(n4 & 3) * 2)) & 3u) & 3u;
        const uint8_t ro = (uint8_t)((upBR >> ((int)(niU & 3) * 2)) & 3u);
        const uint8_t mc = codeH[t];
        __syncthreads();
        codeOldH[t] = ro;
        codeH[t] = mc == 3 ? (uint8_t)3 : ro;
#pragma unroll
        for (int j = 0; j < 8; ++j) Lb[j] = LbR[j];
        __syncthreads();
    }
    uint8_t *Cw = RANGE ? CbR : Cb;
    // the BFS codes are read again only down to depth Ds = D-6 (upper prune levels, k_block_alive, k_concat12's
    // index values, progressive cuts): write back depths D-12 .. D-5; below that they are dead from here on
    if (t >= 1 && t < 64) {                               // heap bytes 1..63 <-> heap nodes 4..255 (whole bytes are mine)
        const int h = t * 4, lq = 31 - __clz(h);
        const uint8_t pk = (uint8_t)(codeH[h] | (codeH[h + 1] << 2) | (codeH[h + 2] << 4) | (codeH[h + 3] << 6));
        const uint8_t po = (uint8_t)(codeOldH[h] | (codeOldH[h + 1] << 2) | (codeOldH[h + 2] << 4) | (codeOldH[h + 3] << 6));
        if (pk != po) Cw[(((int64_t)1 << (D - 12 + lq)) + ((int64_t)blk << lq) + (h - (1 << lq))) >> 2] = pk;
    }
    if (t >= 1 && t < 4 && codeH[t] != codeOldH[t]) {     // heap nodes 1..3 share bytes with neighbouring blocks
        const int lq = 31 - __clz(t);
        cset3(Cw, ((int64_t)1 << (D - 12 + lq)) + ((int64_t)blk << lq) + (t - (1 << lq)));
    }
    if (!RANGE && t == 0) a.subTok[(int64_t)brick * a.nEmitBlk + blk] = cntH[1];
    // ---- top-down: the block's own token string, root assumed live.  Thread t owns the live nodes whose
    // first leaf is its first leaf: ancestors at in-block level lq (depth D-12+lq) when t % 2^(8-lq) == 0
    const int jmin = t ? 8 - (__ffs(t) - 1) : 0;
    bool alive = true;
    if (jmin > 0) alive = codeH[(1 << (jmin - 1)) + (t >> (9 - jmin))] != 3;       // parent not pruned
    uint32_t S = 0;
    int ns = 0, preDs = 0, aliveAtDs = 0;
    for (int lq = jmin; lq <= 7 && alive; ++lq) {
        if (lq == 6) { preDs = ns; aliveAtDs = 1; }
        const uint32_t code = codeH[(1 << lq) + (t >> (8 - lq))];
        S |= code << (2 * ns);
        ++ns;
        if (code == 3) alive = false;
    }
    uint32_t tot;
    const uint32_t pos = block4_excl_scan_u32((uint32_t)(ns + (alive ? cnt4 : 0)), shw, tot);
    uint32_t bitpos = 2u * pos;
    pe_put(W, bitpos, (unsigned long long)S, ns);
    bitpos += 2u * (uint32_t)ns;
#ifdef PE_KO_COMPOSE
    if (false) {
#else
    if (alive) {
#endif
        // preorder of my 31-node subtree; internal tokens wait in (pb, pn) and leave with the next leaf-pair piece
        unsigned long long pb = a4;
        int pn = 1;
        if (a4 != 3u) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t c3 = (a3 >> (8 * r)) & 3u;
                pb |= (unsigned long long)c3 << (2 * pn); ++pn;
                if (c3 == 3u) continue;
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const int q = 2 * r + qq;
                    const uint32_t c2 = (a2 >> (4 * q)) & 3u;
                    pb |= (unsigned long long)c2 << (2 * pn); ++pn;
                    if (c2 == 3u) continue;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const int k = 2 * q + kk;
                        const uint32_t c1 = (a1 >> (2 * k)) & 3u;
                        unsigned long long H = c1;
                        int Hn = 1;
                        if (c1 != 3u) {
                            const int n0 = (int)(nt[k] & 0xFFFFu), n1t = (int)(nt[k] >> 16);
                            H |= ((unsigned long long)(Lb[k] & 0xFFFFu) << 2) | ((unsigned long long)(Lb[k] >> 16) << (2 + 2 * n0));
                            Hn = 1 + n0 + n1t;
                        }
                        pe_put(W, bitpos, pb | (H << (2 * pn)), pn + Hn);
                        bitpos += 2u * (uint32_t)(pn + Hn);
                        pb = 0; pn = 0;
                    }
                }
            }
        }
        pe_put(W, bitpos, pb, pn);
    }
    if (!RANGE && (t & 3) == 0)       // decode side-car index: the depth-Ds node of every 64 leaves, block-local for now
        a.idxOff[(int64_t)brick * a.nIdx + (base >> 6) + (t >> 2)] = aliveAtDs ? pos + (uint32_t)preDs : VR_IDX_DEAD;
    if (!RANGE) {
        // ... and, below it, how many tokens each 4-leaf subtree owns in preorder: its own nodes plus the
        // ancestors (depth >= Ds) whose first leaf is its first leaf.  One byte each (at most 4 + 39 tokens); the
        // decoder turns them into token offsets with a 16-lane prefix sum, so one lane decodes four voxels.
        const int nsIn = jmin <= 6 ? (aliveAtDs ? ns - preDs : 0) : ns;
        const bool l4 = alive && a4 != 3u;
        const bool l30 = l4 && (a3 & 3u) != 3u, l31 = l4 && ((a3 >> 8) & 3u) != 3u;
        const uint32_t f0 = (uint32_t)(nsIn + (alive ? 1 + (l4 ? 1 + (l30 ? cnt2[0] : 0) : 0) : 0));
        const uint32_t f1 = (uint32_t)(l30 ? cnt2[1] : 0);
        const uint32_t f2 = (uint32_t)(l4 ? 1 + (l31 ? cnt2[2] : 0) : 0);
        const uint32_t f3 = (uint32_t)(l31 ? cnt2[3] : 0);
        a.fineIdx[((int64_t)brick * a.nIdx + (base >> 6)) * 4 + t] = f0 | (f1 << 8) | (f2 << 16) | (f3 << 24);
    }
    __syncthreads();
    // the block's slot of the gapped stream buffer; one word past the string too, so that no stale token of an earlier
    // build sits right behind it (the decoders fetch whole words past a run's end; what they fetch there is never used)
    const uint32_t nw = min((tot + 15u) / 16u + 1u, (uint32_t)PE_WORDS);
    uint32_t *slot = (uint32_t *)((RANGE ? a.gapR : a.gap) + (int64_t)brick * a.treeCap) + (size_t)blk * PE_WORDS;
#ifndef PE_KO_COPY
    for (uint32_t i = t; i < nw; i += 256u) slot[i] = W[i];
#else
    if (t == 0) slot[0] = W[0] + nw;
#endif
}


__global__ void __launch_bounds__(EMIT_RANKS_PER_BLOCK)
k_emit_count(EmitArgs a)
{
    __shared__ uint32_t shw[4];
    const int brick = blockIdx.y;
    const Ctrl &c = a.ctrls[brick];
    if (c.constBrick) return;
    const uint8_t *Cb = a.codes + (int64_t)brick * a.codeStride;
    const uint8_t *Tb = a.temp + (int64_t)brick * a.heapStride;
    const uint8_t *Rl = a.rb.b[c.par] + (int64_t)brick * a.leafStride;
    const uint32_t n = 1u << a.D;
    uint32_t r = blockIdx.x * EMIT_RANKS_PER_BLOCK + threadIdx.x;
    uint32_t w = 0;
    if (r < n) {
        Owned o = owned_tokens(Cb, nullptr, Tb, nullptr, Rl, nullptr, a.D, a.maxDepth, a.tol, c.distanceMap, nullptr,
                               a.Ds, r);
        w = o.nSpine + o.nLeaf;
    }
    uint32_t tot;
    block_excl_scan_u32(w, shw, tot);
    if (threadIdx.x == 0) a.blockTot[(int64_t)brick * a.nEmitBlk + blockIdx.x] = tot;
}

// exclusive scan of the per-block token counts of one brick (one block per brick)
__global__ void __launch_bounds__(1024)
k_emit_scan(EmitArgs a, int64_t nblk)
{
    __shared__ uint32_t shw[16];
    __shared__ unsigned long long carrySh;
    const int brick = blockIdx.x;
    if (a.ctrls[brick].constBrick) return;
    const uint32_t *in = a.blockTot + (int64_t)brick * a.nEmitBlk;
    uint32_t *out = a.blockOff + (int64_t)brick * a.nEmitBlk;
    unsigned long long *out64 = a.blockOff64 ? a.blockOff64 + (int64_t)brick * a.nEmitBlk : nullptr;
    if (threadIdx.x == 0) carrySh = 0;
    __syncthreads();
    for (int64_t base = 0; base < nblk; base += 1024) {
        int64_t i = base + threadIdx.x;
        uint32_t v = i < nblk ? in[i] : 0;
        uint32_t tot;
        uint32_t ex = block_excl_scan_u32(v, shw, tot);        // 1024 blocks of at most 9 * 4096 tokens: fits 32 bits
        unsigned long long carry = carrySh;
        if (i < nblk) { out[i] = (uint32_t)(carry + ex); if (out64) out64[i] = carry + ex; }
        __syncthreads();
        if (threadIdx.x == 0) carrySh = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.ctrls[brick].numActive = carrySh;     // numActiveNodes (R.cpp:714)
        if (a.ctrlsR) a.ctrlsR[brick].numActive = carrySh;
    }
}

// Words that contain a block boundary are merged with atomicOr; they are zeroed here
// first (every shared word is the first or last word of some block's token range).
__global__ void __launch_bounds__(256)
k_emit_zero(EmitArgs a, int64_t nblk)
{
    const int brick = blockIdx.y;
    int64_t blk = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (blk >= nblk || a.ctrls[brick].constBrick) return;
    if (a.brickOff && a.brickOff[gridDim.y] > (unsigned long long)a.compactCap) return;       // (did not fit: nothing is written)
    const unsigned long long g0 = a.blockOff64 ? a.blockOff64[(int64_t)brick * a.nEmitBlk + blk] : a.blockOff[(int64_t)brick * a.nEmitBlk + blk];
    uint32_t tot = a.blockTot[(int64_t)brick * a.nEmitBlk + blk];
    const int64_t base = a.brickOff ? (int64_t)a.brickOff[brick] : (int64_t)brick * a.treeCap;
    uint32_t *W = (uint32_t *)(a.tree + base);
    uint32_t *WR = a.treeR ? (uint32_t *)(a.treeR + base) : nullptr;
    if (tot == 0) {
        if (blk == 0 && a.ctrls[brick].numActive == 0) { W[0] = 0; if (WR) WR[0] = 0; }
        return;
    }
    W[g0 >> 4] = 0;
    W[(g0 + tot - 1) >> 4] = 0;
    if (WR) { WR[g0 >> 4] = 0; WR[(g0 + tot - 1) >> 4] = 0; }
}

// byte offset of every brick's contiguous stream in the compact buffers (16-byte aligned, one spare word each) and
// their total; the streams of all bricks back to back take what they take, not B times the worst case
__global__ void __launch_bounds__(1024)
k_brick_offsets(const Ctrl *ctrls, int B, unsigned long long *brickOff, int64_t cap, int32_t *overflow)
{
    __shared__ unsigned long long part[16];
    __shared__ unsigned long long carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < B; base += 1024) {
        const int b = base + t;
        unsigned long long bytes = 0;
        if (b < B) bytes = ((((ctrls[b].numActive + 15ull) >> 4) + 1ull) * 4ull + 15ull) & ~15ull;
        unsigned long long incl = bytes;
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long n = __shfl_up(incl, o); if ((t & 63) >= o) incl += n; }
        if ((t & 63) == 63) part[t >> 6] = incl;
        __syncthreads();
        unsigned long long before = carry;
        for (int i = 0; i < (t >> 6); ++i) before += part[i];
        if (b < B) brickOff[b] = before + incl - bytes;
        __syncthreads();
        if (t == 1023) carry = before + incl;
        __syncthreads();
    }
    if (t == 0) { brickOff[B] = carry; *overflow = carry > (unsigned long long)cap ? 1 : 0; }
}

#define EMIT_LDS_WORDS 256   // 4096 tokens >= 256*9 + 511 + 32 (+15 phase)

__device__ inline void lds_put(uint32_t *W, uint32_t pos, uint32_t bits, int ntok)
{
    if (ntok <= 0) return;
    unsigned long long v = (unsigned long long)bits << ((pos & 15u) * 2u);
    uint32_t w = pos >> 4;
    atomicOr(&W[w], (uint32_t)v);
    uint32_t hi = (uint32_t)(v >> 32);
    if (hi) atomicOr(&W[w + 1], hi);
}

__global__ void __launch_bounds__(EMIT_RANKS_PER_BLOCK)
k_emit_write(EmitArgs a)
{
    __shared__ uint32_t shw[4];
    __shared__ uint32_t W[EMIT_LDS_WORDS], WR[EMIT_LDS_WORDS];
    const int brick = blockIdx.y;
    Ctrl &c = a.ctrls[brick];
    if (c.constBrick) return;
    const bool mr = a.codesR != nullptr;
    const uint8_t *Cb = a.codes + (int64_t)brick * a.codeStride;
    const uint8_t *Tb = a.temp + (int64_t)brick * a.heapStride;
    const uint8_t *Rl = a.rb.b[c.par] + (int64_t)brick * a.leafStride;
    const uint8_t *CbR = mr ? a.codesR + (int64_t)brick * a.codeStride : nullptr;
    const uint8_t *TbR = mr ? a.tempR + (int64_t)brick * a.heapStride : nullptr;
    const uint8_t *RlR = mr ? a.rbR.b[a.ctrlsR[brick].par] + (int64_t)brick * a.leafStride : nullptr;
    const uint8_t *dmapR = mr ? a.ctrlsR[brick].distanceMap : nullptr;
    const uint32_t n = 1u << a.D;
    const uint32_t r = blockIdx.x * EMIT_RANKS_PER_BLOCK + threadIdx.x;
    for (int i = threadIdx.x; i < EMIT_LDS_WORDS; i += blockDim.x) { W[i] = 0; WR[i] = 0; }
    Owned o;
    o.nSpine = o.nLeaf = 0; o.finalErr = -1; o.aliveAtDs = 0; o.preDs = 0; o.zeroRun = 0;
    if (r < n) o = owned_tokens(Cb, CbR, Tb, TbR, Rl, RlR, a.D, a.maxDepth, a.tol, c.distanceMap, dmapR, a.Ds, r);
    if (o.zeroRun) atomicAdd(&c.zeroRun, 1);
    uint32_t w = o.nSpine + o.nLeaf, tot;
    uint32_t lo = block_excl_scan_u32(w, shw, tot); // also orders the LDS clear above
    const uint32_t g0 = a.blockOff[(int64_t)brick * a.nEmitBlk + blockIdx.x];
    const uint32_t phase = g0 & 15u;
    if (w) {
        uint32_t pos = phase + lo;
        lds_put(W, pos, (uint32_t)o.spineBits, o.nSpine < 16 ? o.nSpine : 16);
        if (o.nSpine > 16) lds_put(W, pos + 16, (uint32_t)(o.spineBits >> 32), o.nSpine - 16);
        lds_put(W, pos + o.nSpine, o.leafBits, o.nLeaf);
        if (mr) {
            lds_put(WR, pos, (uint32_t)o.spineBitsR, o.nSpine < 16 ? o.nSpine : 16);
            if (o.nSpine > 16) lds_put(WR, pos + 16, (uint32_t)(o.spineBitsR >> 32), o.nSpine - 16);
            lds_put(WR, pos + o.nSpine, o.leafBitsR, o.nLeaf);
        }
    }
    // decode side-car index: one entry per depth-Ds subtree root
    if (r < n && (r & ((1u << a.K) - 1u)) == 0) {
        const uint32_t s = r >> a.K;
        int val = c.distanceMap[0];              // root scalar (R.cpp:743)
        for (int j = 1; j <= a.Ds; ++j) {
            int code = cget(Cb, ((int64_t)1 << j) + (s >> (a.Ds - j)));
            val = apply_code(val, code, c.distanceMap[j]); // pruned descendants carry code 3: unchanged
        }
        a.idxOff[(int64_t)brick * a.nIdx + s] = o.aliveAtDs ? g0 + lo + (uint32_t)o.preDs : VR_IDX_DEAD;
        a.idxVal[(int64_t)brick * a.nIdx + s] = (uint8_t)val;
    }
    __syncthreads();
    if (tot == 0) return;
    uint32_t *G = (uint32_t *)(a.tree + (int64_t)brick * a.treeCap) + (g0 >> 4);
    uint32_t *GR = mr ? (uint32_t *)(a.treeR + (int64_t)brick * a.treeCap) + (g0 >> 4) : nullptr;
    const uint32_t nw = ((phase + tot - 1) >> 4) + 1;
    for (uint32_t i = threadIdx.x; i < nw; i += blockDim.x) {
        if (i == 0 || i == nw - 1) { if (W[i]) atomicOr(&G[i], W[i]); if (mr && WR[i]) atomicOr(&GR[i], WR[i]); }
        else { G[i] = W[i]; if (mr) GR[i] = WR[i]; }
    }
}

// ---- emit, 4 leaf ranks per thread (VolumeKdtree streams with D >= 12) -----------------
// Thread t of a block owns ranks r0..r0+3 (r0 = 4t): the spine above the quad (first leaf
// r0), the depth-(D-2) node, both depth-(D-1) nodes and the four leaves with their grown
// branches -- at most D+1+32 tokens, assembled in a 128-bit register string.  Leaf arrays
// are read as dwords; one block = 1024 ranks.
#define EMIT4_RANKS 1024
#define EMIT4_LDS_WORDS 640     // >= (1024*8 + 1023 + 28 + 15) / 16

struct Str128 { unsigned long long lo, hi; int n; };
struct Quad { Str128 s; int preDs, aliveAtDs, zeroRun; };

// inner: codes of the block's internal nodes at depths D-10 .. D-3, heap-ordered (node (l, i) at (1<<l)+i)
// lutS: k_chain_lut's table (LDS copy)
// What a thread reads from memory for its quad, fetched in one batch before anything is looked at
struct QuadIn { uint32_t quadB, pairB, clb, tl, rl; };
__device__ __forceinline__ QuadIn quad_load(const uint8_t *__restrict__ Cb, const uint8_t *__restrict__ Tb,
                                            const uint8_t *__restrict__ Rl, int D, uint32_t r0)
{
    QuadIn q;
    q.quadB = Cb[(((int64_t)1 << (D - 2)) + (r0 >> 2)) >> 2];      // byte holding the depth-(D-2) code
    q.pairB = Cb[(((int64_t)1 << (D - 1)) + (r0 >> 1)) >> 2];      // 4 packed depth-(D-1) codes
    q.clb = Cb[(((int64_t)1 << D) + r0) >> 2];                      // my four leaf codes = one byte
    q.tl = *(const uint32_t *)(Tb + ((int64_t)1 << D) + r0);
    q.rl = *(const uint32_t *)(Rl + r0);
    return q;
}

__device__ inline Quad quad_tokens(const QuadIn &in, const uint8_t *inner, const uint32_t *lutS, bool rootLive,
                                   unsigned long long upSpine, int D, int maxDepth, int tol, int Ds, uint32_t r0)
{
    Quad Q;
    Q.s.lo = Q.s.hi = 0; Q.s.n = 0; Q.preDs = 0; Q.aliveAtDs = 0; Q.zeroRun = 0;
    const uint32_t lr = r0 & 1023u;                                  // rank inside the block
    const int quadCode = (int)((in.quadB >> (((r0 >> 2) & 3u) * 2u)) & 3u);
    const uint32_t pk2 = (in.pairB >> (((r0 >> 1) & 3u) * 2u)) & 15u;                // mine: two of the four
    const uint32_t pair = (pk2 & 3u) | ((pk2 >> 2) << 8);
    const uint32_t clb = in.clb;
    const uint32_t cl = (clb & 3u) | (((clb >> 2) & 3u) << 8) | (((clb >> 4) & 3u) << 16) | (((clb >> 6) & 3u) << 24);
    const uint32_t tl = in.tl, rl = in.rl;
    // The thread's string = spine (the live ancestors above the quad node whose first leaf is r0; at most
    // D-2 <= 26 tokens, one 64-bit word) followed by the quad node's subtree, composed bottom-up from
    // fixed-shape pieces: leaf = code + branch (<= 8 tokens), half = pair node + two leaves (<= 17 tokens),
    // quad = node + two halves (<= 35 tokens).  Three wide shifts per thread instead of one per token.
    bool alive;
    int j;
    unsigned long long S = 0;
    int ns = 0;
    if (lr == 0) {                            // first rank of the block: the spine above depth D-10 comes precomputed
        ns = (int)(upSpine >> 56);
        S = upSpine & 0x00FFFFFFFFFFFFFFull;
        alive = rootLive;
        j = D - 10;
    } else {
        const int c0 = __ffs((int)lr) - 1;    // 2..9
        j = D - c0;                           // first spine depth (> D-10)
        const int l = j - 1 - (D - 10);       // parent level inside the block
        alive = inner[(1 << l) + (lr >> (D - j + 1))] != 3;
    }
    for (; alive && j <= D - 3; ++j) {         // spine down to the quad node's parent
        if (j == Ds) { Q.preDs = ns; Q.aliveAtDs = 1; }
        const int l = j - (D - 10);
        const uint32_t code = inner[(1 << l) + (lr >> (D - j))];
        S |= (unsigned long long)code << (2 * ns);
        ++ns;
        if (code == 3) alive = false;
    }
    unsigned long long qlo = 0, qhi = 0;
    int qn = 0;
    if (alive) {
        if (D - 2 == Ds) { Q.preDs = ns; Q.aliveAtDs = 1; }      // (Ds <= D-2: the index granularity K is >= 2 here)
        qlo = (unsigned long long)quadCode;
        qn = 1;
        if (quadCode != 3) {
            unsigned long long H[2];
            int Hn[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t pc = (pair >> (8 * h)) & 255u;
                H[h] = pc; Hn[h] = 1;
                if (pc == 3) continue;
                uint32_t Lb[2];
                int Ln[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int k = 2 * h + e;
                    const uint32_t code = (cl >> (8 * k)) & 255u;
                    uint32_t bits = code;
                    int nt = 1;
                    if (code != 3) {                                         // grown branch, R.cpp:655-704
                        const int t = (int)((tl >> (8 * k)) & 255u);
                        int rec = (int)((rl >> (8 * k)) & 255u);
                        const int m0 = rec > t ? rec - t : t - rec, lim = t < 255 - t ? t : 255 - t;
                        if (m0 <= lim) {                                     // no clamp can matter: table (see k_chain_lut)
                            const uint32_t en = lutS[m0];
                            const uint32_t ch = en >> 16;
                            bits |= (t > rec ? ch : chain_mirror(ch)) << 2;
                            nt += (int)((en >> 8) & 7u);
                        } else {
                            int depth = D;
                            while (depth < maxDepth) {
                                const int err = rec > t ? rec - t : t - rec;
                                if (err > tol) {
                                    ++depth;
                                    const Enc en = encode_node(t, rec, 64 >> (depth - D - 1));   // distanceMap[D+1..] = 64 .. 1
                                    rec = en.recon;
                                    if (depth == maxDepth && en.code == 0) Q.zeroRun += 1;
                                    bits |= (uint32_t)en.code << (2 * nt);
                                    ++nt;
                                } else { bits |= 3u << (2 * nt); ++nt; break; }
                            }
                        }
                    }
                    Lb[e] = bits; Ln[e] = nt;
                }
                H[h] = (unsigned long long)pc | ((unsigned long long)Lb[0] << 2) | ((unsigned long long)Lb[1] << (2 + 2 * Ln[0]));
                Hn[h] = 1 + Ln[0] + Ln[1];
            }
            const int sh1 = 2 + 2 * Hn[0];                  // 4 .. 36
            qlo |= (H[0] << 2) | (H[1] << sh1);
            qhi = H[1] >> (64 - sh1);
            qn = 1 + Hn[0] + Hn[1];
        }
    }
    const int shS = 2 * ns;                                 // <= 52
    Q.s.lo = S | (qlo << shS);
    Q.s.hi = (qhi << shS) | (shS ? qlo >> (64 - shS) : 0ull);
    Q.s.n = ns + qn;
    return Q;
}

// One flag per 1024-rank emit block: does its depth-(D-10) subtree emit anything at all?
// No iff a proper ancestor of that subtree's root was pruned and this block does not hold
// that ancestor's first leaf (the pruned ancestor's own token belongs to the block that does).
// Lets k_emit4 skip the constant regions of a volume without touching their leaf arrays.
__global__ void __launch_bounds__(256)
k_block_alive(EmitArgs a, int64_t nblk, int sub)     // sub: log2 of the leaves per emit block (10: k_emit4, 12: k_concat12)
{
    const int brick = blockIdx.y;
    const int64_t blk = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (blk >= nblk || a.ctrls[brick].constBrick) return;
    const uint8_t *Cb = a.codes + (int64_t)brick * a.codeStride;
    const uint8_t *dmap = a.ctrls[brick].distanceMap;
    const int dl = a.D - sub;
    bool alive = true;
    int val = dmap[0];                       // decoded scalar along the path (root: R.cpp:743)
    unsigned long long spine = 0, spineR = 0; // tokens above depth D-10 owned by the block's first rank (mid / range stream)
    const uint8_t *CbR = a.codesR ? a.codesR + (int64_t)brick * a.codeStride : nullptr;
    int nsp = 0;
    const uint32_t r0 = (uint32_t)blk << sub;
    const int jmin = r0 ? a.D - (__ffs((int)r0) - 1) : 0;     // first spine depth of rank r0 (<= dl)
    bool pathAlive = true;
    for (int j = 0; j < dl; ++j) {
        const int code = cget(Cb, ((int64_t)1 << j) + (blk >> (dl - j)));
        if (j > 0) val = apply_code(val, code, dmap[j]);
        if (pathAlive && j >= jmin) {
            spine |= (unsigned long long)code << (2 * nsp);
            if (CbR) spineR |= (unsigned long long)cget(CbR, ((int64_t)1 << j) + (blk >> (dl - j))) << (2 * nsp);   // 3 wherever the mid code is
            ++nsp;
        }
        if (code == 3 && pathAlive) {
            // first pruned node on the path: it is emitted itself, by the block holding ITS first leaf
            alive = ((blk >> (dl - j)) << (dl - j)) == blk;
            pathAlive = false;
        }
    }
    const int64_t o = (int64_t)brick * a.nEmitBlk + blk;
    a.blockAlive[o] = (uint8_t)((alive ? 1 : 0) | (pathAlive ? 2 : 0));   // bit1: the depth-(D-10) node itself is live
    a.blockVal[o] = (uint8_t)val;            // scalar of the depth-(D-11) parent (codes applied down to depth dl-1)
    a.blockSpine[o] = spine | ((unsigned long long)nsp << 56);
    if (CbR) a.blockSpineR[o] = spineR | ((unsigned long long)nsp << 56);
    // tokens this block emits: the upper spine its first rank owns + its own subtree if that is live
    // (k_prune12 left the subtree's count in blockOff[], which k_emit_scan overwrites afterwards)
    a.blockTot[o] = (uint32_t)nsp + (pathAlive ? a.blockOff[o] : 0u);
}

template <bool WRITE>
__global__ void __launch_bounds__(256)
k_emit4(EmitArgs a)
{
    __shared__ uint32_t shw[4];
    __shared__ uint32_t W[WRITE ? EMIT4_LDS_WORDS : 1];
    const int brick = blockIdx.y;
    Ctrl &c = a.ctrls[brick];
    if (c.constBrick) return;
    const uint8_t *Cb = a.codes + (int64_t)brick * a.codeStride;
    const uint8_t *Tb = a.temp + (int64_t)brick * a.heapStride;
    const uint8_t *Rl = a.rb.b[c.par] + (int64_t)brick * a.leafStride;
    const uint32_t r0 = blockIdx.x * EMIT4_RANKS + threadIdx.x * 4;
    __shared__ uint8_t inner[256];
    __shared__ uint32_t lutS[256];
    const int64_t bo = (int64_t)brick * a.nEmitBlk + blockIdx.x;
    const int bflags = a.blockAlive[bo];
    const int bval = a.blockVal[bo];
    const unsigned long long upSpine = a.blockSpine[bo];
    if (!(bflags & 1)) {          // wave-uniform: nothing to emit here
        if (!WRITE) { if (threadIdx.x == 0) a.blockTot[bo] = 0; return; }
        if ((r0 & ((1u << a.K) - 1u)) == 0) {     // index entries of a dead region: value of the pruned ancestor
            a.idxOff[(int64_t)brick * a.nIdx + (r0 >> a.K)] = VR_IDX_DEAD;
            a.idxVal[(int64_t)brick * a.nIdx + (r0 >> a.K)] = (uint8_t)bval;   // codes below a pruned node are all 3
        }
        return;
    }
    // every global load of the block is issued here, before any is looked at (one memory round trip
    // instead of a chain of them): the block's internal nodes of depths D-10 .. D-3 (255 codes, one per
    // thread, heap index t inside the block), the branch table, the quad's own bytes, the block offset
    const int t = threadIdx.x ? threadIdx.x : 1, l = 31 - __clz(t);
    const int64_t innerIdx = ((int64_t)1 << (a.D - 10 + l)) + (((int64_t)blockIdx.x) << l) + (t - (1 << l));
    const uint32_t innerB = Cb[innerIdx >> 2];
    const uint32_t lutV = a.chainLut[threadIdx.x];
    const QuadIn qin = quad_load(Cb, Tb, Rl, a.D, r0);
    const uint32_t g0 = WRITE ? a.blockOff[bo] : 0u;
    inner[threadIdx.x] = (uint8_t)((innerB >> ((int)(innerIdx & 3) * 2)) & 3u);
    lutS[threadIdx.x] = lutV;
    if (WRITE) for (int i = threadIdx.x; i < EMIT4_LDS_WORDS; i += 256) W[i] = 0;
    __syncthreads();
    const Quad Q = quad_tokens(qin, inner, lutS, (bflags & 2) != 0, upSpine, a.D, a.maxDepth, a.tol, a.Ds, r0);
    uint32_t tot;
    const uint32_t lo = block_excl_scan_u32((uint32_t)Q.s.n, shw, tot);
    if (!WRITE) {
        if (threadIdx.x == 0) a.blockTot[(int64_t)brick * a.nEmitBlk + blockIdx.x] = tot;
        return;
    }
    const uint32_t phase = g0 & 15u;
    if (Q.zeroRun) atomicAdd(&c.zeroRun, Q.zeroRun);
    // never write outside the brick's stream buffer, whatever the counts say (a count/emit mismatch
    // would be a bug; it must surface as a failed parity check, not as a memory fault)
    if (((unsigned long long)g0 + tot + 32ull) * 2ull > (unsigned long long)a.treeCap * 8ull) {
        if (threadIdx.x == 0) atomicMax(&c.emitOverflow, 1);
        return;
    }
    if (Q.s.n) {
        const uint32_t pos = phase + lo;
        const int sh = (int)(pos & 15u) * 2;
        uint32_t w = pos >> 4;
        // 128-bit string shifted into up to five words
        const unsigned long long v0 = Q.s.lo << sh;
        const unsigned long long v1 = (Q.s.hi << sh) | (sh ? (Q.s.lo >> (64 - sh)) : 0ull);
        const uint32_t v2 = sh ? (uint32_t)(Q.s.hi >> (64 - sh)) : 0u;
        const int nbits = sh + 2 * Q.s.n;
        atomicOr(&W[w], (uint32_t)v0);
        if (nbits > 32) atomicOr(&W[w + 1], (uint32_t)(v0 >> 32));
        if (nbits > 64) atomicOr(&W[w + 2], (uint32_t)v1);
        if (nbits > 96) atomicOr(&W[w + 3], (uint32_t)(v1 >> 32));
        if (nbits > 128) atomicOr(&W[w + 4], v2);
    }
    if ((r0 & ((1u << a.K) - 1u)) == 0) {     // decode side-car index entry (K >= 2): depth Ds = D-K >= D-10
        const uint32_t sidx = r0 >> a.K;
        int val = bval;                        // scalar of the parent of the block's depth-(D-10) node
        const uint32_t lr = r0 & 1023u;
        for (int j = a.D - 10; j <= a.Ds; ++j) {
            const int l = j - (a.D - 10);
            const int code = j >= a.D - 2 ? cget(Cb, ((int64_t)1 << j) + (r0 >> (a.D - j)))
                                          : inner[(1 << l) + (lr >> (a.D - j))];
            val = j == 0 ? val : apply_code(val, code, c.distanceMap[j]);
        }
        a.idxOff[(int64_t)brick * a.nIdx + sidx] = Q.aliveAtDs ? g0 + lo + (uint32_t)Q.preDs : VR_IDX_DEAD;
        a.idxVal[(int64_t)brick * a.nIdx + sidx] = (uint8_t)val;
    }
    __syncthreads();
    if (tot == 0) return;
    uint32_t *G = (uint32_t *)(a.tree + (int64_t)brick * a.treeCap) + (g0 >> 4);
    const uint32_t nw = ((phase + tot - 1) >> 4) + 1;
    for (uint32_t i = threadIdx.x; i < nw; i += 256) {
        if (i == 0 || i == nw - 1) { if (W[i]) atomicOr(&G[i], W[i]); }
        else G[i] = W[i];
    }
}

// The decode index of a fused build: block-local offsets -> offsets into the gapped stream buffer (slot b starts at
// token b * PE_TOKENS), the scalar decoded down to depth Ds, and the scalars of the eight depth-(D-3) nodes below it
// (k_decode_quad).  Codes under a pruned node are all 3, so the walks are the same for live and dead entries.
__global__ void __launch_bounds__(256)
k_index12(EmitArgs a, uint32_t nblk)
{   // a wave per block, four per workgroup (one-wave workgroups: two million of them were dispatch-bound)
    const int brick = blockIdx.y, t = threadIdx.x & 63, D = a.D;
    Ctrl &c = a.ctrls[brick];
    if (c.constBrick) return;
    const uint32_t blk = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (blk >= nblk) return;
    const int64_t bo = (int64_t)brick * a.nEmitBlk + blk;
    const int bflags = a.blockAlive[bo];
    const int bval = a.blockVal[bo];
    const uint32_t s = (blk << 6) + (uint32_t)t;                        // my depth-Ds (= D-6) subtree
    const int64_t io = (int64_t)brick * a.nIdx + s;
    const uint8_t *Cb = a.codes + (int64_t)brick * a.codeStride;
    {   // a block whose root is a pruned token (two thirds of the bench volume's: constant regions), or that lies under
        // one: every code below moves nothing (3s, or the "keep"s the level loop left under a root k_prune_emit12
        // pruned), so all 64 entries are dead with the scalar above the block -- one byte decides, nothing else is read
        const int64_t nr = ((int64_t)1 << (D - 12)) + blk;
        const uint32_t rootCode = ((uint32_t)Cb[nr >> 2] >> ((int)(nr & 3) * 2)) & 3u;
        if (rootCode == 3u || !(bflags & 2)) {
            a.idxOff[io] = VR_IDX_DEAD;
            if (a.idxBase && t == 0) a.idxBase[bo] = (unsigned long long)blk * PE_TOKENS;
            a.idxVal[io] = (uint8_t)bval;
            *(uint2 *)(a.idxVal3 + io * 8) = make_uint2((uint32_t)bval * 0x01010101u, (uint32_t)bval * 0x01010101u);
            return;
        }
    }
    const uint32_t local = a.idxOff[io];
    uint32_t cb[7];
    int csh[7], dist[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        const int j = D - 12 + q;
        const int64_t ni = ((int64_t)1 << j) + (s >> (6 - q));
        cb[q] = Cb[ni >> 2];
        csh[q] = (int)(ni & 3) * 2;
        dist[q] = c.distanceMap[j];
    }
    // the three levels below my node (2 + 4 + 8 codes: the level loop's own, pruning only ever turns a 0 into a 3 and
    // neither moves a scalar) for the depth-(D-3) scalars k_decode_quad starts from
    const int64_t n5 = ((int64_t)1 << (D - 5)) + 2 * (int64_t)s, n4 = ((int64_t)1 << (D - 4)) + 4 * (int64_t)s,
                  n3 = ((int64_t)1 << (D - 3)) + 8 * (int64_t)s;
    const uint32_t c5 = (uint32_t)Cb[n5 >> 2] >> ((int)(n5 & 3) * 2), c4 = Cb[n4 >> 2], c3 = *(const uint16_t *)(Cb + (n3 >> 2));
    const int dl5 = c.distanceMap[D - 5], dl4 = c.distanceMap[D - 4], dl3 = c.distanceMap[D - 3];
    int val = bval;                             // scalar of the block root's parent
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        const int code = (int)((cb[q] >> csh[q]) & 3u);
        val = (D - 12 + q) == 0 ? val : apply_code(val, code, dist[q]);
    }
    // (64-bit trees: the entry stays relative to the block, whose slot offset goes to idxBase)
    // an entry whose own token is a 3 is as good as one under a pruned node: all its voxels take its scalar, and a
    // decoder that sees "dead" fills them without touching the stream
    const bool root3 = ((cb[6] >> csh[6]) & 3u) == 3u;
    a.idxOff[io] = ((bflags & 2) && local != VR_IDX_DEAD && !root3) ? (a.idxBase ? 0u : blk * (uint32_t)PE_TOKENS) + local : VR_IDX_DEAD;
    if (a.idxBase && t == 0) a.idxBase[bo] = (unsigned long long)blk * PE_TOKENS;
    a.idxVal[io] = (uint8_t)val;
    uint32_t lo3 = 0, hi3 = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int v = apply_code(val, (int)((c5 >> (2 * (i >> 2))) & 3u), dl5);
        v = apply_code(v, (int)((c4 >> (2 * (i >> 1))) & 3u), dl4);
        v = apply_code(v, (int)((c3 >> (2 * i)) & 3u), dl3);
        if (i < 4) lo3 |= (uint32_t)v << (8 * i); else hi3 |= (uint32_t)v << (8 * (i - 4));
    }
    *(uint2 *)(a.idxVal3 + io * 8) = make_uint2(lo3, hi3);
}

// The reference's contiguous layout (R.cpp:631-718 writes one array): block b contributes the upper spine its first rank
// owns (k_block_alive) and, if its depth-(D-12) root is live, its string from the gapped buffer, at token offset
// blockOff[b] of the brick's stream (a.tree / a.treeR = the destination, brick b at a.brickOff[b]).
// A workgroup takes 16 consecutive blocks = one contiguous piece of the stream: their metadata in one batch of loads;
// the short ones (two thirds of the bench volume's blocks are a spine and one token) by one lane each; the long ones
// by all 256 threads, four destination words per thread and trip from five source words (one 16-byte and one 4-byte
// load, funnel shifts, one 16-byte store).  A word two blocks share is merged in LDS and written once; only the
// piece's very first and last word can belong to a neighbouring workgroup too: those were zeroed by k_emit_zero and
// take atomicOr (two atomics per workgroup instead of two per block: the atomics were a third of this kernel).
#define CC_BLOCKS 16
#ifndef CC_BATCH
#define CC_BATCH 4          // chunks per thread whose loads are in flight together
#endif
template <bool RANGE>        // RANGE: MidRangeTree's second stream -- same offsets and shape, its own strings and spine
__global__ void __launch_bounds__(256)
k_concat12(EmitArgs a, const uint8_t *__restrict__ gap, int64_t nblk)
{
    __shared__ unsigned long long g0S[CC_BLOCKS], spineS[CC_BLOCKS];
    __shared__ uint32_t totS[CC_BLOCKS], cntS[CC_BLOCKS];
    __shared__ unsigned long long bwIdx[2 * CC_BLOCKS];     // boundary words: destination word index (~0: unused slot) ...
    __shared__ uint32_t bwVal[2 * CC_BLOCKS];               // ... and this block's bits of it
    __shared__ uint32_t chunkPre[CC_BLOCKS + 1];            // long blocks: first chunk of every block in the workgroup's chunk list
    __shared__ uint8_t chunkBlk[CC_BLOCKS * (PE_WORDS / 4 + 4)];   // ... and the block of every chunk
    const int brick = blockIdx.y, t = threadIdx.x;
    Ctrl &c = a.ctrls[brick];
    if (c.constBrick) return;
    if (a.brickOff && a.brickOff[gridDim.y] > (unsigned long long)a.compactCap) return;       // (did not fit: the host regrows and repeats)
    const int64_t blk0 = (int64_t)blockIdx.x * CC_BLOCKS;
    const int64_t base = a.brickOff ? (int64_t)a.brickOff[brick] : (int64_t)brick * a.treeCap;
    uint32_t *G0 = (uint32_t *)((RANGE ? a.treeR : a.tree) + base);
    const uint64_t capTok = a.brickOff ? ~0ull >> 2 : (uint64_t)a.treeCap * 4ull;            // tokens the brick's buffer holds
    if (t < 2 * CC_BLOCKS) { bwIdx[t] = ~0ull; bwVal[t] = 0u; }
    if (t < CC_BLOCKS) {
        const int64_t blk = blk0 + t;
        unsigned long long g0 = 0, sp = 0;
        uint32_t tot = 0, cnt = 0;
        if (blk < nblk) {
            const int64_t bo = (int64_t)brick * a.nEmitBlk + blk;
            const int bflags = a.blockAlive[bo];
            sp = RANGE ? a.blockSpineR[bo] : a.blockSpine[bo];
            g0 = a.blockOff64 ? a.blockOff64[bo] : (unsigned long long)a.blockOff[bo];
            tot = (bflags & 1) ? a.blockTot[bo] : 0u;
            cnt = (bflags & 2) && tot ? tot - (uint32_t)(sp >> 56) : 0u;      // tokens of the block's own string
            // never write outside the brick's stream buffer, whatever the counts say (a count/emit mismatch would be a
            // bug; it must surface as a failed parity check, not as a memory fault)
            if (tot && g0 + tot + 32ull > capTok) { atomicMax(&c.emitOverflow, 1); tot = 0; cnt = 0; }
        }
        g0S[t] = g0; spineS[t] = sp; totS[t] = tot; cntS[t] = cnt;
    }
    __syncthreads();
    // ---- short blocks (spine + string of at most 32 tokens): one lane each
    if (t < CC_BLOCKS && totS[t] != 0u && totS[t] <= 32u) {
        const unsigned long long g0 = g0S[t];
        const uint32_t tot = totS[t], cnt = cntS[t];
        const int nsp = (int)(spineS[t] >> 56);
        unsigned long long v = spineS[t] & 0x00FFFFFFFFFFFFFFull;
        if (cnt) {
            const uint32_t *slot = (const uint32_t *)(gap + (int64_t)brick * a.treeCap) + (size_t)(blk0 + t) * PE_WORDS;
            unsigned long long sv = slot[0];
            if (cnt > 16u) sv |= (unsigned long long)slot[1] << 32;
            if (cnt < 32u) sv &= (1ull << (2 * cnt)) - 1ull;
            v |= sv << (2 * nsp);
        }
        const uint32_t sh = (uint32_t)(g0 & 15ull) * 2u;
        const uint64_t w0 = g0 >> 4, wl = (g0 + tot - 1) >> 4;       // first and last word (shared with the neighbours)
        const uint32_t x0 = (uint32_t)(v << sh), x1 = (uint32_t)((v << sh) >> 32), x2 = sh ? (uint32_t)(v >> (64u - sh)) : 0u;
        bwIdx[2 * t] = w0; bwVal[2 * t] = x0;
        if (wl == w0 + 1) { bwIdx[2 * t + 1] = wl; bwVal[2 * t + 1] = x1; }
        else if (wl == w0 + 2) { G0[w0 + 1] = x1; bwIdx[2 * t + 1] = wl; bwVal[2 * t + 1] = x2; }
    }
    // ---- long blocks: their 16-byte destination chunks as ONE list over the workgroup's blocks (a prefix over the
    // blocks' chunk counts, a byte per chunk saying whose it is), four chunks per thread and trip with all their loads
    // issued before the first is used: a thread looping over the blocks one after another had one load in flight and
    // sixteen dependent round trips (2.3 ms for the bench volume's 4.5 GB; the copy itself is 1.6)
    if (t == 0) {
        uint32_t run = 0;
        for (int i = 0; i < CC_BLOCKS; ++i) {
            chunkPre[i] = run;
            if (totS[i] > 32u) {
                const uint64_t w0 = g0S[i] >> 4, wl = (g0S[i] + totS[i] - 1) >> 4;
                run += (uint32_t)((wl - (w0 & ~3ull)) / 4 + 1);
            }
        }
        chunkPre[CC_BLOCKS] = run;
    }
    __syncthreads();
    const uint32_t nchunkAll = chunkPre[CC_BLOCKS];
    for (int i = 0; i < CC_BLOCKS; ++i)
        for (uint32_t j = chunkPre[i] + (uint32_t)t; j < chunkPre[i + 1]; j += 256u) chunkBlk[j] = (uint8_t)i;
    __syncthreads();
    const uint32_t *gapW = (const uint32_t *)(gap + (int64_t)brick * a.treeCap) + (size_t)blk0 * PE_WORDS;
    for (uint32_t j0 = (uint32_t)t; j0 < nchunkAll; j0 += 256u * CC_BATCH) {
        uint32_t sv[CC_BATCH][5];
        // -- the loads of up to CC_BATCH chunks
#pragma unroll
        for (int k = 0; k < CC_BATCH; ++k) {
            const uint32_t j = j0 + 256u * (uint32_t)k;
#pragma unroll
            for (int m = 0; m < 5; ++m) sv[k][m] = 0u;
            if (j >= nchunkAll) continue;
            const int i = chunkBlk[j];
            const unsigned long long g0 = g0S[i];
            const uint32_t cnt = cntS[i];
            const int nsp = (int)(spineS[i] >> 56);
            const int nws = (int)((cnt + 15u) >> 4);
            const uint32_t *slot = gapW + (size_t)i * PE_WORDS;
            const uint64_t wA = (g0 >> 4) & ~3ull;
            const long long so = 2ll * ((long long)(16ull * wA) - (long long)g0 - nsp) + 128ll * (long long)(j - chunkPre[i]);
            const int sw = (int)(so >> 5);                                 // (floor division)
            if (cnt && sw >= 0 && sw + 4 < nws) {                          // the common case: five words inside the string
                struct __attribute__((packed, aligned(4))) U4 { uint32_t x, y, z, w; };
                const U4 u = *(const U4 *)(slot + sw);
                sv[k][0] = u.x; sv[k][1] = u.y; sv[k][2] = u.z; sv[k][3] = u.w; sv[k][4] = slot[sw + 4];
            } else {
#pragma unroll
                for (int m = 0; m < 5; ++m) sv[k][m] = (cnt && sw + m >= 0 && sw + m < nws) ? slot[sw + m] : 0u;
            }
        }
        // -- shift, edges, store
#pragma unroll
        for (int k = 0; k < CC_BATCH; ++k) {
            const uint32_t j = j0 + 256u * (uint32_t)k;
            if (j >= nchunkAll) continue;
            const int i = chunkBlk[j];
            const unsigned long long g0 = g0S[i];
            const uint32_t tot = totS[i], cnt = cntS[i];
            const int nsp = (int)(spineS[i] >> 56);
            const unsigned long long spine = spineS[i] & 0x00FFFFFFFFFFFFFFull;
            const uint64_t w0 = g0 >> 4, wl = (g0 + tot - 1) >> 4;
            const uint64_t wA = w0 & ~3ull;                                  // first destination word of chunk 0 (16-byte aligned)
            const uint32_t q = j - chunkPre[i];
            const uint64_t d0 = wA + 4ull * (uint64_t)q;                     // my four destination words d0 .. d0+3
            // bit offset of word d0 in the block's string (negative: in front of it)
            const long long so = 2ll * ((long long)(16ull * wA) - (long long)g0 - nsp) + 128ll * (long long)q;
            const uint32_t shf = (uint32_t)(so & 31ll);
            uint32_t o[4];
            const int tok0 = (int)(so >> 1);                               // string token at bit 0 of word d0
            const bool edge = tok0 < 0 || tok0 + 64 > (int)cnt;            // the chunk touches the string's start or end
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                o[m] = __builtin_amdgcn_alignbit(sv[k][m + 1], sv[k][m], shf);
                if (edge) {
                    const int tokAt = tok0 + 16 * m;
                    if (tokAt + 16 > (int)cnt) {        // the string's last word may carry stale bits behind the last token: cut them
                        const int keep = (int)cnt - tokAt;
                        o[m] = keep <= 0 ? 0u : (keep >= 16 ? o[m] : (o[m] & ((1u << (2 * keep)) - 1u)));
                    }
                    if (tokAt < 0 && nsp) {             // in front of the string: the spine's bits
                        const long long bo2 = 2ll * ((long long)(16ull * (d0 + m)) - (long long)g0);
                        o[m] |= bo2 >= 0 ? (bo2 < 64 ? (uint32_t)(spine >> bo2) : 0u) : (uint32_t)(spine << (-bo2));
                    }
                }
            }
            if (d0 > w0 && d0 + 3 < wl) {
                *(uint4 *)(G0 + d0) = make_uint4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const uint64_t d = d0 + m;
                    if (d < w0 || d > wl) continue;
                    if (d == w0) { bwIdx[2 * i] = d; bwVal[2 * i] = o[m]; }
                    else if (d == wl) { bwIdx[2 * i + 1] = d; bwVal[2 * i + 1] = o[m]; }
                    else G0[d] = o[m];
                }
            }
        }
    }
    __syncthreads();
    // ---- the boundary words: slots are in stream order, so equal words are neighbours; the first slot of a word writes
    // the OR of all of them
    if (t < 2 * CC_BLOCKS) {
        const unsigned long long w = bwIdx[t];
        if (w != ~0ull) {
            bool first = true;
            uint32_t v = 0;
            unsigned long long wFirst = ~0ull, wLast = 0;
            for (int j = 0; j < 2 * CC_BLOCKS; ++j) {
                const unsigned long long wj = bwIdx[j];
                if (wj == ~0ull) continue;
                if (wFirst == ~0ull) wFirst = wj;
                wLast = wj;
                if (wj == w) { v |= bwVal[j]; if (j < t) first = false; }
            }
            if (first) {
                if (w == wFirst || w == wLast) { if (v) atomicOr(&G0[w], v); }
                else G0[w] = v;
            }
        }
    }
}

__global__ void __launch_bounds__(1024)
k_emit_stats(EmitArgs a, int64_t nblk)
{
    __shared__ unsigned long long sh[16];
    const int brick = blockIdx.x;
    if (a.ctrls[brick].constBrick) return;
    unsigned long long v = 0;
    for (int64_t i = threadIdx.x; i < nblk; i += blockDim.x) v = stat_merge(v, a.blockL1[(int64_t)brick * a.nEmitBlk + i]);
    for (int o = 32; o > 0; o >>= 1) v = stat_merge(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < 16; ++i) t = stat_merge(t, sh[i]);
        Ctrl &c = a.ctrls[brick];
        c.statL1 = t & ((1ull << 40) - 1ull);
        c.maxErrBefore = (int)((t >> 40) & 255ull);
        c.maxErrAfter = (int)((t >> 48) & 255ull);
    }
}

// Closed form for a brick of one value v (tolerance >= 1, maxEpochs >= 1, D >= 1): the root's
// distance is v (running mean of the single mismatch |0 - v|), its code is "add" (or "keep" for
// v = 0), every other node reproduces its parent exactly (distance 0, code 0) and is pruned, so
// the stream is [1][3][3] (or [3] for v = 0), all statistics are 0 and every decode-index entry
// is "dead" with scalar v.  Identical to what the general path produces (tests compare both with
// the oracle); it just skips ~all work for the constant regions of a volume.
__global__ void __launch_bounds__(256)
k_const_finish(int D, Ctrl *ctrls, uint8_t *tree, int64_t treeCap, uint32_t *idxOff, uint8_t *idxVal, int64_t nIdx,
               const unsigned long long *brickOff = nullptr, int64_t compactCap = 0)
{
    const int brick = blockIdx.y;
    Ctrl &c = ctrls[brick];
    if (!c.constBrick) return;
    if (brickOff && brickOff[gridDim.y] > (unsigned long long)compactCap) return;
    const int v = c.constVal;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int d = 0; d <= D; ++d) c.distanceMap[d] = d == 0 ? (uint8_t)v : 0;
        c.numActive = v ? 3 : 1;
        *(uint32_t *)(tree + (brickOff ? (int64_t)brickOff[brick] : (int64_t)brick * treeCap)) = v ? 0x3Du : 0x03u;
        c.numReverts = 0; c.maxErrBefore = 0; c.maxErrAfter = 0; c.statL1 = 0; c.emitOverflow = 0;
    }
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s < nIdx) {
        idxOff[(int64_t)brick * nIdx + s] = VR_IDX_DEAD;
        idxVal[(int64_t)brick * nIdx + s] = (uint8_t)v;
    }
}

// The half-range stream of such a brick (MidRangeTree): every half range is 0, the root's distance 0 and code "keep",
// the shape the mid stream's (M.cpp:864-865): [0][3][3], or [3] where the mid stream is [3]; distanceMap all zero.
__global__ void k_const_finish_range(int D, Ctrl *ctrlsR, uint8_t *treeR, int64_t treeCap, const unsigned long long *brickOff = nullptr,
                                     int64_t compactCap = 0)
{
    Ctrl &c = ctrlsR[blockIdx.x];
    if (threadIdx.x || !c.constBrick) return;
    if (brickOff && brickOff[gridDim.x] > (unsigned long long)compactCap) return;
    const int v = c.constVal;             // the brick's VALUE (k_ctrl_init): decides the shape only
    for (int d = 0; d <= D; ++d) c.distanceMap[d] = 0;
    c.numActive = v ? 3 : 1;
    *(uint32_t *)(treeR + (brickOff ? (int64_t)brickOff[blockIdx.x] : (int64_t)blockIdx.x * treeCap)) = v ? 0x3Cu : 0x03u;
    c.numReverts = 0; c.maxErrBefore = 0; c.maxErrAfter = 0; c.statL1 = 0; c.emitOverflow = 0;
}

// ------------------------------------------------------------ host driver ----
static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// The level loop of the bricks [b0, b0 + nb) of a stream (every array of a set is per brick: a sub-range is the same
// launches on offset pointers)
static void compress_stream(BrickSet *bs, Stream2 &s0, hipStream_t st, const uint8_t *rootMin, const uint8_t *rootMax,
                            int64_t mmStride, SkipBlocks sk, unsigned long long *blockErr, void *estSumm0,
                            int b0 = 0, int nb = -1)
{
    const int D = bs->D, B = nb < 0 ? bs->B : nb;
    struct { Ctrl *ctrl; uint8_t *temp, *codes; uint8_t *recon[3]; } s;
    s.ctrl = s0.ctrl + b0; s.temp = s0.temp + (int64_t)b0 * bs->heapStride; s.codes = s0.codes + (int64_t)b0 * bs->codeStride;
    for (int i = 0; i < 3; ++i) s.recon[i] = s0.recon[i] + (int64_t)b0 * bs->reconStride;
    blockErr += (int64_t)b0 * bs->nErrBlk;
    const int64_t errPlane = (int64_t)bs->B * bs->nErrBlk;     // the minus / plus planes of the partials (leafless leaf level) lie behind the first
    void *estSumm = (uint32_t *)estSumm0 + (int64_t)b0 * bs->estSummStride * (4 * EST_CAND);
    if (sk.flag) sk.flag += (int64_t)b0 * sk.nBlk;
    if (rootMin) { rootMin += (int64_t)b0 * mmStride; rootMax += (int64_t)b0 * mmStride; }
    ReconBufs rb{{s.recon[0], s.recon[1], s.recon[2]}};
    const int guarded = bs->variant != 0; // GUARDED and MIDRANGE both carry the :333/:340 guard
    hipLaunchKernelGGL(k_ctrl_init, dim3(B), dim3(64), 0, st, s.ctrl, rootMin, rootMax, mmStride);
    if (bs->maxEpochs <= 0) { // the loop never runs: tree.resize() / recon.resize() zero-fill is the result
        for (int i = 0; i < 3; ++i) hipMemsetAsync(s.recon[i], 0, (size_t)B * bs->reconStride, st);
        hipMemsetAsync(s.codes, 0, (size_t)B * bs->codeStride, st);
    }
    for (int d = 0; d <= D; ++d) {
        const int64_t n = (int64_t)1 << d;
        hipLaunchKernelGGL(k_est_head, dim3(B), dim3(64), 0, st, d, bs->maxEpochs, s.ctrl, s.temp, bs->heapStride, rb,
                           bs->reconStride, sk);
        if (n > EST_HEAD) {
            const int64_t nseg = n / EST_SEG - EST_HEAD / EST_SEG;
            for (int r = 0; r < EST_ROUNDS; ++r) {
                const int nc = r < 2 ? 4 : EST_CAND, ncNext = r + 1 < 2 ? 4 : EST_CAND;   // two 4-wide windows, then 8-wide ones
                const unsigned gx = r == 0 ? cdiv(nseg, 16) : (cdiv(nseg, 4) < 64 ? cdiv(nseg, 4) : 64);   // four segments per wave first
                hipLaunchKernelGGL(k_est_summ, dim3(gx, B), dim3(256), 0, st, d, nc, s.ctrl, s.temp,
                                   bs->heapStride, rb, bs->reconStride, (uint32_t *)estSumm, bs->estSummStride, sk);
                hipLaunchKernelGGL(k_est_walk, dim3(B), dim3(64), 0, st, d, bs->maxEpochs, nc, ncNext,
                                   r == EST_ROUNDS - 1 ? 1 : 0, s.ctrl, s.temp, bs->heapStride, rb, bs->reconStride,
                                   (const uint32_t *)estSumm, bs->estSummStride, sk);
            }
        }
        for (int e = 0; e < bs->maxEpochs; ++e) {
            if (n >= 4096)
            {
                const int iters = n / 4096 >= 512 ? FILL_ITERS : (n / 4096 >= 64 ? (FILL_ITERS < 4 ? FILL_ITERS : 4) : 1);
                hipLaunchKernelGGL((bs->leafless && d == D) ? k_fill16<false> : k_fill16<true>, dim3(cdiv(n / 4096, iters), B), dim3(256), 0, st,
                                   d, bs->maxEpochs, s.ctrl, s.temp, s.codes, bs->heapStride, bs->codeStride, rb, bs->reconStride, blockErr,
                                   bs->nErrBlk, sk, errPlane, iters);
            }
            else
                hipLaunchKernelGGL(k_fill, dim3(cdiv(n, FILL_NODES_PER_BLOCK), B), dim3(256), 0, st, d, s.ctrl, s.temp,
                                   s.codes, bs->heapStride, bs->codeStride, rb, bs->reconStride, blockErr, bs->nErrBlk);
            hipLaunchKernelGGL(k_control, dim3(B), dim3(64), 0, st, d, bs->maxEpochs, guarded, s.ctrl, s.temp,
                               bs->heapStride, rb, bs->reconStride, blockErr, bs->nErrBlk, errPlane, (bs->leafless && d == D) ? 1 : 0);
        }
        hipLaunchKernelGGL(k_level_end, dim3(B), dim3(64), 0, st, d, s.ctrl);
    }
}

__global__ void k_fix_chain_distances(int D, int maxDepth, Ctrl *ctrls, Ctrl *ctrlsR)
{
    if (threadIdx.x) return;
    const int add[VR_CHAIN_LEVELS] = {64, 32, 16, 8, 4, 2, 1}; // R.cpp:23,94-97
    for (int dd = D + 1, i = 0; dd <= maxDepth; ++dd, ++i) {
        ctrls[blockIdx.x].distanceMap[dd] = (uint8_t)add[i];
        if (ctrlsR) ctrlsR[blockIdx.x].distanceMap[dd] = (uint8_t)add[i];
    }
}

// VRHIP_SYNC=1: synchronise after every phase and report the first failing one (debug aid)
static bool dbg_sync(hipStream_t st, const char *phase)
{
    static const bool on = getenv("VRHIP_SYNC") != nullptr;
    if (!on) return true;
    hipError_t e = hipStreamSynchronize(st);
    fprintf(stderr, "[vrhip] phase %s: %s\n", phase, hipGetErrorString(e));
    return e == hipSuccess;
}

static void fill_emit_args(BrickSet *bs, EmitArgs &a)
{
    const bool mr = bs->variant == 2;
    a.codes = bs->mid.codes; a.codesR = mr ? bs->rng.codes : nullptr;
    a.temp = bs->mid.temp; a.tempR = mr ? bs->rng.temp : nullptr;
    a.rb = ReconBufs{{bs->mid.recon[0], bs->mid.recon[1], bs->mid.recon[2]}};
    a.rbR = ReconBufs{{bs->rng.recon[0], bs->rng.recon[1], bs->rng.recon[2]}};
    a.ctrls = bs->mid.ctrl; a.ctrlsR = mr ? bs->rng.ctrl : nullptr;
    a.heapStride = bs->heapStride; a.leafStride = bs->reconStride; a.codeStride = bs->codeStride;
    a.D = bs->D; a.maxDepth = bs->maxDepth; a.tol = bs->tolerance; a.Ds = bs->Ds; a.K = bs->K;
    a.blockTot = bs->blockTot; a.blockOff = bs->blockOff; a.nEmitBlk = bs->nEmitBlk;
    a.blockOff64 = bs->blockOff64; a.idxBase = bs->idxBase;
    a.blockL1 = bs->blockL1;
    a.blockAlive = bs->blockAlive; a.blockVal = bs->blockVal; a.blockSpine = bs->blockSpine; a.blockSpineR = bs->blockSpineR;
    a.tree = bs->mid.tree; a.treeR = mr ? bs->rng.tree : nullptr; a.treeCap = bs->treeCap;
    a.brickOff = nullptr; a.compactCap = 0;
    a.idxOff = bs->idxOff; a.idxVal = bs->idxVal; a.idxVal3 = bs->idxVal3; a.nIdx = bs->nIdx;
    a.chainLut = bs->chainLut;
}

// The reference's contiguous stream(s) of a fused build, into bs->mid.treeCompact (and rng.treeCompact): the bricks'
// byte offsets (their streams back to back), the words two blocks share zeroed, then spine + string of every block to
// its place.  At the end of build() (BrickSet::compactOnBuild) or when the host first asks for bytes.  The buffers are
// sized from the streams' real lengths: the first build of a set guesses, a stream that does not fit raises
// compactOverflow and writes nothing, and the host (capi.hip contiguous_stream) regrows the buffers and repeats.
int compact_launch(BrickSet *bs, hipStream_t st)
{
    const int D = bs->D, B = bs->B;
    const bool mr = bs->variant == 2;
    if (!bs->brickOff && hipMalloc(&bs->brickOff, (size_t)(B + 1) * sizeof(unsigned long long)) != hipSuccess) return -3;
    if (!bs->compactOverflow && hipMalloc(&bs->compactOverflow, sizeof(int32_t)) != hipSuccess) return -3;
    if (bs->compactCap == 0) {
        // first time: 5/8 byte per voxel (the bench volume takes 0.56; noise takes up to 2.3) -- a guess, corrected on demand
        int64_t cap = (int64_t)B * bs->g.voxels / 8 * 5 + (int64_t)B * 64 + 4096;
        const int64_t worst = (int64_t)B * bs->treeCap;
        if (cap > worst) cap = worst;
        if (hipMalloc(&bs->mid.treeCompact, (size_t)cap) != hipSuccess) return -3;
        if (mr && hipMalloc(&bs->rng.treeCompact, (size_t)cap) != hipSuccess) { hipFree(bs->mid.treeCompact); bs->mid.treeCompact = nullptr; return -3; }
        bs->compactCap = cap;
    }
    EmitArgs a;
    fill_emit_args(bs, a);
    a.tree = bs->mid.treeCompact; a.treeR = mr ? bs->rng.treeCompact : nullptr;
    a.brickOff = bs->brickOff; a.compactCap = bs->compactCap;
    const int64_t nblk = cdiv((int64_t)1 << D, 4096);
    // constant bricks first: their token counts (closed form) are part of the offsets
    hipLaunchKernelGGL(k_brick_offsets, dim3(1), dim3(1024), 0, st, bs->mid.ctrl, B, bs->brickOff, bs->compactCap, bs->compactOverflow);
    hipLaunchKernelGGL(k_emit_zero, dim3(cdiv(nblk, 256), B), dim3(256), 0, st, a, nblk);
    hipLaunchKernelGGL(k_concat12<false>, dim3(cdiv(nblk, CC_BLOCKS), B), dim3(256), 0, st, a, bs->mid.tree, nblk);
    if (mr) hipLaunchKernelGGL(k_concat12<true>, dim3(cdiv(nblk, CC_BLOCKS), B), dim3(256), 0, st, a, bs->rng.tree, nblk);
    // (the closed form's stream word only: its index entries were written when the gapped stream was finished)
    hipLaunchKernelGGL(k_const_finish, dim3(1, B), dim3(64), 0, st, D, bs->mid.ctrl, bs->mid.treeCompact,
                       bs->treeCap, bs->idxOff, bs->idxVal, (int64_t)0, bs->brickOff, bs->compactCap);
    if (mr) hipLaunchKernelGGL(k_const_finish_range, dim3(B), dim3(64), 0, st, D, bs->rng.ctrl, bs->rng.treeCompact, bs->treeCap, bs->brickOff, bs->compactCap);
    bs->compactValid = true;
    return launch_status("compact");
}

// Internal streams for the forks below, created on first need and only as many as are used: a process has few hardware
// queues (4 by default), streams beyond them share one, and a kernel queued behind another stream's long copy on a shared
// queue waits for it (a MidRangeTree build inside the timestep streamer took 236 instead of 68 ms with three idle
// extra streams per handle).
static int ensure_aux(BrickSet *bs, int n)
{
    if (n > 3) n = 3;
    if (!bs->evFork && hipEventCreateWithFlags(&bs->evFork, hipEventDisableTiming) != hipSuccess) return 0;
    int have = 0;
    for (int i = 0; i < n; ++i) {
        if (!bs->auxN[i] && hipStreamCreateWithFlags(&bs->auxN[i], hipStreamNonBlocking) != hipSuccess) break;
        if (!bs->evJoinN[i] && hipEventCreateWithFlags(&bs->evJoinN[i], hipEventDisableTiming) != hipSuccess) break;
        ++have;
    }
    bs->aux = bs->auxN[0]; bs->evJoin = bs->evJoinN[0];
    return have;
}

int encode_launch(BrickSet *bs, const uint8_t *vox, hipStream_t st)
{
    const int D = bs->D, B = bs->B;
    const bool mr = bs->variant == 2;
    hipEventRecord(bs->ev[0], st);
    const uint8_t *rootMinP = nullptr, *rootMaxP = nullptr;   // where the last pyramid round leaves each brick's root (min,max)
    const bool fused = D >= 12 && bs->K == 6 && !bs->sw.noFusedEmit;   // prune + block-local emit in one kernel
    if (bs->leafless != (fused && bs->maxEpochs >= 1)) return -2;      // (the arrays were sized for the other mode: capi alloc_encoder_buffers)
    // SkipBlocks needs k_pyramid12's constant bit in front and k_prune_emit12 behind, a prune that makes such blocks
    // one token (tolerance >= 1) and a level loop that runs
    bool skipOn = fused && bs->blockFlag && (!mr || bs->blockFlagR) && bs->tolerance >= 1 && bs->maxEpochs >= 1 && D >= 14 && !bs->sw.noSkipBlocks;
    const int64_t rootStride = (int64_t)1 << (D > 10 ? D - 10 : 0);
    // ---- BUILD: pyramid.  Bottom 12 levels by k_pyramid12 when x-runs of 16 voxels exist,
    // the rest (and small / thin bricks) in rounds of <= 10 levels.
    {
        int dLeaf = D, round = 0;
        const uint8_t *inMin = nullptr, *inMax = nullptr;
        int64_t inStride = 0;
        const int64_t oStride = (int64_t)1 << (D > 10 ? D - 10 : 0);
        Pyr12Geom pg{};
        bool use12 = D >= 12 && !bs->generalGeom;
        skipOn = skipOn && use12;
        if (use12) {
            for (int q = 0; q < 12; ++q) {
                const int ax = bs->g.axis[D - 12 + q];
                if (ax == 0) pg.ax++; else if (ax == 1) pg.ay++; else pg.az++;
            }
            use12 = pg.ax >= 4;
            skipOn = skipOn && use12;
            for (int i = 0; i < 16 && use12; ++i) {
                uint32_t r = 0;
                for (int q = 0; q < 12; ++q) {
                    const int dd = D - 12 + q;
                    const int cbit = bs->g.axis[dd] == 0 ? (i >> bs->g.bit[dd]) & 1 : 0;
                    r = (r << 1) | (uint32_t)(bs->g.bit[dd] < 4 ? cbit : 0);
                }
                pg.sx[i] = (uint16_t)r;
            }
        }
        if (use12) {
            const int64_t Bx = bs->g.X >> pg.ax, By = bs->g.Y >> pg.ay, Bz = bs->g.Z >> pg.az;
            const int64_t perLine = (bs->g.X < 128 ? bs->g.X : 128) >> pg.ax;
            pg.swz = (perLine == 8 && Bx % 8 == 0 && ((Bx / 8) * By * Bz) % 8 == 0 && !bs->sw.noSwz) ? 1 : 0;
            pg.nbx = (int)Bx; pg.nby = (int)By;
            pg.lnbx = 0; while ((1 << pg.lnbx) < pg.nbx) ++pg.lnbx;
            pg.lnby = 0; while ((1 << pg.lnby) < pg.nby) ++pg.lnby;
            pg.spread = bs->spread;
            hipLaunchKernelGGL(k_pyramid12, dim3((unsigned)((int64_t)1 << (D - 12)), B), dim3(256), 0, st, bs->g, pg, vox,
                               bs->mid.temp, bs->heapStride, mr ? bs->rng.temp : nullptr, bs->mmMin[0], bs->mmMax[0],
                               oStride, skipOn ? bs->blockFlag : nullptr, skipOn && mr ? bs->blockFlagR : nullptr);
            dLeaf = D - 12;
            inMin = bs->mmMin[0]; inMax = bs->mmMax[0]; inStride = oStride;
            rootMinP = inMin; rootMaxP = inMax;
            round = 1;
        }
        while (dLeaf > 0 || round == 0) {
            int L = dLeaf < 10 ? dLeaf : 10;
            int64_t nblk = (int64_t)1 << (dLeaf - L);
            uint8_t *oMin = bs->mmMin[round & 1], *oMax = bs->mmMax[round & 1];
            if (round == 0)
                hipLaunchKernelGGL(k_pyramid<true>, dim3((unsigned)nblk, B), dim3(256), 0, st, bs->g, dLeaf, L, vox,
                                   inMin, inMax, inStride, bs->mid.temp, bs->heapStride, mr ? bs->rng.temp : nullptr,
                                   oMin, oMax, oStride, bs->generalGeom ? bs->srcIdx : nullptr);
            else
                hipLaunchKernelGGL(k_pyramid<false>, dim3((unsigned)nblk, B), dim3(256), 0, st, bs->g, dLeaf, L, vox,
                                   inMin, inMax, inStride, bs->mid.temp, bs->heapStride, mr ? bs->rng.temp : nullptr,
                                   oMin, oMax, oStride, nullptr);
            dLeaf -= L;
            rootMinP = oMin; rootMaxP = oMax;
            if (dLeaf == 0) break;
            inMin = oMin; inMax = oMax; inStride = oStride;
            ++round;
        }
    }
    hipEventRecord(bs->ev[1], st);
    dbg_sync(st, "pyramid");
    // ---- COMPRESS
    // constant bricks take the closed form (both streams of a MidRangeTree too; needs the leaf prune and an epoch to exist)
    const bool constOk = bs->maxEpochs >= 1 && bs->tolerance >= 1 && D >= 1;
    const SkipBlocks sk{skipOn ? bs->blockFlag : nullptr, (int64_t)1 << (D >= 12 ? D - 12 : 0), D - 2};
    const SkipBlocks skR{skipOn && mr ? bs->blockFlagR : nullptr, sk.nBlk, D - 2};
    // MidRangeTree: the two streams' level loops do not depend on each other (M.cpp:399-544 runs them one after the
    // other): the half-range stream's goes to the set's second stream, so its one-wave-per-brick walkers run beside
    // the mid stream's wide kernels and the other way round
    const bool forkR = mr && bs->blockErrR && bs->estSummR && !bs->sw.mrSerial && ensure_aux(bs, 1) == 1;
    if (forkR) {
        hipEventRecord(bs->evFork, st);
        hipStreamWaitEvent(bs->aux, bs->evFork, 0);
        compress_stream(bs, bs->rng, bs->aux, constOk ? rootMinP : nullptr, constOk ? rootMaxP : nullptr, rootStride, skR,
                        bs->blockErrR, bs->estSummR);
        hipEventRecord(bs->evJoin, bs->aux);
    }
    // VolumeKdtree: the same trick over ranges of the bricks (vr_brickset_set_concurrency; 2 by default)
    int parts = bs->levelLoopStreams;
    if (bs->sw.forkBricks > 0) parts = bs->sw.forkBricks;
    if (parts > 4) parts = 4;
    if (mr || B < 16 * parts || parts < 2) parts = 1;
    if (parts > 1) parts = 1 + ensure_aux(bs, parts - 1);
    if (parts > 1) {
        hipEventRecord(bs->evFork, st);
        for (int p = 1; p < parts; ++p) {
            const int b0 = (int)((int64_t)B * p / parts), b1 = (int)((int64_t)B * (p + 1) / parts);
            hipStreamWaitEvent(bs->auxN[p - 1], bs->evFork, 0);
            compress_stream(bs, bs->mid, bs->auxN[p - 1], constOk ? rootMinP : nullptr, constOk ? rootMaxP : nullptr, rootStride, sk,
                            bs->blockErr, bs->estSumm, b0, b1 - b0);
            hipEventRecord(bs->evJoinN[p - 1], bs->auxN[p - 1]);
        }
        compress_stream(bs, bs->mid, st, constOk ? rootMinP : nullptr, constOk ? rootMaxP : nullptr, rootStride, sk, bs->blockErr,
                        bs->estSumm, 0, (int)((int64_t)B / parts));
        for (int p = 1; p < parts; ++p) hipStreamWaitEvent(st, bs->evJoinN[p - 1], 0);
    } else
    compress_stream(bs, bs->mid, st, constOk ? rootMinP : nullptr, constOk ? rootMaxP : nullptr, rootStride, sk, bs->blockErr, bs->estSumm);
    if (forkR) hipStreamWaitEvent(st, bs->evJoin, 0);
    else if (mr) compress_stream(bs, bs->rng, st, constOk ? rootMinP : nullptr, constOk ? rootMaxP : nullptr, rootStride, skR,
                                 bs->blockErr, bs->estSumm);
    hipEventRecord(bs->ev[2], st);
    dbg_sync(st, "compress");
    // ---- PRUNE
    hipLaunchKernelGGL(k_chain_lut, dim3(1), dim3(256), 0, st, bs->tolerance, bs->maxDepth - D, bs->chainLut);
    ReconBufs rb{{bs->mid.recon[0], bs->mid.recon[1], bs->mid.recon[2]}};
    ReconBufs rbR{{bs->rng.recon[0], bs->rng.recon[1], bs->rng.recon[2]}};
    int pruneFrom = D - 1;
    bs->fineHas.assign((size_t)B, 0);
    if (fused) {
        PruneEmitArgs pa;
        pa.sk = sk; pa.skR = skR;
        pa.D = D; pa.tol = bs->tolerance; pa.maxDepth = bs->maxDepth; pa.ctrls = bs->mid.ctrl;
        pa.temp = bs->mid.temp; pa.codes = bs->mid.codes;
        pa.heapStride = bs->heapStride; pa.codeStride = bs->codeStride; pa.leafStride = bs->reconStride;
        pa.rb = rb; pa.subTok = bs->blockOff; pa.nEmitBlk = bs->nEmitBlk; pa.blockL1 = bs->blockL1;
        pa.chainLut = bs->chainLut; pa.idxOff = bs->idxOff; pa.nIdx = bs->nIdx;
        if (!bs->fineIdx && hipMalloc(&bs->fineIdx, (size_t)B * bs->nIdx * 16) != hipSuccess) return -3;
        if (!bs->idxVal3 && hipMalloc(&bs->idxVal3, (size_t)B * bs->nIdx * 8) != hipSuccess) return -3;
        pa.fineIdx = (uint32_t *)bs->fineIdx;
        pa.ctrlsR = mr ? bs->rng.ctrl : nullptr; pa.tempR = mr ? bs->rng.temp : nullptr; pa.codesR = mr ? bs->rng.codes : nullptr;
        pa.rbR = rbR;
        pa.gap = bs->mid.tree; pa.gapR = mr ? bs->rng.tree : nullptr; pa.treeCap = bs->treeCap;
        const dim3 peGrid((unsigned)((int64_t)1 << (D - 12)), B);
        if (bs->leafless) {
            hipLaunchKernelGGL((k_prune_emit12<false, true>), peGrid, dim3(256), 0, st, pa);
            if (mr) hipLaunchKernelGGL((k_prune_emit12<true, true>), peGrid, dim3(256), 0, st, pa);
        } else {
            hipLaunchKernelGGL((k_prune_emit12<false, false>), peGrid, dim3(256), 0, st, pa);
            if (mr) hipLaunchKernelGGL((k_prune_emit12<true, false>), peGrid, dim3(256), 0, st, pa);
        }
        pruneFrom = D - 13;
        bs->fineHas.assign((size_t)B, 1);
    } else if (D >= 12) {
        hipLaunchKernelGGL(k_prune12, dim3((unsigned)((int64_t)1 << (D - 12)), B), dim3(256), 0, st, D, bs->tolerance,
                           bs->mid.ctrl, bs->mid.temp, bs->mid.codes, mr ? bs->rng.codes : nullptr, bs->heapStride,
                           bs->codeStride, rb, bs->reconStride, bs->maxDepth, (!mr && bs->K >= 2) ? bs->blockOff : nullptr,
                           bs->nEmitBlk,    // per-brick stride of the block arrays (same as EmitArgs::nEmitBlk)
                           bs->blockL1, bs->chainLut);
        pruneFrom = D - 13;
    } else
        hipLaunchKernelGGL(k_prune_leaf, dim3(cdiv((int64_t)1 << D, 256), B), dim3(256), 0, st, D, bs->tolerance,
                           bs->mid.ctrl, bs->mid.temp, bs->mid.codes, mr ? bs->rng.codes : nullptr, bs->heapStride,
                           bs->codeStride, rb, bs->reconStride, bs->maxDepth, bs->blockL1, bs->nEmitBlk);
    for (int d = pruneFrom; d >= 0; --d)
        hipLaunchKernelGGL(k_prune_level, dim3(cdiv((int64_t)1 << d, 256), B), dim3(256), 0, st, d, bs->mid.ctrl, bs->mid.codes,
                           mr ? bs->rng.codes : nullptr, bs->codeStride);
    hipLaunchKernelGGL(k_fix_chain_distances, dim3(B), dim3(64), 0, st, D, bs->maxDepth, bs->mid.ctrl,
                       mr ? bs->rng.ctrl : nullptr);
    hipEventRecord(bs->ev[3], st);
    dbg_sync(st, "prune");
    // ---- CONVERT
    EmitArgs a;
    fill_emit_args(bs, a);
    if (bs->idx64 && !fused) {      // the level-synchronous emitters write absolute (32-bit) index entries: zero bases
        a.blockOff64 = nullptr; a.idxBase = nullptr;
        hipMemsetAsync(bs->idxBase, 0, (size_t)B * bs->nEmitBlk * sizeof(unsigned long long), st);
    }
    const bool quad = (!mr || fused) && D >= 12 && bs->K >= 2;
    const int64_t nblk = cdiv((int64_t)1 << D, fused ? 4096 : (quad ? EMIT4_RANKS : EMIT_RANKS_PER_BLOCK));
    if (quad) hipLaunchKernelGGL(k_block_alive, dim3(cdiv(nblk, 256), B), dim3(256), 0, st, a, nblk, fused ? 12 : 10);   // + token counts
    else hipLaunchKernelGGL(k_emit_count, dim3((unsigned)nblk, B), dim3(EMIT_RANKS_PER_BLOCK), 0, st, a);
    dbg_sync(st, "block_alive/count");
    hipLaunchKernelGGL(k_emit_scan, dim3(B), dim3(1024), 0, st, a, nblk);
    dbg_sync(st, "emit_scan");
    // fused: the strings stay in their slots of the gapped buffer, the decode index points there (k_index12); the
    // contiguous stream is made when the host asks for it (compact_launch)
    bs->gapped = fused;
    bs->compactValid = false;
    if (fused) hipLaunchKernelGGL(k_index12, dim3(cdiv(nblk, 4), B), dim3(256), 0, st, a, (uint32_t)nblk);
    else {
        hipLaunchKernelGGL(k_emit_zero, dim3(cdiv(nblk, 256), B), dim3(256), 0, st, a, nblk);
        if (quad) hipLaunchKernelGGL(k_emit4<true>, dim3((unsigned)nblk, B), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_emit_write, dim3((unsigned)nblk, B), dim3(EMIT_RANKS_PER_BLOCK), 0, st, a);
    }
    // statistics records: one per 1024 leaves from k_prune12, one per 256 from k_prune_leaf
    hipLaunchKernelGGL(k_emit_stats, dim3(B), dim3(1024), 0, st, a, (int64_t)(D >= 12 ? cdiv((int64_t)1 << D, 1024) : cdiv((int64_t)1 << D, 256)));
    hipLaunchKernelGGL(k_const_finish, dim3(cdiv(bs->nIdx, 256), B), dim3(256), 0, st, D, bs->mid.ctrl, bs->mid.tree,
                       bs->treeCap, bs->idxOff, bs->idxVal, bs->nIdx);
    if (mr) hipLaunchKernelGGL(k_const_finish_range, dim3(B), dim3(64), 0, st, D, bs->rng.ctrl, bs->rng.tree, bs->treeCap);
    // the reference's build() ends with the contiguous stream (tree.swap(preorderTree), R.cpp:714-718): so does this one,
    // unless the caller turned it off (vr_brickset_set_compaction): the decoders read the block strings where they are
    if (fused && bs->compactOnBuild) {
        const int rc = compact_launch(bs, st);
        if (rc != 0) return rc;
    }
    hipEventRecord(bs->ev[4], st);
    dbg_sync(st, "emit_write");
    return launch_status("encode");
}

} // namespace vr
