// brickset.h -- host-side state behind the opaque vr_brickset handle.
#pragma once
#include "kd_common.h"
#include <string>
#include <vector>

namespace vr {

struct Stream2 {              // one 2-bit stream (mid, or MidRangeTree's half-range stream)
    uint8_t *temp = nullptr;  // B * heapStride   truth heap (midrange / half range), 1-based
    uint8_t *codes = nullptr; // B * codeStride   2-bit codes, four per byte (TwoBitArray packing), 1-based heap (BFS)
    uint8_t *recon[3] = {nullptr, nullptr, nullptr}; // B * reconStride each: parents + two level buffers
    Ctrl *ctrl = nullptr;     // B
    uint8_t *tree = nullptr;  // B * treeCap      the stream the decoders read: the reference's contiguous preorder stream (TwoBitArray
                              //                  packing) or, after a fused build, its block-gapped form (BrickSet::gapped)
    uint8_t *treeCompact = nullptr; // BrickSet::compactCap bytes: fused builds: the reference's contiguous streams, brick b at byte
                                    // BrickSet::brickOff[b] (compact_launch; at the end of build() unless compaction was turned off)
};

// Debugging / experiment switches.  Read from the environment (VRHIP_<NAME>) ONCE, when a set is created, or set
// explicitly with vr_brickset_set_switch -- never in a launch path: the kernels a handle uses do not change behind
// the caller's back, and buffers sized under one setting are not used under another.
struct Switches {
    bool decodeV1 = false;       // VRHIP_DECODE_V1: k_decode_lane for everything
    bool decodeWalk = false;     // VRHIP_DECODE_WALK: k_decode_tile (no per-4-leaf counts)
    bool decodeFineV1 = false;   // VRHIP_DECODE_FINE_V1: round 1's k_decode_fine
    bool decodeQuad = false;     // VRHIP_DECODE_QUAD: round 2's k_decode_quad instead of k_decode_region
    bool noFusedEmit = false;    // VRHIP_NO_FUSED_EMIT
    bool noSkipBlocks = false;   // VRHIP_NO_SKIP_BLOCKS
    bool noSwz = false;          // VRHIP_NOSWZ
    bool mrSerial = false;       // VRHIP_MR_SERIAL
    int forkBricks = 0;          // VRHIP_FORK_BRICKS (0: default)
};

struct BrickSet {
    int32_t B = 0;
    Switches sw;
    Geom g{};
    int32_t D = 0, maxDepth = 0;
    int32_t tolerance = 6, maxEpochs = 5, variant = 0;
    int32_t K = 0;            // decode index granularity: subtrees of 2^K leaves
    int32_t Ds = 0;           // D - K
    int64_t heapStride = 0;   // 2^(D+1)
    int64_t leafStride = 0;   // 2^D
    int64_t codeStride = 0;   // bytes of packed BFS codes per brick: heapStride / 4, 4-byte aligned (leafless builds: the levels above the leaves only)
    int64_t reconStride = 0;  // bytes per brick of a reconstruction buffer: 2^D, or 2^(D-1) in a leafless build
    // A fused build (k_prune_emit12) never stores the leaf level's codes and reconstruction: its leaf-level fills only
    // sum errors, and the prune recomputes both from (truth, parent's reconstruction, the distances the level loop ended
    // with: Ctrl::finalReconDist / finalCodesDist).  Decided when the encoder's buffers are allocated (first build).
    bool leafless = false;
    int64_t treeCap = 0;      // bytes per brick reserved for the preorder stream
    int64_t nIdx = 0;         // 2^Ds index entries per brick

    Stream2 mid, rng;
    uint8_t *mmMin[2] = {nullptr, nullptr}, *mmMax[2] = {nullptr, nullptr}; // pyramid carry arrays
    unsigned long long *blockErr = nullptr; // B * nErrBlk : per-block sum err^2 of the fill pass
    int64_t nErrBlk = 0;
    void *estSumm = nullptr;     // B * estSummStride EstSummary records (estimator, kd_encode.hip)
    int64_t estSummStride = 0;
    unsigned long long *blockL1 = nullptr; // B * nEmitBlk
    uint8_t *blockFlag = nullptr;          // B * 2^(D-12): kd_encode.hip SkipBlocks (constant / skipped 4096-leaf blocks)
    uint8_t *blockFlagR = nullptr;         // ... of a MidRangeTree's half-range stream
    uint8_t *blockAlive = nullptr, *blockVal = nullptr;   // B * nEmitBlk
    unsigned long long *blockSpine = nullptr, *blockSpineR = nullptr; // B * nEmitBlk (R: MidRangeTree's range stream)
    uint32_t *chainLut = nullptr;   // 256: grown branch of a leaf by its initial error (k_chain_lut)
    uint32_t *blockTot = nullptr, *blockOff = nullptr; // B * nEmitBlk
    bool idx64 = false;            // more than 2^32 tokens possible (origTreeDepth > 28; VRHIP_FORCE_IDX64=1 for tests): 64-bit scan,
    unsigned long long *blockOff64 = nullptr, *idxBase = nullptr;   // block-relative index entries + per-block bases (B * nEmitBlk each)
    int64_t nEmitBlk = 0;
    uint32_t *idxOff = nullptr; // B * nIdx  token offset of each depth-Ds subtree root (VR_IDX_DEAD: inside a pruned region)
    uint8_t *idxVal = nullptr;  // B * nIdx  decoded scalar of that root (or of the pruned ancestor)
    uint8_t *idxValCut = nullptr; // B * nIdx  progressive cut above the index level: ancestor scalars
    uint8_t *fineIdx = nullptr;   // B * nIdx * 16  tokens owned by each 4-leaf subtree of a depth-Ds node (fused encoder only)
    std::vector<uint8_t> fineHas; // per brick: fineIdx describes its current stream (all set -> k_decode_fine)
    uint32_t *decTables = nullptr; // B * FD_TABLE_WORDS: k_decode_fine's tables of every brick for the current cut
    uint8_t *idxVal3 = nullptr;   // B * nIdx * 8   decoded scalar of every depth-(D-3) node (k_decode_quad; with fineIdx)
    uint32_t *chainTab = nullptr; // 8 x 16384 entries: k_decode_quad's grown-branch tables, one per number of refining levels (0..7),
    bool chainTabReady = false;   // all written once (never rewritten: decodes of one set on several streams may share them)
    std::vector<std::vector<uint8_t>> hostTree; // foreign streams keep their bytes for progressive cuts
    uint32_t *lut = nullptr;    // 2^K : local rank -> packed (dx | dy<<10 | dz<<20)
    bool foreignRange = false;   // a foreign MidRangeTree file also supplied the range stream
    uint32_t *spread = nullptr;  // rank bits of every x, y, z coordinate: rank(x,y,z) = spread[x] | spread[X+y] | spread[X+Y+z]
    // General extents (any axis not a power of two, or longer than 1024): the reference's split rule (R.cpp:151-162)
    // then gives boxes of unequal size, an axis order that differs from node to node, leaves that span two cells or
    // none.  The tree itself is still the complete binary heap of 2^D leaves, so every rank-domain kernel (level loop,
    // prune, emit, stream parse) runs unchanged; only the two ends that touch voxels go through tables:
    bool generalGeom = false;
    uint32_t *srcIdx = nullptr;    // 2^D: voxel a leaf reads, the min corner of its build box (R.cpp:194-195)
    uint32_t *ownerRank = nullptr; // X*Y*Z: the leaf whose decoded value a voxel ends up with: the last one in preorder
                                   // whose levelCut box covers it (R.cpp:759-766, 790-799, 821-830)
    uint8_t *ownerSurv = nullptr;  // X*Y*Z: how many halvings of its leaf's box along a grown branch the voxel survives
    uint8_t *rankVals = nullptr;   // B * 2^D * 2: decoded value | branch nodes << 8 of every leaf rank (general-extent decode scratch)

    std::vector<Ctrl> hostCtrl; // copied back lazily
    std::vector<int32_t> hostRevertsR;   // MidRangeTree: reverted epochs of the half-range stream, per brick (sync_ctrl)
    uint32_t lutZeroRun = 0;    // chainLut[256]: table entries that would need the zero-run rewrite (always 0)
    bool encoderReady = false;  // every buffer of ensure_encoder_buffers (capi.hip) is allocated
    bool built = false, hostCtrlValid = false;
    bool gapped = false;        // mid.tree / rng.tree hold 4096-leaf block strings in fixed slots (kd_encode.hip PE_WORDS)
    bool compactValid = false;  // treeCompact holds the current build's contiguous stream (unless compactOverflow says it did not fit)
    bool compactOnBuild = true; // build() ends with the contiguous stream, like the reference's (tree.swap(preorderTree), R.cpp:714-718)
    int64_t compactCap = 0;     // bytes of each treeCompact buffer: sized from the streams' real lengths, not from their worst case
    unsigned long long *brickOff = nullptr;   // B + 1 (device): byte offset of every brick's stream in treeCompact (16-byte aligned); [B] = total
    int32_t *compactOverflow = nullptr;       // device flag: the streams did not fit compactCap (nothing was written; the host regrows and repeats)
    std::vector<unsigned long long> hostBrickOff;
    bool foreign = false;       // stream installed by set_tree/open (no encoder state)
    std::vector<int64_t> openTreeBytes; // per brick: tree.bits size as the reference's open() would have it
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // MidRangeTree: the half-range stream's level loop runs on a stream of its own beside the mid stream's (fork / join
    // by events around the two compress_stream calls), with its own partial-sum and estimator scratch
    hipStream_t aux = nullptr, auxN[3] = {nullptr, nullptr, nullptr};     // aux == auxN[0]
    hipEvent_t evFork = nullptr, evJoin = nullptr, evJoinN[3] = {nullptr, nullptr, nullptr};
    int levelLoopStreams = 2;      // VolumeKdtree: brick ranges whose level loops run side by side (1 = none); vr_brickset_set_concurrency
    unsigned long long *blockErrR = nullptr;
    void *estSummR = nullptr;
    float phasesMs[5] = {0, 0, 0, 0, 0};
    bool timingsPending = false, decodeTimingPending = false;
    void *lastStream = nullptr;
};

// kd_encode.hip
int encode_launch(BrickSet *bs, const uint8_t *voxDev, hipStream_t st);
int compact_launch(BrickSet *bs, hipStream_t st);   // fused builds: contiguous stream(s) into Stream2::treeCompact
// kd_decode.hip
int decode_launch(BrickSet *bs, uint8_t *outDev, int cutDepth, hipStream_t st, bool rangeStream = false);
int cut_values_from_stream(BrickSet *bs, const uint8_t *treeHost, int64_t numActive, const uint8_t *dmapHost, int cut,
                           std::vector<uint8_t> &vals);
int build_index_from_stream(BrickSet *bs, int brick, const uint8_t *treeHost, int64_t numActive,
                            const uint8_t *dmapHost, std::vector<uint32_t> &offs, std::vector<uint8_t> &vals,
                            std::vector<uint8_t> &fine, std::vector<uint8_t> &val3);
void make_geom(Geom &g, const int64_t dims[3]);
int build_general_geometry(BrickSet *bs);   // srcIdx / ownerRank for general extents (0, -3 out of memory, -1 device error)
void make_lut(const Geom &g, int K, std::vector<uint32_t> &lut);

} // namespace vr
