// capi.hip -- the extern "C" boundary declared in include/vrhip.h.  Thin: argument
// checks, device memory management, launches (kd_encode/kd_decode/raymarch.hip) and
// the reference's file format.  No CPU compute path exists behind these entry points.
#include "../../include/vrhip.h"
#include "brickset.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <map>
#include <mutex>
#include <new>

using namespace vr;

namespace vr {
int raycast_launch(const uint8_t *, const int64_t dims[3], const vr_camera *, const vr_render_params *, float *, hipStream_t);
int composite_over_launch(float *, const float *, int64_t, hipStream_t);
int skip_grid_launch(const uint8_t *, const int64_t dims[3], int, uint8_t *, hipStream_t);
int composite_finish_launch(const float *, float *, int64_t, hipStream_t);
int composite_slabs_launch(const float *, int, int64_t, int64_t, int, const vr_camera *, const vr_render_params *, float *, hipStream_t);
int assemble_launch(bool, const uint8_t *, uint8_t *, int, const int64_t bd[3], const int64_t *, const int64_t grid[3], hipStream_t);
int measure_error_launch(const uint8_t *, const uint8_t *, int64_t, int *, unsigned long long *, hipStream_t);
int query_error_launch(const uint8_t *, const uint8_t *, int64_t, uint8_t *, hipStream_t);
extern std::atomic<int> g_skipGridV1;
}

struct vr_brickset { BrickSet s; };

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_ == hipErrorOutOfMemory ? VR_ERR_OOM : VR_ERR_NO_DEVICE; } while (0)

static bool device_ok()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if ((e != hipSuccess || n <= 0) && getenv("VRHIP_DEBUG"))
        fprintf(stderr, "[vrhip] hipGetDeviceCount: %s (n=%d)\n", hipGetErrorString(e), n);
    return e == hipSuccess && n > 0;
}

extern "C" {

vr_status vr_device_count(int32_t *count)
{
    if (!count) return VR_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return VR_OK;
}

vr_status vr_set_device(int32_t device)
{
    if (!device_ok()) return VR_ERR_NO_DEVICE;
    HIPCHK(hipSetDevice(device));
    return VR_OK;
}

const char *vr_status_string(vr_status s)
{
    switch (s) {
    case VR_OK: return "ok";
    case VR_ERR_INVALID: return "invalid argument";
    case VR_ERR_NO_DEVICE: return "no usable HIP device (there is no CPU fallback)";
    case VR_ERR_OOM: return "out of device memory";
    case VR_ERR_IO: return "file error";
    case VR_ERR_STATE: return "wrong state (no tree built / loaded)";
    case VR_ERR_FORMAT: return "malformed tree stream";
    case VR_ERR_UNSUPPORTED: return "unsupported dimensions or option";
    default: return "unknown";
    }
}

const char *vr_version(void) { return "vrhip 0.1 (gfx950)"; }

vr_status vr_malloc(void **dev, int64_t bytes)
{
    if (!dev || bytes <= 0) return VR_ERR_INVALID;
    if (!device_ok()) return VR_ERR_NO_DEVICE;
    HIPCHK(hipMalloc(dev, (size_t)bytes));
    return VR_OK;
}
vr_status vr_free(void *dev)
{
    if (dev) hipFree(dev);
    return VR_OK;
}
vr_status vr_upload(void *dst, const void *src, int64_t bytes, void *stream)
{
    if (!dst || !src || bytes <= 0) return VR_ERR_INVALID;
    if (!device_ok()) return VR_ERR_NO_DEVICE;
    HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return VR_OK;
}
vr_status vr_download(void *dst, const void *src, int64_t bytes, void *stream)
{
    if (!dst || !src || bytes <= 0) return VR_ERR_INVALID;
    if (!device_ok()) return VR_ERR_NO_DEVICE;
    HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return VR_OK;
}

static void free_stream2(Stream2 &s)
{
    hipFree(s.treeCompact);
    hipFree(s.temp); hipFree(s.codes);
    for (int i = 0; i < 3; ++i) hipFree(s.recon[i]);
    hipFree(s.ctrl); hipFree(s.tree);
    s = Stream2();
}

static vr_status alloc_stream2(BrickSet &b, Stream2 &s, bool encoder)
{
    const size_t B = (size_t)b.B;
    HIPCHK(hipMalloc(&s.ctrl, B * sizeof(Ctrl)));
    HIPCHK(hipMemset(s.ctrl, 0, B * sizeof(Ctrl)));
    HIPCHK(hipMalloc(&s.tree, B * (size_t)b.treeCap));
    (void)encoder;      // (the encoder's arrays are made by ensure_encoder_buffers at the first build)
    return VR_OK;
}

// everything ensure_encoder_buffers allocates (an opened MidRangeTree file owns rng.ctrl / rng.tree: kept)
static void free_encoder_buffers(BrickSet &b)
{
    auto drop = [](auto *&p) { if (p) hipFree(p); p = nullptr; };
    Stream2 *ss[2] = {&b.mid, &b.rng};
    for (Stream2 *s : ss) {
        drop(s->temp); drop(s->codes);
        for (int i = 0; i < 3; ++i) drop(s->recon[i]);
    }
    for (int i = 0; i < 2; ++i) { drop(b.mmMin[i]); drop(b.mmMax[i]); }
    drop(b.blockErr); drop(b.estSumm); drop(b.blockErrR); drop(b.estSummR); drop(b.blockL1); drop(b.blockFlag); drop(b.blockFlagR); drop(b.blockAlive); drop(b.blockVal); drop(b.blockSpine); drop(b.blockSpineR);
    drop(b.chainLut); drop(b.blockTot); drop(b.blockOff); drop(b.blockOff64); drop(b.idxBase);
    b.encoderReady = false;
}

static vr_status alloc_encoder_buffers(BrickSet &b)
{
    const size_t B = (size_t)b.B;
    // a fused build (k_prune_emit12: D >= 12, 64-leaf index granularity, not switched off) with a level loop that runs
    // keeps nothing of the leaf level: codes and reconstruction arrays end one level higher (BrickSet::leafless).
    // The switch is final here: vr_brickset_set_switch("no_fused_emit") is refused after the first build.
    b.leafless = b.D >= 12 && b.K == 6 && !b.sw.noFusedEmit && b.maxEpochs >= 1;
    b.reconStride = b.leafless ? b.leafStride / 2 : b.leafStride;
    b.codeStride = b.leafless ? b.leafStride / 4 + 16 : ((b.heapStride + 15) / 16) * 4;
    HIPCHK(hipMalloc(&b.mid.temp, B * (size_t)b.heapStride));
    HIPCHK(hipMalloc(&b.mid.codes, B * (size_t)b.codeStride));
    for (int i = 0; i < 3; ++i) HIPCHK(hipMalloc(&b.mid.recon[i], B * (size_t)b.reconStride));
    if (b.variant == VR_VARIANT_MIDRANGE) {
        if (!b.rng.ctrl) HIPCHK(hipMalloc(&b.rng.ctrl, B * sizeof(Ctrl)));       // (an opened MidRangeTree file has these already)
        if (!b.rng.tree) HIPCHK(hipMalloc(&b.rng.tree, B * (size_t)b.treeCap));
        HIPCHK(hipMalloc(&b.rng.temp, B * (size_t)b.heapStride));
        HIPCHK(hipMalloc(&b.rng.codes, B * (size_t)b.codeStride));
        for (int i = 0; i < 3; ++i) HIPCHK(hipMalloc(&b.rng.recon[i], B * (size_t)b.reconStride));
    }
    const int64_t mm = (int64_t)1 << (b.D > 10 ? b.D - 10 : 0);
    for (int i = 0; i < 2; ++i) {
        HIPCHK(hipMalloc(&b.mmMin[i], B * (size_t)mm));
        HIPCHK(hipMalloc(&b.mmMax[i], B * (size_t)mm));
    }
    b.nErrBlk = ((int64_t)1 << b.D) / 1024 > 0 ? ((int64_t)1 << b.D) / 1024 : 1;
    HIPCHK(hipMalloc(&b.blockErr, 3 * B * (size_t)b.nErrBlk * sizeof(unsigned long long)));     // + the central difference's two planes (Ctrl::altSel)
    b.estSummStride = ((int64_t)1 << b.D) / 1024 > 0 ? ((int64_t)1 << b.D) / 1024 : 1;
    HIPCHK(hipMalloc(&b.estSumm, B * (size_t)b.estSummStride * 128));
    if (b.variant == VR_VARIANT_MIDRANGE) {
        HIPCHK(hipMalloc(&b.blockErrR, 3 * B * (size_t)b.nErrBlk * sizeof(unsigned long long)));
        HIPCHK(hipMalloc(&b.estSummR, B * (size_t)b.estSummStride * 128));
    }
    b.nEmitBlk = (((int64_t)1 << b.D) + 255) / 256;
    HIPCHK(hipMalloc(&b.blockL1, B * (size_t)b.nEmitBlk * sizeof(unsigned long long)));
    if (b.D >= 12) HIPCHK(hipMalloc(&b.blockFlag, B * ((size_t)1 << (b.D - 12))));
    if (b.D >= 12 && b.variant == VR_VARIANT_MIDRANGE) HIPCHK(hipMalloc(&b.blockFlagR, B * ((size_t)1 << (b.D - 12))));
    HIPCHK(hipMalloc(&b.blockAlive, B * (size_t)b.nEmitBlk));
    HIPCHK(hipMalloc(&b.blockVal, B * (size_t)b.nEmitBlk));
    HIPCHK(hipMalloc(&b.blockSpine, B * (size_t)b.nEmitBlk * sizeof(unsigned long long)));
    if (b.variant == VR_VARIANT_MIDRANGE) HIPCHK(hipMalloc(&b.blockSpineR, B * (size_t)b.nEmitBlk * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&b.chainLut, (260 + 512) * sizeof(uint32_t)));   // 256 entries + [256]: entries that would need the zero-run rewrite; + 512 keyed by the signed error
    HIPCHK(hipMalloc(&b.blockTot, B * (size_t)b.nEmitBlk * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&b.blockOff, B * (size_t)b.nEmitBlk * sizeof(uint32_t)));
    if (b.idx64) {
        HIPCHK(hipMalloc(&b.blockOff64, B * (size_t)b.nEmitBlk * sizeof(unsigned long long)));
        if (!b.idxBase) HIPCHK(hipMalloc(&b.idxBase, B * (size_t)b.nEmitBlk * sizeof(unsigned long long)));    // (set_tree may have made it)
    }
    return VR_OK;
}

// "ready" is a flag of its own, set after the last allocation: a failure part-way (out of memory) releases what
// was taken, so the next build retries from scratch and reports the same error instead of launching kernels on
// null side buffers.
static vr_status ensure_encoder_buffers(BrickSet &b)
{
    if (b.encoderReady) return VR_OK;
    const vr_status rc = alloc_encoder_buffers(b);
    if (rc != VR_OK) { (void)hipGetLastError(); free_encoder_buffers(b); return rc; }
    b.encoderReady = true;
    return VR_OK;
}

vr_status vr_brickset_destroy(vr_brickset *h)
{
    if (!h) return VR_OK;
    BrickSet &b = h->s;
    free_encoder_buffers(b);
    free_stream2(b.mid);
    free_stream2(b.rng);
    hipFree(b.brickOff); hipFree(b.compactOverflow);
    hipFree(b.idxOff); hipFree(b.idxVal); hipFree(b.idxValCut); hipFree(b.fineIdx); hipFree(b.idxVal3); hipFree(b.chainTab); hipFree(b.decTables); hipFree(b.lut); hipFree(b.spread); hipFree(b.srcIdx); hipFree(b.ownerRank); hipFree(b.ownerSurv); hipFree(b.rankVals);
    for (int i = 0; i < 8; ++i) if (b.ev[i]) hipEventDestroy(b.ev[i]);
    if (b.evFork) hipEventDestroy(b.evFork);
    for (int i = 0; i < 3; ++i) { if (b.evJoinN[i]) hipEventDestroy(b.evJoinN[i]); if (b.auxN[i]) hipStreamDestroy(b.auxN[i]); }
    delete h;
    return VR_OK;
}

static bool pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

vr_status vr_brickset_create(vr_brickset **out, int32_t num_bricks, const int64_t dims[3], int32_t tolerance,
                             int32_t max_epochs, int32_t variant)
{
    if (!out || !dims || num_bricks <= 0) return VR_ERR_INVALID;
    if (tolerance < 0 || max_epochs < 0 || variant < 0 || variant > 2) return VR_ERR_INVALID;
    for (int k = 0; k < 3; ++k) if (dims[k] <= 0) return VR_ERR_INVALID;
    // power-of-two extents up to 1024 per axis take the tiled kernels; anything else (the reference accepts any
    // extents, R.cpp:26-36,151-162) goes through the general-extent tables.  Limits: origTreeDepth <= 31, fewer than
    // 2^32 voxels per tree, and above depth 28 (64-bit token offsets) one tree per set.
    bool general = !pow2(dims[0]) || !pow2(dims[1]) || !pow2(dims[2]) || dims[0] > 1024 || dims[1] > 1024 || dims[2] > 1024;
    if (dims[0] > (1ll << 20) || dims[1] > (1ll << 20) || dims[2] > (1ll << 20) || dims[0] * dims[1] * dims[2] >= (1ll << 32))
        return VR_ERR_UNSUPPORTED;
    if (!device_ok()) return VR_ERR_NO_DEVICE;
    vr_brickset *h = new (std::nothrow) vr_brickset();
    if (!h) return VR_ERR_OOM;
    BrickSet &b = h->s;
    b.B = num_bricks;
    {   // the VRHIP_* debugging switches are read here, once per set (brickset.h Switches)
        Switches &w = b.sw;
        w.decodeV1 = getenv("VRHIP_DECODE_V1") != nullptr;
        w.decodeWalk = getenv("VRHIP_DECODE_WALK") != nullptr;
        w.decodeFineV1 = getenv("VRHIP_DECODE_FINE_V1") != nullptr;
        w.decodeQuad = getenv("VRHIP_DECODE_QUAD") != nullptr;
        w.noFusedEmit = getenv("VRHIP_NO_FUSED_EMIT") != nullptr;
        w.noSkipBlocks = getenv("VRHIP_NO_SKIP_BLOCKS") != nullptr;
        w.noSwz = getenv("VRHIP_NOSWZ") != nullptr;
        w.mrSerial = getenv("VRHIP_MR_SERIAL") != nullptr;
        if (const char *e = getenv("VRHIP_FORK_BRICKS")) { const long v = strtol(e, nullptr, 10); w.forkBricks = v >= 1 && v <= 4 ? (int)v : 0; }
    }
    make_geom(b.g, dims);
    b.D = b.g.D;
    if (b.D > 31) { delete h; return VR_ERR_UNSUPPORTED; }
    // deeper than 28 (the reference's own 2048x2048x768 tree is 31 deep, main.cpp:242-251): a stream can pass 2^32
    // tokens, so the emitter scans in 64 bits and the decode index goes block-relative; table-driven geometry only
    b.idx64 = b.D > 28 || (getenv("VRHIP_FORCE_IDX64") && b.D >= 12);      // (create-time only, like the switches above)
    if (b.idx64) general = true;       // the tiled decoders read absolute 32-bit index entries
    if (b.idx64 && num_bricks != 1 && b.D > 28) { delete h; return VR_ERR_UNSUPPORTED; }
    b.generalGeom = general;
    b.maxDepth = b.D + VR_CHAIN_LEVELS;
    b.tolerance = tolerance; b.maxEpochs = max_epochs; b.variant = variant;
    b.K = b.D < 6 ? b.D : 6;   // decode index granularity: 4x4x4 voxel subtrees
    b.Ds = b.D - b.K;
    b.heapStride = (int64_t)1 << (b.D + 1);
    b.leafStride = (int64_t)1 << b.D;
    b.codeStride = ((b.heapStride + 15) / 16) * 4;     // (both final when the encoder's buffers are made: alloc_encoder_buffers)
    b.reconStride = b.leafStride;
    const int64_t numMax = b.heapStride - 1 + VR_CHAIN_LEVELS * b.leafStride; // numMaxNodes R.cpp:35
    b.treeCap = ((numMax + 15) / 16 + 2) * 4 + 256;   // slack: the decoder stages whole words past a run's end
    if (b.D >= 12) {   // a fused build keeps every 4096-leaf block's string in a fixed slot of 2320 words (kd_encode.hip PE_WORDS)
        const int64_t gappedBytes = ((int64_t)1 << (b.D - 12)) * 2320 * 4 + 256;
        if (gappedBytes > b.treeCap) b.treeCap = gappedBytes;
    }
    b.nIdx = (int64_t)1 << b.Ds;
    vr_status rc = alloc_stream2(b, b.mid, false);
    if (rc == VR_OK) {
        hipError_t e = hipMalloc(&b.idxOff, (size_t)b.B * b.nIdx * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&b.idxVal, (size_t)b.B * b.nIdx);
        if (e == hipSuccess) e = hipMalloc(&b.idxValCut, (size_t)b.B * b.nIdx);
        if (e == hipSuccess && general) {
            const int grc = build_general_geometry(&b);
            if (grc != 0) e = grc == -3 ? hipErrorOutOfMemory : hipErrorUnknown;
        }
        if (e == hipSuccess) e = hipMalloc(&b.lut, ((size_t)1 << b.K) * sizeof(uint32_t));
        if (e == hipSuccess && !general) {
            std::vector<uint32_t> lut;
            make_lut(b.g, b.K, lut);
            e = hipMemcpy(b.lut, lut.data(), lut.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess && !general) e = hipMalloc(&b.spread, (size_t)(b.g.X + b.g.Y + b.g.Z) * sizeof(uint32_t));
        if (e == hipSuccess && !general) {
            // kernels never walk Geom::axis/bit (a dependent chain of loads from the kernel-argument
            // segment): a coordinate's contribution to the Morton rank comes from this table
            std::vector<uint32_t> sp((size_t)(b.g.X + b.g.Y + b.g.Z), 0u);
            const int off[3] = {0, b.g.X, b.g.X + b.g.Y}, ext[3] = {b.g.X, b.g.Y, b.g.Z};
            for (int ax = 0; ax < 3; ++ax)
                for (int v = 0; v < ext[ax]; ++v) {
                    uint32_t r = 0;
                    for (int d = 0; d < b.g.D; ++d) {
                        r <<= 1;
                        if (b.g.axis[d] == ax) r |= (uint32_t)(v >> b.g.bit[d]) & 1u;
                    }
                    sp[(size_t)off[ax] + v] = r;
                }
            e = hipMemcpy(b.spread, sp.data(), sp.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        }
        for (int i = 0; i < 8 && e == hipSuccess; ++i) e = hipEventCreate(&b.ev[i]);
        // (the internal streams of a build are created when a build first needs them: kd_encode.hip ensure_aux)
        if (e != hipSuccess) rc = e == hipErrorOutOfMemory ? VR_ERR_OOM : VR_ERR_NO_DEVICE;
    }
    if (rc != VR_OK) { vr_brickset_destroy(h); return rc; }
    b.hostCtrl.resize(b.B);
    b.openTreeBytes.assign(b.B, -1);
    *out = h;
    return VR_OK;
}

vr_status vr_brickset_set_switch(vr_brickset *h, const char *name, int32_t value)
{
    if (!h || !name) return VR_ERR_INVALID;
    Switches &w = h->s.sw;
    const bool on = value != 0;
    if (!strcmp(name, "decode_v1")) w.decodeV1 = on;
    else if (!strcmp(name, "decode_walk")) w.decodeWalk = on;
    else if (!strcmp(name, "decode_fine_v1")) w.decodeFineV1 = on;
    else if (!strcmp(name, "decode_quad")) w.decodeQuad = on;
    else if (!strcmp(name, "no_skip_blocks")) w.noSkipBlocks = on;
    else if (!strcmp(name, "noswz")) w.noSwz = on;
    else if (!strcmp(name, "mr_serial")) w.mrSerial = on;
    else if (!strcmp(name, "fork_bricks")) { if (value < 0 || value > 4) return VR_ERR_INVALID; w.forkBricks = value; }
    else if (!strcmp(name, "no_fused_emit")) {
        // the emitter decides the layout of the stream buffer and which side-cars exist: only before the first build
        if (h->s.built) return VR_ERR_STATE;
        w.noFusedEmit = on;
    } else return VR_ERR_INVALID;
    return VR_OK;
}

vr_status vr_brickset_set_compaction(vr_brickset *h, int32_t on_build)
{
    if (!h) return VR_ERR_INVALID;
    h->s.compactOnBuild = on_build != 0;
    return VR_OK;
}

vr_status vr_debug_set(const char *name, int32_t value)
{
    if (!name) return VR_ERR_INVALID;
    if (!strcmp(name, "skip_grid_v1")) { vr::g_skipGridV1.store(value ? 1 : 0); return VR_OK; }
    return VR_ERR_INVALID;
}

vr_status vr_brickset_set_error_tolerance(vr_brickset *h, int32_t tol)
{
    if (!h || tol < 0) return VR_ERR_INVALID;
    h->s.tolerance = tol;
    return VR_OK;
}
vr_status vr_brickset_set_max_epochs(vr_brickset *h, int32_t e)
{
    if (!h || e < 0) return VR_ERR_INVALID;
    h->s.maxEpochs = e;
    return VR_OK;
}

vr_status vr_brickset_build(vr_brickset *h, const uint8_t *vox, void *stream)
{
    if (!h || !vox) return VR_ERR_INVALID;
    BrickSet &b = h->s;
    // (vr_brickset_set_max_epochs(0) <-> >= 1 between two builds changes what the level loop keeps of the leaf level:
    // the encoder's arrays are made again for the other mode)
    if (b.encoderReady && b.leafless != (b.D >= 12 && b.K == 6 && !b.sw.noFusedEmit && b.maxEpochs >= 1)) {
        HIPCHK(hipDeviceSynchronize());
        free_encoder_buffers(b);
    }
    vr_status rc = ensure_encoder_buffers(b);
    if (rc != VR_OK) return rc;
    b.hostCtrlValid = false;
    b.hostBrickOff.clear();
    b.foreign = false;
    std::fill(b.openTreeBytes.begin(), b.openTreeBytes.end(), -1);
    {   // -3: a lazily allocated side buffer did not fit
        const int rc = encode_launch(&b, vox, (hipStream_t)stream);
        if (rc != 0) return rc == -3 ? VR_ERR_OOM : VR_ERR_NO_DEVICE;
    }
    b.built = true;
    b.timingsPending = true;
    b.lastStream = stream;
    return VR_OK;
}

static vr_status sync_ctrl(BrickSet &b)
{
    if (!b.built) return VR_ERR_STATE;
    if (b.hostCtrlValid) return VR_OK;
    HIPCHK(hipStreamSynchronize((hipStream_t)b.lastStream));
    HIPCHK(hipMemcpy(b.hostCtrl.data(), b.mid.ctrl, (size_t)b.B * sizeof(Ctrl), hipMemcpyDeviceToHost));
    b.lutZeroRun = 0;
    if (b.chainLut && !b.foreign) HIPCHK(hipMemcpy(&b.lutZeroRun, b.chainLut + 256, sizeof(uint32_t), hipMemcpyDeviceToHost));
    // MidRangeTree: the half-range stream's level loop reverts epochs of its own (M.cpp:399-544)
    b.hostRevertsR.assign((size_t)b.B, 0);
    if (b.variant == VR_VARIANT_MIDRANGE && !b.foreign && b.rng.ctrl) {
        std::vector<Ctrl> r((size_t)b.B);
        HIPCHK(hipMemcpy(r.data(), b.rng.ctrl, (size_t)b.B * sizeof(Ctrl), hipMemcpyDeviceToHost));
        for (int i = 0; i < b.B; ++i) b.hostRevertsR[(size_t)i] = r[(size_t)i].constBrick ? 0 : r[(size_t)i].numReverts;
    }
    b.hostCtrlValid = true;
    return VR_OK;
}

vr_status vr_brickset_info(vr_brickset *h, int32_t brick, vr_tree_info *info)
{
    if (!h || !info || brick < 0 || brick >= h->s.B) return VR_ERR_INVALID;
    BrickSet &b = h->s;
    vr_status rc = sync_ctrl(b);
    if (rc != VR_OK) return rc;
    const Ctrl &c = b.hostCtrl[brick];
    if (c.emitOverflow) return VR_ERR_STATE;   // internal count/emit mismatch: no valid stream
    memset(info, 0, sizeof(*info));
    info->X = b.g.X; info->Y = b.g.Y; info->Z = b.g.Z;
    info->orig_tree_depth = b.D;
    info->max_tree_depth = b.maxDepth;
    info->num_active_nodes = (int64_t)c.numActive;
    info->tree_bytes = ((int64_t)c.numActive + 3) / 4;
    info->tolerance = b.tolerance; info->max_epochs = b.maxEpochs; info->variant = b.variant;
    info->num_reverts = c.numReverts + ((size_t)brick < b.hostRevertsR.size() ? b.hostRevertsR[(size_t)brick] : 0);
    info->max_error_before = c.maxErrBefore;
    info->max_error_after = c.maxErrAfter;
    info->mean_l1_after = (double)c.statL1 / (double)b.leafStride;
    info->zero_run_rewrites = c.constBrick ? 0 : c.zeroRun + (int32_t)b.lutZeroRun;
    info->est_exact_segments = c.constBrick ? 0 : c.estFallbacks;
    return VR_OK;
}

// Device address of brick `brick`'s contiguous preorder stream (the reference's tree.bits) of stream s.  After a fused
// build the decoders read the block-gapped form; the reference's layout is in treeCompact, made at the end of build() or
// here, on demand.  Its buffers are sized from the streams' real lengths: when a build's streams did not fit (the first
// build of a set guesses; later ones may grow), the buffers are regrown to what is needed and the copy repeated.
static vr_status contiguous_stream(BrickSet &b, Stream2 &s, int brick, const uint8_t **ptr)
{
    if (!s.tree) return VR_ERR_STATE;
    if (!b.gapped) { *ptr = s.tree + (size_t)brick * b.treeCap; return VR_OK; }
    const bool mr = b.variant == VR_VARIANT_MIDRANGE;
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIPCHK(hipStreamSynchronize((hipStream_t)b.lastStream));
        if (!b.compactValid) {
            const int rc = compact_launch(&b, (hipStream_t)b.lastStream);
            if (rc != 0) return rc == -3 ? VR_ERR_OOM : VR_ERR_NO_DEVICE;
            HIPCHK(hipStreamSynchronize((hipStream_t)b.lastStream));
            b.hostBrickOff.clear();
        }
        if (b.hostBrickOff.size() != (size_t)b.B + 1) {
            b.hostBrickOff.resize((size_t)b.B + 1);
            HIPCHK(hipMemcpy(b.hostBrickOff.data(), b.brickOff, ((size_t)b.B + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            // (an emit overflow would have been flagged in the control blocks: re-read them)
            HIPCHK(hipMemcpy(b.hostCtrl.data(), b.mid.ctrl, (size_t)b.B * sizeof(Ctrl), hipMemcpyDeviceToHost));
            b.hostCtrlValid = true;
        }
        const int64_t total = (int64_t)b.hostBrickOff[(size_t)b.B];
        if (total <= b.compactCap) break;
        // did not fit: nothing was written.  Regrow (some headroom: the next timestep's streams differ) and repeat
        hipFree(b.mid.treeCompact); b.mid.treeCompact = nullptr;
        hipFree(b.rng.treeCompact); b.rng.treeCompact = nullptr;
        const int64_t cap = total + total / 8 + 4096;
        HIPCHK(hipMalloc(&b.mid.treeCompact, (size_t)cap));
        if (mr) HIPCHK(hipMalloc(&b.rng.treeCompact, (size_t)cap));
        b.compactCap = cap;
        b.compactValid = false;
        if (attempt == 1) return VR_ERR_STATE;
    }
    *ptr = s.treeCompact + (size_t)b.hostBrickOff[(size_t)brick];
    return VR_OK;
}

static vr_status get_tree_common(BrickSet &b, Stream2 &s, int brick, uint8_t *dst, int64_t cap)
{
    vr_status rc = sync_ctrl(b);
    if (rc != VR_OK) return rc;
    int64_t bytes = ((int64_t)b.hostCtrl[brick].numActive + 3) / 4;
    if (!dst || cap < bytes) return VR_ERR_INVALID;
    const uint8_t *src = nullptr;
    rc = contiguous_stream(b, s, brick, &src);
    if (rc != VR_OK) return rc;
    if (b.hostCtrl[brick].emitOverflow) return VR_ERR_STATE;
    HIPCHK(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return VR_OK;
}

vr_status vr_brickset_get_tree(vr_brickset *h, int32_t brick, uint8_t *dst, int64_t cap)
{
    if (!h || brick < 0 || brick >= h->s.B) return VR_ERR_INVALID;
    return get_tree_common(h->s, h->s.mid, brick, dst, cap);
}

vr_status vr_brickset_get_distance_map(vr_brickset *h, int32_t brick, uint8_t *dst, int32_t cap)
{
    if (!h || !dst || brick < 0 || brick >= h->s.B) return VR_ERR_INVALID;
    BrickSet &b = h->s;
    vr_status rc = sync_ctrl(b);
    if (rc != VR_OK) return rc;
    if (cap < b.maxDepth + 1) return VR_ERR_INVALID;
    memcpy(dst, b.hostCtrl[brick].distanceMap, (size_t)b.maxDepth + 1);
    return VR_OK;
}

vr_status vr_brickset_get_tree_range(vr_brickset *h, int32_t brick, uint8_t *dst, int64_t cap)
{
    if (!h || brick < 0 || brick >= h->s.B) return VR_ERR_INVALID;
    if (h->s.variant != VR_VARIANT_MIDRANGE || (h->s.foreign && !h->s.foreignRange)) return VR_ERR_STATE;
    return get_tree_common(h->s, h->s.rng, brick, dst, cap);
}

vr_status vr_brickset_get_distance_map_range(vr_brickset *h, int32_t brick, uint8_t *dst, int32_t cap)
{
    if (!h || !dst || brick < 0 || brick >= h->s.B) return VR_ERR_INVALID;
    BrickSet &b = h->s;
    if (b.variant != VR_VARIANT_MIDRANGE || (b.foreign && !b.foreignRange)) return VR_ERR_STATE;
    vr_status rc = sync_ctrl(b);
    if (rc != VR_OK) return rc;
    if (cap < b.maxDepth + 1) return VR_ERR_INVALID;
    Ctrl c;
    HIPCHK(hipMemcpy(&c, b.rng.ctrl + brick, sizeof(Ctrl), hipMemcpyDeviceToHost));
    memcpy(dst, c.distanceMap, (size_t)b.maxDepth + 1);
    return VR_OK;
}

// M.cpp:1095-1128 convertToByteArray.  A byte-shuffle of two host-visible streams:
// done on the host from the bytes the GPU encoder produced (format conversion, not
// part of the compute path).
vr_status vr_brickset_get_packed4(vr_brickset *h, int32_t brick, uint8_t *dst, int64_t cap, int64_t *length)
{
    if (!h || !length || brick < 0 || brick >= h->s.B) return VR_ERR_INVALID;
    BrickSet &b = h->s;
    if (b.variant != VR_VARIANT_MIDRANGE || (b.foreign && !b.foreignRange)) return VR_ERR_STATE;
    vr_status rc = sync_ctrl(b);
    if (rc != VR_OK) return rc;
    const int64_t n = (int64_t)b.hostCtrl[brick].numActive;
    int64_t v = (int64_t)ceil((double)n / 2.0);
    v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v++;
    *length = v;
    if (!dst) return VR_OK;
    if (cap < v) return VR_ERR_INVALID;
    const int64_t bytes = (n + 3) / 4;
    std::vector<uint8_t> m((size_t)bytes), r((size_t)bytes);
    const uint8_t *baseM = nullptr, *baseR = nullptr;
    rc = contiguous_stream(b, b.mid, brick, &baseM);
    if (rc == VR_OK) rc = contiguous_stream(b, b.rng, brick, &baseR);
    if (rc != VR_OK) return rc;
    HIPCHK(hipMemcpy(m.data(), baseM, (size_t)bytes, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(r.data(), baseR, (size_t)bytes, hipMemcpyDeviceToHost));
    memset(dst, 0, (size_t)v);
    auto get = [](const std::vector<uint8_t> &a, int64_t i) { return (a[(size_t)(i >> 2)] >> ((i & 3) * 2)) & 3; };
    int64_t o = 0;
    for (int64_t i = 0; i < n; i += 2) {
        int first = get(m, i), second = get(r, i), third = 0, fourth = 0;
        if (i + 1 < n) { third = get(m, i + 1); fourth = get(r, i + 1); }
        dst[o++] = (uint8_t)((first << 6) | (second << 4) | (third << 2) | fourth);
    }
    return VR_OK;
}

vr_status vr_brickset_decode(vr_brickset *h, int32_t cut_depth, uint8_t *out, void *stream)
{
    if (!h || !out) return VR_ERR_INVALID;
    BrickSet &b = h->s;
    if (!b.built) return VR_ERR_STATE;
    if (cut_depth > b.maxDepth) return VR_ERR_INVALID;
    const int cut = cut_depth < 0 ? b.maxDepth : cut_depth;
    if (cut < b.Ds && b.foreign) {
        // ancestor scalars at the cut depth, from the stream bytes kept at set_tree/open time
        vr_status rc = sync_ctrl(b);
        if (rc != VR_OK) return rc;
        for (int br = 0; br < b.B; ++br) {
            std::vector<uint8_t> vals((size_t)b.nIdx, 0);
            if (br < (int)b.hostTree.size() && !b.hostTree[br].empty() &&
                cut_values_from_stream(&b, b.hostTree[br].data(), (int64_t)b.hostCtrl[br].numActive,
                                       b.hostCtrl[br].distanceMap, cut, vals) != 0)
                return VR_ERR_FORMAT;
            HIPCHK(hipMemcpy(b.idxValCut + (size_t)br * b.nIdx, vals.data(), vals.size(), hipMemcpyHostToDevice));
        }
    }
    {
        const int rc = decode_launch(&b, out, cut, (hipStream_t)stream);
        if (rc != 0) return rc == -3 ? VR_ERR_OOM : VR_ERR_NO_DEVICE;
    }
    b.decodeTimingPending = true;
    b.lastStream = stream;
    return VR_OK;
}

vr_status vr_brickset_decode_range(vr_brickset *h, int32_t cut_depth, uint8_t *out, void *stream)
{
    if (!h || !out) return VR_ERR_INVALID;
    BrickSet &b = h->s;
    if (!b.built) return VR_ERR_STATE;
    if (b.variant != VR_VARIANT_MIDRANGE) return VR_ERR_STATE;
    if (b.foreign) return VR_ERR_UNSUPPORTED;          // an opened file carries no BFS codes to seed the index scalars from
    if (cut_depth > b.maxDepth) return VR_ERR_INVALID;
    const int cut = cut_depth < 0 ? b.maxDepth : cut_depth;
    if (decode_launch(&b, out, cut, (hipStream_t)stream, true) != 0) return VR_ERR_NO_DEVICE;
    b.decodeTimingPending = true;
    b.lastStream = stream;
    return VR_OK;
}

vr_status vr_brickset_set_tree(vr_brickset *h, int32_t brick, const uint8_t *tree, int64_t tree_bytes,
                               int64_t num_active, const uint8_t *dmap, int32_t map_len)
{
    if (!h || !tree || !dmap || brick < 0 || brick >= h->s.B) return VR_ERR_INVALID;
    BrickSet &b = h->s;
    if (map_len < b.maxDepth + 1 || num_active <= 0) return VR_ERR_INVALID;
    if (num_active >= (1ll << 32)) return VR_ERR_UNSUPPORTED;      // the host-side index of a foreign stream is 32-bit
    if (b.built && !b.foreign) return VR_ERR_STATE;                // streams are installed into a fresh set, not beside built trees
    const int64_t need = (num_active + 3) / 4;
    if (tree_bytes < need || need > b.treeCap) return VR_ERR_FORMAT;
    std::vector<uint32_t> offs;
    std::vector<uint8_t> vals, fine, val3;
    if (build_index_from_stream(&b, brick, tree, num_active, dmap, offs, vals, fine, val3) != 0) return VR_ERR_FORMAT;
    // the fine decoders take the grown-branch distances as the constants the reference writes (R.cpp:94-97); a
    // file that says otherwise is decoded by the walking kernel, which reads them from the map
    for (int i = 0; i < VR_CHAIN_LEVELS; ++i)
        if (dmap[b.D + 1 + i] != (uint8_t)(64 >> i)) { fine.clear(); val3.clear(); }
    if (!b.built) { // first foreign tree: other bricks stay empty until set
        b.fineHas.assign((size_t)b.B, 1);   // (their index is all "pruned": no counts are read)
        HIPCHK(hipMemset(b.mid.ctrl, 0, (size_t)b.B * sizeof(Ctrl)));
        std::vector<uint32_t> dead((size_t)b.B * b.nIdx, VR_IDX_DEAD);
        HIPCHK(hipMemcpy(b.idxOff, dead.data(), dead.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemset(b.idxVal, 0, (size_t)b.B * b.nIdx));
        for (auto &c : b.hostCtrl) memset(&c, 0, sizeof(Ctrl));
    } else {
        vr_status rc = sync_ctrl(b);
        if (rc != VR_OK) return rc;
    }
    Ctrl &c = b.hostCtrl[brick];
    memset(&c, 0, sizeof(Ctrl));
    c.numActive = (unsigned long long)num_active;
    memcpy(c.distanceMap, dmap, (size_t)b.maxDepth + 1);
    HIPCHK(hipMemcpy(b.mid.ctrl + brick, &c, sizeof(Ctrl), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(b.mid.tree + (size_t)brick * b.treeCap, 0, (size_t)b.treeCap));
    HIPCHK(hipMemcpy(b.mid.tree + (size_t)brick * b.treeCap, tree, (size_t)need, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b.idxOff + (size_t)brick * b.nIdx, offs.data(), offs.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b.idxVal + (size_t)brick * b.nIdx, vals.data(), vals.size(), hipMemcpyHostToDevice));
    b.built = true;
    b.hostCtrlValid = true;
    b.foreign = true;
    b.gapped = false;          // an installed stream is the contiguous one; its index points into it
    if (b.idx64) {      // absolute 32-bit offsets from the host parse: bases are zero
        if (!b.idxBase) { b.nEmitBlk = (((int64_t)1 << b.D) + 255) / 256; HIPCHK(hipMalloc(&b.idxBase, (size_t)b.B * b.nEmitBlk * sizeof(unsigned long long))); }
        HIPCHK(hipMemset(b.idxBase, 0, (size_t)b.B * b.nEmitBlk * sizeof(unsigned long long)));
    }
    if ((int)b.fineHas.size() != b.B) b.fineHas.assign((size_t)b.B, 0);
    b.fineHas[(size_t)brick] = 0;
    if (!fine.empty()) {
        if (!b.fineIdx) HIPCHK(hipMalloc(&b.fineIdx, (size_t)b.B * b.nIdx * 16));
        HIPCHK(hipMemcpy(b.fineIdx + (size_t)brick * b.nIdx * 16, fine.data(), fine.size(), hipMemcpyHostToDevice));
        if (!b.idxVal3) HIPCHK(hipMalloc(&b.idxVal3, (size_t)b.B * b.nIdx * 8));
        HIPCHK(hipMemcpy(b.idxVal3 + (size_t)brick * b.nIdx * 8, val3.data(), val3.size(), hipMemcpyHostToDevice));
        b.fineHas[(size_t)brick] = 1;
    }
    if ((int)b.hostTree.size() != b.B) b.hostTree.assign((size_t)b.B, std::vector<uint8_t>());
    b.hostTree[brick].assign(tree, tree + need);
    return VR_OK;
}

// File layout of VolumeKdtree::save (R.cpp:535-544).
vr_status vr_brickset_save(vr_brickset *h, int32_t brick, const char *path)
{
    if (!h || !path || brick < 0 || brick >= h->s.B) return VR_ERR_INVALID;
    BrickSet &b = h->s;
    vr_status rc = sync_ctrl(b);
    if (rc != VR_OK) return rc;                       // "ERROR! No tree to save." R.cpp:526-530
    const Ctrl &c = b.hostCtrl[brick];
    const int64_t bytes = ((int64_t)c.numActive + 3) / 4;
    if (bytes == 0) return VR_ERR_STATE;
    std::vector<uint8_t> tree((size_t)bytes);
    const uint8_t *baseM = nullptr;
    rc = contiguous_stream(b, b.mid, brick, &baseM);
    if (rc != VR_OK) return rc;
    HIPCHK(hipMemcpy(tree.data(), baseM, (size_t)bytes, hipMemcpyDeviceToHost));
    // MidRangeTree::save (M.cpp:753-785): same header, then distanceMap, distanceMap_range, tree, tree_range
    const bool mrFile = b.variant == VR_VARIANT_MIDRANGE;
    std::vector<uint8_t> treeR;
    Ctrl cr;
    if (mrFile) {
        if (b.foreign && !b.foreignRange) return VR_ERR_STATE;
        treeR.resize((size_t)bytes);
        const uint8_t *baseR = nullptr;
        rc = contiguous_stream(b, b.rng, brick, &baseR);
        if (rc != VR_OK) return rc;
        HIPCHK(hipMemcpy(treeR.data(), baseR, (size_t)bytes, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&cr, b.rng.ctrl + brick, sizeof(Ctrl), hipMemcpyDeviceToHost));
    }
    FILE *f = fopen(path, "wb");
    if (!f) return VR_ERR_IO;
    int64_t rootMin[3] = {0, 0, 0}, rootMax[3] = {b.g.X, b.g.Y, b.g.Z};
    int32_t mtd = b.maxDepth, otd = b.D;
    int64_t X = b.g.X, Y = b.g.Y, Z = b.g.Z, na = (int64_t)c.numActive;
    bool ok = fwrite(rootMin, 8, 3, f) == 3 && fwrite(rootMax, 8, 3, f) == 3 && fwrite(&mtd, 4, 1, f) == 1 &&
              fwrite(&otd, 4, 1, f) == 1 && fwrite(&X, 8, 1, f) == 1 && fwrite(&Y, 8, 1, f) == 1 &&
              fwrite(&Z, 8, 1, f) == 1 && fwrite(&na, 8, 1, f) == 1 &&
              fwrite(c.distanceMap, 1, (size_t)mtd + 1, f) == (size_t)mtd + 1 &&
              (!mrFile || fwrite(cr.distanceMap, 1, (size_t)mtd + 1, f) == (size_t)mtd + 1) &&
              fwrite(tree.data(), 1, (size_t)bytes, f) == (size_t)bytes &&
              (!mrFile || fwrite(treeR.data(), 1, (size_t)bytes, f) == (size_t)bytes);
    fclose(f);
    return ok ? VR_OK : VR_ERR_IO;
}

// VolumeKdtree::open (R.cpp:554-594).  A missing file is an error code here (the
// reference waits for Enter and calls exit(-1)).
vr_status vr_brickset_open(vr_brickset **out, const char *path)
{
    if (!out || !path) return VR_ERR_INVALID;
    FILE *f = fopen(path, "rb");
    if (!f) return VR_ERR_IO;
    fseek(f, 0, SEEK_END);
    const int64_t fileSize = ftell(f);
    fseek(f, 0, SEEK_SET);
    int64_t rootMin[3], rootMax[3], X, Y, Z, na;
    int32_t mtd, otd;
    bool ok = fread(rootMin, 8, 3, f) == 3 && fread(rootMax, 8, 3, f) == 3 && fread(&mtd, 4, 1, f) == 1 &&
              fread(&otd, 4, 1, f) == 1 && fread(&X, 8, 1, f) == 1 && fread(&Y, 8, 1, f) == 1 &&
              fread(&Z, 8, 1, f) == 1 && fread(&na, 8, 1, f) == 1;
    if (!ok || mtd < VR_CHAIN_LEVELS || mtd >= VR_MAX_DEPTH || na <= 0) { fclose(f); return VR_ERR_FORMAT; }
    std::vector<uint8_t> dmap((size_t)mtd + 1);
    // R.cpp:581 subtracts only three of the four int64 fields: tree.bits ends up 8 bytes
    // longer than what was saved (SURVEY C-6); numActiveNodes is authoritative.
    const int64_t openBytes = fileSize - (2 * 24 + 2 * 4 + mtd + 1 + 3 * 8);
    const int64_t have = fileSize - (88 + mtd + 1);
    if (have < (na + 3) / 4) { fclose(f); return VR_ERR_FORMAT; }
    std::vector<uint8_t> tree((size_t)have);
    ok = fread(dmap.data(), 1, dmap.size(), f) == dmap.size() && fread(tree.data(), 1, tree.size(), f) == tree.size();
    fclose(f);
    if (!ok) return VR_ERR_IO;
    int64_t dims[3] = {X, Y, Z};
    vr_brickset *h = nullptr;
    vr_status rc = vr_brickset_create(&h, 1, dims, 6, 5, VR_VARIANT_RECOVER); // ctor defaults R.h:89-94
    if (rc != VR_OK) return rc;
    if (h->s.D != otd || h->s.maxDepth != mtd) { vr_brickset_destroy(h); return VR_ERR_FORMAT; }
    rc = vr_brickset_set_tree(h, 0, tree.data(), (int64_t)tree.size(), na, dmap.data(), mtd + 1);
    if (rc != VR_OK) { vr_brickset_destroy(h); return rc; }
    h->s.openTreeBytes[0] = openBytes;
    *out = h;
    return VR_OK;
}

// MidRangeTree files (M.cpp:753-785).  The reference's own reader (M.cpp:787-833) mis-sizes both streams by
// 4 bytes (it subtracts three of the four int64 header fields before halving), so the range stream it reads
// back is shifted; there is nothing to match there.  This reader returns exactly what save() wrote.
vr_status vr_brickset_open_variant(vr_brickset **out, const char *path, int32_t variant)
{
    if (!out || !path) return VR_ERR_INVALID;
    if (variant != VR_VARIANT_MIDRANGE) {
        if (variant != VR_VARIANT_RECOVER && variant != VR_VARIANT_GUARDED) return VR_ERR_INVALID;
        return vr_brickset_open(out, path);
    }
    FILE *f = fopen(path, "rb");
    if (!f) return VR_ERR_IO;
    fseek(f, 0, SEEK_END);
    const int64_t fileSize = ftell(f);
    fseek(f, 0, SEEK_SET);
    int64_t rootMin[3], rootMax[3], X, Y, Z, na;
    int32_t mtd, otd;
    bool ok = fread(rootMin, 8, 3, f) == 3 && fread(rootMax, 8, 3, f) == 3 && fread(&mtd, 4, 1, f) == 1 &&
              fread(&otd, 4, 1, f) == 1 && fread(&X, 8, 1, f) == 1 && fread(&Y, 8, 1, f) == 1 &&
              fread(&Z, 8, 1, f) == 1 && fread(&na, 8, 1, f) == 1;
    if (!ok || mtd < VR_CHAIN_LEVELS || mtd >= VR_MAX_DEPTH || na <= 0) { fclose(f); return VR_ERR_FORMAT; }
    const int64_t have = fileSize - (88 + 2 * ((int64_t)mtd + 1));
    const int64_t T = have / 2;
    if (have < 0 || (have & 1) || T != (na + 3) / 4) { fclose(f); return VR_ERR_FORMAT; }
    std::vector<uint8_t> dmap((size_t)mtd + 1), dmapR((size_t)mtd + 1), tree((size_t)T), treeR((size_t)T);
    ok = fread(dmap.data(), 1, dmap.size(), f) == dmap.size() && fread(dmapR.data(), 1, dmapR.size(), f) == dmapR.size() &&
         fread(tree.data(), 1, tree.size(), f) == tree.size() && fread(treeR.data(), 1, treeR.size(), f) == treeR.size();
    fclose(f);
    if (!ok) return VR_ERR_IO;
    int64_t dims[3] = {X, Y, Z};
    vr_brickset *h = nullptr;
    vr_status rc = vr_brickset_create(&h, 1, dims, 6, 5, VR_VARIANT_MIDRANGE);
    if (rc != VR_OK) return rc;
    if (h->s.D != otd || h->s.maxDepth != mtd) { vr_brickset_destroy(h); return VR_ERR_FORMAT; }
    rc = vr_brickset_set_tree(h, 0, tree.data(), (int64_t)tree.size(), na, dmap.data(), mtd + 1);
    if (rc != VR_OK) { vr_brickset_destroy(h); return rc; }
    BrickSet &b = h->s;
    Ctrl cr;
    memset(&cr, 0, sizeof(Ctrl));
    cr.numActive = (unsigned long long)na;
    memcpy(cr.distanceMap, dmapR.data(), (size_t)mtd + 1);
    hipError_t e = hipSuccess;
    if (!b.rng.ctrl) e = hipMalloc(&b.rng.ctrl, sizeof(Ctrl));
    if (e == hipSuccess && !b.rng.tree) e = hipMalloc(&b.rng.tree, (size_t)b.treeCap);
    if (e == hipSuccess) e = hipMemcpy(b.rng.ctrl, &cr, sizeof(Ctrl), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(b.rng.tree, 0, (size_t)b.treeCap);
    if (e == hipSuccess) e = hipMemcpy(b.rng.tree, treeR.data(), (size_t)T, hipMemcpyHostToDevice);
    if (e != hipSuccess) { vr_brickset_destroy(h); return VR_ERR_NO_DEVICE; }
    b.foreignRange = true;
    *out = h;
    return VR_OK;
}

vr_status vr_measure_error(const uint8_t *dec, const uint8_t *orig, int64_t n, int32_t *max_error, double *mean_error,
                           void *stream)
{
    if (!dec || !orig || n <= 0) return VR_ERR_INVALID;
    if (!device_ok()) return VR_ERR_NO_DEVICE;
    int *dMax = nullptr;
    unsigned long long *dSum = nullptr;
    HIPCHK(hipMalloc(&dMax, sizeof(int)));
    HIPCHK(hipMalloc(&dSum, sizeof(unsigned long long)));
    hipMemsetAsync(dMax, 0, sizeof(int), (hipStream_t)stream);
    hipMemsetAsync(dSum, 0, sizeof(unsigned long long), (hipStream_t)stream);
    int rc = measure_error_launch(dec, orig, n, dMax, dSum, (hipStream_t)stream);
    int hm = 0;
    unsigned long long hs = 0;
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e == hipSuccess) e = hipMemcpy(&hm, dMax, sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&hs, dSum, sizeof(hs), hipMemcpyDeviceToHost);
    hipFree(dMax); hipFree(dSum);
    if (rc != 0 || e != hipSuccess) return VR_ERR_NO_DEVICE;
    if (max_error) *max_error = hm;
    if (mean_error) *mean_error = (double)hs / (double)n;
    return VR_OK;
}

vr_status vr_query_error(const uint8_t *dec, const uint8_t *orig, int64_t n, uint8_t *err, void *stream)
{
    if (!dec || !orig || !err || n <= 0) return VR_ERR_INVALID;
    if (!device_ok()) return VR_ERR_NO_DEVICE;
    return query_error_launch(dec, orig, n, err, (hipStream_t)stream) == 0 ? VR_OK : VR_ERR_NO_DEVICE;
}

static vr_status assemble_common(bool toVolume, const uint8_t *src, int32_t nb, const int64_t bd[3], const int64_t *ijk,
                                 const int64_t grid[3], uint8_t *dst, void *stream)
{
    if (!src || !dst || !bd || !ijk || !grid || nb <= 0) return VR_ERR_INVALID;
    if (bd[0] % 16 != 0) return VR_ERR_UNSUPPORTED;
    for (int b = 0; b < nb; ++b)
        for (int k = 0; k < 3; ++k)
            if (ijk[3 * b + k] < 0 || ijk[3 * b + k] >= grid[k]) return VR_ERR_INVALID;
    if (!device_ok()) return VR_ERR_NO_DEVICE;
    // the brick map lives on the device: uploaded once per distinct map and kept for the life of the process (a
    // streaming loop calls this every frame with the same map; allocating, copying and synchronising each time would
    // serialise the pipeline).  Cached maps are never freed, so a launch needs no lock; past 64 distinct maps a call
    // uploads its own copy and releases it after its launch has run.
    static std::mutex mu;
    static std::map<std::vector<int64_t>, int64_t *> cache;
    int64_t *d = nullptr;
    bool mine = false;
    {
        std::lock_guard<std::mutex> lk(mu);
        std::vector<int64_t> key(ijk, ijk + (size_t)nb * 3);
        int dev = 0;
        hipGetDevice(&dev);
        key.push_back(dev);
        auto it = cache.find(key);
        if (it == cache.end()) {
            HIPCHK(hipMalloc(&d, (size_t)nb * 3 * sizeof(int64_t)));
            if (hipMemcpy(d, ijk, (size_t)nb * 3 * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) { hipFree(d); return VR_ERR_NO_DEVICE; }
            if (cache.size() < 64) cache.emplace(std::move(key), d);
            else mine = true;
        } else d = it->second;
    }
    const int rc = assemble_launch(toVolume, src, dst, nb, bd, d, grid, (hipStream_t)stream);
    if (mine) { hipStreamSynchronize((hipStream_t)stream); hipFree(d); }
    return rc == 0 ? VR_OK : VR_ERR_NO_DEVICE;
}

vr_status vr_assemble_bricks(const uint8_t *bricks, int32_t nb, const int64_t bd[3], const int64_t *ijk,
                             const int64_t grid[3], uint8_t *volume, void *stream)
{
    return assemble_common(true, bricks, nb, bd, ijk, grid, volume, stream);
}
vr_status vr_disassemble_bricks(const uint8_t *volume, int32_t nb, const int64_t bd[3], const int64_t *ijk,
                                const int64_t grid[3], uint8_t *bricks, void *stream)
{
    return assemble_common(false, volume, nb, bd, ijk, grid, bricks, stream);
}

vr_status vr_raycast(const uint8_t *vol, const int64_t dims[3], const vr_camera *cam, const vr_render_params *P,
                     float *rgba, void *stream)
{
    if (!vol || !dims || !cam || !P || !rgba) return VR_ERR_INVALID;
    if (P->width <= 0 || P->height <= 0 || P->max_samples < 0 || P->mode < 0 || P->mode > 2) return VR_ERR_INVALID;
    if (!device_ok()) return VR_ERR_NO_DEVICE;
    return raycast_launch(vol, dims, cam, P, rgba, (hipStream_t)stream) == 0 ? VR_OK : VR_ERR_NO_DEVICE;
}

vr_status vr_skip_grid_build(const uint8_t *vol, const int64_t dims[3], int32_t cell, uint8_t *grid, void *stream)
{
    if (!vol || !dims || !grid || cell <= 0 || cell > 64) return VR_ERR_INVALID;
    for (int k = 0; k < 3; ++k) if (dims[k] <= 0 || dims[k] >= (1ll << 31)) return VR_ERR_INVALID;
    if (!device_ok()) return VR_ERR_NO_DEVICE;
    return skip_grid_launch(vol, dims, cell, grid, (hipStream_t)stream) == 0 ? VR_OK : VR_ERR_NO_DEVICE;
}

vr_status vr_composite_over(float *front, const float *back, int64_t n, void *stream)
{
    if (!front || !back || n <= 0) return VR_ERR_INVALID;
    if (!device_ok()) return VR_ERR_NO_DEVICE;
    return composite_over_launch(front, back, n, (hipStream_t)stream) == 0 ? VR_OK : VR_ERR_NO_DEVICE;
}
vr_status vr_composite_finish(const float *partial, float *rgba, int64_t n, void *stream)
{
    if (!partial || !rgba || n <= 0) return VR_ERR_INVALID;
    if (!device_ok()) return VR_ERR_NO_DEVICE;
    return composite_finish_launch(partial, rgba, n, (hipStream_t)stream) == 0 ? VR_OK : VR_ERR_NO_DEVICE;
}

vr_status vr_composite_slabs(const float *partials, int32_t num_slabs, int64_t num_pixels, int64_t first_pixel, int32_t axis,
                             const vr_camera *cam, const vr_render_params *P, float *rgba, void *stream)
{
    if (!partials || !cam || !P || !rgba || num_slabs <= 0 || num_pixels <= 0 || first_pixel < 0 || axis < 0 || axis > 2)
        return VR_ERR_INVALID;
    if (P->width <= 0 || P->height <= 0 || first_pixel + num_pixels > (int64_t)P->width * P->height) return VR_ERR_INVALID;
    if (!device_ok()) return VR_ERR_NO_DEVICE;
    return composite_slabs_launch(partials, num_slabs, num_pixels, first_pixel, axis, cam, P, rgba, (hipStream_t)stream) == 0
               ? VR_OK : VR_ERR_NO_DEVICE;
}

vr_status vr_brickset_set_concurrency(vr_brickset *h, int32_t level_loop_streams)
{
    if (!h || level_loop_streams < 1 || level_loop_streams > 4) return VR_ERR_INVALID;
    h->s.levelLoopStreams = level_loop_streams;
    return VR_OK;
}

vr_status vr_brickset_last_timings(vr_brickset *h, float ms[5])
{
    if (!h || !ms) return VR_ERR_INVALID;
    BrickSet &b = h->s;
    if (b.timingsPending) {
        HIPCHK(hipEventSynchronize(b.ev[4]));
        for (int i = 0; i < 4; ++i) hipEventElapsedTime(&b.phasesMs[i], b.ev[i], b.ev[i + 1]);
        b.timingsPending = false;
    }
    if (b.decodeTimingPending) {
        HIPCHK(hipEventSynchronize(b.ev[6]));
        hipEventElapsedTime(&b.phasesMs[4], b.ev[5], b.ev[6]);
        b.decodeTimingPending = false;
    }
    for (int i = 0; i < 5; ++i) ms[i] = b.phasesMs[i];
    return VR_OK;
}

} // extern "C"
