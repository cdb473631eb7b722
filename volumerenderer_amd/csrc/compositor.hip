// compositor.hip -- sort-last compositing across the GPUs of a node behind the C ABI (include/vrhip.h):
// the one real exchange step of the path.  Rank r holds the partial (c, tau) image of slab r of the
// volume; the frame is cut into `world` row tiles; ONE grouped RCCL call moves tile t of every rank's partial image
// to rank t (direct send: ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd over xGMI, 4 MB per peer at 1080p and
// eight ranks); rank t combines its `world` partials per pixel in that pixel's view order (k_composite_slabs:
// "over" is associative but not commutative, raycaster.frag:69-72); a second grouped call gathers the finished tiles
// on rank 0.  The reference is single-GPU: this is new, in the form SURVEY 8b proposed (vr_composite_init(comm) /
// vr_composite).
//
// RCCL is bound at run time (dlopen of the copy already in the process -- PyTorch brings its own -- or of
// librccl.so.1), so libvrhip.so loads and every single-GPU entry point works where RCCL is absent.
#include "../../include/vrhip.h"
#include "kd_common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <new>

namespace vr {
int composite_slabs_launch(const float *, int, int64_t, int64_t, int, const vr_camera *, const vr_render_params *, float *, hipStream_t);
}

namespace {

struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (const char *n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);     // the copy already loaded
        for (const char *n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
        r.Send = (decltype(r.Send))dlsym(h, "ncclSend");
        r.Recv = (decltype(r.Recv))dlsym(h, "ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv;
    });
    return r;
}

void rows_of(int H, int rank, int world, int &lo, int &hi)
{   // contiguous block partition, the first ranks take the remainder (distributed.shard_range)
    const int q = H / world, r = H % world;
    lo = rank * q + (rank < r ? rank : r);
    hi = lo + q + (rank < r ? 1 : 0);
}

bool nccl_ok(ncclResult_t e, const char *what)
{
    if (e == ncclSuccess) return true;
    if (getenv("VRHIP_DEBUG")) fprintf(stderr, "[vrhip] %s: %s\n", what, rccl().GetErrorString ? rccl().GetErrorString(e) : "rccl error");
    return false;
}

} // namespace

struct vr_compositor {
    int rank = 0, world = 1, W = 0, H = 0;
    ncclComm_t comm = nullptr;
    bool ownComm = false;
    float *recv = nullptr;      // world * (my tile's pixels) * 4
    float *tile = nullptr;      // my finished tile
};

extern "C" {

vr_status vr_rccl_unique_id(uint8_t id[128])
{
    if (!id) return VR_ERR_INVALID;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (!rccl().ok) return VR_ERR_UNSUPPORTED;
    ncclUniqueId u;
    if (!nccl_ok(rccl().GetUniqueId(&u), "ncclGetUniqueId")) return VR_ERR_NO_DEVICE;
    memcpy(id, &u, 128);
    return VR_OK;
}

static vr_status compositor_alloc(vr_compositor *c)
{
    int lo, hi;
    rows_of(c->H, c->rank, c->world, lo, hi);
    const size_t npix = (size_t)(hi - lo) * c->W;
    if (c->world > 1 && npix) {
        if (hipMalloc(&c->recv, (size_t)c->world * npix * 4 * sizeof(float)) != hipSuccess) return VR_ERR_OOM;
        if (hipMalloc(&c->tile, npix * 4 * sizeof(float)) != hipSuccess) return VR_ERR_OOM;
    }
    return VR_OK;
}

vr_status vr_compositor_create(vr_compositor **out, const uint8_t id[128], int32_t rank, int32_t world, int32_t width, int32_t height)
{
    if (!out || world < 1 || rank < 0 || rank >= world || width <= 0 || height < world) return VR_ERR_INVALID;
    if (world > 1 && !id) return VR_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return VR_ERR_NO_DEVICE;
    vr_compositor *c = new (std::nothrow) vr_compositor();
    if (!c) return VR_ERR_OOM;
    c->rank = rank; c->world = world; c->W = width; c->H = height;
    if (world > 1) {
        if (!rccl().ok) { delete c; return VR_ERR_UNSUPPORTED; }
        ncclUniqueId u;
        memcpy(&u, id, 128);
        if (!nccl_ok(rccl().CommInitRank(&c->comm, world, u, rank), "ncclCommInitRank")) { delete c; return VR_ERR_NO_DEVICE; }
        c->ownComm = true;
    }
    const vr_status rc = compositor_alloc(c);
    if (rc != VR_OK) { vr_compositor_destroy(c); return rc; }
    *out = c;
    return VR_OK;
}

vr_status vr_compositor_create_from_comm(vr_compositor **out, void *nccl_comm, int32_t rank, int32_t world, int32_t width, int32_t height)
{
    if (!out || world < 1 || rank < 0 || rank >= world || width <= 0 || height < world) return VR_ERR_INVALID;
    if (world > 1 && !nccl_comm) return VR_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return VR_ERR_NO_DEVICE;
    if (world > 1 && !rccl().ok) return VR_ERR_UNSUPPORTED;
    vr_compositor *c = new (std::nothrow) vr_compositor();
    if (!c) return VR_ERR_OOM;
    c->rank = rank; c->world = world; c->W = width; c->H = height;
    c->comm = (ncclComm_t)nccl_comm;
    const vr_status rc = compositor_alloc(c);
    if (rc != VR_OK) { vr_compositor_destroy(c); return rc; }
    *out = c;
    return VR_OK;
}

vr_status vr_compositor_destroy(vr_compositor *c)
{
    if (!c) return VR_OK;
    hipFree(c->recv); hipFree(c->tile);
    if (c->ownComm && c->comm && rccl().ok) rccl().CommDestroy(c->comm);
    delete c;
    return VR_OK;
}

vr_status vr_compositor_composite(vr_compositor *c, const float *partial_dev, int32_t axis, const vr_camera *cam,
                                  const vr_render_params *params, float *rgba_dev, void *stream)
{
    if (!c || !partial_dev || !cam || !params || axis < 0 || axis > 2) return VR_ERR_INVALID;
    if (params->width != c->W || params->height != c->H) return VR_ERR_INVALID;
    if (c->rank == 0 && !rgba_dev) return VR_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int64_t frame = (int64_t)c->W * c->H;
    if (c->world == 1)
        return vr::composite_slabs_launch(partial_dev, 1, frame, 0, axis, cam, params, rgba_dev, st) == 0 ? VR_OK : VR_ERR_NO_DEVICE;
    Rccl &R = rccl();
    int myLo, myHi;
    rows_of(c->H, c->rank, c->world, myLo, myHi);
    const size_t npix = (size_t)(myHi - myLo) * c->W;
    // ---- tile t of my partial image to rank t, my tile of every rank's image to me (slab order = rank order)
    if (!nccl_ok(R.GroupStart(), "ncclGroupStart")) return VR_ERR_NO_DEVICE;
    bool ok = true;
    for (int peer = 0; peer < c->world && ok; ++peer) {
        if (peer == c->rank) continue;
        int lo, hi;
        rows_of(c->H, peer, c->world, lo, hi);
        ok = ok && nccl_ok(R.Send(partial_dev + (size_t)lo * c->W * 4, (size_t)(hi - lo) * c->W * 4, ncclFloat, peer, c->comm, st), "ncclSend");
        ok = ok && nccl_ok(R.Recv(c->recv + (size_t)peer * npix * 4, npix * 4, ncclFloat, peer, c->comm, st), "ncclRecv");
    }
    ok = nccl_ok(R.GroupEnd(), "ncclGroupEnd") && ok;
    if (!ok) return VR_ERR_NO_DEVICE;
    if (hipMemcpyAsync(c->recv + (size_t)c->rank * npix * 4, partial_dev + (size_t)myLo * c->W * 4, npix * 4 * sizeof(float),
                       hipMemcpyDeviceToDevice, st) != hipSuccess) return VR_ERR_NO_DEVICE;
    // ---- my tile: every pixel combines the slabs front to back in ITS view order
    float *dst = c->rank == 0 ? rgba_dev + (size_t)myLo * c->W * 4 : c->tile;
    if (vr::composite_slabs_launch(c->recv, c->world, (int64_t)npix, (int64_t)myLo * c->W, axis, cam, params, dst, st) != 0) return VR_ERR_NO_DEVICE;
    // ---- the finished tiles to rank 0
    if (!nccl_ok(R.GroupStart(), "ncclGroupStart")) return VR_ERR_NO_DEVICE;
    if (c->rank != 0) ok = nccl_ok(R.Send(c->tile, npix * 4, ncclFloat, 0, c->comm, st), "ncclSend");
    else
        for (int peer = 1; peer < c->world && ok; ++peer) {
            int lo, hi;
            rows_of(c->H, peer, c->world, lo, hi);
            ok = nccl_ok(R.Recv(rgba_dev + (size_t)lo * c->W * 4, (size_t)(hi - lo) * c->W * 4, ncclFloat, peer, c->comm, st), "ncclRecv");
        }
    ok = nccl_ok(R.GroupEnd(), "ncclGroupEnd") && ok;
    return ok ? VR_OK : VR_ERR_NO_DEVICE;
}

// ---- streams, events, pinned host memory: what a C++ host needs to overlap the stages of a timestep stream
// (include/vrhip/TimestepStreamer.hpp) without seeing a HIP header
vr_status vr_stream_create(void **stream)
{
    if (!stream) return VR_ERR_INVALID;
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return VR_ERR_NO_DEVICE;
    *stream = (void *)s;
    return VR_OK;
}
vr_status vr_stream_destroy(void *stream) { if (stream) hipStreamDestroy((hipStream_t)stream); return VR_OK; }
vr_status vr_stream_synchronize(void *stream) { return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? VR_OK : VR_ERR_NO_DEVICE; }
vr_status vr_event_create(void **event)
{
    if (!event) return VR_ERR_INVALID;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return VR_ERR_NO_DEVICE;
    *event = (void *)e;
    return VR_OK;
}
vr_status vr_event_destroy(void *event) { if (event) hipEventDestroy((hipEvent_t)event); return VR_OK; }
vr_status vr_event_record(void *event, void *stream)
{
    if (!event) return VR_ERR_INVALID;
    return hipEventRecord((hipEvent_t)event, (hipStream_t)stream) == hipSuccess ? VR_OK : VR_ERR_NO_DEVICE;
}
vr_status vr_stream_wait_event(void *stream, void *event)
{
    if (!event) return VR_ERR_INVALID;
    return hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0) == hipSuccess ? VR_OK : VR_ERR_NO_DEVICE;
}
vr_status vr_event_synchronize(void *event)
{
    if (!event) return VR_ERR_INVALID;
    return hipEventSynchronize((hipEvent_t)event) == hipSuccess ? VR_OK : VR_ERR_NO_DEVICE;
}
vr_status vr_event_elapsed_ms(void *start, void *end, float *ms)
{
    if (!start || !end || !ms) return VR_ERR_INVALID;
    return hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)end) == hipSuccess ? VR_OK : VR_ERR_NO_DEVICE;
}
vr_status vr_malloc_host(void **host, int64_t bytes)
{
    if (!host || bytes <= 0) return VR_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return VR_ERR_NO_DEVICE;
    return hipHostMalloc(host, (size_t)bytes, hipHostMallocDefault) == hipSuccess ? VR_OK : VR_ERR_OOM;
}
vr_status vr_free_host(void *host) { if (host) hipHostFree(host); return VR_OK; }
vr_status vr_upload_async(void *dst_dev, const void *src_host, int64_t bytes, void *stream)
{
    if (!dst_dev || !src_host || bytes <= 0) return VR_ERR_INVALID;
    return hipMemcpyAsync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream) == hipSuccess ? VR_OK : VR_ERR_NO_DEVICE;
}
vr_status vr_download_async(void *dst_host, const void *src_dev, int64_t bytes, void *stream)
{
    if (!dst_host || !src_dev || bytes <= 0) return VR_ERR_INVALID;
    return hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess ? VR_OK : VR_ERR_NO_DEVICE;
}

} // extern "C"
