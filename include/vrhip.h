/*
 * vrhip.h -- C ABI of the MI355X-native kd-tree volume codec + ray-march compositor.
 *
 * This is the drop-in boundary for ONE hot path of AugmentariumLab/VolumeRenderer:
 *   ingest   VolumeReader<T>::LoadBricksToTexture        volume_renderer/VolumeReader.h:151-223
 *   encode   VolumeKdtree::build                          volume_renderer/VolumeKdTree_recover.cpp:17-140
 *   decode   VolumeKdtree::levelCut                       volume_renderer/VolumeKdTree_recover.cpp:726-835
 *   file     VolumeKdtree::save / open                    volume_renderer/VolumeKdTree_recover.cpp:521-594
 *   4-bit    MidRangeTree::build / convertToByteArray     volume_renderer/MidRangeTree.cpp:17-176,1095-1128
 *   render   raycaster.frag / isosurface.frag + UnitBrick::Draw
 *                                                         volume_renderer/raycaster.frag:18-86,
 *                                                         volume_renderer/isosurface.frag:77-159,
 *                                                         volume_renderer/UnitBrick.h:98-100
 * The reference has no FFI layer; its boundary is the public surface of those C++
 * classes, called from main.cpp:142-290,358-404.  The headers in include/vrhip/ re-create
 * that surface (same class and method names) as a header-only facade over the
 * functions below.  Everything behind this header is hand-written HIP for gfx950;
 * there is NO CPU fallback: every compute entry point returns VR_ERR_NO_DEVICE
 * when no HIP device is usable.
 *
 * Conventions
 *  - plain C, no exceptions across the boundary, every function returns vr_status;
 *  - volumes are uint8, x fastest: cell(x,y,z) = x + X*y + X*Y*z (R.cpp:4-6);
 *  - "dev" pointers are HIP device pointers, "host" pointers ordinary memory;
 *    the caller owns every buffer it passes;
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls
 *    that return data to the host synchronise that stream before returning;
 *    device-to-device calls are asynchronous on it.
 *  - a vr_brickset is a batch of B independent bricks of identical dimensions, one
 *    kd-tree per brick (a single VolumeKdtree is a brickset with B = 1).  All B
 *    trees are built / decoded by the same batched kernel launches.
 */
#ifndef VRHIP_H
#define VRHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t vr_status;
enum {
    VR_OK = 0,
    VR_ERR_INVALID = -1,     /* bad argument (null pointer, non-positive dims, negative tolerance ...) */
    VR_ERR_NO_DEVICE = -2,   /* no usable HIP device / HIP call failed; there is no CPU fallback */
    VR_ERR_OOM = -3,
    VR_ERR_IO = -4,          /* save/open: file missing or short (reference: exit(-1), R.cpp:560-565) */
    VR_ERR_STATE = -5,       /* e.g. decode before build, save of an empty tree (R.cpp:526-530) */
    VR_ERR_FORMAT = -6,      /* malformed tree stream */
    VR_ERR_UNSUPPORTED = -7
};

/* Which reference class the brickset mirrors. */
enum {
    VR_VARIANT_RECOVER = 0,  /* VolumeKdTree_recover.cpp (live copy)                         */
    VR_VARIANT_GUARDED = 1,  /* VolumeKdtree.cpp:333 guard -- byte-identical output, less work */
    VR_VARIANT_MIDRANGE = 2  /* MidRangeTree.cpp: second 2-bit stream + 4-bit packing         */
};

typedef struct vr_brickset vr_brickset;

/* Mirrors the public data members of class VolumeKdtree (VolumeKdtree_recover.h:57-79). */
typedef struct vr_tree_info {
    int64_t X, Y, Z;             /* brick dimensions                               */
    int32_t orig_tree_depth;     /* origTreeDepth  (R.cpp:29)                      */
    int32_t max_tree_depth;      /* maxTreeDepth = orig + 7 (R.cpp:30)             */
    int64_t num_active_nodes;    /* numActiveNodes (R.cpp:714)                     */
    int64_t tree_bytes;          /* tree.bytes() = ceil(numActiveNodes / 4)        */
    int32_t tolerance, max_epochs, variant;
    int32_t num_reverts;         /* gradient-descent reverts taken (defect C-2 indicator); MidRangeTree: both streams' */
    int32_t max_error_before;    /* encoder's own leaf max error before branch growth (R.cpp:71-76) */
    int32_t max_error_after;     /* ... after branch growth (R.cpp:115-120)        */
    double  mean_l1_after;       /* (R.cpp:122-129)                                */
    int32_t zero_run_rewrites;   /* grown branches that ended on an evaluated "keep": the reference rewrites such a run
                                  * of zeros to 3s (R.cpp:662-669,686-688).  Provably impossible for tolerance >= 0, so
                                  * the GPU emitters count instead of rewriting; anything but 0 here means the stream
                                  * differs from the reference's */
    int32_t est_exact_segments;  /* diagnostic: 1024-node segments the distance estimator (R.cpp:415-455) had to walk node by
                                  * node instead of taking from their summaries; 0 for opened / set trees */
} vr_tree_info;

/* ---- library / device ------------------------------------------------------- */
vr_status vr_device_count(int32_t *count);                 /* VR_OK + 0 devices is possible (CPU-only box) */
vr_status vr_set_device(int32_t device);
const char *vr_status_string(vr_status s);
const char *vr_version(void);

/* ---- device buffers for host code that does not include HIP headers ----------------
 * (the reference's host code hands over std::vector<byte>; the facade in include/vrhip/
 * stages it through these).  Copies synchronise `stream` before returning. */
vr_status vr_malloc(void **dev, int64_t bytes);
vr_status vr_free(void *dev);
vr_status vr_upload(void *dst_dev, const void *src_host, int64_t bytes, void *stream);
vr_status vr_download(void *dst_host, const void *src_dev, int64_t bytes, void *stream);

/* ---- brickset life cycle ------------------------------------------------------
 * VolumeKdtree(std::vector<byte>&, x, y, z) + setErrorTolerance + setMaxEpochs
 * (VolumeKdtree_recover.h:103-112, R.cpp:9-15).  Any extents are accepted, as in the reference
 * (R.cpp:26-36: origTreeDepth = the sum of the floors of the three log2; R.cpp:151-162: the split rule that
 * follows from it -- unequal halves, an axis order that differs from node to node, leaves that read the min
 * corner of a box of several cells, R.cpp:194-195, and levelCut's boxes that leave some voxels unwritten,
 * R.cpp:759-766 with 806-833; all reproduced bit for bit).  Power-of-two extents up to 1024 per axis (the
 * reference's brick sizes 256x256x128 / 256^3) take the tiled fast kernels, everything else table-driven ones.
 * Limits of one tree: origTreeDepth <= 31, fewer than 2^32 voxels and at most 2^20 cells per axis; deeper than 28
 * levels (a stream can then pass 2^32 tokens: 64-bit token offsets, table-driven kernels) only with num_bricks == 1
 * -> VR_ERR_UNSUPPORTED beyond.  The reference program's own 2048x2048x768 tree (main.cpp:242-251) is 31 levels. */
vr_status vr_brickset_create(vr_brickset **out, int32_t num_bricks, const int64_t dims[3],
                             int32_t tolerance, int32_t max_epochs, int32_t variant);
vr_status vr_brickset_destroy(vr_brickset *bs);
vr_status vr_brickset_set_error_tolerance(vr_brickset *bs, int32_t tolerance);   /* R.cpp:9-11  */
vr_status vr_brickset_set_max_epochs(vr_brickset *bs, int32_t max_epochs);       /* R.cpp:13-15 */

/* ---- encode: VolumeKdtree::build (R.cpp:17-140) --------------------------------
 * voxels_dev: num_bricks * X*Y*Z bytes, brick b at offset b*X*Y*Z, each x-fastest.
 * Unlike the reference (R.cpp:51-52) the input buffer is left untouched.
 * Asynchronous on `stream`; vr_brickset_info()/get_* synchronise. */
vr_status vr_brickset_build(vr_brickset *bs, const uint8_t *voxels_dev, void *stream);

/* Public members after build(): per-brick info, tree bytes (TwoBitArray::bits of
 * the preorder stream), distanceMap (maxTreeDepth+1 bytes). */
vr_status vr_brickset_info(vr_brickset *bs, int32_t brick, vr_tree_info *info);
vr_status vr_brickset_get_tree(vr_brickset *bs, int32_t brick, uint8_t *dst_host, int64_t capacity);
vr_status vr_brickset_get_distance_map(vr_brickset *bs, int32_t brick, uint8_t *dst_host, int32_t capacity);
/* MidRangeTree only: tree_range.bits, distanceMap_range, convertToByteArray (M.cpp:1095-1128);
 * packed length is returned through *length (pass dst_host = NULL to query it). */
vr_status vr_brickset_get_tree_range(vr_brickset *bs, int32_t brick, uint8_t *dst_host, int64_t capacity);
vr_status vr_brickset_get_distance_map_range(vr_brickset *bs, int32_t brick, uint8_t *dst_host, int32_t capacity);
vr_status vr_brickset_get_packed4(vr_brickset *bs, int32_t brick, uint8_t *dst_host, int64_t capacity, int64_t *length);

/* ---- decode: VolumeKdtree::levelCut (R.cpp:726-835) ----------------------------
 * out_dev: num_bricks * X*Y*Z bytes.  cut_depth == max_tree_depth (or < 0) is the reference's
 * levelCut, bit-exact.  0 <= cut_depth < max_tree_depth is a PROGRESSIVE cut with defined
 * semantics (new: the reference's walk de-synchronises there): the stream is
 * parsed completely, refinement stops below the cut, every voxel gets the decoded scalar of its
 * ancestor at depth min(cut_depth, depth of its terminal node).  Asynchronous on `stream`. */
vr_status vr_brickset_decode(vr_brickset *bs, int32_t cut_depth, uint8_t *out_dev, void *stream);

/* MidRangeTree only (new: the reference builds the half-range stream, MidRangeTree.cpp:399-544, 871-982, but
 * never decodes it -- its levelCut, :984-1093, reads the mid stream alone): the same progressive
 * decode applied to the range stream, i.e. per voxel the half range of its terminal node's box as the
 * encoder reconstructed it (distanceMap_range, codes of tree_range).  With vr_brickset_decode this gives
 * [mid - range, mid + range] bounds at any cut depth (coarse-to-fine refinement, empty-space tests).
 * VR_ERR_STATE for other variants, VR_ERR_UNSUPPORTED for a set opened from a file. */
vr_status vr_brickset_decode_range(vr_brickset *bs, int32_t cut_depth, uint8_t *out_dev, void *stream);

/* Install a foreign preorder stream (e.g. read from a reference-written file) as
 * brick `brick`: builds the decode side-car index from the bytes alone. */
vr_status vr_brickset_set_tree(vr_brickset *bs, int32_t brick, const uint8_t *tree_host, int64_t tree_bytes,
                               int64_t num_active_nodes, const uint8_t *distance_map_host, int32_t map_len);

/* ---- file format: VolumeKdtree::save / open (R.cpp:521-594) --------------------
 * Byte-identical to the reference's file: rootMin,rootMax (3x int64 each),
 * maxTreeDepth, origTreeDepth (int32), X,Y,Z,numActiveNodes (int64), distanceMap,
 * tree bytes.  vr_brickset_open creates a 1-brick set. */
vr_status vr_brickset_save(vr_brickset *bs, int32_t brick, const char *path);
vr_status vr_brickset_open(vr_brickset **out, const char *path);
/* MidRangeTree::save / open (MidRangeTree.cpp:753-833).  A VR_VARIANT_MIDRANGE set saves the
 * reference's MidRangeTree layout byte for byte: the same 88-byte header, distanceMap,
 * distanceMap_range, tree bytes, tree_range bytes.  vr_brickset_open_variant(…, VR_VARIANT_MIDRANGE)
 * reads such a file back exactly (the reference's reader mis-sizes the streams by 4 bytes and returns
 * a shifted range stream: nothing to match); other variants forward to vr_brickset_open. */
vr_status vr_brickset_open_variant(vr_brickset **out, const char *path, int32_t variant);

/* ---- error helpers: measureMaxError / measureMeanError / queryError (R.cpp:386-411)
 * The reference dereferences the input it has already cleared (SURVEY C-7); here the
 * original volume is passed explicitly.  n = number of voxels. */
vr_status vr_measure_error(const uint8_t *decoded_dev, const uint8_t *original_dev, int64_t n,
                           int32_t *max_error, double *mean_error, void *stream);
vr_status vr_query_error(const uint8_t *decoded_dev, const uint8_t *original_dev, int64_t n,
                         uint8_t *error_dev, void *stream);

/* ---- ingest: VolumeReader<T>::LoadBricksToTexture (VolumeReader.h:151-223) ------
 * Places brick b (brick_dims, x-fastest, contiguous at bricks_dev + b*brick_voxels)
 * at grid cell brick_ijk[3*b..3*b+2] of a global x-fastest volume of
 * (I*X, J*Y, K*Z) voxels: volume_dev must hold I*J*K*X*Y*Z bytes whatever num_bricks is (grid cells
 * without a brick are left untouched; the reference sizes its array by numBricks, VolumeReader.h:163-168,
 * and overruns it for sparse brick lists -- not reproduced).  64-bit indices (the reference's 32-bit ones wrap above
 * 2^32 voxels, VolumeReader.h:171).  vr_disassemble_bricks is the inverse (global
 * volume -> contiguous bricks), used to feed per-brick trees. */
vr_status vr_assemble_bricks(const uint8_t *bricks_dev, int32_t num_bricks, const int64_t brick_dims[3],
                             const int64_t *brick_ijk, const int64_t grid[3], uint8_t *volume_dev, void *stream);
vr_status vr_disassemble_bricks(const uint8_t *volume_dev, int32_t num_bricks, const int64_t brick_dims[3],
                                const int64_t *brick_ijk, const int64_t grid[3], uint8_t *bricks_dev, void *stream);

/* ---- render: raycaster.frag / isosurface.frag on the UnitBrick proxy cube --------
 * Camera = the values main.cpp feeds glm::lookAt / glm::perspectiveFov (main.cpp:33-40,396-397). */
typedef struct vr_camera {
    float pos[3];      /* cameraPos   (0,0,-0.75)  */
    float front[3];    /* cameraFront (0,0,1)      */
    float up[3];       /* cameraUp    (0,1,0)      */
    float fov_deg;     /* fov 50                   */
    float z_near, z_far; /* 0.1, 100               */
} vr_camera;

enum { VR_RENDER_COMPOSITE = 0 /* raycaster.frag */, VR_RENDER_ISOSURFACE = 1 /* isosurface.frag */,
       VR_RENDER_PARTIAL = 2 /* raycaster.frag accumulation as an (rgb-premultiplied c, transmittance) pair for sort-last compositing */ };

typedef struct vr_render_params {
    int32_t width, height;   /* 1600x1200 in the reference (main.cpp:27); bench uses 1920x1080 */
    float step_size[3];      /* uniform step_size = 1/BRICK_DIM (main.cpp:330-331)             */
    float iso_value;         /* uniform isoValue = currIsoVal/255 (main.cpp:334)               */
    int32_t max_samples;     /* MAX_SAMPLES = 300 (raycaster.frag:14)                          */
    int32_t mode;            /* VR_RENDER_*                                                    */
    /* Sort-last multi-GPU path (VR_RENDER_PARTIAL): the rank owns the samples whose texture-space
     * position lies in [box_min, box_max); volume_dev holds the voxels [vol_origin, vol_origin+dims)
     * of a global volume of global_dims voxels (own slab plus halo layers for filtering).
     * Single-GPU path: box {0,0,0}-{1,1,1}, global_dims {0,0,0} (= dims), vol_origin {0,0,0}. */
    float box_min[3], box_max[3];
    int64_t global_dims[3];
    int64_t vol_origin[3];
    int32_t no_early_exit;   /* 1: ignore the alpha>0.99 exit (reference for the sort-last path)   */
    /* Empty-space skipping (new; the reference only hints at it, isosurface_compressed.frag:23-29): skip_grid_dev = a
     * grid built by vr_skip_grid_build from THE SAME volume_dev, skip_cell its cell size in voxels; NULL / 0 = off.
     * A sample whose eight taps are provably all zero (compositor) or provably all on one side of the iso value is not
     * fetched; the ray still advances sample by sample, so the frame is bit-identical to the one without the grid. */
    int32_t skip_cell;
    const uint8_t *skip_grid_dev;
} vr_render_params;

/* volume_dev: X*Y*Z uint8 (the 3-D texture contents, GL_RED/GL_UNSIGNED_BYTE, GL_LINEAR,
 * clamp-to-edge: VolumeReader.h:114-127).  rgba_dev: height*width*4 float32, row 0 = top.
 * Pixels not covered by the cube are white (main.cpp:392). */
vr_status vr_raycast(const uint8_t *volume_dev, const int64_t dims[3], const vr_camera *cam,
                     const vr_render_params *params, float *rgba_dev, void *stream);

/* (min, max) of every skip_cell^3 cell of the volume, widened by the one voxel a trilinear fetch reaches beyond its
 * base voxel: grid_dev holds 2 bytes per cell, cells x fastest, ceil(dims / skip_cell) cells per axis.  Exact bounds of
 * the DECODED voxels: MidRangeTree's half-range stream would give bounds of the original data one level above, but the
 * decoded scalar of an internal node is a prediction with no error bound, so mid +- range at a coarse cut is not a safe
 * bracket (vr_brickset_decode_range remains available for previews). */
vr_status vr_skip_grid_build(const uint8_t *volume_dev, const int64_t dims[3], int32_t skip_cell, uint8_t *grid_dev,
                             void *stream);

/* Sort-last compositing of VR_RENDER_PARTIAL images: front = front OVER back, per pixel
 * (c1 + t1*c2, t1*t2); and the final colour transfer of raycaster.frag:82-85. */
vr_status vr_composite_over(float *front_dev, const float *back_dev, int64_t num_pixels, void *stream);
vr_status vr_composite_finish(const float *partial_dev, float *rgba_dev, int64_t num_pixels, void *stream);
/* Direct-send sort-last compositing of one image tile: partials_dev holds num_slabs partial images
 * of this tile back to back (num_pixels*4 floats each, slab s = the s-th slab along `axis` of the
 * volume).  Every pixel combines them front to back in ITS view order (ascending slab index where
 * the pixel's ray direction along `axis` is >= 0, descending otherwise -- "over" is associative but
 * not commutative) and applies the colour transfer.  first_pixel = row-major index of the tile's
 * first pixel in the width*height frame described by params. */
vr_status vr_composite_slabs(const float *partials_dev, int32_t num_slabs, int64_t num_pixels, int64_t first_pixel,
                             int32_t axis, const vr_camera *cam, const vr_render_params *params, float *rgba_dev,
                             void *stream);

/* ---- sort-last compositing across the GPUs of a node (new: the reference is single-GPU) ----------------
 * One process per GPU.  Rank r ray-marches slab r of the volume along `axis` into a VR_RENDER_PARTIAL image
 * (width * height * 4 floats); vr_compositor_composite moves tile t (a block of rows) of every rank's image to rank t
 * in ONE grouped RCCL call (ncclSend / ncclRecv over xGMI), combines the `world` partials of its tile per pixel in that
 * pixel's view order (vr_composite_slabs) and gathers the finished RGBA tiles on rank 0 (rgba_dev: width * height * 4
 * floats there, ignored elsewhere).  All of it is queued on `stream`.  The communicator is either the library's own,
 * created from an ncclUniqueId the ranks share (vr_rccl_unique_id on one rank, passed on by whatever the host uses
 * to talk between its processes), or the caller's ncclComm_t.  RCCL is bound at run time: world == 1 needs none, and
 * VR_ERR_UNSUPPORTED comes back where it cannot be found.  The alpha > 0.99 early exit of raycaster.frag:76 cannot be
 * honoured across slabs (bounded difference, <= 0.017 per channel). */
typedef struct vr_compositor vr_compositor;
vr_status vr_rccl_unique_id(uint8_t id[128]);
vr_status vr_compositor_create(vr_compositor **out, const uint8_t id[128], int32_t rank, int32_t world,
                               int32_t width, int32_t height);
vr_status vr_compositor_create_from_comm(vr_compositor **out, void *nccl_comm, int32_t rank, int32_t world,
                                         int32_t width, int32_t height);
vr_status vr_compositor_composite(vr_compositor *c, const float *partial_dev, int32_t axis, const vr_camera *cam,
                                  const vr_render_params *params, float *rgba_dev, void *stream);
vr_status vr_compositor_destroy(vr_compositor *c);

/* ---- streams, events, pinned host memory (what include/vrhip/TimestepStreamer.hpp overlaps the stages of a timestep
 * stream with: main.cpp:242-290 runs them one after another).  Streams and events travel as void*. */
vr_status vr_stream_create(void **stream);
vr_status vr_stream_destroy(void *stream);
vr_status vr_stream_synchronize(void *stream);
vr_status vr_stream_wait_event(void *stream, void *event);
vr_status vr_event_create(void **event);
vr_status vr_event_destroy(void *event);
vr_status vr_event_record(void *event, void *stream);
vr_status vr_event_synchronize(void *event);
vr_status vr_event_elapsed_ms(void *start_event, void *end_event, float *ms);
vr_status vr_malloc_host(void **host, int64_t bytes);
vr_status vr_free_host(void *host);
vr_status vr_upload_async(void *dst_dev, const void *src_host, int64_t bytes, void *stream);
vr_status vr_download_async(void *dst_host, const void *src_dev, int64_t bytes, void *stream);

/* ---- instrumentation (the reference's DebugTimer phases, R.cpp:47-113) ----------
 * Milliseconds of the last build / decode measured with hipEvents on the call's stream:
 * phases[0..4] = BUILD(pyramid), COMPRESS, PRUNE, CONVERT, DECODE. */
vr_status vr_brickset_last_timings(vr_brickset *bs, float phases_ms[5]);

/* ---- concurrency inside one build (new; the reference's build(useThreads) is its nearest relative, R.cpp:17) ----
 * The level loop (compressGradientDescent) of a VolumeKdtree set is a chain of wide kernels and of one-wave-per-brick
 * control steps; run over `level_loop_streams` ranges of the bricks side by side (internal streams, forked from and
 * joined to the call's stream by events) the control steps of one range hide behind the wide kernels of another.
 * 1 = off, 2 = default, up to 4.  One build alone on the device: 2 is 5 % and 4 is 8 % faster than 1; with several
 * bricksets in flight on streams of their own, 1 is the right choice (the sets already fill each other's gaps).
 * Results do not depend on it.  MidRangeTree sets always run their two streams' level loops side by side. */
vr_status vr_brickset_set_concurrency(vr_brickset *bs, int32_t level_loop_streams);

/* ---- the contiguous stream (R.cpp:631-718, tree.swap(preorderTree)) ----------------------------------------------
 * A build of a brick of 4096 leaves or more keeps every 4096-leaf block's token string in a slot of its own (what the
 * decoders read, in place) and, as its last step, writes the reference's contiguous byte stream beside it: the bytes
 * vr_brickset_get_tree / save / get_packed4 hand out.  on_build = 0 leaves that copy to the first call that asks for
 * bytes (a pipeline that only ever decodes on the device saves ~4 % of a build); 1 (default) is the reference's build().
 * The contiguous streams of all bricks lie back to back in one buffer sized from their real lengths (about 0.6 byte
 * per voxel for the bench volume), regrown on demand. */
vr_status vr_brickset_set_compaction(vr_brickset *bs, int32_t on_build);

/* ---- debugging switches (new) ----------------------------------------------------------------------------------
 * Which kernel serves a call is decided by the set's geometry and by a few switches kept IN THE HANDLE: they are
 * initialised from the environment (VRHIP_DECODE_WALK, VRHIP_DECODE_FINE_V1, VRHIP_DECODE_QUAD, VRHIP_DECODE_V1,
 * VRHIP_NO_FUSED_EMIT, VRHIP_NO_SKIP_BLOCKS, VRHIP_NOSWZ, VRHIP_MR_SERIAL, VRHIP_FORK_BRICKS) once, when the set is
 * created, and changed afterwards only through this call -- never by the environment at launch time.  Names:
 * "decode_walk", "decode_fine_v1", "decode_quad", "decode_v1", "no_skip_blocks", "noswz", "mr_serial",
 * "fork_bricks" (0..4), "no_fused_emit" (before the first build only: VR_ERR_STATE afterwards).  Results never
 * depend on a switch; the tests use them to check the kernels against each other.
 * vr_debug_set: process-wide switches that belong to no set: "skip_grid_v1". */
vr_status vr_brickset_set_switch(vr_brickset *bs, const char *name, int32_t value);
vr_status vr_debug_set(const char *name, int32_t value);

#ifdef __cplusplus
}
#endif
#endif /* VRHIP_H */
