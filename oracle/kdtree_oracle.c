/*
 * kdtree_oracle.c -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by
 * or called from the product path (volumerenderer_amd/, include/, libvrhip.so).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * A literal, serial, double-precision CPU restatement of the reference's
 * progressive kd-tree codec (class VolumeKdtree, the live copy):
 *   /root/reference/volume_renderer/VolumeKdTree_recover.cpp  ("R.cpp" below)
 *   /root/reference/volume_renderer/VolumeKdtree_recover.h    ("R.h")
 *   /root/reference/volume_renderer/TwoBitArray.h
 * plus the second (half-range) stream and 4-bit packing of
 *   /root/reference/volume_renderer/MidRangeTree.cpp          ("M.cpp")
 * Each function cites the lines it follows.  Written from the algorithm, not
 * copied: no Eigen, no PPL, no std::stack; plain C99.
 *
 * PARITY UNPINNED (see DESIGN.md "Oracle"): the reference needs <ppl.h> (MSVC) and
 * Eigen, neither present in this image, and building it with stand-in headers
 * is not permitted, so oracle/_ref does not exist; the reference ships no
 * tests or golden vectors.  The only reference outputs available are the
 * known-answer values the survey session recorded (SURVEY.md section 8c /
 * Appendix B: numActiveNodes, full distanceMap, FNV-1a-64 of tree bytes and of
 * decoded voxels for 16^3/128^3/256^3 sphere_n3, the MidRangeTree 32^3 hashes)
 * -- from a build with stand-in headers, which does not count as a pin.
 * tests/test_oracle_golden.py reproduces all of them all the same.
 *
 * Reference defects reproduced on purpose:
 *  C-1 currentError/currentDF/currentStepSize start at 0.0 and carry over
 *      across epochs AND levels (R.cpp:225,311) -- the zero-init reading.
 *  C-2 a gradient-descent revert restores recon but not the 2-bit codes
 *      (R.cpp:323-331).
 *  C-3 parity is with the serial path build(false).
 *  C-4 levelCut is only meaningful at cutDepth == maxTreeDepth.
 *  C-6 open() over-allocates the tree by 8 bytes (R.cpp:581).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint8_t byte;

#define VKO_MAX_DIM 3
#define VKO_MAX_TRACE 4096

/* One record per gradient-descent epoch, for tests that need to know whether a
 * revert (defect C-2) happened and for debugging the GPU control kernel. */
typedef struct {
    int32_t depth, epoch, kind; /* kind: 0 start estimate, 1 fill pass, 2 revert, 3 break(err<1), 4 break(same dist), 5 df */
    double distance, error, df, step;
} vko_trace_rec;

typedef struct vko_tree {
    int64_t X, Y, Z;
    int64_t rootMin[3], rootMax[3];
    int32_t tolerance, maxEpochs, guarded;
    int32_t origTreeDepth, maxTreeDepth;
    int64_t numOrigNodes, numMaxNodes, firstOrigLeaf, numActiveNodes;
    byte *distanceMap;
    int32_t distanceMapLen;
    byte *tree; /* TwoBitArray::bits */
    int64_t treeBytes;
    byte *temp; /* BFS midrange array; leaves only after the compress stage */
    int64_t tempLen;
    byte *recon; /* leaf-level reconstruction */
    int64_t reconLen;
    byte *reconAll; /* debugging aid: final recon of EVERY level, BFS order (not in the reference) */
    const byte *data; /* caller's voxels, x fastest (R.cpp:4-6); not owned */
    /* MidRangeTree second stream (M.cpp) */
    int32_t midrange; /* 0 = VolumeKdtree, 1 = MidRangeTree */
    byte *distanceMapRange;
    byte *treeRange;
    int64_t treeRangeBytes;
    byte *tempRange;
    byte *reconRange;
    /* encoder self-reported leaf statistics (R.cpp:71-84,115-129) */
    int32_t maxErrBefore, maxErrAfter;
    double meanL1Before, meanL2Before, meanL1After, meanL2After;
    int32_t zeroRunRewrites; /* times R.cpp:686-688 fired (believed unreachable) */
    int32_t numReverts;
    vko_trace_rec *trace;
    int32_t traceLen;
    int32_t stage; /* 0 none, 1 pyramid, 2 compressed, 3 pruned, 4 converted */
} vko_tree;

/* ---- TwoBitArray (TwoBitArray.h:30-53): element i in byte i/4, bits 2*(i&3) */
static inline int tb_get(const byte *bits, int64_t i) { return (bits[i >> 2] >> ((i & 3) * 2)) & 3; }
static inline void tb_set(byte *bits, int64_t i, int v)
{
    byte mask = (byte)~(3 << ((i & 3) * 2));
    bits[i >> 2] = (byte)((bits[i >> 2] & mask) | (v << ((i & 3) * 2)));
}
static inline int64_t tb_bytes(int64_t n) { return (n + 3) / 4; }

static void trace_add(vko_tree *t, int depth, int epoch, int kind, double dist, double err, double df, double step)
{
    if (!t->trace || t->traceLen >= VKO_MAX_TRACE) return;
    vko_trace_rec *r = &t->trace[t->traceLen++];
    r->depth = depth; r->epoch = epoch; r->kind = kind;
    r->distance = dist; r->error = err; r->df = df; r->step = step;
}

/* R.h:103-112 constructor defaults; R.h:89-94 */
vko_tree *vko_create(const byte *voxels, int64_t x, int64_t y, int64_t z)
{
    vko_tree *t = (vko_tree *)calloc(1, sizeof(vko_tree));
    if (!t) return NULL;
    t->data = voxels;
    t->X = x; t->Y = y; t->Z = z;
    t->rootMax[0] = x; t->rootMax[1] = y; t->rootMax[2] = z;
    t->tolerance = 6;
    t->maxEpochs = 5;
    t->trace = (vko_trace_rec *)calloc(VKO_MAX_TRACE, sizeof(vko_trace_rec));
    return t;
}

void vko_destroy(vko_tree *t)
{
    if (!t) return;
    free(t->distanceMap); free(t->tree); free(t->temp); free(t->recon);
    free(t->distanceMapRange); free(t->treeRange); free(t->tempRange); free(t->reconRange);
    free(t->trace);
    free(t->reconAll);
    free(t);
}

void vko_set_error_tolerance(vko_tree *t, int tol) { t->tolerance = tol; }  /* R.cpp:9-11 */
void vko_set_max_epochs(vko_tree *t, int e) { t->maxEpochs = e; }           /* R.cpp:13-15 */
/* guarded=1: VolumeKdtree.cpp:333 / M.cpp:340 skip the DF evaluation in the last epoch */
void vko_set_guarded(vko_tree *t, int g) { t->guarded = g; }
void vko_set_midrange(vko_tree *t, int m) { t->midrange = m; }

static inline int64_t get_cell(const vko_tree *t, int64_t x, int64_t y, int64_t z)
{
    return x + t->X * y + t->X * t->Y * z; /* R.cpp:4-6 */
}

/* Split-axis rule shared by build (R.cpp:151-159) and decode (R.cpp:793-797). */
static int split_dim_build(int depth, const int64_t *mn, const int64_t *mx)
{
    int splitDim = depth % VKO_MAX_DIM;
    int64_t ext[3] = { mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2] };
    int64_t numCells = ext[0] * ext[1] * ext[2];
    int i = 0;
    while (numCells > 1 && ext[splitDim] == 1) splitDim = (depth + ++i) % VKO_MAX_DIM;
    return splitDim;
}

/* R.cpp:143-201 (and M.cpp:180-238 for the half-range array) */
static void build_recursive(vko_tree *t, int64_t idx, int depth, const int64_t *mnIn, const int64_t *mxIn,
                            byte *outMin, byte *outMax)
{
    double thisMin = 0.0, thisMax = 0.0;
    int64_t mn[3] = { mnIn[0], mnIn[1], mnIn[2] };
    int64_t mx[3] = { mxIn[0], mxIn[1], mxIn[2] };
    if (depth < t->origTreeDepth) {
        int sd = split_dim_build(depth, mn, mx);
        int64_t thisMid = (mn[sd] + mx[sd]) / 2;
        int64_t thisMaxB = mx[sd];
        byte lmin, lmax, rmin, rmax;
        mx[sd] = thisMid;
        build_recursive(t, 2 * idx + 1, depth + 1, mn, mx, &lmin, &lmax);
        mn[sd] = thisMid;
        mx[sd] = thisMaxB;
        build_recursive(t, 2 * idx + 2, depth + 1, mn, mx, &rmin, &rmax);
        thisMin = fmin((double)lmin, (double)rmin);
        thisMax = fmax((double)lmax, (double)rmax);
    } else if (depth == t->origTreeDepth) {
        thisMax = thisMin = (double)t->data[get_cell(t, mn[0], mn[1], mn[2])];
    }
    t->temp[idx] = (byte)((thisMax + thisMin) / 2.0); /* R.cpp:198 midrange */
    if (t->midrange) t->tempRange[idx] = (byte)((thisMax - thisMin) / 2.0); /* M.cpp:235 */
    *outMin = (byte)thisMin;
    *outMax = (byte)thisMax;
}

/* R.cpp:415-455 -- running-mean distance estimator.  Literal double arithmetic. */
static void encode_node_estimate(byte truth, byte parent, double *estimateSum, double *estimateCount)
{
    double nodeTruth = (double)truth;
    double parentEstimate = (double)parent;
    double parentDistance = fabs(parentEstimate - nodeTruth);
    double masterDistance = (*estimateSum + parentDistance) / (*estimateCount + 1.0);
    double noneError = parentDistance;
    double addEstimate = fmin(255.0, parentEstimate + masterDistance);
    double addError = fabs(addEstimate - nodeTruth);
    double subEstimate = fmax(0.0, parentEstimate - masterDistance);
    double subError = fabs(subEstimate - nodeTruth);
    double minError = fmin(subError, fmin(noneError, addError));
    if (minError == noneError) return;
    /* add or sub wins: both branches update the running mean identically */
    *estimateSum += parentDistance;
    *estimateCount += 1.0;
}

/* R.cpp:457-502 -- ties resolved keep(0) > add(1) > sub(2). */
static inline byte encode_node(byte truth, byte parent, byte distanceVal, int *codeOut, double *errOut)
{
    double nodeTruth = (double)truth;
    double parentEstimate = (double)parent;
    double parentDistance = fabs(parentEstimate - nodeTruth);
    double masterDistance = (double)distanceVal;
    double noneEstimate = parentEstimate;
    double noneError = parentDistance;
    double addEstimate = fmin(255.0, parentEstimate + masterDistance);
    double addError = fabs(addEstimate - nodeTruth);
    double subEstimate = fmax(0.0, parentEstimate - masterDistance);
    double subError = fabs(subEstimate - nodeTruth);
    double minError = fmin(subError, fmin(noneError, addError));
    if (errOut) *errOut = minError;
    if (minError == noneError) { *codeOut = 0; return (byte)noneEstimate; }
    if (minError == addError) { *codeOut = 1; return (byte)addEstimate; }
    *codeOut = 2;
    return (byte)subEstimate;
}

static inline double clampd(double v, double lo, double hi) { return fmin(hi, fmax(lo, v)); }

/* R.cpp:206-384 (M.cpp:241-397 and :399-544 are the same loop on temp / temp_range).
 * truthBFS: BFS array of node truths; treeBits: BFS 2-bit codes (written);
 * distMap: per-level distance (written 0..D); reconOut: leaf-level recon (malloc'd). */
static byte *compress_gradient_descent(vko_tree *t, const byte *truthBFS, byte *treeBits, byte *distMap,
                                       byte *reconAllOut)
{
    const double gamma = 1.25, h = 1.0, maxAbsStepSize = 4.0;
    const int D = t->origTreeDepth;
    int64_t maxNodes = (int64_t)1 << D;
    byte *recon = (byte *)calloc((size_t)maxNodes, 1);
    byte *parents = (byte *)calloc((size_t)maxNodes, 1);
    byte *reconPrev = (byte *)calloc((size_t)maxNodes, 1);
    int64_t startingNodeIdx = 0, endingNodeIdx = 0, parentStartingNodeIdx = 0;
    /* C-1: declared uninitialised at R.cpp:225; pinned to 0.0 */
    double currentDistance = 0.0, currentError = 0.0, currentDF = 0.0, currentStepSize = 0.0;
    double previousStepSize, previousDistance, previousDF = 0.0, previousError;

    for (int depth = 0; depth < D + 1; depth++) {
        int epoch = 0;
        double distanceSum = 0.0, distanceCount = 0.0;
        int64_t numNodes = (int64_t)1 << depth; /* pow(2, depth) R.cpp:234 */
        endingNodeIdx = startingNodeIdx + numNodes;
        memset(recon, 0, (size_t)numNodes); /* recon.resize after clear -> zeros */

        /* starting distance: serial running mean (R.cpp:254-266) */
        for (int64_t n = startingNodeIdx; n < endingNodeIdx; n++) {
            byte p = n == 0 ? 0 : parents[((n - 1) / 2) - parentStartingNodeIdx];
            encode_node_estimate(truthBFS[n], p, &distanceSum, &distanceCount);
        }
        if (distanceCount > 0) currentDistance = round(distanceSum / distanceCount);
        else currentDistance = 0.0;
        trace_add(t, depth, -1, 0, currentDistance, distanceSum, distanceCount, 0.0);

        previousDistance = 0.0;
        previousStepSize = 255.0;
        previousError = 65025.0;
        while (epoch < t->maxEpochs && fabs(previousStepSize) >= 0.5) {
            if (epoch != 0) {
                previousDistance = currentDistance;
                previousError = currentError;
                previousDF = currentDF;
                previousStepSize = currentStepSize;
                currentDistance = round(fmin(255.0, fmax(0.0, previousDistance + previousStepSize)));
                if (currentDistance == previousDistance) {
                    trace_add(t, depth, epoch, 4, currentDistance, currentError, currentDF, currentStepSize);
                    break;
                }
            }
            /* fill pass: writes codes + recon, accumulates err^2 WITHOUT reset (R.cpp:307-315) */
            {
                byte dv = (byte)currentDistance;
                for (int64_t n = startingNodeIdx; n < endingNodeIdx; n++) {
                    byte p = n == 0 ? 0 : parents[((n - 1) / 2) - parentStartingNodeIdx];
                    int code; double L1;
                    byte r = encode_node(truthBFS[n], p, dv, &code, &L1);
                    tb_set(treeBits, n, code);
                    currentError += L1 * L1; /* pow(L1,2.0) is exact for these integers */
                    recon[n - startingNodeIdx] = r;
                }
            }
            currentError /= (double)numNodes;
            trace_add(t, depth, epoch, 1, currentDistance, currentError, currentDF, currentStepSize);

            if (currentError < 1.0) {
                trace_add(t, depth, epoch, 3, currentDistance, currentError, currentDF, currentStepSize);
                break;
            }
            if (epoch != 0 && currentError > previousError) { /* revert, R.cpp:323-331 (C-2) */
                currentError = previousError;
                currentDistance = previousDistance;
                currentDF = previousDF;
                currentStepSize = previousStepSize / 2.0;
                { byte *tmp = recon; recon = reconPrev; reconPrev = tmp; } /* vector::swap */
                t->numReverts++;
                trace_add(t, depth, epoch, 2, currentDistance, currentError, currentDF, currentStepSize);
                epoch++;
                continue;
            }
            if (!t->guarded || epoch + 1 < t->maxEpochs) {
                byte est[2] = { (byte)fmax(0.0, currentDistance - h), (byte)fmin(255.0, currentDistance + h) };
                double estErr[2] = { 0.0, 0.0 };
                for (int i = 0; i < 2; i++) {
                    for (int64_t n = startingNodeIdx; n < endingNodeIdx; n++) {
                        byte p = n == 0 ? 0 : parents[((n - 1) / 2) - parentStartingNodeIdx];
                        int code; double L1;
                        encode_node(truthBFS[n], p, est[i], &code, &L1);
                        estErr[i] += L1 * L1;
                    }
                    estErr[i] /= (double)numNodes;
                }
                currentDF = (estErr[1] - estErr[0]) / (2.0 * h);
                currentStepSize = fmax(-maxAbsStepSize, fmin(maxAbsStepSize, -gamma * currentDF));
                memcpy(reconPrev, recon, (size_t)numNodes); /* reconPreviousEpoch = recon */
                trace_add(t, depth, epoch, 5, currentDistance, currentError, currentDF, currentStepSize);
            }
            epoch++;
        }
        distMap[depth] = (byte)currentDistance; /* R.cpp:369 */
        if (reconAllOut) memcpy(reconAllOut + startingNodeIdx, recon, (size_t)numNodes);

        if (depth < D) { /* R.cpp:374-381 */
            byte *tmp = parents; parents = recon; recon = tmp;
            parentStartingNodeIdx = startingNodeIdx;
            startingNodeIdx = endingNodeIdx;
        }
    }
    free(parents);
    free(reconPrev);
    return recon;
}

/* R.cpp:596-629, serial (C-3).  Iterative post-order by levels is equivalent
 * because a node only depends on its children; we keep the recursive form's
 * semantics: result(node) = pruned?  MidRangeTree (M.cpp:835-869) decides on
 * the mid stream only and marks the range code 3 alongside. */
static int prune_recursive(vko_tree *t, int64_t rootIdx, int rootDepth)
{
    int leftSub = 1, rightSub = 1, meets = 1;
    if (rootDepth < t->origTreeDepth) {
        leftSub = prune_recursive(t, 2 * rootIdx + 1, rootDepth + 1);
        rightSub = prune_recursive(t, 2 * rootIdx + 2, rootDepth + 1);
    }
    if (rootDepth == t->origTreeDepth) {
        int64_t ri = rootIdx - t->firstOrigLeaf;
        meets = abs((int)t->recon[ri] - (int)t->temp[ri]) < t->tolerance; /* R.cpp:620 */
    }
    if (leftSub && rightSub && tb_get(t->tree, rootIdx) == 0 && meets) {
        tb_set(t->tree, rootIdx, 3);
        if (t->midrange) tb_set(t->treeRange, rootIdx, 3); /* M.cpp:864-865: range code follows mid */
        return 1;
    }
    return 0;
}

typedef struct { int64_t inputIdx; int depth; int eval; int64_t zeroStartIdx; } cv_item;

/* R.cpp:631-724; M.cpp:871-982 emits the range stream in lock-step. */
static void convert_to_preorder(vko_tree *t)
{
    int64_t bytes = tb_bytes(t->numMaxNodes);
    byte *pre = (byte *)calloc((size_t)bytes, 1);
    byte *preR = t->midrange ? (byte *)calloc((size_t)bytes, 1) : NULL;
    cv_item *stack = (cv_item *)malloc(sizeof(cv_item) * (size_t)(t->maxTreeDepth + 8) * 2);
    int sp = 0;
    int64_t outputIdx = 0;
    stack[sp++] = (cv_item){ 0, 0, 0, -1 };
    while (sp > 0) {
        cv_item it = stack[sp - 1];
        int64_t inputIdx = it.inputIdx, zeroStartIdx = it.zeroStartIdx, reconIdx = 0;
        int depth = it.depth, eval = it.eval;
        int code = tb_get(t->tree, inputIdx);
        int codeR = t->midrange ? tb_get(t->treeRange, inputIdx) : 0;
        if (depth >= t->origTreeDepth) {
            reconIdx = inputIdx - t->firstOrigLeaf;
            if (eval) {
                int c;
                t->recon[reconIdx] = encode_node(t->temp[reconIdx], t->recon[reconIdx], t->distanceMap[depth], &c, NULL);
                tb_set(t->tree, inputIdx, c);
                code = c;
                if (t->midrange) {
                    int cr;
                    t->reconRange[reconIdx] = encode_node(t->tempRange[reconIdx], t->reconRange[reconIdx],
                                                          t->distanceMapRange[depth], &cr, NULL);
                    tb_set(t->treeRange, inputIdx, cr);
                    codeR = cr;
                }
                if (zeroStartIdx != -1) { if (code != 0) zeroStartIdx = -1; }
                else { if (code == 0) zeroStartIdx = outputIdx; }
            } else {
                if (depth > t->origTreeDepth) { code = 3; codeR = 3; }
            }
        }
        tb_set(pre, outputIdx, code);
        if (preR) tb_set(preR, outputIdx, codeR);
        outputIdx++;
        sp--;
        if (depth >= t->maxTreeDepth || code == 3) {
            if (zeroStartIdx != -1) {
                for (int64_t i = zeroStartIdx; i < outputIdx; i++) { tb_set(pre, i, 3); if (preR) tb_set(preR, i, 3); }
                t->zeroRunRewrites++;
            }
            continue;
        }
        if (depth >= t->origTreeDepth) {
            if (abs((int)t->recon[reconIdx] - (int)t->temp[reconIdx]) > t->tolerance)
                stack[sp++] = (cv_item){ inputIdx, depth + 1, 1, zeroStartIdx };
            else
                stack[sp++] = (cv_item){ inputIdx, depth + 1, 0, zeroStartIdx };
            continue;
        }
        stack[sp++] = (cv_item){ 2 * inputIdx + 2, depth + 1, 0, zeroStartIdx };
        stack[sp++] = (cv_item){ 2 * inputIdx + 1, depth + 1, 0, zeroStartIdx };
    }
    free(stack);
    t->numActiveNodes = outputIdx;
    t->treeBytes = tb_bytes(outputIdx); /* preorderTree.resize(numActiveNodes) R.cpp:715 */
    free(t->tree);
    t->tree = (byte *)realloc(pre, (size_t)(t->treeBytes > 0 ? t->treeBytes : 1));
    if (preR) {
        free(t->treeRange);
        t->treeRangeBytes = t->treeBytes;
        t->treeRange = (byte *)realloc(preR, (size_t)(t->treeBytes > 0 ? t->treeBytes : 1));
    }
}

static void leaf_stats(const vko_tree *t, int *maxErr, double *meanL1, double *meanL2)
{
    int64_t n = t->tempLen;
    int m = 0; double s1 = 0.0, s2 = 0.0;
    for (int64_t i = 0; i < n; i++) {
        double e = fabs((double)t->temp[i] - (double)t->recon[i]);
        if ((int)e > m) m = (int)e;
        s1 += e; s2 += e * e;
    }
    *maxErr = m; *meanL1 = s1 / (double)n; *meanL2 = s2 / (double)n;
}

/* Runs build() up to and including `stage` (1 pyramid, 2 compress, 3 prune, 4 convert).
 * R.cpp:17-140.  Stages exist so kernel-level GPU tests can diff intermediates. */
int vko_build_to_stage(vko_tree *t, int stage)
{
    if (t->stage != 0) return -1;
    const int maxAddLevels = 7;
    static const int addLevelDistance[7] = { 64, 32, 16, 8, 4, 2, 1 };
    int nx = (int)(log((double)t->X) / log(2.0)); /* R.cpp:26-28, float log on purpose */
    int ny = (int)(log((double)t->Y) / log(2.0));
    int nz = (int)(log((double)t->Z) / log(2.0));
    t->origTreeDepth = nx + ny + nz;
    t->maxTreeDepth = t->origTreeDepth + maxAddLevels;
    t->distanceMapLen = t->maxTreeDepth + 1;
    t->distanceMap = (byte *)calloc((size_t)t->distanceMapLen, 1);
    t->numOrigNodes = ((int64_t)1 << (t->origTreeDepth + 1)) - 1;
    t->numMaxNodes = t->numOrigNodes + ((int64_t)1 << t->origTreeDepth) * maxAddLevels;
    t->temp = (byte *)calloc((size_t)t->numOrigNodes, 1);
    t->tempLen = t->numOrigNodes;
    if (t->midrange) {
        t->tempRange = (byte *)calloc((size_t)t->numOrigNodes, 1);
        t->distanceMapRange = (byte *)calloc((size_t)t->distanceMapLen, 1);
    }
    if (!t->temp) return -2;

    byte mn, mx;
    build_recursive(t, 0, 0, t->rootMin, t->rootMax, &mn, &mx);
    t->data = NULL; /* data->clear() R.cpp:51-52: the input is consumed */
    t->stage = 1;
    if (stage <= 1) return 0;

    t->treeBytes = tb_bytes(t->numOrigNodes);
    t->tree = (byte *)calloc((size_t)t->treeBytes, 1);
    t->firstOrigLeaf = ((int64_t)1 << t->origTreeDepth) - 1;
    t->reconAll = (byte *)calloc((size_t)t->numOrigNodes, 1);
    t->recon = compress_gradient_descent(t, t->temp, t->tree, t->distanceMap, t->reconAll);
    t->reconLen = (int64_t)1 << t->origTreeDepth;
    if (t->midrange) {
        t->treeRangeBytes = t->treeBytes;
        t->treeRange = (byte *)calloc((size_t)t->treeRangeBytes, 1);
        t->reconRange = compress_gradient_descent(t, t->tempRange, t->treeRange, t->distanceMapRange, NULL);
    }
    /* temp.erase(begin, begin+firstOrigLeaf) R.cpp:64 */
    memmove(t->temp, t->temp + t->firstOrigLeaf, (size_t)t->reconLen);
    t->tempLen = t->reconLen;
    if (t->midrange) memmove(t->tempRange, t->tempRange + t->firstOrigLeaf, (size_t)t->reconLen);
    leaf_stats(t, &t->maxErrBefore, &t->meanL1Before, &t->meanL2Before);
    t->stage = 2;
    if (stage <= 2) return 0;

    prune_recursive(t, 0, 0);
    for (int d = t->origTreeDepth + 1, a = 0; d < t->maxTreeDepth + 1; d++, a++) {
        t->distanceMap[d] = (byte)addLevelDistance[a]; /* R.cpp:94-97 */
        if (t->midrange) t->distanceMapRange[d] = (byte)addLevelDistance[a];
    }
    t->stage = 3;
    if (stage <= 3) return 0;

    convert_to_preorder(t);
    leaf_stats(t, &t->maxErrAfter, &t->meanL1After, &t->meanL2After);
    t->stage = 4;
    return 0;
}

int vko_build(vko_tree *t) { return vko_build_to_stage(t, 4); }

typedef struct { int64_t idx; int depth; byte scalar; int64_t mn[3], mx[3]; } lc_item;

/* R.cpp:726-835.  `out` must hold X*Y*Z bytes; like vector::resize on a fresh
 * vector it should be zero-filled by the caller (cells the walk never writes --
 * non-power-of-two extents, C-10 -- keep their previous content).
 * Returns 0, or <0 if the stream is malformed (the reference would hit UB). */
int vko_level_cut(const vko_tree *t, int cutDepth, byte *out)
{
    if (t->numActiveNodes <= 0 || !t->tree) return -1;
    lc_item *stack = (lc_item *)malloc(sizeof(lc_item) * (size_t)(t->maxTreeDepth + 8));
    int sp = 0, rc = 0;
    lc_item root;
    root.idx = 0; root.depth = 0; root.scalar = t->distanceMap[0];
    memcpy(root.mn, t->rootMin, sizeof root.mn);
    memcpy(root.mx, t->rootMax, sizeof root.mx);
    stack[sp++] = root;
    while (sp > 0) {
        lc_item cur = stack[sp - 1];
        if (cur.idx >= t->numActiveNodes) { rc = -2; break; }
        int code = tb_get(t->tree, cur.idx);
        if (code == 3 || cur.depth == cutDepth) {
            for (int64_t x = cur.mn[0]; x < cur.mx[0]; x++)
                for (int64_t y = cur.mn[1]; y < cur.mx[1]; y++)
                    for (int64_t z = cur.mn[2]; z < cur.mx[2]; z++)
                        out[get_cell(t, x, y, z)] = cur.scalar;
            sp--;
            int64_t nextRight = cur.idx + 1;
            if (nextRight < t->numActiveNodes) {
                if (sp == 0) { rc = -3; break; }
                lc_item par = stack[--sp];
                int c = tb_get(t->tree, nextRight);
                byte scalar = par.scalar;
                if (c == 1) scalar = (byte)fmin(255.0, (double)scalar + (double)t->distanceMap[par.depth + 1]);
                else if (c == 2) scalar = (byte)fmax(0.0, (double)scalar - (double)t->distanceMap[par.depth + 1]);
                int64_t ext[3] = { par.mx[0] - par.mn[0], par.mx[1] - par.mn[1], par.mx[2] - par.mn[2] };
                if (ext[0] * ext[1] * ext[2] > 1) {
                    int sd = par.depth % VKO_MAX_DIM, i = 0;
                    while (ext[sd] == 1) sd = (par.depth + ++i) % VKO_MAX_DIM;
                    par.mn[sd] = (par.mn[sd] + par.mx[sd]) / 2;
                }
                par.idx = nextRight; par.depth += 1; par.scalar = scalar;
                if (par.depth > t->maxTreeDepth) { rc = -4; break; }
                stack[sp++] = par;
            }
        } else {
            if (cur.depth >= t->origTreeDepth) sp--;
            int64_t nextLeft = cur.idx + 1;
            if (nextLeft >= t->numActiveNodes) { rc = -5; break; }
            int c = tb_get(t->tree, nextLeft);
            byte scalar = cur.scalar;
            if (c == 1) scalar = (byte)fmin(255.0, (double)scalar + (double)t->distanceMap[cur.depth + 1]);
            else if (c == 2) scalar = (byte)fmax(0.0, (double)scalar - (double)t->distanceMap[cur.depth + 1]);
            int64_t ext[3] = { cur.mx[0] - cur.mn[0], cur.mx[1] - cur.mn[1], cur.mx[2] - cur.mn[2] };
            if (ext[0] * ext[1] * ext[2] > 1) {
                int sd = cur.depth % VKO_MAX_DIM, i = 0;
                while (ext[sd] == 1) sd = (cur.depth + ++i) % VKO_MAX_DIM;
                cur.mx[sd] = (cur.mn[sd] + cur.mx[sd]) / 2;
            }
            cur.idx = nextLeft; cur.depth += 1; cur.scalar = scalar;
            if (cur.depth > t->maxTreeDepth) { rc = -4; break; }
            stack[sp++] = cur;
        }
    }
    free(stack);
    return rc;
}

/* NOT IN THE REFERENCE (its levelCut de-synchronises for cutDepth < maxTreeDepth, defect C-4):
 * a well-defined progressive cut, used as the oracle of vr_brickset_decode(cut_depth < max).
 * The stream is parsed completely (so subtrees below the cut are skipped correctly) but scalar
 * updates stop below the cut: every voxel gets the decoded scalar of its ancestor at depth
 * min(cutDepth, depth of its terminal node).  cutDepth == maxTreeDepth equals vko_level_cut. */
int vko_level_cut_progressive(const vko_tree *t, int cutDepth, byte *out)
{
    if (t->numActiveNodes <= 0 || !t->tree) return -1;
    lc_item *stack = (lc_item *)malloc(sizeof(lc_item) * (size_t)(t->maxTreeDepth + 8));
    int sp = 0, rc = 0;
    lc_item root;
    root.idx = 0; root.depth = 0; root.scalar = t->distanceMap[0];
    memcpy(root.mn, t->rootMin, sizeof root.mn);
    memcpy(root.mx, t->rootMax, sizeof root.mx);
    stack[sp++] = root;
    while (sp > 0) {
        lc_item cur = stack[sp - 1];
        if (cur.idx >= t->numActiveNodes) { rc = -2; break; }
        int code = tb_get(t->tree, cur.idx);
        if (code == 3 || cur.depth == t->maxTreeDepth) {
            for (int64_t x = cur.mn[0]; x < cur.mx[0]; x++)
                for (int64_t y = cur.mn[1]; y < cur.mx[1]; y++)
                    for (int64_t z = cur.mn[2]; z < cur.mx[2]; z++)
                        out[get_cell(t, x, y, z)] = cur.scalar;
            sp--;
            int64_t nextRight = cur.idx + 1;
            if (nextRight < t->numActiveNodes) {
                if (sp == 0) { rc = -3; break; }
                lc_item par = stack[--sp];
                int c = tb_get(t->tree, nextRight);
                byte scalar = par.scalar;
                if (par.depth + 1 <= cutDepth) {
                    if (c == 1) scalar = (byte)fmin(255.0, (double)scalar + (double)t->distanceMap[par.depth + 1]);
                    else if (c == 2) scalar = (byte)fmax(0.0, (double)scalar - (double)t->distanceMap[par.depth + 1]);
                }
                int64_t ext[3] = { par.mx[0] - par.mn[0], par.mx[1] - par.mn[1], par.mx[2] - par.mn[2] };
                if (ext[0] * ext[1] * ext[2] > 1) {
                    int sd = par.depth % VKO_MAX_DIM, i = 0;
                    while (ext[sd] == 1) sd = (par.depth + ++i) % VKO_MAX_DIM;
                    par.mn[sd] = (par.mn[sd] + par.mx[sd]) / 2;
                }
                par.idx = nextRight; par.depth += 1; par.scalar = scalar;
                stack[sp++] = par;
            }
        } else {
            if (cur.depth >= t->origTreeDepth) sp--;
            int64_t nextLeft = cur.idx + 1;
            if (nextLeft >= t->numActiveNodes) { rc = -5; break; }
            int c = tb_get(t->tree, nextLeft);
            byte scalar = cur.scalar;
            if (cur.depth + 1 <= cutDepth) {
                if (c == 1) scalar = (byte)fmin(255.0, (double)scalar + (double)t->distanceMap[cur.depth + 1]);
                else if (c == 2) scalar = (byte)fmax(0.0, (double)scalar - (double)t->distanceMap[cur.depth + 1]);
            }
            int64_t ext[3] = { cur.mx[0] - cur.mn[0], cur.mx[1] - cur.mn[1], cur.mx[2] - cur.mn[2] };
            if (ext[0] * ext[1] * ext[2] > 1) {
                int sd = cur.depth % VKO_MAX_DIM, i = 0;
                while (ext[sd] == 1) sd = (cur.depth + ++i) % VKO_MAX_DIM;
                cur.mx[sd] = (cur.mn[sd] + cur.mx[sd]) / 2;
            }
            cur.idx = nextLeft; cur.depth += 1; cur.scalar = scalar;
            if (cur.depth > t->maxTreeDepth) { rc = -4; break; }
            stack[sp++] = cur;
        }
    }
    free(stack);
    return rc;
}

/* NOT IN THE REFERENCE (MidRangeTree builds tree_range, M.cpp:399-544,871-982, but its levelCut,
 * M.cpp:984-1093, never reads it): the progressive cut above applied to the range stream.  Both streams
 * are emitted in lock step (same structure, same numActiveNodes), so this is the same walk over
 * treeRange / distanceMapRange.  Oracle of vr_brickset_decode_range. */
int vko_level_cut_range(const vko_tree *t, int cutDepth, byte *out)
{
    if (!t->midrange || !t->treeRange || !t->distanceMapRange) return -10;
    vko_tree r = *t;
    r.tree = t->treeRange;
    r.distanceMap = t->distanceMapRange;
    return vko_level_cut_progressive(&r, cutDepth, out);
}

/* R.cpp:386-411: error helpers.  The reference dereferences the (cleared)
 * input, C-7; here the original volume is passed explicitly. */
int vko_measure_max_error(const byte *decoded, const byte *original, int64_t n)
{
    int m = 0;
    for (int64_t i = 0; i < n; i++) { int e = abs((int)decoded[i] - (int)original[i]); if (e > m) m = e; }
    return m;
}
double vko_measure_mean_error(const byte *decoded, const byte *original, int64_t n)
{
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) s += fabs((double)decoded[i] - (double)original[i]);
    return s / (double)n;
}
void vko_query_error(const byte *decoded, const byte *original, int64_t n, byte *out)
{
    for (int64_t i = 0; i < n; i++) out[i] = (byte)abs((int)decoded[i] - (int)original[i]);
}

/* R.cpp:521-552: 88-byte header | distanceMap | tree bytes, little endian. */
int vko_save(const vko_tree *t, const char *filename)
{
    if (t->treeBytes == 0 || !t->tree) return -1; /* "ERROR! No tree to save." */
    FILE *f = fopen(filename, "wb");
    if (!f) return -2;
    int32_t mtd = t->maxTreeDepth, otd = t->origTreeDepth;
    fwrite(t->rootMin, 8, 3, f);
    fwrite(t->rootMax, 8, 3, f);
    fwrite(&mtd, 4, 1, f);
    fwrite(&otd, 4, 1, f);
    fwrite(&t->X, 8, 1, f); fwrite(&t->Y, 8, 1, f); fwrite(&t->Z, 8, 1, f);
    fwrite(&t->numActiveNodes, 8, 1, f);
    fwrite(t->distanceMap, 1, (size_t)(t->maxTreeDepth + 1), f);
    if (t->midrange) {   /* MidRangeTree::save, M.cpp:753-785: both maps, then both streams */
        fwrite(t->distanceMapRange, 1, (size_t)(t->maxTreeDepth + 1), f);
        fwrite(t->tree, 1, (size_t)t->treeBytes, f);
        fwrite(t->treeRange, 1, (size_t)t->treeBytes, f);
        fclose(f);
        return 0;
    }
    fwrite(t->tree, 1, (size_t)t->treeBytes, f);
    fclose(f);
    return 0;
}

/* MidRangeTree::open, M.cpp:787-833, restated literally.  The size computation subtracts three of
 * the four int64 fields (:815, like R.cpp:581) and halves: treeSize comes out 4 bytes too large, so
 * `tree` swallows the first 4 bytes of the range stream and `tree_range` starts 4 bytes late and comes up
 * 8 bytes short (zero here).  The mid stream still decodes (numActiveNodes bounds the walk); the range
 * stream read back this way is NOT what was saved.  Test infrastructure documents this; the product's
 * reader (vr_brickset_open_variant) reads what save() wrote. */
vko_tree *vko_open_midrange(const char *filename)
{
    FILE *f = fopen(filename, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    int64_t fileSize = ftell(f);
    fseek(f, 0, SEEK_SET);
    vko_tree *t = (vko_tree *)calloc(1, sizeof(vko_tree));
    t->tolerance = 6; t->maxEpochs = 5; t->midrange = 1;
    int32_t mtd = 0, otd = 0;
    size_t ok = 0;
    ok += fread(t->rootMin, 8, 3, f);
    ok += fread(t->rootMax, 8, 3, f);
    ok += fread(&mtd, 4, 1, f);
    ok += fread(&otd, 4, 1, f);
    ok += fread(&t->X, 8, 1, f); ok += fread(&t->Y, 8, 1, f); ok += fread(&t->Z, 8, 1, f);
    ok += fread(&t->numActiveNodes, 8, 1, f);
    if (ok != 12 || mtd < 0 || mtd > 255) { fclose(f); free(t); return NULL; }
    t->maxTreeDepth = mtd; t->origTreeDepth = otd;
    int64_t treeSize = fileSize - (2 * 24 + 2 * 4 + 2 * (mtd + 1) + 3 * 8); /* sic */
    treeSize /= 2;
    if (treeSize < 0) { fclose(f); free(t); return NULL; }
    t->distanceMapLen = mtd + 1;
    t->distanceMap = (byte *)calloc((size_t)t->distanceMapLen, 1);
    t->distanceMapRange = (byte *)calloc((size_t)t->distanceMapLen, 1);
    t->tree = (byte *)calloc((size_t)treeSize + 1, 1);
    t->treeRange = (byte *)calloc((size_t)treeSize + 1, 1);
    t->treeBytes = treeSize; t->treeRangeBytes = treeSize;
    ok = fread(t->distanceMap, 1, (size_t)(mtd + 1), f);
    ok = fread(t->distanceMapRange, 1, (size_t)(mtd + 1), f);
    ok = fread(t->tree, 1, (size_t)treeSize, f);
    ok = fread(t->treeRange, 1, (size_t)treeSize, f);   /* short read */
    (void)ok;
    fclose(f);
    t->stage = 4;
    return t;
}

/* R.cpp:554-594 incl. the 8-byte over-allocation of tree.bits (C-6). */
vko_tree *vko_open(const char *filename)
{
    FILE *f = fopen(filename, "rb");
    if (!f) return NULL; /* reference: prints, waits for Enter, exit(-1) */
    fseek(f, 0, SEEK_END);
    int64_t fileSize = ftell(f);
    fseek(f, 0, SEEK_SET);
    vko_tree *t = (vko_tree *)calloc(1, sizeof(vko_tree));
    t->tolerance = 6; t->maxEpochs = 5;
    int32_t mtd = 0, otd = 0;
    size_t ok = 0;
    ok += fread(t->rootMin, 8, 3, f);
    ok += fread(t->rootMax, 8, 3, f);
    ok += fread(&mtd, 4, 1, f);
    ok += fread(&otd, 4, 1, f);
    ok += fread(&t->X, 8, 1, f); ok += fread(&t->Y, 8, 1, f); ok += fread(&t->Z, 8, 1, f);
    ok += fread(&t->numActiveNodes, 8, 1, f);
    if (ok != 12 || mtd < 0 || mtd > 255) { fclose(f); free(t); return NULL; }
    t->maxTreeDepth = mtd; t->origTreeDepth = otd;
    int64_t treeSize = fileSize - (2 * 24 + 2 * 4 + mtd + 1 + 3 * 8); /* sic: 3, not 4 */
    if (treeSize < 0) { fclose(f); free(t); return NULL; }
    t->distanceMapLen = mtd + 1;
    t->distanceMap = (byte *)calloc((size_t)t->distanceMapLen, 1);
    t->tree = (byte *)calloc((size_t)treeSize + 1, 1);
    t->treeBytes = treeSize;
    ok = fread(t->distanceMap, 1, (size_t)(mtd + 1), f);
    ok = fread(t->tree, 1, (size_t)treeSize, f); /* comes up 8 bytes short; rest stays 0 */
    (void)ok;
    fclose(f);
    t->stage = 4;
    return t;
}

/* M.cpp:1095-1128 convertToByteArray: two nodes per byte,
 * byte = mid_i<<6 | rng_i<<4 | mid_{i+1}<<2 | rng_{i+1}; length = next power of
 * two >= ceil(n/2) (the 32-bit bit-smear, :1098-1106), zero padded.
 * Returns the length; writes at most cap bytes. */
int64_t vko_mid_convert_to_byte_array(const vko_tree *t, byte *out, int64_t cap)
{
    if (!t->midrange || !t->treeRange) return -1;
    int64_t v = (int64_t)ceil((double)t->numActiveNodes / 2.0);
    v--;
    v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    v++;
    if (!out) return v;
    if (cap < v) return -2;
    memset(out, 0, (size_t)v);
    int64_t idx = 0, outIdx = 0;
    while (idx < t->numActiveNodes) {
        byte first = (byte)tb_get(t->tree, idx), second = (byte)tb_get(t->treeRange, idx), third = 0, fourth = 0;
        if (idx + 1 < t->numActiveNodes) { third = (byte)tb_get(t->tree, idx + 1); fourth = (byte)tb_get(t->treeRange, idx + 1); }
        out[outIdx++] = (byte)((first << 6) | (second << 4) | (third << 2) | fourth);
        idx += 2;
    }
    return v;
}

/* ---- accessors for the ctypes wrapper ---- */
int32_t vko_orig_tree_depth(const vko_tree *t) { return t->origTreeDepth; }
int32_t vko_max_tree_depth(const vko_tree *t) { return t->maxTreeDepth; }
int64_t vko_num_active_nodes(const vko_tree *t) { return t->numActiveNodes; }
int64_t vko_num_orig_nodes(const vko_tree *t) { return t->numOrigNodes; }
int64_t vko_first_orig_leaf(const vko_tree *t) { return t->firstOrigLeaf; }
int64_t vko_tree_bytes(const vko_tree *t) { return t->treeBytes; }
const byte *vko_tree_ptr(const vko_tree *t) { return t->tree; }
const byte *vko_distance_map_ptr(const vko_tree *t) { return t->distanceMap; }
int64_t vko_temp_len(const vko_tree *t) { return t->tempLen; }
const byte *vko_temp_ptr(const vko_tree *t) { return t->temp; }
int64_t vko_recon_len(const vko_tree *t) { return t->reconLen; }
const byte *vko_recon_ptr(const vko_tree *t) { return t->recon; }
const byte *vko_recon_all_ptr(const vko_tree *t) { return t->reconAll; }
const byte *vko_tree_range_ptr(const vko_tree *t) { return t->treeRange; }
int64_t vko_tree_range_bytes(const vko_tree *t) { return t->treeRangeBytes; }
const byte *vko_distance_map_range_ptr(const vko_tree *t) { return t->distanceMapRange; }
const byte *vko_temp_range_ptr(const vko_tree *t) { return t->tempRange; }
const byte *vko_recon_range_ptr(const vko_tree *t) { return t->reconRange; }
void vko_dims(const vko_tree *t, int64_t *d) { d[0] = t->X; d[1] = t->Y; d[2] = t->Z; }
int32_t vko_num_reverts(const vko_tree *t) { return t->numReverts; }
int32_t vko_zero_run_rewrites(const vko_tree *t) { return t->zeroRunRewrites; }
int32_t vko_trace_len(const vko_tree *t) { return t->traceLen; }
const vko_trace_rec *vko_trace_ptr(const vko_tree *t) { return t->trace; }
void vko_leaf_stats(const vko_tree *t, int32_t *maxBefore, int32_t *maxAfter, double *l1Before, double *l1After)
{
    *maxBefore = t->maxErrBefore; *maxAfter = t->maxErrAfter;
    *l1Before = t->meanL1Before; *l1After = t->meanL1After;
}

/* FNV-1a-64 over raw bytes (SURVEY.md Appendix B hash convention). */
uint64_t vko_fnv1a64(const byte *p, int64_t n)
{
    uint64_t h = 1469598103934665603ULL;
    for (int64_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ULL; }
    return h;
}

/* Survey's sphere generator: 32-bit LCG seed 12345,
 * advanced once per voxel before use, z outer / x inner.  noiseMask 7 =
 * sphere_n3, 0 = sphere_n0, >=256 = random bytes. */
void vko_gen_sphere(byte *vol, int n, int noiseMask, uint32_t seed)
{
    uint32_t s = seed;
    for (int z = 0; z < n; z++)
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++) {
                s = s * 1664525u + 1013904223u;
                double r = sqrt((x - n / 2.0) * (x - n / 2.0) + (y - n / 2.0) * (y - n / 2.0) +
                                (z - n / 2.0) * (z - n / 2.0)) / (n / 2.0);
                int v = noiseMask >= 256 ? (int)(s >> 24)
                                         : (int)(255.0 * fmax(0.0, 1.0 - r)) + (int)((s >> 24) & (uint32_t)noiseMask);
                vol[x + (size_t)n * y + (size_t)n * n * z] = (byte)(v < 255 ? v : 255);
            }
}
