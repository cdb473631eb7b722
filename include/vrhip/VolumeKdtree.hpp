// VolumeKdtree.hpp -- header-only C++ facade with the reference's class name, method names
// and argument meaning (volume_renderer/VolumeKdtree_recover.h:51-175), over the C ABI of
// include/vrhip.h.  Host code that used the reference class compiles against this instead:
//
//     VolumeKdtree *myTree = new VolumeKdtree(volume.data, dims[0], dims[1], dims[2]);   // main.cpp:251
//     myTree->setMaxEpochs(2); myTree->setErrorTolerance(1);                              // :253-254
//     myTree->build(true);                                                               // :257
//     myTree->save("tree.bin");                                                           // :267
//     std::vector<unsigned char> treeData; myTree->levelCut(myTree->maxTreeDepth, treeData); // :280-281
//
// Differences, all forced by reference defects: build() does not
// empty the caller's vector (C-7: the error helpers need it); levelCut() at cutDepth ==
// maxTreeDepth is the reference's result, shallower cuts are a well-defined progressive decode
// instead of the reference's de-synchronising walk (C-4); open() of a missing file throws
// instead of exit(-1).
#pragma once
#include "../vrhip.h"
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

typedef unsigned char byte;

namespace vrhip_detail {
inline void check(vr_status s, const char *what)
{
    if (s != VR_OK) throw std::runtime_error(std::string(what) + ": " + vr_status_string(s));
}
struct DeviceBuffer {
    void *p = nullptr;
    int64_t bytes = 0;
    void ensure(int64_t n) { if (n > bytes) { vr_free(p); p = nullptr; check(vr_malloc(&p, n), "vr_malloc"); bytes = n; } }
    ~DeviceBuffer() { vr_free(p); }
};
} // namespace vrhip_detail

// Mirrors TwoBitArray (TwoBitArray.h:11-88) as far as callers use it: .bits and .bytes().
struct TwoBitArrayView {
    std::vector<byte> bits;
    int64_t bytes() const { return (int64_t)bits.size(); }
    int operator[](int64_t i) const { return (bits[(size_t)(i >> 2)] >> ((i & 3) * 2)) & 3; }
};

class VolumeKdtree {
public:
    // public data members of the reference class (VolumeKdtree_recover.h:57-84)
    int64_t rootMin[3] = {0, 0, 0}, rootMax[3] = {0, 0, 0};
    TwoBitArrayView tree;
    std::vector<byte> distanceMap;
    int maxTreeDepth = 0, origTreeDepth = 0;
    int64_t numActiveNodes = 0;
    int64_t X = 0, Y = 0, Z = 0;
    int tolerance = 6, maxEpochs = 5;          // VolumeKdtree_recover.h:92-93
    std::vector<byte> *output = nullptr;

    VolumeKdtree() {}
    VolumeKdtree(std::vector<byte> &inData, int64_t x, int64_t y, int64_t z) : X(x), Y(y), Z(z), data(&inData)
    {
        rootMax[0] = x; rootMax[1] = y; rootMax[2] = z;
    }
    virtual ~VolumeKdtree() { vr_brickset_destroy(bs); }
    VolumeKdtree(const VolumeKdtree &) = delete;
    VolumeKdtree &operator=(const VolumeKdtree &) = delete;

    void setErrorTolerance(int errorTolerance) { tolerance = errorTolerance; }   // R.cpp:9-11
    void setMaxEpochs(int epochs) { maxEpochs = epochs; }                        // R.cpp:13-15

    virtual int variant() const { return VR_VARIANT_RECOVER; }

    // R.cpp:17-140.  useThreads is meaningless here: the whole build is batched GPU kernels.
    void build(bool useThreads = true)
    {
        (void)useThreads;
        using namespace vrhip_detail;
        if (!data || (int64_t)data->size() != X * Y * Z) throw std::runtime_error("build(): data does not match X*Y*Z");
        vr_brickset_destroy(bs);
        bs = nullptr;
        const int64_t dims[3] = {X, Y, Z};
        check(vr_brickset_create(&bs, 1, dims, tolerance, maxEpochs, variant()), "vr_brickset_create");
        vox.ensure(X * Y * Z);
        check(vr_upload(vox.p, data->data(), X * Y * Z, nullptr), "vr_upload");
        check(vr_brickset_build(bs, (const uint8_t *)vox.p, nullptr), "vr_brickset_build");
        refresh();
    }

    // R.cpp:726-835.  outData is resized to X*Y*Z like the reference does (R.cpp:733).
    void levelCut(int cutDepth, std::vector<byte> &outData)
    {
        using namespace vrhip_detail;
        if (!bs) throw std::runtime_error("levelCut(): no tree");
        output = &outData;
        outData.resize((size_t)(X * Y * Z));
        dec.ensure(X * Y * Z);
        check(vr_brickset_decode(bs, cutDepth, (uint8_t *)dec.p, nullptr), "vr_brickset_decode");
        check(vr_download(outData.data(), dec.p, X * Y * Z, nullptr), "vr_download");
    }

    // R.cpp:386-411 -- need the original data and a previous levelCut()
    int measureMaxError() { int m = 0; double mean = 0; errors(&m, &mean); return m; }
    double measureMeanError() { int m = 0; double mean = 0; errors(&m, &mean); return mean; }
    void queryError(std::vector<byte> &outData)
    {
        using namespace vrhip_detail;
        need_both();
        DeviceBuffer e;
        e.ensure(X * Y * Z);
        check(vr_query_error((const uint8_t *)dec.p, (const uint8_t *)vox.p, X * Y * Z, (uint8_t *)e.p, nullptr), "vr_query_error");
        outData.resize((size_t)(X * Y * Z));
        check(vr_download(outData.data(), e.p, X * Y * Z, nullptr), "vr_download");
    }

    void save(std::string filename)            // R.cpp:521-552, byte-identical file
    {
        if (!bs) { return; }                   // "ERROR! No tree to save." (R.cpp:526-530)
        vrhip_detail::check(vr_brickset_save(bs, 0, filename.c_str()), "vr_brickset_save");
    }
    void open(std::string filename)            // R.cpp:554-594
    {
        vr_brickset_destroy(bs);
        bs = nullptr;
        vrhip_detail::check(vr_brickset_open(&bs, filename.c_str()), "vr_brickset_open");
        refresh();
    }

    vr_brickset *handle() { return bs; }

protected:
    vr_brickset *bs = nullptr;
    std::vector<byte> *data = nullptr;
    vrhip_detail::DeviceBuffer vox, dec;

    void refresh()
    {
        using namespace vrhip_detail;
        vr_tree_info ti;
        check(vr_brickset_info(bs, 0, &ti), "vr_brickset_info");
        X = ti.X; Y = ti.Y; Z = ti.Z;
        rootMax[0] = X; rootMax[1] = Y; rootMax[2] = Z;
        origTreeDepth = ti.orig_tree_depth;
        maxTreeDepth = ti.max_tree_depth;
        numActiveNodes = ti.num_active_nodes;
        tree.bits.resize((size_t)ti.tree_bytes);
        check(vr_brickset_get_tree(bs, 0, tree.bits.data(), ti.tree_bytes), "vr_brickset_get_tree");
        distanceMap.resize((size_t)maxTreeDepth + 1);
        check(vr_brickset_get_distance_map(bs, 0, distanceMap.data(), maxTreeDepth + 1), "vr_brickset_get_distance_map");
    }
    void need_both()
    {
        if (!bs || !output || dec.p == nullptr || vox.p == nullptr)
            throw std::runtime_error("error helpers need build() data and a previous levelCut()");
    }
    void errors(int *m, double *mean)
    {
        need_both();
        int32_t mm = 0;
        vrhip_detail::check(vr_measure_error((const uint8_t *)dec.p, (const uint8_t *)vox.p, X * Y * Z, &mm, mean, nullptr),
                            "vr_measure_error");
        *m = mm;
    }
};

// volume_renderer/MidRangeTree.h:51-271: second (half-range) stream + 4-bit packing
class MidRangeTree : public VolumeKdtree {
public:
    using VolumeKdtree::VolumeKdtree;
    TwoBitArrayView tree_range;
    std::vector<byte> distanceMap_range;
    int variant() const override { return VR_VARIANT_MIDRANGE; }
    void build(bool useThreads = true)
    {
        VolumeKdtree::build(useThreads);
        using namespace vrhip_detail;
        tree_range.bits.resize(tree.bits.size());
        check(vr_brickset_get_tree_range(bs, 0, tree_range.bits.data(), (int64_t)tree_range.bits.size()), "get_tree_range");
        distanceMap_range.resize((size_t)maxTreeDepth + 1);
        check(vr_brickset_get_distance_map_range(bs, 0, distanceMap_range.data(), maxTreeDepth + 1), "get_distance_map_range");
    }
    void open(std::string filename)            // M.cpp:787-833 (reads back exactly what save() wrote, see vrhip.h)
    {
        using namespace vrhip_detail;
        vr_brickset_destroy(bs);
        bs = nullptr;
        check(vr_brickset_open_variant(&bs, filename.c_str(), VR_VARIANT_MIDRANGE), "vr_brickset_open_variant");
        refresh();
        tree_range.bits.resize(tree.bits.size());
        check(vr_brickset_get_tree_range(bs, 0, tree_range.bits.data(), (int64_t)tree_range.bits.size()), "get_tree_range");
        distanceMap_range.resize((size_t)maxTreeDepth + 1);
        check(vr_brickset_get_distance_map_range(bs, 0, distanceMap_range.data(), maxTreeDepth + 1), "get_distance_map_range");
    }
    // New: the reference builds tree_range but its levelCut (M.cpp:984-1093) never reads it.  The half
    // range per voxel at a cut depth; with levelCut() this bounds every voxel by [mid - range, mid + range].
    void levelCutRange(int cutDepth, std::vector<byte> &outData)
    {
        using namespace vrhip_detail;
        if (!bs) throw std::runtime_error("levelCutRange(): no tree");
        outData.resize((size_t)(X * Y * Z));
        DeviceBuffer r;
        r.ensure(X * Y * Z);
        check(vr_brickset_decode_range(bs, cutDepth, (uint8_t *)r.p, nullptr), "vr_brickset_decode_range");
        check(vr_download(outData.data(), r.p, X * Y * Z, nullptr), "vr_download");
    }
    // save(): VolumeKdtree::save on a VR_VARIANT_MIDRANGE set writes MidRangeTree's layout (M.cpp:753-785)
    void convertToByteArray(std::vector<byte> &byteArray)      // M.cpp:1095-1128
    {
        using namespace vrhip_detail;
        int64_t n = 0;
        check(vr_brickset_get_packed4(bs, 0, nullptr, 0, &n), "get_packed4");
        byteArray.resize((size_t)n);
        check(vr_brickset_get_packed4(bs, 0, byteArray.data(), n, &n), "get_packed4");
    }
};
