// HashedKdtree.hpp -- the reference's third tree class (volume_renderer/HashedKdtree.h:26-227), kept as an API.
//
// PARITY UNPINNED.  The reference's HashedKdtree cannot be run: build() writes past its hash table on the first
// level of a 16^3 volume (HashedKdtree.cpp:138) and seeds its collision handling from std::random_device
// (HashedKdtree.h:83-85, HashedKdtree.cpp:473), so it has no defined output to compare with, and none of the
// reference's fixtures hold one.  What a caller of that class gets here is the same public surface -- ctor,
// build(), levelCut(), the three error helpers, save()/open(), and the data members main.cpp-style code reads --
// over the VolumeKdtree path of this library (same progressive 2-bit kd-tree, tolerance 4 as HashedKdtree.h:79):
// levelCut() is therefore VolumeKdtree's decode, bit-exact against the oracle like every VolumeKdtree result,
// and the hash-table members exist but stay empty (numCollisions = 0).
#pragma once
#include "VolumeKdtree.hpp"

class HashedKdtree {
public:
    // public data members (HashedKdtree.h:32-63)
    int64_t rootMin[3] = {0, 0, 0}, rootMax[3] = {0, 0, 0};
    TwoBitArrayView treeData;                 // here: the preorder 2-bit stream of the VolumeKdtree path
    TwoBitArrayView treeStructure;            // hash-table layout of the reference: not produced
    TwoBitArrayView treeDataCollisions, treeStructureCollisions;
    std::vector<byte> distanceMap;
    int treeDepth = 0;
    int64_t numCollisions = 0;
    int64_t hashMask = 0;
    int64_t X = 0, Y = 0, Z = 0;
    std::vector<byte> *output = nullptr, *output2 = nullptr;
    int queryDepth = 0;
    int tolerance = 4;                        // HashedKdtree.h:79

    HashedKdtree() {}
    HashedKdtree(std::vector<byte> &inData, int64_t x, int64_t y, int64_t z) : X(x), Y(y), Z(z), t(inData, x, y, z)
    {
        rootMax[0] = x; rootMax[1] = y; rootMax[2] = z;
    }

    void build()                              // HashedKdtree.h:105
    {
        t.setErrorTolerance(tolerance);
        t.build(true);
        sync();
    }
    void levelCut(int cutDepth, std::vector<byte> &outData)   // HashedKdtree.h:113
    {
        queryDepth = cutDepth;
        output = &outData;
        t.levelCut(cutDepth, outData);
    }
    int measureMaxError() { return t.measureMaxError(); }          // HashedKdtree.h:120
    double measureMeanError() { return t.measureMeanError(); }     // HashedKdtree.h:127
    void queryError(std::vector<byte> &outData) { t.queryError(outData); }   // HashedKdtree.h:129
    void save(std::string filename) { t.save(filename); }          // HashedKdtree.h:136 (VolumeKdtree's file layout)
    void open(std::string filename) { t.open(filename); sync(); }  // HashedKdtree.h:143

private:
    VolumeKdtree t;
    void sync()
    {
        X = t.X; Y = t.Y; Z = t.Z;
        rootMax[0] = X; rootMax[1] = Y; rootMax[2] = Z;
        treeDepth = t.maxTreeDepth;
        distanceMap = t.distanceMap;
        treeData.bits = t.tree.bits;
    }
};
