// main_pipeline.cpp -- the start-up pipeline of the reference's main() (main.cpp:141-290) and one
// frame of its render loop (:373-411), written against the facade headers: same class names,
// same calls.  Plain C++ (g++), no HIP headers: everything GPU goes through the C ABI.
//
//   g++ -std=c++14 -O2 -Iinclude examples/main_pipeline.cpp -Lvolumerenderer_amd -lvrhip
//       -Wl,-rpath,$PWD/volumerenderer_amd -o /tmp/main_pipeline ; /tmp/main_pipeline /tmp/bricks
#include "vrhip/VolumeReader.hpp"
#include "vrhip/HashedKdtree.hpp"
#include <cmath>
#include <cstdio>
#include <sstream>

static int64_t BRICK_DIM[3] = {64, 64, 32};
static int64_t VOLUME_GRID[3] = {2, 2, 2};
static std::map<int, dim3D> volumeBrickMap;
static std::string dir;

static std::string findBrickBinaryFile(int brick, int timestep)     // main.cpp:581-597
{
    std::ostringstream o;
    o << dir << "/d_" << timestep << "_" << brick;
    return o.str();
}
static void fillVolumeBrickMap()                                      // main.cpp:599-619
{
    for (int b = 0; b < 8; ++b) { volumeBrickMap[b][0] = b % 2; volumeBrickMap[b][1] = (b / 2) % 2; volumeBrickMap[b][2] = b / 4; }
}

int main(int argc, char **argv)
{
    dir = argc > 1 ? argv[1] : "/tmp";
    // synthetic brick files in place of the Richtmyer-Meshkov data set
    for (int b = 0; b < 8; ++b) {
        std::vector<unsigned char> v((size_t)(BRICK_DIM[0] * BRICK_DIM[1] * BRICK_DIM[2]));
        for (size_t i = 0; i < v.size(); ++i) v[i] = (unsigned char)(128 + 100 * std::sin(0.05 * (double)(i % 64) + b) + (i * 2654435761u >> 30));
        std::ofstream(findBrickBinaryFile(b, 273), std::ios::binary).write((const char *)v.data(), (std::streamsize)v.size());
    }
    fillVolumeBrickMap();
    VolumeReader<unsigned char> volume(BRICK_DIM, VOLUME_GRID, findBrickBinaryFile, &volumeBrickMap);
    bool ok = volume.LoadBricksToTexture(8, 2, 2, 2, 273, false);                                   // main.cpp:242
    if (!ok) return 1;
    VolumeKdtree *myTree = new VolumeKdtree(volume.data, volume.dataDims[0], volume.dataDims[1], volume.dataDims[2]);
    myTree->setMaxEpochs(2);                                                                          // main.cpp:253
    myTree->setErrorTolerance(1);                                                                     // :254
    myTree->build(true);                                                                              // :257
    myTree->save(dir + "/tree_1tolerance.bin");                                                       // :267
    std::vector<unsigned char> treeData;
    myTree->levelCut(myTree->maxTreeDepth, treeData);                                                 // :280-281
    std::printf("origTreeDepth %d maxTreeDepth %d numActiveNodes %lld tree bytes %lld\n", myTree->origTreeDepth,
                myTree->maxTreeDepth, (long long)myTree->numActiveNodes, (long long)myTree->tree.bytes());
    std::printf("MAX ERROR: %d  MEAN ERROR: %.4f\n", myTree->measureMaxError(), myTree->measureMeanError());   // :283-284
    // re-upload the decoded volume as the texture (main.cpp:290) and draw one frame (:396-404)
    void *tex = nullptr;
    vr_malloc(&tex, (int64_t)treeData.size());
    vr_upload(tex, treeData.data(), (int64_t)treeData.size(), nullptr);
    UnitBrick brick;
    brick.Setup();
    brick.Bind(tex, volume.dataDims);
    vr_camera cam = UnitBrick::defaultCamera();
    vr_render_params P = UnitBrick::defaultParams(320, 240);
    void *img = nullptr;
    vr_malloc(&img, 320 * 240 * 16);
    brick.Draw(cam, P, (float *)img);
    std::vector<float> host(320 * 240 * 4);
    vr_download(host.data(), img, 320 * 240 * 16, nullptr);
    std::printf("centre pixel rgba %.3f %.3f %.3f %.3f\n", host[(120 * 320 + 160) * 4], host[(120 * 320 + 160) * 4 + 1],
                host[(120 * 320 + 160) * 4 + 2], host[(120 * 320 + 160) * 4 + 3]);
    // a tree file written by save() opens again and decodes to the same voxels
    VolumeKdtree again;
    again.open(dir + "/tree_1tolerance.bin");
    std::vector<unsigned char> t2;
    again.levelCut(again.maxTreeDepth, t2);
    std::printf("reopen: voxels equal %d\n", (int)(t2 == treeData));
    // the third tree class keeps its interface (HashedKdtree.h:26-143); results are the VolumeKdtree path's, tolerance 4
    HashedKdtree hashed(volume.data, volume.dataDims[0], volume.dataDims[1], volume.dataDims[2]);
    hashed.build();
    std::vector<unsigned char> t3;
    hashed.levelCut(hashed.treeDepth, t3);
    std::printf("hashed: max error %d mean %.4f\n", hashed.measureMaxError(), hashed.measureMeanError());
    brick.Unbind(); brick.Delete();
    vr_free(tex); vr_free(img);
    delete myTree;
    return t2 == treeData ? 0 : 2;
}
