// VolumeReader.hpp / UnitBrick -- header-only facade of the reference's ingest and proxy-geometry
// classes (volume_renderer/VolumeReader.h:28-290, volume_renderer/UnitBrick.h:17-119) over the
// C ABI.  The GL texture becomes a device buffer; UnitBrick::Draw() becomes the ray-march launch.
#pragma once
#include "../vrhip.h"
#include "VolumeKdtree.hpp"
#include <fstream>
#include <functional>
#include <iostream>
#include <map>

typedef int64_t dim3D[3];   // (X,Y,Z)  VolumeReader.h:19

template <typename T>
class VolumeReader {
public:
    dim3D brickDims = {0, 0, 0};
    std::function<std::string(int, int)> findSourceFile;
    void *textureId = nullptr;                 // device copy of `data` (the reference's GLuint texture)
    std::map<int, dim3D> *brickMap = nullptr;
    std::vector<T> data;
    dim3D dataDims = {0, 0, 0};

    VolumeReader()
    {
        findSourceFile = [](int, int) -> std::string { throw std::runtime_error("\n\nERROR! findSourceFile function not defined.\n"); };
    }
    VolumeReader(dim3D brick, dim3D /*grid: ignored by the reference too, VolumeReader.h:68-76*/,
                 std::function<std::string(int, int)> findFileFunct, std::map<int, dim3D> *bMap)
    {
        for (int k = 0; k < 3; ++k) brickDims[k] = brick[k];
        findSourceFile = findFileFunct;
        brickMap = bMap;
    }
    ~VolumeReader() { vr_free(textureId); }

    bool LoadBrickToTexture(int brick, int timestep, bool dealloc, bool toGPU = true)      // VolumeReader.h:91-107
    {
        bool ok = LoadVolumeFromBinaryFile(findSourceFile(brick, timestep));
        tempBrick.swap(data);
        for (int k = 0; k < 3; ++k) dataDims[k] = brickDims[k];
        if (ok) { if (toGPU) transferToGPU(dealloc); }
        else std::cout << "ERROR! Texture load failure!" << std::endl;
        return ok;
    }

    void transferToGPU(bool dealloc = true)                                                 // VolumeReader.h:114-138
    {
        const int64_t n = (int64_t)(dataDims[0] * dataDims[1] * dataDims[2] * (int64_t)sizeof(T));
        vr_free(textureId);
        textureId = nullptr;
        vrhip_detail::check(vr_malloc(&textureId, n), "vr_malloc");
        vrhip_detail::check(vr_upload(textureId, data.data(), n, nullptr), "vr_upload");
        if (dealloc) { std::vector<T>().swap(data); std::vector<T>().swap(tempBrick); }
    }

    // VolumeReader.h:151-223.  Brick placement runs on the device with 64-bit indices
    // (the reference's 32-bit ones wrap above 2^32 voxels, :171).
    bool LoadBricksToTexture(int64_t numBricks, int64_t I, int64_t J, int64_t K, int timestep, bool dealloc, bool toGPU = true)
    {
        using namespace vrhip_detail;
        const int64_t X = brickDims[0], Y = brickDims[1], Z = brickDims[2], XYZ = X * Y * Z;
        std::vector<T> all((size_t)(XYZ * numBricks));
        std::vector<int64_t> ijk((size_t)numBricks * 3);
        for (int b = 0; b < numBricks; ++b) {
            for (int k = 0; k < 3; ++k) ijk[(size_t)b * 3 + k] = (*brickMap)[b][k];
            if (!LoadVolumeFromBinaryFile(findSourceFile(b, timestep))) {
                std::cout << "Load error. Brick loading terminated." << std::endl;
                return false;
            }
            std::copy(tempBrick.begin(), tempBrick.end(), all.begin() + (size_t)b * XYZ);
        }
        DeviceBuffer bricks;
        bricks.ensure(XYZ * numBricks);
        check(vr_upload(bricks.p, all.data(), XYZ * numBricks, nullptr), "vr_upload");
        vr_free(textureId);
        textureId = nullptr;
        check(vr_malloc(&textureId, I * J * K * XYZ), "vr_malloc");
        const int64_t grid[3] = {I, J, K};
        check(vr_assemble_bricks((const uint8_t *)bricks.p, (int32_t)numBricks, brickDims, ijk.data(), grid,
                                 (uint8_t *)textureId, nullptr), "vr_assemble_bricks");
        dataDims[0] = I * X; dataDims[1] = J * Y; dataDims[2] = K * Z;
        std::cout << "TEXTURE SIZE: " << (double)(dataDims[0] * dataDims[1] * dataDims[2]) / 1e9 << " GB" << std::endl;
        data.resize((size_t)(I * J * K * XYZ));
        check(vr_download(data.data(), textureId, I * J * K * XYZ, nullptr), "vr_download");
        if (toGPU && dealloc) std::vector<T>().swap(data);
        return true;
    }

private:
    std::vector<T> tempBrick;
    bool LoadVolumeFromBinaryFile(std::string filename)                                      // VolumeReader.h:244-289
    {
        std::ifstream is(filename, std::ios::in | std::ios::binary);
        if (!is.good()) return false;
        const int64_t expected = (int64_t)sizeof(T) * brickDims[0] * brickDims[1] * brickDims[2];
        is.seekg(0, is.end);
        const int64_t fileSize = is.tellg();
        is.seekg(0, is.beg);
        if (expected != fileSize) throw std::runtime_error("File size does not match expected dataset size!");
        tempBrick.resize((size_t)(brickDims[0] * brickDims[1] * brickDims[2]));
        is.read((char *)tempBrick.data(), fileSize);
        return (bool)is;
    }
};

// volume_renderer/UnitBrick.h:17-119.  Bind() takes what the GL state held implicitly (the bound
// 3-D texture); Draw() renders one frame of the unit cube with the uniforms of main.cpp:319-402.
class UnitBrick {
public:
    void Setup() {}
    void Bind(const void *volume_dev = nullptr, const int64_t *dims = nullptr)
    {
        if (volume_dev) { vol = volume_dev; for (int k = 0; k < 3; ++k) d[k] = dims[k]; }
        bound = true;
    }
    void Unbind() { bound = false; }
    void Delete() { vol = nullptr; }
    // rgba_dev: height*width*4 floats on the device
    void Draw(const vr_camera &cam, const vr_render_params &params, float *rgba_dev, void *stream = nullptr)
    {
        if (!bound || !vol) throw std::runtime_error("UnitBrick::Draw without Bind(volume, dims)");
        vrhip_detail::check(vr_raycast((const uint8_t *)vol, d, &cam, &params, rgba_dev, stream), "vr_raycast");
    }
    static vr_camera defaultCamera()           // main.cpp:33-40
    {
        vr_camera c = {{0.f, 0.f, -0.75f}, {0.f, 0.f, 1.f}, {0.f, 1.f, 0.f}, 50.f, 0.1f, 100.f};
        return c;
    }
    static vr_render_params defaultParams(int w = 1600, int h = 1200, int mode = VR_RENDER_COMPOSITE)   // main.cpp:27,330-334
    {
        vr_render_params p = {};
        p.width = w; p.height = h;
        p.step_size[0] = 1.f / 256.f; p.step_size[1] = 1.f / 256.f; p.step_size[2] = 1.f / 128.f;
        p.iso_value = 40.f / 255.f; p.max_samples = 300; p.mode = mode;
        for (int k = 0; k < 3; ++k) { p.box_min[k] = 0.f; p.box_max[k] = 1.f; }
        return p;
    }

private:
    const void *vol = nullptr;
    int64_t d[3] = {0, 0, 0};
    bool bound = false;
};
