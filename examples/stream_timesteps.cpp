// stream_timesteps.cpp -- the reference's start-up sequence (main.cpp:242-290) over several timesteps, overlapped:
// vrhip::TimestepStreamer (include/vrhip/TimestepStreamer.hpp) reads brick files, uploads, builds, decodes and hands
// every decoded timestep to a callback that draws a frame; the same run in the reference's order must give the same
// bytes.  Plain C++ (g++), no HIP headers.
//
//   g++ -std=c++14 -O2 -pthread -Iinclude examples/stream_timesteps.cpp -Lvolumerenderer_amd -lvrhip
//       -Wl,-rpath,$PWD/volumerenderer_amd -o /tmp/stream_timesteps ; /tmp/stream_timesteps /tmp/bricks
#include "vrhip/TimestepStreamer.hpp"
#include "vrhip/VolumeReader.hpp"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

static int64_t BRICK_DIM[3] = {32, 32, 32};
static std::string dir;

static std::string findBrickBinaryFile(int brick, int timestep)     // main.cpp:581-597
{
    std::ostringstream o;
    o << dir << "/d_" << timestep << "_" << brick;
    return o.str();
}

static unsigned long long fnv1a64(const std::vector<uint8_t> &v)
{
    unsigned long long h = 0xcbf29ce484222325ull;
    for (uint8_t b : v) { h ^= b; h *= 0x100000001b3ull; }
    return h;
}

int main(int argc, char **argv)
{
    dir = argc > 1 ? argv[1] : "/tmp";
    const int B = 8, T = 5;
    const int64_t V = BRICK_DIM[0] * BRICK_DIM[1] * BRICK_DIM[2];
    std::vector<int> timesteps;
    for (int t = 0; t < T; ++t) {
        timesteps.push_back(270 + t);
        for (int b = 0; b < B; ++b) {
            std::vector<unsigned char> v((size_t)V);
            for (size_t i = 0; i < v.size(); ++i)
                v[i] = (unsigned char)(128 + 90 * std::sin(0.07 * (double)(i % 32) + b + 0.3 * t) + ((i * 2654435761u + (unsigned)t) >> 30));
            std::ofstream(findBrickBinaryFile(b, 270 + t), std::ios::binary).write((const char *)v.data(), (std::streamsize)v.size());
        }
    }
    vrhip::BrickFileSource src(findBrickBinaryFile, B, BRICK_DIM, timesteps);
    std::vector<unsigned long long> hashes[2];
    for (int pass = 0; pass < 2; ++pass) {
        vrhip::TimestepStreamer st(B, BRICK_DIM, 1, 2);
        std::vector<std::vector<uint8_t>> host((size_t)T, std::vector<uint8_t>((size_t)(B * V)));
        float *frame = nullptr;
        vr_malloc((void **)&frame, 160 * 120 * 4 * (int64_t)sizeof(float));
        vr_camera cam = UnitBrick::defaultCamera();
        vr_render_params P = UnitBrick::defaultParams(160, 120);
        st.run(src, [&](size_t t, const uint8_t *vol, void *stream) {
            // "draw": one frame of the first brick, then keep the decoded bytes (asynchronous: same stream)
            vrhip_detail::check(vr_raycast(vol, BRICK_DIM, &cam, &P, frame, stream), "vr_raycast");
            vrhip_detail::check(vr_download_async(host[t].data(), vol, B * V, stream), "vr_download_async");
        }, pass == 0);
        vr_free(frame);
        for (int t = 0; t < T; ++t) hashes[pass].push_back(fnv1a64(host[(size_t)t]));
    }
    bool same = true;
    for (int t = 0; t < T; ++t) {
        std::printf("timestep %d decoded fnv1a64 %016llx\n", 270 + t, hashes[0][(size_t)t]);
        same = same && hashes[0][(size_t)t] == hashes[1][(size_t)t];
    }
    std::printf("overlapped == sequential: %d\n", same ? 1 : 0);
    // a brick file of the wrong size must raise like VolumeReader.h:258-260
    std::ofstream(findBrickBinaryFile(3, 270), std::ios::binary).write("short", 5);
    bool raised = false;
    try {
        vrhip::TimestepStreamer st(B, BRICK_DIM, 1, 2);
        st.run(src, nullptr);
    } catch (const std::exception &ex) {
        raised = std::strstr(ex.what(), "File size does not match") != nullptr;
    }
    std::printf("wrong file size raises: %d\n", raised ? 1 : 0);
    return same && raised ? 0 : 1;
}
