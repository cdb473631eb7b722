// TimestepStreamer.hpp -- header-only C++ counterpart of volumerenderer_amd/pipeline.py over the C ABI: the start-up
// sequence of the reference's main() (main.cpp:242-290: LoadBricksToTexture -> build -> levelCut -> draw, strictly one
// after another) as a pipeline over timesteps.  Disk -> pinned host memory (a reader thread, two staging buffers) ->
// device (copy stream, two volumes) -> build -> levelCut -> the caller's frames (compute stream): the upload of
// timestep t+1 runs beside the build / decode / frames of timestep t, and nothing synchronises the device per timestep.
// Compiles with plain g++ (no HIP header): streams and events are the C ABI's void*.
#pragma once
#include "../vrhip.h"
#include "VolumeKdtree.hpp"
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace vrhip {

// The disk stage: VolumeReader<T>::LoadVolumeFromBinaryFile for every brick of a timestep (VolumeReader.h:244-289, called
// from LoadBricksToTexture :172-182).  findSourceFile(brick, timestep) -> path, as main.cpp:581-597.
class BrickFileSource {
public:
    std::function<std::string(int, int)> findSourceFile;
    int numBricks = 0;
    int64_t brickBytes = 0;
    std::vector<int> timesteps;

    BrickFileSource(std::function<std::string(int, int)> find, int bricks, const int64_t brickDims[3], std::vector<int> ts)
        : findSourceFile(std::move(find)), numBricks(bricks), brickBytes(brickDims[0] * brickDims[1] * brickDims[2]), timesteps(std::move(ts)) {}

    size_t size() const { return timesteps.size(); }

    // timestep i -> host buffer (numBricks * brickBytes), brick after brick; a file of the wrong size throws like
    // VolumeReader.h:258-260
    void readInto(size_t i, uint8_t *dst) const
    {
        for (int b = 0; b < numBricks; ++b) {
            const std::string path = findSourceFile(b, timesteps[i]);
            FILE *f = std::fopen(path.c_str(), "rb");
            if (!f) throw std::runtime_error("cannot open " + path);
            std::fseek(f, 0, SEEK_END);
            const long sz = std::ftell(f);
            std::fseek(f, 0, SEEK_SET);
            if ((int64_t)sz != brickBytes) { std::fclose(f); throw std::runtime_error("File size does not match expected dataset size!"); }
            const size_t got = std::fread(dst + (size_t)b * brickBytes, 1, (size_t)brickBytes, f);
            std::fclose(f);
            if ((int64_t)got != brickBytes) throw std::runtime_error("short read: " + path);
        }
    }
};

class TimestepStreamer {
public:
    vr_brickset *set = nullptr;
    int numBricks = 0;
    int64_t bytes = 0;                      // one timestep
    void *vox[2] = {nullptr, nullptr};      // device: raw bricks of timesteps t, t+1
    void *out[2] = {nullptr, nullptr};      // device: decoded bricks
    void *copyStream = nullptr, *computeStream = nullptr;

    TimestepStreamer(int bricks, const int64_t brickDims[3], int tolerance = 1, int maxEpochs = 2, int variant = VR_VARIANT_RECOVER)
        : numBricks(bricks), bytes((int64_t)bricks * brickDims[0] * brickDims[1] * brickDims[2])
    {
        vrhip_detail::check(vr_brickset_create(&set, bricks, brickDims, tolerance, maxEpochs, variant), "vr_brickset_create");
        for (int i = 0; i < 2; ++i) {
            vrhip_detail::check(vr_malloc(&vox[i], bytes), "vr_malloc");
            vrhip_detail::check(vr_malloc(&out[i], bytes), "vr_malloc");
            vrhip_detail::check(vr_malloc_host(&stage[i], bytes), "vr_malloc_host");
        }
        vrhip_detail::check(vr_stream_create(&copyStream), "vr_stream_create");
        vrhip_detail::check(vr_stream_create(&computeStream), "vr_stream_create");
    }
    ~TimestepStreamer()
    {
        if (computeStream) vr_stream_synchronize(computeStream);
        if (copyStream) vr_stream_synchronize(copyStream);
        for (int i = 0; i < 2; ++i) { vr_free(vox[i]); vr_free(out[i]); vr_free_host(stage[i]); }
        vr_stream_destroy(copyStream); vr_stream_destroy(computeStream);
        vr_brickset_destroy(set);
    }
    TimestepStreamer(const TimestepStreamer &) = delete;
    TimestepStreamer &operator=(const TimestepStreamer &) = delete;

    // onDecoded(t, decoded device bricks, compute stream): launch the frames of timestep t on that stream.
    // overlap = false: the reference's order (upload, build, decode, draw, one after another).
    void run(const BrickFileSource &src, const std::function<void(size_t, const uint8_t *, void *)> &onDecoded, bool overlap = true,
             int cutDepth = -1)
    {
        const size_t T = src.size();
        if (T == 0) return;
        std::vector<void *> uploaded(T, nullptr), consumed(T, nullptr);
        for (size_t t = 0; t < T; ++t) {
            vrhip_detail::check(vr_event_create(&uploaded[t]), "vr_event_create");
            vrhip_detail::check(vr_event_create(&consumed[t]), "vr_event_create");
        }
        // ---- the disk stage: a reader thread fills the two pinned staging buffers in turn; buffer t & 1 is free again
        // once the upload of timestep t - 2 out of it has finished
        std::mutex mu;
        std::condition_variable cv;
        std::vector<char> ready(T, 0), issued(T, 0);
        std::string err;
        std::thread reader([&] {
            try {
                for (size_t t = 0; t < T; ++t) {
                    if (t >= 2) {
                        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return issued[t - 2] != 0; }); }
                        vrhip_detail::check(vr_event_synchronize(uploaded[t - 2]), "vr_event_synchronize");
                    }
                    src.readInto(t, (uint8_t *)stage[t & 1]);
                    { std::lock_guard<std::mutex> lk(mu); ready[t] = 1; }
                    cv.notify_all();
                }
            } catch (const std::exception &ex) {
                std::lock_guard<std::mutex> lk(mu);
                err = ex.what();
                for (auto &r : ready) r = 1;
                cv.notify_all();
            }
        });
        const auto upload = [&](size_t t) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return ready[t] != 0; }); }
            if (!err.empty()) return;
            if (t >= 2) vrhip_detail::check(vr_stream_wait_event(copyStream, consumed[t - 2]), "vr_stream_wait_event");   // the device buffer's previous user
            vrhip_detail::check(vr_upload_async(vox[t & 1], stage[t & 1], bytes, copyStream), "vr_upload_async");
            vrhip_detail::check(vr_event_record(uploaded[t], copyStream), "vr_event_record");
            { std::lock_guard<std::mutex> lk(mu); issued[t] = 1; }
            cv.notify_all();
        };
        std::string fail;
        try {
            upload(0);
            for (size_t t = 0; t < T && err.empty(); ++t) {
                if (overlap && t + 1 < T) upload(t + 1);
                if (!err.empty()) break;
                vrhip_detail::check(vr_stream_wait_event(computeStream, uploaded[t]), "vr_stream_wait_event");
                vrhip_detail::check(vr_brickset_build(set, (const uint8_t *)vox[t & 1], computeStream), "vr_brickset_build");
                vrhip_detail::check(vr_brickset_decode(set, cutDepth, (uint8_t *)out[t & 1], computeStream), "vr_brickset_decode");
                if (onDecoded) onDecoded(t, (const uint8_t *)out[t & 1], computeStream);
                vrhip_detail::check(vr_event_record(consumed[t], computeStream), "vr_event_record");
                if (!overlap && t + 1 < T) { upload(t + 1); vrhip_detail::check(vr_stream_synchronize(copyStream), "vr_stream_synchronize"); }
            }
        } catch (const std::exception &ex) {
            fail = ex.what();
        }
        {   // let the reader finish whatever happens (it may be waiting for an upload that will never be issued)
            std::lock_guard<std::mutex> lk(mu);
            for (auto &i : issued) i = 1;
        }
        cv.notify_all();
        vr_stream_synchronize(copyStream);
        vr_stream_synchronize(computeStream);
        reader.join();
        for (size_t t = 0; t < T; ++t) { vr_event_destroy(uploaded[t]); vr_event_destroy(consumed[t]); }
        if (!fail.empty()) throw std::runtime_error(fail);
        if (!err.empty()) throw std::runtime_error(err);
    }

private:
    void *stage[2] = {nullptr, nullptr};    // pinned host staging buffers of the disk stage
};

} // namespace vrhip
