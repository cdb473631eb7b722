// store-bandwidth micro-benchmark: who limits k_decode_quad's dead-tile path (4.4 TB/s) against a fill (6.9 TB/s)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// mode 0: tile pattern (128 x 8 x 4 tile of a 256 x 256 x 128 brick), 16 tiles per wave, workgroup = 256 consecutive tiles
// mode 1: 4 KB contiguous per tile
// mode 2: tile pattern, but the wave's 16 tiles are taken round-robin (tile = k * 16 + wave)
template <int LDSKB>
__global__ void __launch_bounds__(1024) k_store(uint8_t *out, int mode, int tilesPerWave)
{
    __shared__ uint32_t pad[LDSKB * 256];
    if (threadIdx.x == 0) pad[0] = 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per = 16 * tilesPerWave;
    const int64_t tile0 = (int64_t)blockIdx.x * per;
    const int brick = blockIdx.y;
    const uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
    for (int k = 0; k < tilesPerWave; ++k) {
        const int64_t tileId = tile0 + (mode == 2 ? k * 16 + wave : wave * tilesPerWave + k);
        uint8_t *B = out + (int64_t)brick * (256 * 256 * 128);
        if (mode == 1) {
            uint8_t *L = B + tileId * 4096 + lane * 16;
            for (int z = 0; z < 4; ++z) *(uint4 *)(L + 1024 * z) = v;
        } else {
            const int tx = tileId & 1, ty = (tileId >> 1) & 31, tz = tileId >> 6;
            const int c = lane & 7, y = lane >> 3;
            uint8_t *O = B + tx * 128 + c * 16 + 256 * (ty * 8 + y + 256 * (tz * 4));
            for (int z = 0; z < 4; ++z) *(uint4 *)(O + 65536 * z) = v;
        }
    }
}

// mode 3/4: a wave writes one 16 x 16 x 16 cube of a 256 x 256 x 128 brick as 16-byte row pieces (4 stores of 64 lanes);
// cubes enumerated x fastest.  order 0: workgroup b takes cubes 16b .. 16b+15 (16 waves: a full 256-byte row of cubes);
// order 1: XCD-aware -- workgroup ids that land on one XCD (id % 8) take x-neighbouring cubes one after the other
template <int LDSKB>
__global__ void __launch_bounds__(1024) k_cube(uint8_t *out, int order, int cubesPerWave)
{
    __shared__ uint32_t pad[LDSKB * 256];
    if (threadIdx.x == 0) pad[0] = 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int brick = blockIdx.y;
    uint8_t *B = out + (int64_t)brick * (256 * 256 * 128);
    const uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
    const int ncubes = 16 * 16 * 8;                       // per brick
    for (int k = 0; k < cubesPerWave; ++k) {
        int cube;
        if (order == 0) cube = (blockIdx.x * cubesPerWave + k) * 16 + wave;          // the 16 waves = 16 cubes along x
        else cube = (blockIdx.x * 16 + wave) * cubesPerWave + k;                    // a wave walks consecutive cubes
        if (cube >= ncubes) return;
        const int cx = cube & 15, cy = (cube >> 4) & 15, cz = cube >> 8;
        // lane -> row piece: y = lane & 15, z = (lane >> 4) * 4 + store index
        const int y = lane & 15;
        uint8_t *O = B + cx * 16 + 256 * (cy * 16 + y + 256 * (cz * 16 + (lane >> 4) * 4));
        for (int z = 0; z < 4; ++z) *(uint4 *)(O + 65536 * z) = v;
    }
}

__global__ void __launch_bounds__(256) k_fill(uint4 *out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = make_uint4(1, 2, 3, 4);
}

int main()
{
    const int B = 960;
    const int64_t bytes = (int64_t)B * 256 * 256 * 128;
    uint8_t *out;
    CK(hipMalloc(&out, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        float best = 1e9;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-52s %.3f ms  %.2f TB/s\n", name, best, bytes / best / 1e9);
    };
    const int ntiles = 2048;
    run("fill, grid-stride 16 B/thread, 8192 WGs x 256", [&] { hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (uint4 *)out, bytes / 16); });
    for (int mode = 0; mode < 3; ++mode) {
        char nm[128];
        snprintf(nm, sizeof nm, "mode %d, 1 WG/CU (140 KB LDS), 16 tiles/wave", mode);
        run(nm, [&] { hipLaunchKernelGGL(k_store<140>, dim3(ntiles / 256, B), dim3(1024), 0, 0, out, mode, 16); });
        snprintf(nm, sizeof nm, "mode %d, 2 WG/CU (64 KB LDS), 16 tiles/wave", mode);
        run(nm, [&] { hipLaunchKernelGGL(k_store<64>, dim3(ntiles / 256, B), dim3(1024), 0, 0, out, mode, 16); });
        snprintf(nm, sizeof nm, "mode %d, 1 WG/CU (140 KB LDS), 64 tiles/wave", mode);
        run(nm, [&] { hipLaunchKernelGGL(k_store<140>, dim3(ntiles / 1024, B), dim3(1024), 0, 0, out, mode, 64); });
    }
    for (int order = 0; order < 2; ++order)
        for (int cpw = 1; cpw <= 8; cpw *= 8) {
            char nm[128];
            snprintf(nm, sizeof nm, "cube pieces, order %d, %d cubes/wave, 1 WG/CU", order, cpw);
            run(nm, [&] { hipLaunchKernelGGL(k_cube<140>, dim3(2048 / 16 / cpw, B), dim3(1024), 0, 0, out, order, cpw); });
        }
    return 0;
}
