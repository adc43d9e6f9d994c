// decode_lab.hip -- variants of the BC1 block decoder's store side, timed against each other on one MI355X.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/decode_lab tools/decode_lab.hip && /tmp/decode_lab [pixel GiB]
// Not part of the library; records the experiments behind csrc/bcn_decode.hip.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../dxt-lossless-transform_amd/csrc/bcn_decode.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
using namespace dxtlt;

#define CHECK(x)                                                                     \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
            std::exit(1);                                                            \
        }                                                                            \
    } while (0)

constexpr int T = 256, ROW = 64 + 4, WAVE = 4 * ROW;

__device__ __forceinline__ void store_sc1nt(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_plain(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
__device__ __forceinline__ void store_nt(void* p, u32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); }

// POLICY 0 nt builtin, 1 sc1 nt, 2 plain.  SYNC 0 __syncthreads, 1 wave barrier.  COMPUTE 0: pixels = copies of the block words
template <int POLICY, int SYNC, int COMPUTE, int PER_LANE>
__global__ void __launch_bounds__(T) lab_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n)
{
    __shared__ u32x4 stage[(T / 64) * WAVE * PER_LANE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int u = 0; u < PER_LANE; ++u) {
        const uint64_t wave_first = ((uint64_t)blockIdx.x * PER_LANE + u) * T + 64 * wave;
        const uint64_t b = wave_first + lane;
        uint32_t q[4] = {0, 0, 0, 0}, px[16];
        if (b < n) {
            const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(in) + b);
            q[0] = v.x, q[1] = v.y;
        }
        if (COMPUTE)
            decode_block_px<1>(q, px);
        else
            for (int i = 0; i < 16; ++i)
                px[i] = q[i & 1] + i;
        u32x4* mine = stage + (wave * PER_LANE + u) * WAVE;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            mine[r * ROW + lane] = u32x4{px[4 * r], px[4 * r + 1], px[4 * r + 2], px[4 * r + 3]};
    }
    if (SYNC == 0)
        __syncthreads();
    else
        __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < PER_LANE; ++u) {
        const uint64_t wave_first = ((uint64_t)blockIdx.x * PER_LANE + u) * T + 64 * wave;
        u32x4* mine = stage + (wave * PER_LANE + u) * WAVE;
        u32x4* dst = reinterpret_cast<u32x4*>(out) + 4 * wave_first;
        const uint64_t chunks = n > wave_first ? 4 * (n - wave_first) : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = 64 * k + lane;
            if ((uint64_t)j < chunks) {
                const u32x4 v = mine[(j & 3) * ROW + (j >> 2)];
                if (POLICY == 0)
                    store_nt(dst + j, v);
                else if (POLICY == 1)
                    store_sc1nt(dst + j, v);
                else
                    store_plain(dst + j, v);
            }
        }
    }
}

// 16-byte blocks (BC2 / BC3 input rate): DEC 0 = BC1 decode of the colour half (isolates the extra 8 bytes read),
// 2 = BC2, 3 = BC3.  WG = workgroup threads (256 or 128)
template <int DEC, int WG>
__global__ void __launch_bounds__(WG) lab16_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n)
{
    __shared__ u32x4 stage[(WG / 64) * WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t wave_first = (uint64_t)blockIdx.x * WG + 64 * wave;
    const uint64_t b = wave_first + lane;
    uint32_t q[4] = {0, 0, 0, 0}, px[16];
    if (b < n) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in) + b);
        q[0] = v.x, q[1] = v.y, q[2] = v.z, q[3] = v.w;
    }
    if (DEC == 0)
        decode_bc1_block_px(q[2] ^ q[0], q[3] ^ q[1], px);
    else if (DEC == 2)
        decode_block_px<2>(q, px);
    else
        decode_block_px<3>(q, px);
    u32x4* mine = stage + wave * WAVE;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        mine[r * ROW + lane] = u32x4{px[4 * r], px[4 * r + 1], px[4 * r + 2], px[4 * r + 3]};
    __syncthreads();
    u32x4* dst = reinterpret_cast<u32x4*>(out) + 4 * wave_first;
    const uint64_t chunks = n > wave_first ? 4 * (n - wave_first) : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = 64 * k + lane;
        if ((uint64_t)j < chunks)
            store_sc1nt(dst + j, mine[(j & 3) * ROW + (j >> 2)]);
    }
}

// store side only, no LDS: chunk j of the wave gets a value made from the lane's own block words
template <int POLICY, int XCD>
__global__ void __launch_bounds__(T) nolds_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t wg = blockIdx.x;
    if (XCD) {
        const uint64_t per = gridDim.x / 8;   // grid is a multiple of 8 in this lab
        wg = (blockIdx.x % 8) * per + blockIdx.x / 8;
    }
    const uint64_t wave_first = wg * T + 64 * wave;
    const uint64_t b = wave_first + lane;
    u32x2 v = {0, 0};
    if (b < n)
        v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(in) + b);
    u32x4* dst = reinterpret_cast<u32x4*>(out) + 4 * wave_first;
    const uint64_t chunks = n > wave_first ? 4 * (n - wave_first) : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = 64 * k + lane;
        if ((uint64_t)j < chunks) {
            const u32x4 w = {v.x + k, v.y, v.x ^ v.y, (uint32_t)j};
            if (POLICY == 0)
                store_nt(dst + j, w);
            else if (POLICY == 1)
                store_sc1nt(dst + j, w);
            else
                store_plain(dst + j, w);
        }
    }
}

// the decoder with the workgroup -> tile map made XCD-contiguous
template <int POLICY>
__global__ void __launch_bounds__(T) xcd_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n)
{
    __shared__ u32x4 stage[(T / 64) * WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t per = gridDim.x / 8;
    const uint64_t wg = (blockIdx.x % 8) * per + blockIdx.x / 8;
    const uint64_t wave_first = wg * T + 64 * wave;
    const uint64_t b = wave_first + lane;
    uint32_t q[4] = {0, 0, 0, 0}, px[16];
    if (b < n) {
        const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(in) + b);
        q[0] = v.x, q[1] = v.y;
    }
    decode_block_px<1>(q, px);
    u32x4* mine = stage + wave * WAVE;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        mine[r * ROW + lane] = u32x4{px[4 * r], px[4 * r + 1], px[4 * r + 2], px[4 * r + 3]};
    __builtin_amdgcn_wave_barrier();
    u32x4* dst = reinterpret_cast<u32x4*>(out) + 4 * wave_first;
    const uint64_t chunks = n > wave_first ? 4 * (n - wave_first) : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = 64 * k + lane;
        if ((uint64_t)j < chunks) {
            const u32x4 v = mine[(j & 3) * ROW + (j >> 2)];
            if (POLICY == 1)
                store_sc1nt(dst + j, v);
            else
                store_nt(dst + j, v);
        }
    }
}

// the decoder as a fixed grid walking the tiles (tile = workgroup + k * grid): bounded write concurrency, a moving window
template <int POLICY, int PREFETCH>
__global__ void __launch_bounds__(T) walk_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n)
{
    __shared__ u32x4 stage[(T / 64) * WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t tiles = (n + T - 1) / T;
    u32x4* mine = stage + wave * WAVE;
    uint64_t tile = blockIdx.x;
    u32x2 next = {0, 0};
    if (PREFETCH && tile < tiles && tile * T + threadIdx.x < n)
        next = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(in) + tile * T + threadIdx.x);
    for (; tile < tiles; tile += gridDim.x) {
        const uint64_t wave_first = tile * T + 64 * wave;
        const uint64_t b = wave_first + lane;
        uint32_t q[4] = {0, 0, 0, 0}, px[16];
        if (PREFETCH) {
            q[0] = next.x, q[1] = next.y;
            const uint64_t nb = b + (uint64_t)gridDim.x * T;
            if (nb < n)
                next = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(in) + nb);
        } else if (b < n) {
            const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(in) + b);
            q[0] = v.x, q[1] = v.y;
        }
        decode_block_px<1>(q, px);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            mine[r * ROW + lane] = u32x4{px[4 * r], px[4 * r + 1], px[4 * r + 2], px[4 * r + 3]};
        __builtin_amdgcn_wave_barrier();
        u32x4* dst = reinterpret_cast<u32x4*>(out) + 4 * wave_first;
        const uint64_t chunks = n > wave_first ? 4 * (n - wave_first) : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = 64 * k + lane;
            if ((uint64_t)j < chunks) {
                const u32x4 v = mine[(j & 3) * ROW + (j >> 2)];
                if (POLICY == 1)
                    store_sc1nt(dst + j, v);
                else if (POLICY == 0)
                    store_nt(dst + j, v);
                else
                    store_plain(dst + j, v);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// each lane stores its own 64 bytes (no LDS): the layout this kernel avoids
template <int POLICY>
__global__ void __launch_bounds__(T) direct_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n)
{
    const uint64_t b = (uint64_t)blockIdx.x * T + threadIdx.x;
    if (b >= n)
        return;
    const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(in) + b);
    uint32_t q[4] = {v.x, v.y, 0, 0}, px[16];
    decode_block_px<1>(q, px);
    u32x4* dst = reinterpret_cast<u32x4*>(out) + 4 * b;
    for (int r = 0; r < 4; ++r) {
        const u32x4 w = {px[4 * r], px[4 * r + 1], px[4 * r + 2], px[4 * r + 3]};
        if (POLICY == 1)
            store_sc1nt(dst + r, w);
        else
            store_nt(dst + r, w);
    }
}

__global__ void fill_kernel(uint32_t* p, uint64_t words)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) {
        uint64_t z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
        p[i] = (uint32_t)(z ^ (z >> 31));
    }
}

template <typename F>
double time_ms(F launch, int steps = 10)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    launch();
    launch();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < steps; ++i)
        launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / steps;
}

int main(int argc, char** argv)
{
    const double gib = argc > 1 ? std::atof(argv[1]) : 8.0;
    const uint64_t n = (uint64_t)(gib * (1ull << 30)) / 64;
    uint8_t *in, *out;
    CHECK(hipMalloc(&in, n * 8));
    CHECK(hipMalloc(&out, n * 64));
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n * 2 + 255) / 256)), dim3(256), 0, 0, (uint32_t*)in, n * 2);
    CHECK(hipDeviceSynchronize());
    const double bytes = (double)n * 72;
    auto report = [&](const char* name, double ms) { std::printf("%-44s %8.4f ms  %.4f of 8 TB/s\n", name, ms, bytes / (ms * 1e-3) / 8e12); };
    const unsigned g1 = (unsigned)((n + T - 1) / T), g2 = (unsigned)((n + 2 * T - 1) / (2 * T)), g4 = (unsigned)((n + 4 * T - 1) / (4 * T));
#define RUN(name, kern, grid) report(name, time_ms([&] { hipLaunchKernelGGL((kern), dim3(grid), dim3(T), 0, 0, in, out, n); }))
    RUN("nt builtin, syncthreads (shipped first)", (lab_kernel<0, 0, 1, 1>), g1);
    RUN("sc1 nt, syncthreads", (lab_kernel<1, 0, 1, 1>), g1);
    RUN("plain store, syncthreads", (lab_kernel<2, 0, 1, 1>), g1);
    RUN("sc1 nt, wave barrier", (lab_kernel<1, 1, 1, 1>), g1);
    RUN("nt, wave barrier", (lab_kernel<0, 1, 1, 1>), g1);
    RUN("sc1 nt, wave barrier, 2 blocks per lane", (lab_kernel<1, 1, 1, 2>), g2);
    RUN("nt, wave barrier, 2 blocks per lane", (lab_kernel<0, 1, 1, 2>), g2);
    RUN("sc1 nt, wave barrier, 4 blocks per lane", (lab_kernel<1, 1, 1, 4>), g4);
    RUN("no decode (store side only), sc1 nt", (lab_kernel<1, 1, 0, 1>), g1);
    RUN("no decode (store side only), nt", (lab_kernel<0, 1, 0, 1>), g1);
    RUN("store side, no LDS, sc1 nt", (nolds_kernel<1, 0>), g1);
    RUN("store side, no LDS, nt", (nolds_kernel<0, 0>), g1);
    RUN("store side, no LDS, plain", (nolds_kernel<2, 0>), g1);
    RUN("store side, no LDS, sc1 nt, XCD-contiguous", (nolds_kernel<1, 1>), g1);
    RUN("store side, no LDS, plain, XCD-contiguous", (nolds_kernel<2, 1>), g1);
    RUN("decode, sc1 nt, wave barrier, XCD-contiguous", (xcd_kernel<1>), g1);
    RUN("decode, nt, wave barrier, XCD-contiguous", (xcd_kernel<0>), g1);
    for (unsigned per_cu : {2u, 4u, 8u, 16u, 32u}) {
        char name[96];
        std::snprintf(name, sizeof name, "walk, %u WGs per CU, sc1 nt", per_cu);
        RUN(name, (walk_kernel<1, 0>), 256 * per_cu);
        std::snprintf(name, sizeof name, "walk, %u WGs per CU, sc1 nt, prefetch", per_cu);
        RUN(name, (walk_kernel<1, 1>), 256 * per_cu);
        std::snprintf(name, sizeof name, "walk, %u WGs per CU, plain, prefetch", per_cu);
        RUN(name, (walk_kernel<2, 1>), 256 * per_cu);
    }
    {
        uint8_t* in16;
        CHECK(hipMalloc(&in16, n * 16));
        hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0, 0, (uint32_t*)in16, n * 4);
        CHECK(hipDeviceSynchronize());
        const double bytes16 = (double)n * 80;
        auto report16 = [&](const char* name, double ms) { std::printf("%-44s %8.4f ms  %.4f of 8 TB/s\n", name, ms, bytes16 / (ms * 1e-3) / 8e12); };
#define RUN16(name, kern, wg) report16(name, time_ms([&] { hipLaunchKernelGGL((kern), dim3((unsigned)((n + (wg) - 1) / (wg))), dim3(wg), 0, 0, in16, out, n); }))
        RUN16("16-byte blocks, BC1 decode of half, 256 thr", (lab16_kernel<0, 256>), 256);
        RUN16("16-byte blocks, BC2 decode, 256 thr", (lab16_kernel<2, 256>), 256);
        RUN16("16-byte blocks, BC3 decode, 256 thr", (lab16_kernel<3, 256>), 256);
        RUN16("16-byte blocks, BC2 decode, 128 thr", (lab16_kernel<2, 128>), 128);
        RUN16("16-byte blocks, BC3 decode, 128 thr", (lab16_kernel<3, 128>), 128);
        RUN16("16-byte blocks, BC1 decode of half, 128 thr", (lab16_kernel<0, 128>), 128);
        CHECK(hipFree(in16));
    }
    RUN("direct 64 B per lane, nt", (direct_kernel<0>), g1);
    RUN("direct 64 B per lane, sc1 nt", (direct_kernel<1>), g1);
    report("hipMemsetAsync of the pixel buffer (64/72)", time_ms([&] { CHECK(hipMemsetAsync(out, 0x5A, n * 64, 0)); }) * 72.0 / 64.0);
    return 0;
}
