// channel_lab.hip -- how do addresses map to HBM channels on MI355X, and what does it cost when streams meet on one?
//
// Round 3 left four placement effects "measured, not explained" (power-of-two strides between the buffers of a batch, BC7's
// two placement levels, the BC1 no-split halo forward, pinned host arrays).  This lab measures the address -> channel
// behaviour directly, with the access shape of the transform kernels (one 16-byte vector per lane, 4 KiB per workgroup,
// thousands of short workgroups, nt loads, sc1 nt stores):
//
//   stride   read-only: workgroup k reads 4 KiB at k * S (S = 4 KiB .. 64 MiB, and S + 256 / S + 4352): every power of two
//            that keeps all concurrent accesses on a subset of the channels shows as a drop -- the channel interleave
//   rw       copy of 1 GiB: the write stream starts D bytes after the read stream's start modulo 2^k (D = 0, 256 B .. 64 MiB
//            and D + small offsets): do a read stream and a write stream that walk the channels in lock step hurt each other?
//   streams  the transform's output shape: every workgroup reads 4 KiB and writes six slices (256 + 256 + 1536 + 512 + 512 +
//            1024 B, BC3 with both splits) to six streams that start P bytes apart; P = N * width (the real layout) for
//            N = 2^k blocks and N = 2^k + 17, + 272 ...: is it the STREAM bases being a power of two apart that costs?
//   batch    B buffers of 2^k bytes at stride 2^k against stride 2^k + 4352: reads and writes of one buffer `stride` apart
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/channel_lab tools/channel_lab.hip ; tools/channel_lab [experiment]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                          \
    do {                                                                  \
        hipError_t e_ = (x);                                              \
        if (e_ != hipSuccess) {                                           \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));  \
            std::exit(1);                                                 \
        }                                                                 \
    } while (0)

__device__ __forceinline__ void store_wt(void* p, u32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// workgroup k reads its 4 KiB at base + k * stride (mod span)
__global__ void __launch_bounds__(256) stride_read(const uint8_t* __restrict__ base, uint64_t stride, uint64_t span, uint32_t* sink)
{
    const uint64_t off = ((uint64_t)blockIdx.x * stride) % span + threadIdx.x * 16;
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + off));
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u)
        sink[0] = 1;
}

__global__ void __launch_bounds__(256) stride_write(uint8_t* __restrict__ base, uint64_t stride, uint64_t span)
{
    const uint64_t off = ((uint64_t)blockIdx.x * stride) % span + threadIdx.x * 16;
    store_wt(base + off, u32x4{threadIdx.x, 1, 2, 3});
}

// plain copy, 4 KiB per workgroup
__global__ void __launch_bounds__(256) copy4k(const uint8_t* __restrict__ in, uint8_t* __restrict__ out)
{
    const uint64_t off = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    store_wt(out + off, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + off)));
}

// the transform's output shape: 4 KiB in, six stream slices out (widths 1, 1, 6, 2, 2, 4 bytes per block, 256 blocks per tile)
struct Streams6 {
    uint64_t base[6];
};
__global__ void __launch_bounds__(256) six_streams(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, Streams6 s)
{
    const int t = threadIdx.x;
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + ((uint64_t)blockIdx.x * 256 + t) * 16));
    // image byte 16 t of the tile: which stream, which offset (as in soa_offset_of_image_byte)
    const int o = t * 16;
    const int lo[7] = {0, 256, 512, 2048, 2560, 3072, 4096};
    const int w[6] = {1, 1, 6, 2, 2, 4};
    uint64_t g = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (o >= lo[i] && o < lo[i + 1])
            g = s.base[i] + (uint64_t)blockIdx.x * (uint64_t)(w[i] * 256) + (uint64_t)(o - lo[i]);
    store_wt(out + g, v);
}

// batch6: buffer b = blockIdx.y holds N = tiles * 256 blocks; 4 KiB tiles in, six stream slices out at out + b * stride_out + off_s * N
__global__ void __launch_bounds__(256) batch_six_streams(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t stride_in,
                                                         uint64_t stride_out, uint64_t n_blocks)
{
    const int t = threadIdx.x;
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + blockIdx.y * stride_in + ((uint64_t)blockIdx.x * 256 + t) * 16));
    const int o = t * 16;
    const int lo[7] = {0, 256, 512, 2048, 2560, 3072, 4096};
    const int w[6] = {1, 1, 6, 2, 2, 4};
    const int offs[6] = {0, 1, 2, 8, 10, 12};
    uint64_t g = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (o >= lo[i] && o < lo[i + 1])
            g = (uint64_t)offs[i] * n_blocks + (uint64_t)blockIdx.x * (uint64_t)(w[i] * 256) + (uint64_t)(o - lo[i]);
    store_wt(out + blockIdx.y * stride_out + g, v);
}

// batch: buffer b = blockIdx.y at in + b * stride_in, out + b * stride_out; tiles blockIdx.x
__global__ void __launch_bounds__(256) batch_copy(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t stride_in,
                                                  uint64_t stride_out)
{
    const uint64_t off = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    store_wt(out + blockIdx.y * stride_out + off,
             __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + blockIdx.y * stride_in + off)));
}

template <typename F>
static double time_ms(F&& launch, int reps = 6)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return ms / reps;
}

static void clock_warm(const uint8_t* in, uint8_t* out)
{
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(copy4k, dim3(1 << 18), dim3(256), 0, 0, in, out);
    CHECK(hipDeviceSynchronize());
}

int main(int argc, char** argv)
{
    const char* which = argc > 1 ? argv[1] : "all";
    auto want = [&](const char* n) { return !std::strcmp(which, "all") || !std::strcmp(which, n); };
    const uint64_t span = 64ull << 30;   // one 64 GiB arena: strides up to 64 MiB still touch 1024 distinct places
    uint8_t* arena;
    uint32_t* sink;
    CHECK(hipMalloc(&arena, span + (1ull << 30)));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMemset(arena, 0x5A, span + (1ull << 30)));
    std::printf("arena %p (%llu GiB)\n", (void*)arena, (unsigned long long)(span >> 30));
    clock_warm(arena, arena + (2ull << 30));

    if (want("stride")) {
        // 2^18 workgroups x 4 KiB = 1 GiB moved per launch whatever the stride
        const uint32_t nwg = 1u << 18;
        std::printf("== stride: workgroup k touches 4 KiB at k * S (mod 64 GiB); fraction of 8 TB/s, read / write\n");
        for (int e = 12; e <= 26; ++e) {
            for (uint64_t extra : {0ull, 256ull, 4352ull}) {
                const uint64_t S = (1ull << e) + extra;
                const double r = time_ms([&] { hipLaunchKernelGGL(stride_read, dim3(nwg), dim3(256), 0, 0, arena, S, span, sink); });
                const double w = time_ms([&] { hipLaunchKernelGGL(stride_write, dim3(nwg), dim3(256), 0, 0, arena, S, span); });
                std::printf("S = 2^%d + %-5llu read %.3f  write %.3f\n", e, (unsigned long long)extra, (double)nwg * 4096 / (r * 1e-3) / 8e12,
                            (double)nwg * 4096 / (w * 1e-3) / 8e12);
                std::fflush(stdout);
            }
        }
    }
    if (want("rw")) {
        std::printf("== rw: copy of 1 GiB, out = in + 8 GiB + D; fraction of 8 TB/s on 2 x 1 GiB\n");
        const uint32_t nwg = 1u << 18;
        std::vector<uint64_t> ds = {0};
        for (int e = 8; e <= 26; ++e) ds.push_back(1ull << e);
        for (uint64_t d : std::vector<uint64_t>{(1ull << 13) + 256, (1ull << 13) + 4352, (1ull << 20) + 256, (1ull << 20) + 4352, 3ull << 12, 5ull << 12,
                                                7ull << 12, 3ull << 19})
            ds.push_back(d);
        for (uint64_t D : ds) {
            const double ms = time_ms([&] { hipLaunchKernelGGL(copy4k, dim3(nwg), dim3(256), 0, 0, arena, arena + (8ull << 30) + D); });
            std::printf("D = %-10llu copy %.3f\n", (unsigned long long)D, 2.0 * nwg * 4096 / (ms * 1e-3) / 8e12);
            std::fflush(stdout);
        }
    }
    if (want("streams")) {
        std::printf("== streams: 1 GiB in (2^26 BC3 blocks' worth of tiles), six slices out, stream bases off_s * N; fraction on 2 x 1 GiB\n");
        const uint32_t nwg = 1u << 18;
        const int offs[6] = {0, 1, 2, 8, 10, 12};
        for (uint64_t extra : {0ull, 16ull, 64ull, 128ull, 272ull, 1024ull, 4352ull, 65536ull + 272ull}) {
            const uint64_t N = (1ull << 26) + extra;   // blocks (a multiple of 16: bases stay 16-byte aligned)
            Streams6 s;
            for (int i = 0; i < 6; ++i) s.base[i] = (uint64_t)offs[i] * N;
            const double ms = time_ms([&] { hipLaunchKernelGGL(six_streams, dim3(nwg), dim3(256), 0, 0, arena, arena + (8ull << 30), s); });
            std::printf("N = 2^26 + %-6llu six streams %.3f\n", (unsigned long long)extra, 2.0 * nwg * 4096 / (ms * 1e-3) / 8e12);
            std::fflush(stdout);
        }
    }
    if (want("stride2")) {
        // which strides of the form 2^a + 2^b put the ~2048 workgroups in flight on few channels?  (read-only, fraction of 8 TB/s)
        const uint32_t nwg = 1u << 18;
        std::printf("== stride2: S = 2^a + 2^b, read-only; rows a = 12..26, columns b = 8..a-1 (then b = none)\n");
        for (int a = 12; a <= 26; ++a) {
            std::printf("a=%2d:", a);
            for (int b = 8; b <= a; ++b) {
                const uint64_t S = (1ull << a) + (b < a ? (1ull << b) : 0);
                const double r = time_ms([&] { hipLaunchKernelGGL(stride_read, dim3(nwg), dim3(256), 0, 0, arena, S, span, sink); }, 3);
                std::printf(" %.2f", (double)nwg * 4096 / (r * 1e-3) / 8e12);
            }
            std::printf("\n");
            std::fflush(stdout);
        }
    }
    if (want("batch6")) {
        std::printf("== batch6: 1 GiB as B buffers of N = 2^k BC3 blocks (16 N bytes): 4 KiB tiles in, six stream slices out (bases off_s * N inside the buffer);\n"
                    "           in / out stride = size + pad; fraction on 2 x 1 GiB\n");
        for (int e : {14, 16, 18}) {
            const uint64_t N = 1ull << e, size = N * 16;
            const uint32_t B = (uint32_t)((1ull << 30) / size);
            for (uint64_t pin : {0ull, 4352ull})
                for (uint64_t pout : {0ull, 256ull, 2304ull, 4352ull, 131328ull}) {
                    const double ms = time_ms([&] {
                        hipLaunchKernelGGL(batch_six_streams, dim3((unsigned)(N / 256), B), dim3(256), 0, 0, arena, arena + (8ull << 30), size + pin,
                                           size + pout, N);
                    });
                    std::printf("N 2^%d x %-5u pad in %-5llu out %-6llu six streams %.3f\n", e, B, (unsigned long long)pin, (unsigned long long)pout,
                                2.0 * (double)size * B / (ms * 1e-3) / 8e12);
                    std::fflush(stdout);
                }
        }
    }
    if (want("batch")) {
        std::printf("== batch: 1 GiB as B buffers of 2^k bytes, copy; in / out stride = size + pad; fraction on 2 x 1 GiB\n");
        for (int e : {18, 20, 22}) {
            const uint64_t size = 1ull << e;
            const uint32_t B = (uint32_t)((1ull << 30) >> e);
            for (uint64_t pin : {0ull, 4352ull})
                for (uint64_t pout : {0ull, 256ull, 4352ull, 65536ull + 4352ull}) {
                    const double ms = time_ms([&] {
                        hipLaunchKernelGGL(batch_copy, dim3((unsigned)(size / 4096), B), dim3(256), 0, 0, arena, arena + (8ull << 30), size + pin,
                                           size + pout);
                    });
                    std::printf("size 2^%d x %-5u pad in %-5llu out %-6llu copy %.3f\n", e, B, (unsigned long long)pin, (unsigned long long)pout,
                                2.0 * (double)size * B / (ms * 1e-3) / 8e12);
                    std::fflush(stdout);
                }
        }
    }
    return 0;
}
