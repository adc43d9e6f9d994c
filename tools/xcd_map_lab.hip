// xcd_map_lab.hip -- does it matter WHICH XCD touches which part of the address space?  Workgroups are dealt round robin
// over the eight XCDs (blockIdx % 8 labels the XCD); the identity tile order therefore gives XCD x every eighth tile.  This
// lab runs a read-only, a write-only and a copy kernel over 4 GiB with the tile handed to a workgroup chosen by a mapping:
//   identity | rotate r (tile's XCD label shifted by r) | xor m | chunk G (an XCD takes G consecutive tiles) | contiguous
// for tiles of 1 / 2 / 4 / 8 KiB.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/xcd_map_lab tools/xcd_map_lab.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
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

enum { kIdentity = 0, kRotate = 1, kXor = 2, kChunk = 3, kContiguous = 4 };

__device__ __forceinline__ uint32_t map_tile(uint32_t wg, uint32_t nwg, int mode, uint32_t p)
{
    const uint32_t xcd = wg & 7, slot = wg >> 3;
    switch (mode) {
    case kRotate: return slot * 8 + ((xcd + p) & 7);
    case kXor: return slot * 8 + (xcd ^ p);
    case kChunk: return (slot / p) * 8 * p + xcd * p + slot % p;   // nwg is a multiple of 8 * p here
    case kContiguous: return xcd * (nwg >> 3) + slot;
    default: return wg;
    }
}

__device__ __forceinline__ void store_wt(void* p, u32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// OP: 0 read, 1 write, 2 copy.  One 16-byte vector per lane; blockDim.x * 16 bytes per tile.
template <int OP>
__global__ void kern(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t* __restrict__ sink, int mode, uint32_t p)
{
    const uint64_t tile = map_tile(blockIdx.x, gridDim.x, mode, p);
    const uint64_t off = (tile * blockDim.x + threadIdx.x) * 16;
    u32x4 v = {threadIdx.x, 1, 2, 3};
    if (OP != 1)
        v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + off));
    if (OP != 0)
        store_wt(out + off, v);
    else if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u)
        sink[0] = 1;
}

__global__ void xcc_ids(uint32_t* out)
{
    if (threadIdx.x == 0) {
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        out[blockIdx.x] = id;
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
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return ms / steps;
}

int main(int argc, char** argv)
{
    const double gib = argc > 1 ? std::atof(argv[1]) : 4.0;
    const uint64_t bytes = (uint64_t)(gib * (1ull << 30));
    uint8_t *in, *out;
    uint32_t* sink;
    CHECK(hipMalloc(&in, bytes));
    CHECK(hipMalloc(&out, bytes));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMemset(in, 0x5A, bytes));
    CHECK(hipMemset(out, 0, bytes));
    hipLaunchKernelGGL(xcc_ids, dim3(32), dim3(64), 0, 0, sink);
    std::vector<uint32_t> ids(32);
    CHECK(hipMemcpy(ids.data(), sink, 32 * 4, hipMemcpyDeviceToHost));
    std::printf("XCC id of workgroups 0..31:");
    for (uint32_t v : ids) std::printf(" %u", v & 0xF);
    std::printf("\n");
    struct Map { const char* name; int mode; uint32_t p; };
    const Map maps[] = {{"identity", kIdentity, 0}, {"rotate 1", kRotate, 1}, {"rotate 2", kRotate, 2}, {"rotate 3", kRotate, 3},
                        {"rotate 4", kRotate, 4}, {"rotate 5", kRotate, 5}, {"rotate 6", kRotate, 6}, {"rotate 7", kRotate, 7},
                        {"xor 1", kXor, 1}, {"xor 2", kXor, 2}, {"xor 4", kXor, 4}, {"xor 7", kXor, 7},
                        {"chunk 2", kChunk, 2}, {"chunk 4", kChunk, 4}, {"chunk 16", kChunk, 16}, {"contiguous", kContiguous, 0}};
    const char* ops[] = {"read", "write", "copy"};
    for (int threads : {64, 128, 256, 512}) {
        const uint32_t nwg = (uint32_t)(bytes / ((uint64_t)threads * 16));
        for (const Map& m : maps) {
            double f[3];
            for (int op = 0; op < 3; ++op) {
                auto launch = [&] {
                    if (op == 0) hipLaunchKernelGGL(kern<0>, dim3(nwg), dim3(threads), 0, 0, in, out, sink, m.mode, m.p);
                    if (op == 1) hipLaunchKernelGGL(kern<1>, dim3(nwg), dim3(threads), 0, 0, in, out, sink, m.mode, m.p);
                    if (op == 2) hipLaunchKernelGGL(kern<2>, dim3(nwg), dim3(threads), 0, 0, in, out, sink, m.mode, m.p);
                };
                const double ms = time_ms(launch);
                f[op] = (double)bytes * (op == 2 ? 2 : 1) / (ms * 1e-3) / 8e12;
            }
            std::printf("tile %4d B  %-12s %s %.3f  %s %.3f  %s %.3f\n", threads * 16, m.name, ops[0], f[0], ops[1], f[1], ops[2], f[2]);
            std::fflush(stdout);
        }
    }
    return 0;
}
