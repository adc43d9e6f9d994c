// unaligned_lab.hip -- do 16-byte vector loads / stores at addresses that are NOT 16-byte aligned work on gfx950 under
// ROCm's default (unaligned) memory mode, and what do they cost?  A streaming copy, one 16-byte vector per lane, source
// and destination displaced by 0, 4, 8, 1, 2 bytes; checked byte for byte, timed with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -o tools/unaligned_lab tools/unaligned_lab.hip && ./tools/unaligned_lab
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) copy16(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t vectors)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= vectors)
        return;
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src + 16 * i) : "memory");
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst + 16 * i), "v"(v) : "memory");
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    const uint64_t bytes = 1ull << 30, vectors = bytes / 16;
    uint8_t *a = nullptr, *b = nullptr;
    CHECK(hipMalloc(&a, bytes + 256));
    CHECK(hipMalloc(&b, bytes + 256));
    std::vector<uint8_t> h(1 << 20), back(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 131 + (i >> 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int offs[][2] = {{0, 0}, {4, 0}, {0, 4}, {4, 4}, {8, 8}, {1, 0}, {0, 1}, {2, 2}, {1, 3}};
    for (auto& o : offs) {
        CHECK(hipMemset(b, 0xEE, bytes + 256));
        CHECK(hipMemcpy(a + o[0], h.data(), h.size(), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(copy16, dim3((unsigned)(vectors / 256)), dim3(256), 0, 0, a + o[0], b + o[1], vectors);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(back.data(), b + o[1], back.size(), hipMemcpyDeviceToHost));
        const bool ok = std::memcmp(h.data(), back.data(), h.size()) == 0;
        uint8_t guard[2] = {0, 0};
        if (o[1]) CHECK(hipMemcpy(guard, b + o[1] - 1, 1, hipMemcpyDeviceToHost)); else guard[0] = 0xEE;
        CHECK(hipMemcpy(guard + 1, b + o[1] + bytes, 1, hipMemcpyDeviceToHost));
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 10; ++r)
            hipLaunchKernelGGL(copy16, dim3((unsigned)(vectors / 256)), dim3(256), 0, 0, a + o[0], b + o[1], vectors);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("src +%d dst +%d: %s guards %s  %.3f of 8 TB/s\n", o[0], o[1], ok ? "exact" : "WRONG",
               guard[0] == 0xEE && guard[1] == 0xEE ? "intact" : "CLOBBERED", 2.0 * bytes * 10 / (ms * 1e-3) / 8e12);
        fflush(stdout);
    }
    return 0;
}
