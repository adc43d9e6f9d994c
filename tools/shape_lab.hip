// shape_lab.hip -- what does the SHAPE of a workgroup's memory traffic cost on MI355X, with no work at all?
// The BC7 kernels move 16 KiB per workgroup (256 lanes x 4 vectors): every load first, every store at the end, and run
// at the speed of a work-free copy of that shape (0.74-0.78 of the HBM peak) where the 4 KiB tiles of BC1-3 (one vector
// per lane) reach 0.85.  This program separates the two things that differ: bytes per workgroup and the order of the
// workgroup's loads and stores.  4 GiB in, 4 GiB out, contiguous; `nt` loads, `sc1 nt` stores like the product kernels.
//   K  vectors per lane (workgroup = 256 lanes x K x 16 bytes)
//   burst      all K loads, then all K stores                                  (the BC7 shape at K = 4)
//   pairs      load, store, load, store, ...                                   (K tiles of BC1-3 run one after the other)
//   halves     K/2 loads, K/2 stores, twice
//   barrier    burst with a __syncthreads between the loads and the stores     (what a sort in LDS forces)
//   hipcc --offload-arch=gfx950 -O3 -o tools/shape_lab tools/shape_lab.hip && ./tools/shape_lab
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 ld(const u32x4* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st(u32x4* p, u32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

enum { kBurst = 0, kPairs = 1, kHalves = 2, kBarrier = 3 };

template <int K, int ORDER>
__global__ void __launch_bounds__(256) copy_shape(const u32x4* __restrict__ src, u32x4* __restrict__ dst)
{
    const uint64_t base = (uint64_t)blockIdx.x * (256 * K) + threadIdx.x;
    u32x4 v[K];
    if constexpr (ORDER == kPairs) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            v[j] = ld(src + base + 256 * j);
            st(dst + base + 256 * j, v[j]);
        }
    } else if constexpr (ORDER == kHalves) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int j = h * K / 2; j < (h + 1) * K / 2; ++j) v[j] = ld(src + base + 256 * j);
#pragma unroll
            for (int j = h * K / 2; j < (h + 1) * K / 2; ++j) st(dst + base + 256 * j, v[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < K; ++j) v[j] = ld(src + base + 256 * j);
        if constexpr (ORDER == kBarrier) __syncthreads();
#pragma unroll
        for (int j = 0; j < K; ++j) st(dst + base + 256 * j, v[j]);
    }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// wgs_per_cu: occupancy cap through dynamic LDS nobody touches (160 KiB per CU / cap), 0 = the 8 the wave slots allow
template <int K, int ORDER>
int run(const char* name, const u32x4* a, u32x4* b, uint64_t bytes, hipEvent_t e0, hipEvent_t e1, int wgs_per_cu = 0)
{
    const uint64_t wgs = bytes / (256ull * K * 16);
    const unsigned lds = wgs_per_cu ? (160u << 10) / (unsigned)wgs_per_cu - 512 : 0;
    if (lds > (64u << 10))
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&copy_shape<K, ORDER>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copy_shape<K, ORDER>), dim3((unsigned)wgs), dim3(256), lds, 0, a, b);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((copy_shape<K, ORDER>), dim3((unsigned)wgs), dim3(256), lds, 0, a, b);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("K=%d %-8s wgs/CU %d  %6.3f ms  %.3f of 8 TB/s on 2*len\n", K, name, wgs_per_cu ? wgs_per_cu : 8, ms, 2.0 * bytes / (ms * 1e-3) / 8e12);
    return 0;
}

int main()
{
    const uint64_t bytes = 4ull << 30;
    u32x4 *a = nullptr, *b = nullptr;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMemset(a, 0x5A, bytes));
    CHECK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int pass = 0; pass < 1; ++pass) {
        if (run<1, kBurst>("burst", a, b, bytes, e0, e1)) return 1;
        if (run<2, kBurst>("burst", a, b, bytes, e0, e1)) return 1;
        if (run<2, kPairs>("pairs", a, b, bytes, e0, e1)) return 1;
        if (run<4, kBurst>("burst", a, b, bytes, e0, e1)) return 1;
        if (run<4, kBarrier>("barrier", a, b, bytes, e0, e1)) return 1;
        if (run<4, kHalves>("halves", a, b, bytes, e0, e1)) return 1;
        if (run<4, kPairs>("pairs", a, b, bytes, e0, e1)) return 1;
        if (run<8, kBurst>("burst", a, b, bytes, e0, e1)) return 1;
        if (run<8, kPairs>("pairs", a, b, bytes, e0, e1)) return 1;
    }
    // fewer workgroups per CU: is the big shape slow because too much is in flight?
    for (int cap : {6, 4, 3, 2, 1}) {
        if (run<1, kBurst>("burst", a, b, bytes, e0, e1, cap)) return 1;
        if (run<2, kBurst>("burst", a, b, bytes, e0, e1, cap)) return 1;
        if (run<4, kBurst>("burst", a, b, bytes, e0, e1, cap)) return 1;
        if (run<8, kBurst>("burst", a, b, bytes, e0, e1, cap)) return 1;
    }
    // a byte-for-byte check of the last variant's output
    uint64_t first = 0, last = 0;
    CHECK(hipMemcpy(&first, b, 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&last, reinterpret_cast<uint8_t*>(b) + bytes - 8, 8, hipMemcpyDeviceToHost));
    printf("check %s\n", first == 0x5A5A5A5A5A5A5A5Aull && last == 0x5A5A5A5A5A5A5A5Aull ? "ok" : "WRONG");
    return 0;
}
