// kernel_lab.hip -- A/B bench of BC1 default-mode (YCoCg var1 + split) kernel structures on one MI355X.
// Not part of the product: variants that win are folded into dxt-lossless-transform_amd/csrc/bcn_kernels.hip.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/kernel_lab tools/kernel_lab.hip
//   ./tools/kernel_lab [GiB=8] [reps=10]
//
// Every variant is timed interleaved (round-robin over variants, `reps` rounds) with HIP events and its output is
// compared on the device with the first forward variant (bit-exact).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../dxt-lossless-transform_amd/csrc/ycocg_swar.h"

using dxtlt::decorrelate2;
using dxtlt::recorrelate2;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

template <bool NT>
__device__ __forceinline__ u32x4 ld16(const void* p)
{
    if (NT) return __builtin_nontemporal_load((const u32x4*)p);
    return *(const u32x4*)p;
}
template <bool NT>
__device__ __forceinline__ void st16(void* p, u32x4 v)
{
    if (NT) __builtin_nontemporal_store(v, (u32x4*)p);
    else *(u32x4*)p = v;
}
template <bool NT>
__device__ __forceinline__ void st8(void* p, u32x2 v)
{
    if (NT) __builtin_nontemporal_store(v, (u32x2*)p);
    else *(u32x2*)p = v;
}
template <bool NT>
__device__ __forceinline__ void st4(void* p, uint32_t v)
{
    if (NT) __builtin_nontemporal_store(v, (uint32_t*)p);
    else *(uint32_t*)p = v;
}

// ---------------------------------------------------------------------------------------------------------
// 0. plain copy: the ceiling for "read len, write len"
// ---------------------------------------------------------------------------------------------------------
template <bool NT, int U>
__global__ void __launch_bounds__(256) copy_k(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t nvec)
{
    // each WG iteration moves U * 256 vectors (U KiB*4)
    const uint64_t per = (uint64_t)U * 256;
    for (uint64_t base = (uint64_t)blockIdx.x * per; base < nvec; base += (uint64_t)gridDim.x * per) {
        u32x4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = ld16<NT>(in + (base + threadIdx.x + 256 * j) * 16);
#pragma unroll
        for (int j = 0; j < U; ++j) st16<NT>(out + (base + threadIdx.x + 256 * j) * 16, v[j]);
    }
}

// copy into three output streams with the BC1 split proportions (2:2:4 of every 8 bytes) -- same byte counts and
// store shapes as the transform, no arithmetic, no LDS: isolates the cost of 3 write streams.
template <bool NT>
__global__ void __launch_bounds__(256) copy3_k(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t ntiles,
                                               uint64_t N)
{
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ld16<NT>(in + tile * 16384 + (uint64_t)(threadIdx.x + 256 * j) * 16);
        st16<NT>(out + 0 * N + tile * 4096 + threadIdx.x * 16, v[0]);
        st16<NT>(out + 2 * N + tile * 4096 + threadIdx.x * 16, v[1]);
        st16<NT>(out + 4 * N + tile * 8192 + threadIdx.x * 16, v[2]);
        st16<NT>(out + 4 * N + tile * 8192 + 4096 + threadIdx.x * 16, v[3]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// 1. workgroup tile through LDS (the product structure).  PF = prefetch next tile before the store phase.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bc1_scatter(uint8_t* lds, int T2 /*2*T bytes = c1 offset*/, int u, u32x4 q)
{
    const uint32_t ca = decorrelate2<1>(q.x), cb = decorrelate2<1>(q.z);
    *(uint32_t*)(lds + 4 * u) = (ca & 0xFFFFu) | (cb << 16);
    *(uint32_t*)(lds + T2 + 4 * u) = (ca >> 16) | (cb & 0xFFFF0000u);
    *(u32x2*)(lds + 2 * T2 + 8 * u) = u32x2{q.y, q.w};
}

template <bool NT, bool PF, int VECS>
__global__ void __launch_bounds__(256) fwd_wg(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t ntiles,
                                              uint64_t N)
{
    constexpr int TB = VECS * 4096;     // tile bytes
    constexpr int T = TB / 8;           // blocks per tile
    __shared__ __attribute__((aligned(16))) uint8_t lds[TB];
    const int t = threadIdx.x;
    uint64_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    u32x4 q[VECS];
#pragma unroll
    for (int j = 0; j < VECS; ++j) q[j] = ld16<NT>(in + tile * TB + (uint64_t)(t + 256 * j) * 16);
    for (;;) {
#pragma unroll
        for (int j = 0; j < VECS; ++j) bc1_scatter(lds, 2 * T, t + 256 * j, q[j]);
        __syncthreads();
        const uint64_t next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        if (PF && has_next) {
#pragma unroll
            for (int j = 0; j < VECS; ++j) q[j] = ld16<NT>(in + next * TB + (uint64_t)(t + 256 * j) * 16);
        }
        // image: c0 [0,2T) c1 [2T,4T) idx [4T,8T); 2T = VECS KiB
#pragma unroll
        for (int k = 0; k < VECS; ++k) {
            const int o = (t + 256 * k) * 16;  // image byte
            const u32x4 v = *(u32x4*)(lds + o);
            uint64_t g;
            if (o < 2 * T) g = 0 * N + tile * (2 * T) + o;
            else if (o < 4 * T) g = 2 * N + tile * (2 * T) + (o - 2 * T);
            else g = 4 * N + tile * (4 * T) + (o - 4 * T);
            st16<NT>(out + g, v);
        }
        if (!has_next) break;
        if (!PF) {
#pragma unroll
            for (int j = 0; j < VECS; ++j) q[j] = ld16<NT>(in + next * TB + (uint64_t)(t + 256 * j) * 16);
        }
        __syncthreads();
        tile = next;
    }
}

// ---------------------------------------------------------------------------------------------------------
// 2. wave-private tile: each wave owns 4 KiB (512 blocks), its own LDS slice, no workgroup barrier.
//    stores: c0 1 KiB (one dwordx4 per lane), c1 1 KiB, idx 2 KiB.
// ---------------------------------------------------------------------------------------------------------
template <bool NT, bool PF>
__global__ void __launch_bounds__(256) fwd_wave(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                uint64_t nwtiles, uint64_t N)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[4 * 4096];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint8_t* lds = lds_all + wave * 4096;
    const uint64_t stride = (uint64_t)gridDim.x * 4;
    uint64_t wt = (uint64_t)blockIdx.x * 4 + wave;  // wave-tile index (4 KiB of input)
    if (wt >= nwtiles) return;
    u32x4 q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = ld16<NT>(in + wt * 4096 + (uint64_t)(lane + 64 * j) * 16);
    for (;;) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bc1_scatter(lds, 1024, lane + 64 * j, q[j]);
        __builtin_amdgcn_wave_barrier();
        const uint64_t next = wt + stride;
        const bool has_next = next < nwtiles;
        if (PF && has_next) {
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] = ld16<NT>(in + next * 4096 + (uint64_t)(lane + 64 * j) * 16);
        }
        const u32x4 v0 = *(u32x4*)(lds + lane * 16);
        const u32x4 v1 = *(u32x4*)(lds + 1024 + lane * 16);
        const u32x4 v2 = *(u32x4*)(lds + 2048 + lane * 16);
        const u32x4 v3 = *(u32x4*)(lds + 3072 + lane * 16);
        st16<NT>(out + 0 * N + wt * 1024 + lane * 16, v0);
        st16<NT>(out + 2 * N + wt * 1024 + lane * 16, v1);
        st16<NT>(out + 4 * N + wt * 2048 + lane * 16, v2);
        st16<NT>(out + 4 * N + wt * 2048 + 1024 + lane * 16, v3);
        if (!has_next) break;
        if (!PF) {
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] = ld16<NT>(in + next * 4096 + (uint64_t)(lane + 64 * j) * 16);
        }
        __builtin_amdgcn_wave_barrier();
        wt = next;
    }
}

// ---------------------------------------------------------------------------------------------------------
// 3. no LDS: lane keeps its two blocks, stores 4 B (c0 pair), 4 B (c1 pair), 8 B (idx pair) per load
// ---------------------------------------------------------------------------------------------------------
template <bool NT, int U>
__global__ void __launch_bounds__(256) fwd_direct(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t nvec,
                                                  uint64_t N)
{
    const uint64_t per = (uint64_t)U * 256;
    for (uint64_t base = (uint64_t)blockIdx.x * per; base < nvec; base += (uint64_t)gridDim.x * per) {
        u32x4 q[U];
#pragma unroll
        for (int j = 0; j < U; ++j) q[j] = ld16<NT>(in + (base + threadIdx.x + 256 * j) * 16);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint64_t u = base + threadIdx.x + 256 * j;  // vector index = block pair index
            const uint32_t ca = decorrelate2<1>(q[j].x), cb = decorrelate2<1>(q[j].z);
            st4<NT>(out + 0 * N + 4 * u, (ca & 0xFFFFu) | (cb << 16));
            st4<NT>(out + 2 * N + 4 * u, (ca >> 16) | (cb & 0xFFFF0000u));
            st8<NT>(out + 4 * N + 8 * u, u32x2{q[j].y, q[j].w});
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// inverse, workgroup tile through LDS
// ---------------------------------------------------------------------------------------------------------
template <bool NT, bool PF, int VECS>
__global__ void __launch_bounds__(256) inv_wg(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t ntiles,
                                              uint64_t N)
{
    constexpr int TB = VECS * 4096;
    constexpr int T = TB / 8;
    __shared__ __attribute__((aligned(16))) uint8_t lds[TB];
    const int t = threadIdx.x;
    uint64_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    auto src = [&](uint64_t tl, int k) -> const uint8_t* {
        const int o = (t + 256 * k) * 16;
        uint64_t g;
        if (o < 2 * T) g = 0 * N + tl * (2 * T) + o;
        else if (o < 4 * T) g = 2 * N + tl * (2 * T) + (o - 2 * T);
        else g = 4 * N + tl * (4 * T) + (o - 4 * T);
        return in + g;
    };
    u32x4 v[VECS];
#pragma unroll
    for (int k = 0; k < VECS; ++k) v[k] = ld16<NT>(src(tile, k));
    for (;;) {
#pragma unroll
        for (int k = 0; k < VECS; ++k) *(u32x4*)(lds + (t + 256 * k) * 16) = v[k];
        __syncthreads();
        const uint64_t next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        if (PF && has_next) {
#pragma unroll
            for (int k = 0; k < VECS; ++k) v[k] = ld16<NT>(src(next, k));
        }
#pragma unroll
        for (int j = 0; j < VECS; ++j) {
            const int u = t + 256 * j;
            const uint32_t c0 = *(uint32_t*)(lds + 4 * u), c1 = *(uint32_t*)(lds + 2 * T + 4 * u);
            const u32x2 idx = *(u32x2*)(lds + 4 * T + 8 * u);
            u32x4 q;
            q.x = recorrelate2<1>((c0 & 0xFFFFu) | (c1 << 16));
            q.y = idx.x;
            q.z = recorrelate2<1>((c0 >> 16) | (c1 & 0xFFFF0000u));
            q.w = idx.y;
            st16<NT>(out + tile * TB + (uint64_t)u * 16, q);
        }
        if (!has_next) break;
        if (!PF) {
#pragma unroll
            for (int k = 0; k < VECS; ++k) v[k] = ld16<NT>(src(next, k));
        }
        __syncthreads();
        tile = next;
    }
}

// inverse without LDS: 4 B + 4 B + 8 B loads per lane
template <bool NT, int U>
__global__ void __launch_bounds__(256) inv_direct(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t nvec,
                                                  uint64_t N)
{
    const uint64_t per = (uint64_t)U * 256;
    for (uint64_t base = (uint64_t)blockIdx.x * per; base < nvec; base += (uint64_t)gridDim.x * per) {
        uint32_t c0[U], c1[U];
        u32x2 idx[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint64_t u = base + threadIdx.x + 256 * j;
            c0[j] = *(const uint32_t*)(in + 4 * u);
            c1[j] = *(const uint32_t*)(in + 2 * N + 4 * u);
            idx[j] = *(const u32x2*)(in + 4 * N + 8 * u);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint64_t u = base + threadIdx.x + 256 * j;
            u32x4 q;
            q.x = recorrelate2<1>((c0[j] & 0xFFFFu) | (c1[j] << 16));
            q.y = idx[j].x;
            q.z = recorrelate2<1>((c0[j] >> 16) | (c1[j] & 0xFFFF0000u));
            q.w = idx[j].y;
            st16<NT>(out + u * 16, q);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// lab2: one-shot / blocked variants.  THREADS per workgroup, VECS 16-B vectors per lane per tile; a workgroup
// handles `tpw` consecutive tiles (tpw = 1 -> one tile per workgroup, huge grid).
// ---------------------------------------------------------------------------------------------------------
template <bool NT, int THREADS, int U>
__global__ void __launch_bounds__(THREADS) copy_blk(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t nvec)
{
    const uint64_t base = (uint64_t)blockIdx.x * (THREADS * U);
    u32x4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j)
        if (base + threadIdx.x + THREADS * j < nvec) v[j] = ld16<NT>(in + (base + threadIdx.x + THREADS * j) * 16);
#pragma unroll
    for (int j = 0; j < U; ++j)
        if (base + threadIdx.x + THREADS * j < nvec) st16<NT>(out + (base + threadIdx.x + THREADS * j) * 16, v[j]);
}

template <bool NT, int THREADS, int VECS>
__global__ void __launch_bounds__(THREADS) fwd_blk(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                   uint64_t ntiles, uint64_t N, int tpw)
{
    constexpr int TB = VECS * THREADS * 16;
    constexpr int T = TB / 8;
    __shared__ __attribute__((aligned(16))) uint8_t lds[TB];
    const int t = threadIdx.x;
    uint64_t tile = (uint64_t)blockIdx.x * tpw;
    const uint64_t end = tile + tpw < ntiles ? tile + tpw : ntiles;
    for (; tile < end; ++tile) {
        u32x4 q[VECS];
#pragma unroll
        for (int j = 0; j < VECS; ++j) q[j] = ld16<NT>(in + tile * TB + (uint64_t)(t + THREADS * j) * 16);
#pragma unroll
        for (int j = 0; j < VECS; ++j) bc1_scatter(lds, 2 * T, t + THREADS * j, q[j]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < VECS; ++k) {
            const int o = (t + THREADS * k) * 16;
            const u32x4 v = *(u32x4*)(lds + o);
            uint64_t g;
            if (o < 2 * T) g = 0 * N + tile * (2 * T) + o;
            else if (o < 4 * T) g = 2 * N + tile * (2 * T) + (o - 2 * T);
            else g = 4 * N + tile * (4 * T) + (o - 4 * T);
            st16<NT>(out + g, v);
        }
        if (tile + 1 < end) __syncthreads();
    }
}

template <bool NT, int THREADS, int VECS>
__global__ void __launch_bounds__(THREADS) inv_blk(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                   uint64_t ntiles, uint64_t N, int tpw)
{
    constexpr int TB = VECS * THREADS * 16;
    constexpr int T = TB / 8;
    __shared__ __attribute__((aligned(16))) uint8_t lds[TB];
    const int t = threadIdx.x;
    uint64_t tile = (uint64_t)blockIdx.x * tpw;
    const uint64_t end = tile + tpw < ntiles ? tile + tpw : ntiles;
    for (; tile < end; ++tile) {
        u32x4 v[VECS];
#pragma unroll
        for (int k = 0; k < VECS; ++k) {
            const int o = (t + THREADS * k) * 16;
            uint64_t g;
            if (o < 2 * T) g = 0 * N + tile * (2 * T) + o;
            else if (o < 4 * T) g = 2 * N + tile * (2 * T) + (o - 2 * T);
            else g = 4 * N + tile * (4 * T) + (o - 4 * T);
            v[k] = ld16<NT>(in + g);
        }
#pragma unroll
        for (int k = 0; k < VECS; ++k) *(u32x4*)(lds + (t + THREADS * k) * 16) = v[k];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < VECS; ++j) {
            const int u = t + THREADS * j;
            const uint32_t c0 = *(uint32_t*)(lds + 4 * u), c1 = *(uint32_t*)(lds + 2 * T + 4 * u);
            const u32x2 idx = *(u32x2*)(lds + 4 * T + 8 * u);
            u32x4 q;
            q.x = recorrelate2<1>((c0 & 0xFFFFu) | (c1 << 16));
            q.y = idx.x;
            q.z = recorrelate2<1>((c0 >> 16) | (c1 & 0xFFFF0000u));
            q.w = idx.y;
            st16<NT>(out + tile * TB + (uint64_t)u * 16, q);
        }
        if (tile + 1 < end) __syncthreads();
    }
}

// no LDS, one shot
template <bool NT, int THREADS, int U>
__global__ void __launch_bounds__(THREADS) fwd_direct_blk(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                          uint64_t nvec, uint64_t N)
{
    const uint64_t base = (uint64_t)blockIdx.x * (THREADS * U);
    u32x4 q[U];
#pragma unroll
    for (int j = 0; j < U; ++j) q[j] = ld16<NT>(in + (base + threadIdx.x + THREADS * j) * 16);
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const uint64_t u = base + threadIdx.x + THREADS * j;
        const uint32_t ca = decorrelate2<1>(q[j].x), cb = decorrelate2<1>(q[j].z);
        st4<NT>(out + 0 * N + 4 * u, (ca & 0xFFFFu) | (cb << 16));
        st4<NT>(out + 2 * N + 4 * u, (ca >> 16) | (cb & 0xFFFF0000u));
        st8<NT>(out + 4 * N + 8 * u, u32x2{q[j].y, q[j].w});
    }
}

template <bool NT, int THREADS, int U>
__global__ void __launch_bounds__(THREADS) inv_direct_blk(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                          uint64_t nvec, uint64_t N)
{
    const uint64_t base = (uint64_t)blockIdx.x * (THREADS * U);
    uint32_t c0[U], c1[U];
    u32x2 idx[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const uint64_t u = base + threadIdx.x + THREADS * j;
        if (NT) {
            c0[j] = __builtin_nontemporal_load((const uint32_t*)(in + 4 * u));
            c1[j] = __builtin_nontemporal_load((const uint32_t*)(in + 2 * N + 4 * u));
            idx[j] = __builtin_nontemporal_load((const u32x2*)(in + 4 * N + 8 * u));
        } else {
            c0[j] = *(const uint32_t*)(in + 4 * u);
            c1[j] = *(const uint32_t*)(in + 2 * N + 4 * u);
            idx[j] = *(const u32x2*)(in + 4 * N + 8 * u);
        }
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const uint64_t u = base + threadIdx.x + THREADS * j;
        u32x4 q;
        q.x = recorrelate2<1>((c0[j] & 0xFFFFu) | (c1[j] << 16));
        q.y = idx[j].x;
        q.z = recorrelate2<1>((c0[j] >> 16) | (c1[j] & 0xFFFF0000u));
        q.w = idx[j].y;
        st16<NT>(out + u * 16, q);
    }
}

// inverse, one wave per workgroup, stream slices fetched straight into the LDS image by LDS-DMA
// (global_load_lds_dwordx4: per-lane global source, LDS destination = wave base + lane*16), no VGPR round trip
template <int NTAUX>
__global__ void __launch_bounds__(64) inv_dma64(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t ntiles,
                                                uint64_t N)
{
    constexpr int TB = 1024, T = 128;
    __shared__ __attribute__((aligned(16))) uint8_t lds[TB];
    const int t = threadIdx.x;
    const uint64_t tile = blockIdx.x;
    const int o = t * 16;
    uint64_t g;
    if (o < 2 * T) g = 0 * N + tile * (2 * T) + o;
    else if (o < 4 * T) g = 2 * N + tile * (2 * T) + (o - 2 * T);
    else g = 4 * N + tile * (4 * T) + (o - 4 * T);
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(in + g),
                                     (void __attribute__((address_space(3)))*)lds, 16, 0, NTAUX);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int u = t;
    const uint32_t c0 = *(uint32_t*)(lds + 4 * u), c1 = *(uint32_t*)(lds + 2 * T + 4 * u);
    const u32x2 idx = *(u32x2*)(lds + 4 * T + 8 * u);
    u32x4 q;
    q.x = recorrelate2<1>((c0 & 0xFFFFu) | (c1 << 16));
    q.y = idx.x;
    q.z = recorrelate2<1>((c0 >> 16) | (c1 & 0xFFFF0000u));
    q.w = idx.y;
    st16<true>(out + tile * TB + (uint64_t)u * 16, q);
}

// cache-policy experiment: one-shot copy / forward with the store (and load) policy chosen by inline asm
// POLICY: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0 sc1 nt
template <int POLICY>
__device__ __forceinline__ void st16_policy(void* p, u32x4 v)
{
    if (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    if (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    if (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    if (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    if (POLICY == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <int POLICY>
__device__ __forceinline__ u32x4 ld16_policy(const void* p)
{
    u32x4 v;
    if (POLICY == 0) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int LP, int SP>
__global__ void __launch_bounds__(128) fwd_policy(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t ntiles,
                                                  uint64_t N)
{
    constexpr int TB = 2048, T = 256;
    __shared__ __attribute__((aligned(16))) uint8_t lds[TB];
    const int t = threadIdx.x;
    const uint64_t tile = blockIdx.x;
    const u32x4 q = ld16_policy<LP>(in + tile * TB + (uint64_t)t * 16);
    bc1_scatter(lds, 2 * T, t, q);
    __syncthreads();
    const int o = t * 16;
    const u32x4 v = *(u32x4*)(lds + o);
    uint64_t g;
    if (o < 2 * T) g = 0 * N + tile * (2 * T) + o;
    else if (o < 4 * T) g = 2 * N + tile * (2 * T) + (o - 2 * T);
    else g = 4 * N + tile * (4 * T) + (o - 4 * T);
    st16_policy<SP>(out + g, v);
}

// plain one-shot copy with the same load / store policies: the ceiling for "read len, write len" under that policy
template <int LP, int SP, int THREADS>
__global__ void __launch_bounds__(THREADS) copy_policy(const uint8_t* __restrict__ in, uint8_t* __restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    st16_policy<SP>(out + i * 16, ld16_policy<LP>(in + i * 16));
}

__global__ void fill_k(uint64_t* p, uint64_t n, uint64_t seed)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = z ^ (z >> 31);
    }
}

__global__ void diff_k(const uint64_t* a, const uint64_t* b, uint64_t n, unsigned long long* cnt)
{
    unsigned long long local = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        local += a[i] != b[i];
    if (local) atomicAdd(cnt, local);
}

struct Variant {
    std::string name;
    int kind;  // 0 copy-like (no check), 1 forward, 2 inverse
    std::function<void()> launch;
    std::vector<float> ms;
};

int main(int argc, char** argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 8.0;
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    uint64_t len = (uint64_t)(gib * (1ull << 30));
    len -= len % (1 << 16);
    const uint64_t N = len / 8, nvec = len / 16;
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("len %.2f GiB, %d CUs, reps %d\n", len / 1073741824.0, cus, reps);

    uint8_t *x, *y, *yref, *z;
    CK(hipMalloc(&x, len));
    CK(hipMalloc(&y, len));
    CK(hipMalloc(&yref, len));
    CK(hipMalloc(&z, len));
    unsigned long long* cnt;
    CK(hipMalloc(&cnt, 8));
    fill_k<<<cus * 16, 256>>>((uint64_t*)x, len / 8, 0x0BC10002);
    CK(hipDeviceSynchronize());

    std::vector<Variant> vs;
    auto add = [&](std::string n, int kind, std::function<void()> f) { vs.push_back({n, kind, f, {}}); };

    // reference forward result
    fwd_wg<true, true, 4><<<cus * 8, 256>>>(x, yref, len / 16384, N);
    CK(hipDeviceSynchronize());





#define ADD_COPY(LP, SP, TH) add("copy_policy T" #TH " load" #LP " store" #SP, 0, [=] { copy_policy<LP, SP, TH><<<(unsigned)(len / (16 * TH)), TH>>>(x, y); })
    ADD_COPY(1, 1, 128);
    ADD_COPY(1, 4, 128);
    ADD_COPY(1, 4, 256);
    ADD_COPY(1, 4, 64);
    ADD_COPY(0, 0, 128);
#define ADD_POL(LP, SP) add("fwd_policy T128 load" #LP " store" #SP, 1, [=] { fwd_policy<LP, SP><<<(unsigned)(len / 2048), 128>>>(x, y, len / 2048, N); })
    ADD_POL(1, 1); ADD_POL(0, 0); ADD_POL(1, 0); ADD_POL(0, 1); ADD_POL(1, 2); ADD_POL(1, 3); ADD_POL(1, 4); ADD_POL(1, 5);
    ADD_POL(2, 1); ADD_POL(4, 1); ADD_POL(3, 1); ADD_POL(5, 5);
#define ADD_FWD(NT, TH, V, TPW) add("fwd_blk " #NT " T" #TH " V" #V " tpw" #TPW, 1, [=] { \
        const uint64_t nt = len / (V * TH * 16); fwd_blk<NT, TH, V><<<(unsigned)((nt + TPW - 1) / TPW), TH>>>(x, y, nt, N, TPW); })
    ADD_FWD(true, 128, 1, 1);

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    // correctness first
    for (auto& v : vs) {
        if (v.kind == 0) continue;
        uint8_t* outp = v.kind == 1 ? y : z;
        CK(hipMemset(outp, 0xEE, len));
        v.launch();
        CK(hipGetLastError());
        CK(hipMemset(cnt, 0, 8));
        diff_k<<<cus * 8, 256>>>((const uint64_t*)outp, (const uint64_t*)(v.kind == 1 ? yref : x), len / 8, cnt);
        unsigned long long h = 0;
        CK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost));
        if (h) printf("MISMATCH %-28s %llu qwords differ\n", v.name.c_str(), h);
    }

    for (int r = 0; r < reps + 1; ++r) {
        for (auto& v : vs) {
            CK(hipEventRecord(e0, 0));
            v.launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) v.ms.push_back(ms);
        }
    }
    printf("%-30s %9s %9s %9s  %s\n", "variant", "med ms", "min ms", "GB/s(med)", "frac of 8 TB/s");
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        const double gbps = 2.0 * len / (med * 1e-3) / 1e9;
        printf("%-30s %9.4f %9.4f %9.1f  %.3f\n", v.name.c_str(), med, mn, gbps, gbps / 8000.0);
    }
    return 0;
}
