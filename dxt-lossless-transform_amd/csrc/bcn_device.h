// bcn_device.h -- device code of the BCn block-transform kernels (gfx950): the LDS image and the aligned / halo / shifted / edge
// tiles, shared by the single-buffer kernels (bcn_kernels.hip) and the batch kernels (batch_kernels.hip).  The design notes
// are at the top of bcn_kernels.hip.
//
// -DDXTLT_EXPERIMENTS (a SIDE build, never the shipped library: `DXTLT_EXTRA_HIPCC_FLAGS=-DDXTLT_EXPERIMENTS tools/ab_build_rev.sh
// WORKTREE exp`, run through DXTLT_LIB_PATH) adds what the measurements in profiles/ were taken with and the product does not
// need: the element-granular kernel, the first form of the forward shifted tiles, run-time store policies and tile orders, and
// a wrong-output timing switch (Shifts::skip_partial).  Without the flag none of that is compiled: the kernels carry no switch
// that is not a property of the data.  The two retired kernel families live in bcn_experiments.h, included at the end of this file
// in that build only.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <type_traits>

#include "bc1_normalize.h"
#include "bcn_launch.h"
#include "streaming_store.h"
#include "ycocg_swar.h"

namespace dxtlt {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;        // element-granular / fill kernels
// Default tile workgroup size (tile = threads * 16 bytes) per format and direction, chosen by measurement on the
// 8 GiB workloads with the final store policy (profiles/r01_q_tile_threads_sweep_sc1_stores.txt): fraction of peak, fwd / inv
//   BC1  64: .82/.83   128: .85/.84   256: .84/.82   512: .81/.78
//   BC2  64: .80/.81   128: .83/.82   256: .84/.82   512: .82/.79
//   BC3  64: .71/.72   128: .80/.78   256: .84/.81   512: .82/.80
// (with plain `nt` stores the optimum was smaller for the inverse: profiles/r01_g_tile_threads_sweep.txt)
constexpr int default_tile_threads(int fmt, bool inverse)
{
    (void)inverse;
    return fmt == kBc1 ? 128 : 256;
}

#ifndef DXTLT_NONTEMPORAL
#define DXTLT_NONTEMPORAL 1
#endif

#ifdef DXTLT_WG_TIMING
// EXPERIMENT build only (tools/wg_timing_probe.py, tools/wg_phase_single.py; built by DXTLT_EXTRA_HIPCC_FLAGS=-DDXTLT_WG_TIMING
// tools/ab_build_rev.sh WORKTREE timing): phase marks of lane 0 of every workgroup, 100 MHz ticks (low 32 bits).  Slot 0: kernel
// start (single-buffer aligned kernels), 1: the tile begins, 2: its loads have arrived, 3: behind the barrier, 4: stores issued.
__device__ uint32_t g_wg_marks[8 << 20];
__device__ __forceinline__ void wg_mark(int i)
{
    if (threadIdx.x == 0 && blockIdx.x < (1u << 20))
        g_wg_marks[8 * blockIdx.x + i] = (uint32_t)__builtin_amdgcn_s_memrealtime();
}
#define WG_MARK(i) wg_mark(i)
#define WG_MARK_LOADS_DONE(i)                                  \
    do {                                                       \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       \
        wg_mark(i);                                            \
    } while (0)
#else
#define WG_MARK(i)
#define WG_MARK_LOADS_DONE(i)
#endif

__device__ __forceinline__ u32x4 gload16(const void* p)
{
#if DXTLT_NONTEMPORAL
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
#else
    return *reinterpret_cast<const u32x4*>(p);
#endif
}

// Streaming 16-byte store: streaming_store.h (policy measurements and the wait states its inline asm needs)
__device__ __forceinline__ void gstore16(void* p, u32x4 v)
{
#if DXTLT_NONTEMPORAL
    store_streaming16(p, v);
#else
    *reinterpret_cast<u32x4*>(p) = v;
#endif
}

// AoS-side store of the inverse kernels: the write-through streaming store when the block array is 16-byte aligned (a
// tile's 4 KiB are whole lines); plain `nt` when it is not -- every tile then shares its first and last line with its
// neighbours, and write-through on shared lines is what collapsed to 0.39-0.50 in round 1
__device__ __forceinline__ void gstore16_aos(uint8_t* aos_base, void* p, u32x4 v)
{
    if ((reinterpret_cast<uintptr_t>(aos_base) & 15) == 0)   // uniform
        gstore16(p, v);
    else
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// f(std::integral_constant<int, W>) for the calling wave's number W (uniform), W < WAVES: an if / else-if chain with the last wave
// as the default, like a switch -- mutually exclusive paths, so that the waits one path needs for its loads are not merged into
// the others (a run of independent `if (w == W)` made the compiler put an s_waitcnt vmcnt(0) between wave 0's two loads of the
// inverse shifted tile: tests/test_isa_invariants.py)
template <int W, int WAVES, typename F>
__device__ __forceinline__ void for_this_wave_from(int w, F&& f)
{
    if constexpr (W + 1 >= WAVES) {
        f(std::integral_constant<int, W>{});
    } else {
        if (w == W)
            f(std::integral_constant<int, W>{});
        else
            for_this_wave_from<W + 1, WAVES>(w, f);
    }
}
template <int WAVES, typename F>
__device__ __forceinline__ void for_this_wave(int t, F&& f)
{
    for_this_wave_from<0, WAVES>(__builtin_amdgcn_readfirstlane(t >> 6), f);
}

__host__ __device__ constexpr int fmt_block(int fmt) { return fmt == kBc1 ? 8 : 16; }
// one 16-byte vector per lane: a THREADS-wide workgroup owns THREADS*16 bytes of blocks
__host__ __device__ constexpr int tile_blocks(int fmt, int threads) { return threads * 16 / fmt_block(fmt); }

// Byte offset, inside the whole transformed buffer, of LDS-image byte `o` (a lane's 16-byte segment) for
// the tile whose first block is `blk0` (global block index).  Stream boundaries inside the image are
// multiples of 256 bytes, so a 16-byte segment never straddles two streams.
template <int FMT, bool SA, bool SC, int T>
__device__ __forceinline__ uint64_t soa_offset_of_image_byte(int o, uint64_t total_blocks, uint64_t blk0)
{
    constexpr Streams S = make_streams(FMT, SA, SC);
    uint64_t r = 0;
#pragma unroll
    for (int s = 0; s < S.n; ++s) {
        const int lo = S.off[s] * T;
        const int hi = lo + S.width[s] * T;
        if (o >= lo && o < hi)
            r = (uint64_t)S.off[s] * total_blocks + (uint64_t)S.width[s] * blk0 + (uint64_t)(o - lo);
    }
    return r;
}

// The same for a lane of wave W of the workgroup (W known at compile time: for_this_wave).  A wave's 1 KiB of the image meets at
// most three streams (BC3 with split alphas, wave 0) and usually one, so the per-lane select chain over all streams -- six
// compares and six 64-bit multiply-adds per lane, 40 % of the aligned inverse kernel's instructions, all in front of its load --
// shrinks to the streams the wave can meet, and the stream bases stay on the scalar unit.
#ifndef DXTLT_WAVE_OFFSETS
#define DXTLT_WAVE_OFFSETS 1   // 0: the per-lane select chain of rounds 1-4 (A/B: profiles/r05_wave_offsets.txt)
#endif
template <int FMT, bool SA, bool SC, int T, int W>
__device__ __forceinline__ uint64_t soa_offset_of_image_byte_in_wave(int o, uint64_t total_blocks, uint64_t blk0)
{
    constexpr Streams S = make_streams(FMT, SA, SC);
    constexpr int wave_lo = W * 1024, wave_hi = wave_lo + 1024;
    uint64_t r = 0;
    static_for<0, S.n>([&](auto si) {
        constexpr int s = decltype(si)::value;
        constexpr int lo = S.off[s] * T;
        constexpr int hi = lo + S.width[s] * T;
        if constexpr (lo < wave_hi && hi > wave_lo) {
            constexpr bool only = lo <= wave_lo && hi >= wave_hi;   // the whole wave sits in this stream
            const uint64_t base = (uint64_t)S.off[s] * total_blocks + (uint64_t)S.width[s] * blk0;   // uniform
            if (only || (o >= lo && o < hi))
                r = base + (uint64_t)(uint32_t)(o - lo);
        }
    });
    return r;
}

template <int FMT, bool SA, bool SC, int THREADS>
__device__ __forceinline__ uint64_t soa_offset_of_lane(int t, uint64_t total_blocks, uint64_t blk0)
{
    constexpr int T = tile_blocks(FMT, THREADS);
#if DXTLT_WAVE_OFFSETS
    uint64_t o = 0;
    for_this_wave<THREADS / 64>(t, [&](auto wi) {
        o = soa_offset_of_image_byte_in_wave<FMT, SA, SC, T, decltype(wi)::value>(t * 16, total_blocks, blk0);
    });
    return o;
#else
    return soa_offset_of_image_byte<FMT, SA, SC, T>(t * 16, total_blocks, blk0);
#endif
}

// ------------------------------------------------------------------------------------------------
// LDS image <-> block registers.  `u` is the index of the lane's 16-byte vector inside the tile.
// BC1: the vector holds blocks 2u and 2u+1; BC2/BC3: block u.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T& lds_at(uint8_t* lds, int byte_off)
{
    return *reinterpret_cast<T*>(lds + byte_off);
}

// volatile halfword view in the LDS address space (a plain volatile pointer would become flat_* instructions)
typedef volatile uint16_t __attribute__((address_space(3))) lds_halfword;
__device__ __forceinline__ lds_halfword* lds_halfwords(uint8_t* lds, int byte_off)
{
    return (lds_halfword*)(lds + byte_off);
}

// The 6-byte alpha-index record of a BC3 block (bytes 2..7 of the block) in the LDS image: records are 6 bytes apart, so a
// record's address is a multiple of 4 for every other block and 2 further for the blocks in between.  Round 1 moved it as three
// halfwords (three DS instructions per lane and direction); here it is TWO: an aligned dword and a halfword, the order picked
// by the address's bit 1 with selects (no divergent branch) going in, and one ds_read2_b32 of the two aligned dwords that
// cover the record plus a funnel shift (v_alignbit) coming out.  Every access is naturally aligned (DS accesses at addresses
// that are not multiples of their width run several times slower on gfx950, profiles/r01_s), and the pointers are volatile /
// declared 4-byte aligned so that clang neither fuses them into such accesses nor widens the pair into a ds_read_b64.
// Reference for the bytes: bc3/src/transform/standard/transform/portable32.rs:46-64 (alpha indices = block bytes 2..7, verbatim).
#ifndef DXTLT_BC3_RECORD6
#define DXTLT_BC3_RECORD6 1   // 0: the three-halfword form (A/B: profiles/r05_bc3_record6.txt)
#endif
typedef uint32_t u32x2_align4 __attribute__((ext_vector_type(2), aligned(4)));
typedef volatile uint32_t __attribute__((address_space(3))) lds_dword;

// `addr` even; lo16 = record bytes 0..1 (low half), hi32 = record bytes 2..5
__device__ __forceinline__ void lds_put_record6(uint8_t* lds, int addr, uint32_t lo16, uint32_t hi32)
{
#if DXTLT_BC3_RECORD6
    const bool odd = (addr & 2) != 0;
    const int a32 = odd ? addr + 2 : addr;
    const int a16 = odd ? addr : addr + 4;
    const uint32_t v32 = odd ? hi32 : (lo16 & 0xFFFFu) | (hi32 << 16);
    const uint32_t v16 = odd ? lo16 : hi32 >> 16;
    *(lds_dword*)(lds + a32) = v32;
    *lds_halfwords(lds, a16) = (uint16_t)v16;
#else
    lds_halfwords(lds, addr)[0] = (uint16_t)lo16;
    lds_halfwords(lds, addr)[1] = (uint16_t)hi32;
    lds_halfwords(lds, addr)[2] = (uint16_t)(hi32 >> 16);
#endif
}

__device__ __forceinline__ void lds_get_record6(uint8_t* lds, int addr, uint32_t& lo16, uint32_t& hi32)
{
#if DXTLT_BC3_RECORD6
    const u32x2_align4 w = *reinterpret_cast<const u32x2_align4*>(lds + (addr & ~3));
    const uint32_t sh = (uint32_t)(addr & 2) * 8u;
    const uint32_t rec = __builtin_amdgcn_alignbit(w.y, w.x, sh);   // record bytes 0..3
    lo16 = rec & 0xFFFFu;
    hi32 = (rec >> 16) | ((w.y >> sh) << 16);                       // bytes 2..3, then bytes 4..5
#else
    lo16 = lds_at<uint16_t>(lds, addr + 0);
    hi32 = (uint32_t)lds_at<uint16_t>(lds, addr + 2) | ((uint32_t)lds_at<uint16_t>(lds, addr + 4) << 16);
#endif
}

template <int FMT, int VARIANT, bool SA, bool SC, int T>
__device__ __forceinline__ void scatter_to_image(uint8_t* lds, int u, u32x4 q)
{
    if constexpr (FMT == kBc1) {
        // q = { colours A, indices A, colours B, indices B }
        const uint32_t ca = decorrelate2<VARIANT>(q.x);
        const uint32_t cb = decorrelate2<VARIANT>(q.z);
        if constexpr (SC) {
            lds_at<uint32_t>(lds, 0 * T + 4 * u) = (ca & 0xFFFFu) | (cb << 16);          // c0 of A, B
            lds_at<uint32_t>(lds, 2 * T + 4 * u) = (ca >> 16) | (cb & 0xFFFF0000u);      // c1 of A, B
        } else {
            lds_at<u32x2>(lds, 0 * T + 8 * u) = u32x2{ca, cb};
        }
        lds_at<u32x2>(lds, 4 * T + 8 * u) = u32x2{q.y, q.w};
    } else if constexpr (FMT == kBc2) {
        // q = { alpha lo, alpha hi, colours, indices }
        const uint32_t c = decorrelate2<VARIANT>(q.z);
        lds_at<u32x2>(lds, 0 * T + 8 * u) = u32x2{q.x, q.y};
        if constexpr (SC) {
            lds_at<uint16_t>(lds, 8 * T + 2 * u) = (uint16_t)c;
            lds_at<uint16_t>(lds, 10 * T + 2 * u) = (uint16_t)(c >> 16);
        } else {
            lds_at<uint32_t>(lds, 8 * T + 4 * u) = c;
        }
        lds_at<uint32_t>(lds, 12 * T + 4 * u) = q.w;
    } else {
        // q = { a0 a1 i0 i1, i2 i3 i4 i5, colours, indices }
        const uint32_t c = decorrelate2<VARIANT>(q.z);
        if constexpr (SA) {
            lds_at<uint8_t>(lds, 0 * T + u) = (uint8_t)q.x;
            lds_at<uint8_t>(lds, 1 * T + u) = (uint8_t)(q.x >> 8);
        } else {
            lds_at<uint16_t>(lds, 0 * T + 2 * u) = (uint16_t)q.x;
        }
        lds_put_record6(lds, 2 * T + 6 * u, q.x >> 16, q.y);   // the 6-byte alpha-index record, only 2-byte aligned
        if constexpr (SC) {
            lds_at<uint16_t>(lds, 8 * T + 2 * u) = (uint16_t)c;
            lds_at<uint16_t>(lds, 10 * T + 2 * u) = (uint16_t)(c >> 16);
        } else {
            lds_at<uint32_t>(lds, 8 * T + 4 * u) = c;
        }
        lds_at<uint32_t>(lds, 12 * T + 4 * u) = q.w;
    }
}

template <int FMT, int VARIANT, bool SA, bool SC, int T>
__device__ __forceinline__ u32x4 gather_from_image(uint8_t* lds, int u)
{
    u32x4 q;
    if constexpr (FMT == kBc1) {
        uint32_t ca, cb;
        if constexpr (SC) {
            const uint32_t c0 = lds_at<uint32_t>(lds, 0 * T + 4 * u);
            const uint32_t c1 = lds_at<uint32_t>(lds, 2 * T + 4 * u);
            ca = (c0 & 0xFFFFu) | (c1 << 16);
            cb = (c0 >> 16) | (c1 & 0xFFFF0000u);
        } else {
            const u32x2 p = lds_at<u32x2>(lds, 0 * T + 8 * u);
            ca = p.x;
            cb = p.y;
        }
        const u32x2 idx = lds_at<u32x2>(lds, 4 * T + 8 * u);
        q.x = recorrelate2<VARIANT>(ca);
        q.y = idx.x;
        q.z = recorrelate2<VARIANT>(cb);
        q.w = idx.y;
    } else if constexpr (FMT == kBc2) {
        const u32x2 a = lds_at<u32x2>(lds, 0 * T + 8 * u);
        uint32_t c;
        if constexpr (SC)
            c = (uint32_t)lds_at<uint16_t>(lds, 8 * T + 2 * u) | ((uint32_t)lds_at<uint16_t>(lds, 10 * T + 2 * u) << 16);
        else
            c = lds_at<uint32_t>(lds, 8 * T + 4 * u);
        q.x = a.x;
        q.y = a.y;
        q.z = recorrelate2<VARIANT>(c);
        q.w = lds_at<uint32_t>(lds, 12 * T + 4 * u);
    } else {
        uint32_t a01;
        if constexpr (SA)
            a01 = (uint32_t)lds_at<uint8_t>(lds, 0 * T + u) | ((uint32_t)lds_at<uint8_t>(lds, 1 * T + u) << 8);
        else
            a01 = lds_at<uint16_t>(lds, 0 * T + 2 * u);
        uint32_t i01, i2345;
        lds_get_record6(lds, 2 * T + 6 * u, i01, i2345);
        uint32_t c;
        if constexpr (SC)
            c = (uint32_t)lds_at<uint16_t>(lds, 8 * T + 2 * u) | ((uint32_t)lds_at<uint16_t>(lds, 10 * T + 2 * u) << 16);
        else
            c = lds_at<uint32_t>(lds, 8 * T + 4 * u);
        q.x = a01 | (i01 << 16);
        q.y = i2345;
        q.z = recorrelate2<VARIANT>(c);
        q.w = lds_at<uint32_t>(lds, 12 * T + 4 * u);
    }
    return q;
}

// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD group; observed, used for speed
// only).  This bijective remap hands each group one contiguous eighth of the tiles, so two neighbouring tiles --
// which share a 128-byte line whenever a stream base is misaligned -- meet in the same XCD's L2 and leave (arrive)
// as one full line instead of two partial ones.
__device__ __forceinline__ uint64_t xcd_contiguous_tile(uint32_t orig, uint32_t nwg)
{
    // every product below is smaller than nwg: 32-bit arithmetic is exact (and half the scalar instructions)
    const uint32_t q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const uint32_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (orig >> 3);
}

// Tried and dropped (profiles/r01_z/shift_probe_rotation_and_chunks.txt): handing each XCD chunks of 4 / 16 / 64
// consecutive tiles in turn, so that the chip keeps one moving window -- slower than both this order and the identity
// on every alignment class (BC3 forward, 128-byte aligned bases: 0.75 against 0.84 identity / 0.78 contiguous).
// Neither did staggering the eight XCDs' starting points inside their eighths by 7 / 61 / 509 tiles (the idea: eighths of
// a power-of-two buffer start on the same memory channel): no change (profiles/r01_z/shift_probe_xcd_stagger.txt).

// ------------------------------------------------------------------------------------------------
// Tiled kernels: one tile per workgroup.  `aos` points at the range's first block, `soa` at byte 0 of the
// whole transformed buffer.  Preconditions (checked on the host): both pointers 16-byte aligned; every
// stream base off*total_blocks + width*first_block is a multiple of 16; gridDim.x == number of FULL tiles
// at the start of the range.
// ------------------------------------------------------------------------------------------------
// BC1 block normalisation fused into the forward kernels (experimental module of the reference,
// transform_bc1_with_normalize_blocks, experimental/normalize_blocks/transform.rs:65-166): the two blocks of the
// lane's vector are normalised in registers right after the load, so the fused path moves the same 2*len bytes.
template <int FMT, int NORM>
__device__ __forceinline__ u32x4 normalize_vector(u32x4 q)
{
    if constexpr (FMT == kBc1 && NORM != kNormNone) {
        uint32_t ca = q.x, xa = q.y, cb = q.z, xb = q.w;
        normalize_bc1_block<NORM>(ca, xa);
        normalize_bc1_block<NORM>(cb, xb);
        q = u32x4{ca, xa, cb, xb};
    }
    return q;
}

// bit 1 of the tiled kernels' `xcd_remap` argument (bit 0, experiments build only: XCD-contiguous tile order): the launch is two-dimensional, blockIdx.y numbers the buffers of a regular
// array whose sources / destinations lie `*_stride` bytes apart (first pointer argument, second pointer argument)
constexpr int kTiledArray = 2;

// One aligned tile (the body of fwd_tiled / inv_tiled; the batch kernel runs it for buffers whose stream bases are aligned).
// `lds`: THREADS * 16 bytes.
template <int FMT, int VARIANT, bool SA, bool SC, int THREADS, int NORM = kNormNone>
__device__ __forceinline__ void fwd_aligned_tile(const uint8_t* __restrict__ aos, uint8_t* __restrict__ soa,
                                                 uint64_t total_blocks, uint64_t first_block, uint64_t tile, uint8_t* lds)
{
    constexpr int T = tile_blocks(FMT, THREADS);
    const int t = threadIdx.x;
    WG_MARK(1);
    const u32x4 q = normalize_vector<FMT, NORM>(gload16(aos + tile * (THREADS * 16) + t * 16));
    WG_MARK_LOADS_DONE(2);
    scatter_to_image<FMT, VARIANT, SA, SC, T>(lds, t, q);
    __syncthreads();
    WG_MARK(3);
    const u32x4 v = lds_at<u32x4>(lds, t * 16);
    const uint64_t o = soa_offset_of_lane<FMT, SA, SC, THREADS>(t, total_blocks, first_block + tile * T);
    gstore16(soa + o, v);
    WG_MARK(4);
}

template <int FMT, int VARIANT, bool SA, bool SC, int THREADS>
__device__ __forceinline__ void inv_aligned_tile(const uint8_t* __restrict__ soa, uint8_t* __restrict__ aos,
                                                 uint64_t total_blocks, uint64_t first_block, uint64_t tile, uint8_t* lds)
{
    constexpr int T = tile_blocks(FMT, THREADS);
    const int t = threadIdx.x;
    const uint64_t o = soa_offset_of_lane<FMT, SA, SC, THREADS>(t, total_blocks, first_block + tile * T);
    lds_at<u32x4>(lds, t * 16) = gload16(soa + o);
    __syncthreads();
    const u32x4 q = gather_from_image<FMT, VARIANT, SA, SC, T>(lds, t);
    gstore16_aos(aos, aos + tile * (THREADS * 16) + t * 16, q);
}

template <int FMT, int VARIANT, bool SA, bool SC, int THREADS, int NORM = kNormNone>
__global__ void __launch_bounds__(THREADS)
fwd_tiled(const uint8_t* __restrict__ aos, uint8_t* __restrict__ soa, uint64_t total_blocks, uint64_t first_block,
          int xcd_remap, int64_t aos_stride, int64_t soa_stride)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[THREADS * 16];
    WG_MARK(0);
    if (xcd_remap & kTiledArray) {   // a regular array of buffers: blockIdx.y is the buffer (launch_batch)
        aos += (int64_t)blockIdx.y * aos_stride;
        soa += (int64_t)blockIdx.y * soa_stride;
    }
#ifdef DXTLT_EXPERIMENTS
    const uint64_t tile = (xcd_remap & 1) ? xcd_contiguous_tile(blockIdx.x, gridDim.x) : (uint64_t)blockIdx.x;
#else
    const uint64_t tile = blockIdx.x;   // identity order: the XCD-contiguous one costs aligned tiles 0.01-0.03 (profiles/r01_i_*)
#endif
    fwd_aligned_tile<FMT, VARIANT, SA, SC, THREADS, NORM>(aos, soa, total_blocks, first_block, tile, lds);
}

template <int FMT, int VARIANT, bool SA, bool SC, int THREADS>
__global__ void __launch_bounds__(THREADS)
inv_tiled(const uint8_t* __restrict__ soa, uint8_t* __restrict__ aos, uint64_t total_blocks, uint64_t first_block,
          int xcd_remap, int64_t soa_stride, int64_t aos_stride)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[THREADS * 16];
    if (xcd_remap & kTiledArray) {
        soa += (int64_t)blockIdx.y * soa_stride;
        aos += (int64_t)blockIdx.y * aos_stride;
    }
#ifdef DXTLT_EXPERIMENTS
    const uint64_t tile = (xcd_remap & 1) ? xcd_contiguous_tile(blockIdx.x, gridDim.x) : (uint64_t)blockIdx.x;
#else
    const uint64_t tile = blockIdx.x;
#endif
    inv_aligned_tile<FMT, VARIANT, SA, SC, THREADS>(soa, aos, total_blocks, first_block, tile, lds);
}

// ------------------------------------------------------------------------------------------------
// Shifted tiles: the tiled structure for transformed buffers whose stream bases are NOT 16-byte aligned
// (odd block counts -- e.g. a DDS payload with a full mip chain -- or ranges that start at an odd block).
// Stream s lands at global address G_s = soa + off_s*N + w_s*first_block with misalignment d_s = G_s & 15.
// Its slice of the LDS image is stored d_s bytes further in (each stream gets 16 bytes of padding), so an
// LDS byte and the global byte it maps to have the same address modulo 16: the body of every slice still
// leaves (arrives) as aligned 16-byte vectors, and only the first and last 16-byte segment of a slice are
// partial; those are moved with 1/2/4/8-byte accesses that touch exactly the slice's own bytes, so the
// neighbouring tile (which owns the rest of that 16-byte segment) never races with it.
// The AoS pointer is 16-byte aligned in the common case; any other address works (unaligned vector accesses), a little
// slower.  256-thread tiles.
// ------------------------------------------------------------------------------------------------
struct Shifts {
    int d[6];
    int natural;      // 1: every d[s] is a multiple of stream s's element width (always so when the SoA pointer is 8-byte aligned)
    int halo_vecs;    // halo tiles: 16-byte vectors in front of a tile whose blocks have bytes in the tile's windows
    // offset from the SoA pointer of the aligned segment that holds the first byte of stream s of the RANGE:
    // off_s * total_blocks + w_s * first_block - d[s] (may be "negative": wraps).  Filled by the host (launch_transform)
    // or by the batch kernel from its table entry: the tiles then add one multiple of a constant per stream instead of
    // redoing two 64-bit products and a difference per stream and workgroup on the scalar unit (PMC: 107 scalar
    // instructions per wave in the inverse shifted tile, 95 in the halo tile, against 36-40 in the aligned tiles).
    uint64_t gbase[6];
    // Edge tiles (halo tiles forward, shifted tiles inverse): the range holds `range_blocks` blocks = `full_tiles` whole tiles
    // and a ragged rest.  Workgroups [0, full_tiles) run whole tiles, one more -- when the launch has it -- the edge tile
    // behind them (the rest, and forward the last bytes of every stream that the whole tiles' moved-back windows leave out).
    uint32_t full_tiles;
    uint64_t range_blocks;
#ifdef DXTLT_EXPERIMENTS
    int xcd_remap;    // 1: consecutive tiles stay on one XCD (see xcd_contiguous_tile); the product fixes it per kernel
    int line_policy;  // forward store policy (first form: 1 = write-through for lines one wave instruction writes whole, 2 = plain
                      // stores on shared lines; halo tiles: 3 = write-through, else plain nt); the product always runs 3
    int skip_partial; // TIMING experiment, WRONG OUTPUT: 1 = leave out the partial head / tail segments and the halo load
#endif
};

// The tile order of the halo (forward) and shifted (inverse) kernels, and the halo tiles' store policy: fixed in the product
// (identity order + write-through forward -- every window starts on a 64-byte sector, nothing is shared; XCD-contiguous order
// inverse -- neighbouring tiles share lines), run-time values in the experiments build.
__device__ __forceinline__ bool shifts_xcd_contiguous(const Shifts& sh, bool product_default)
{
#ifdef DXTLT_EXPERIMENTS
    (void)product_default;
    return sh.xcd_remap != 0;
#else
    (void)sh;
    return product_default;
#endif
}
__device__ __forceinline__ bool shifts_halo_write_through(const Shifts& sh)
{
#ifdef DXTLT_EXPERIMENTS
    return sh.line_policy == 3;
#else
    (void)sh;
    return true;
#endif
}
__device__ __forceinline__ bool shifts_skip_partial(const Shifts& sh)
{
#ifdef DXTLT_EXPERIMENTS
    return sh.skip_partial != 0;
#else
    (void)sh;
    return false;
#endif
}

template <typename STREAMS>
__host__ __device__ inline void fill_gbase(Shifts& sh, const STREAMS& S, uint64_t total_blocks, uint64_t first_block)
{
    for (int i = 0; i < 6; ++i)
        sh.gbase[i] = i < S.n ? (uint64_t)S.off[i] * total_blocks + (uint64_t)S.width[i] * first_block - (uint64_t)sh.d[i] : 0;
}

// A copy of the kernel's Shifts argument whose every field is needed HERE (an empty, non-volatile asm makes each value opaque
// at this point), so that all of them leave the kernel-argument segment in ONE scalar round trip at the top of the kernel.
// Left alone the compiler fetches an argument in the block that first uses it: the arguments of the first branch first, the
// whole tile's own -- tile order, store policy, bases -- behind a second and third wait: dependent round trips in front of the
// tile's load.  The batch kernel, whose regular-array path happens to fetch everything at once, ran a buffer as its only entry
// at 0.80 where the single call ran it at 0.77 (tools/batch_vs_single_probe.py, profiles/r04_batch_edge_tiles.txt).
typedef const __attribute__((address_space(1))) uint8_t* global_cptr;
typedef __attribute__((address_space(1))) uint8_t* global_ptr;

// the same for a pointer argument (through an integer: a pointer that went through an asm would come back as a generic one,
// and every access of the tile would become a flat_* instruction)
__device__ __forceinline__ const uint8_t* fetched_now(const uint8_t* p)
{
    uint64_t v = reinterpret_cast<uintptr_t>(p);
    asm("" : "+s"(v));
    return (const uint8_t*)(global_cptr)v;
}
__device__ __forceinline__ uint8_t* fetched_now(uint8_t* p)
{
    uint64_t v = reinterpret_cast<uintptr_t>(p);
    asm("" : "+s"(v));
    return (uint8_t*)(global_ptr)v;
}

__device__ __forceinline__ Shifts shifts_fetched_at_once(const Shifts& in)
{
    Shifts s = in;
    // ONE statement: with several, the compiler fetches what the first one needs, waits, and only then asks for the next one's
#ifdef DXTLT_EXPERIMENTS
    asm("" : "+s"(s.d[0]), "+s"(s.d[1]), "+s"(s.d[2]), "+s"(s.d[3]), "+s"(s.d[4]), "+s"(s.d[5]), "+s"(s.gbase[0]), "+s"(s.gbase[1]),
             "+s"(s.gbase[2]), "+s"(s.gbase[3]), "+s"(s.gbase[4]), "+s"(s.gbase[5]), "+s"(s.xcd_remap), "+s"(s.line_policy),
             "+s"(s.skip_partial), "+s"(s.natural), "+s"(s.halo_vecs), "+s"(s.full_tiles), "+s"(s.range_blocks));
#else
    asm("" : "+s"(s.d[0]), "+s"(s.d[1]), "+s"(s.d[2]), "+s"(s.d[3]), "+s"(s.d[4]), "+s"(s.d[5]), "+s"(s.gbase[0]), "+s"(s.gbase[1]),
             "+s"(s.gbase[2]), "+s"(s.gbase[3]), "+s"(s.gbase[4]), "+s"(s.gbase[5]), "+s"(s.natural), "+s"(s.halo_vecs),
             "+s"(s.full_tiles), "+s"(s.range_blocks));
#endif
    return s;
}

// element width of a stream: its bytes per block, except the 6-byte alpha index records, which move as three halfwords
__host__ __device__ constexpr int stream_element_width(int bytes_per_block) { return bytes_per_block == 6 ? 2 : bytes_per_block; }

template <typename STREAMS>
__host__ __device__ inline int shifts_are_natural(const STREAMS& S, const int (&d)[6])
{
    int ok = 1;
    for (int i = 0; i < 6; ++i)
        if (i < S.n && (d[i] & (stream_element_width(S.width[i]) - 1)) != 0)
            ok = 0;
    return ok;
}


// LDS accesses of the shifted tiles.  A field of W bytes sits at an address that is only known to be congruent to
// `addr & (W - 1)` (the stream's shift, uniform over the wave).  DS instructions at addresses that are not multiples
// of their width execute on gfx950, but several times slower than aligned ones (profiles/r01_s), and clang -- which
// assumes they are free -- fuses neighbouring narrow accesses into exactly such instructions.  So the field is moved
// as the shortest sequence of naturally aligned 1/2/4/8-byte pieces for its misalignment, through volatile pointers
// in the LDS address space (volatile stops the fusion; a plain volatile pointer would become flat_* instructions).
template <typename T>
using lds_volatile = volatile T __attribute__((address_space(3)));

template <int SIZE>
__device__ __forceinline__ void lds_put_piece(uint8_t* lds, int addr, uint64_t v)
{
    if constexpr (SIZE == 1) *(lds_volatile<uint8_t>*)(lds + addr) = (uint8_t)v;
    if constexpr (SIZE == 2) *(lds_volatile<uint16_t>*)(lds + addr) = (uint16_t)v;
    if constexpr (SIZE == 4) *(lds_volatile<uint32_t>*)(lds + addr) = (uint32_t)v;
    if constexpr (SIZE == 8) *(lds_volatile<uint64_t>*)(lds + addr) = v;
}

template <int SIZE>
__device__ __forceinline__ uint64_t lds_get_piece(uint8_t* lds, int addr)
{
    if constexpr (SIZE == 1) return *(lds_volatile<uint8_t>*)(lds + addr);
    if constexpr (SIZE == 2) return *(lds_volatile<uint16_t>*)(lds + addr);
    if constexpr (SIZE == 4) return *(lds_volatile<uint32_t>*)(lds + addr);
    if constexpr (SIZE == 8) return *(lds_volatile<uint64_t>*)(lds + addr);
    return 0;
}

// size of the piece that starts P bytes into a W-byte field whose address is K modulo W
constexpr int lds_piece_size(int W, int K, int P)
{
    const int a = K + P;
    int align = a == 0 ? W : (a & -a);
    if (align > W) align = W;
    int size = 1;
    while (size * 2 <= align && size * 2 <= W - P) size *= 2;
    return size;
}

template <int W, int K, int P = 0>
__device__ __forceinline__ void lds_put_seq(uint8_t* lds, int addr, uint64_t v)
{
    if constexpr (P < W) {
        constexpr int size = lds_piece_size(W, K, P);
        lds_put_piece<size>(lds, addr + P, v >> (8 * P));
        lds_put_seq<W, K, P + size>(lds, addr, v);
    }
}

template <int W, int K, int P = 0>
__device__ __forceinline__ uint64_t lds_get_seq(uint8_t* lds, int addr)
{
    if constexpr (P < W) {
        constexpr int size = lds_piece_size(W, K, P);
        return (lds_get_piece<size>(lds, addr + P) << (8 * P)) | lds_get_seq<W, K, P + size>(lds, addr);
    }
    return 0;
}

// NAT: the caller guarantees addr % W == 0 (every stream shift is a multiple of its field width: Shifts::natural)
template <int W, bool NAT = false>
__device__ __forceinline__ void lds_put(uint8_t* lds, int addr, uint64_t v)
{
    if constexpr (W == 1 || NAT) {
        lds_put_piece<W>(lds, addr, v);
    } else {
        switch (addr & (W - 1)) {
        case 0: lds_put_seq<W, 0>(lds, addr, v); break;
        case 1: lds_put_seq<W, 1>(lds, addr, v); break;
        case 2: if constexpr (W > 2) lds_put_seq<W, 2>(lds, addr, v); break;
        case 3: if constexpr (W > 2) lds_put_seq<W, 3>(lds, addr, v); break;
        case 4: if constexpr (W > 4) lds_put_seq<W, 4>(lds, addr, v); break;
        case 5: if constexpr (W > 4) lds_put_seq<W, 5>(lds, addr, v); break;
        case 6: if constexpr (W > 4) lds_put_seq<W, 6>(lds, addr, v); break;
        default: if constexpr (W > 4) lds_put_seq<W, 7>(lds, addr, v); break;
        }
    }
}

template <int W, bool NAT = false>
__device__ __forceinline__ uint64_t lds_get(uint8_t* lds, int addr)
{
    if constexpr (W == 1 || NAT) {
        return lds_get_piece<W>(lds, addr);
    } else {
        switch (addr & (W - 1)) {
        case 0: return lds_get_seq<W, 0>(lds, addr);
        case 1: return lds_get_seq<W, 1>(lds, addr);
        case 2: if constexpr (W > 2) return lds_get_seq<W, 2>(lds, addr); break;
        case 3: if constexpr (W > 2) return lds_get_seq<W, 3>(lds, addr); break;
        case 4: if constexpr (W > 4) return lds_get_seq<W, 4>(lds, addr); break;
        case 5: if constexpr (W > 4) return lds_get_seq<W, 5>(lds, addr); break;
        case 6: if constexpr (W > 4) return lds_get_seq<W, 6>(lds, addr); break;
        default: if constexpr (W > 4) return lds_get_seq<W, 7>(lds, addr); break;
        }
        return 0;
    }
}

// stream indices of the fields (make_streams order)
template <int FMT, bool SA, bool SC>
struct FieldStreams {
    static constexpr int alpha = 0;                                       // BC2 alpha, BC3 a0 or (a0,a1)
    static constexpr int a1 = 1;                                          // BC3 split alphas
    static constexpr int aidx = SA ? 2 : 1;                               // BC3
    static constexpr int col = FMT == kBc1 ? 0 : FMT == kBc2 ? 1 : (SA ? 3 : 2);  // c0 or (c0,c1)
    static constexpr int c1 = col + 1;                                    // split colours
    static constexpr int idx = col + (SC ? 2 : 1);
};

// NAT (Shifts::natural): every stream's shift is a multiple of its element width, so each element goes out as one aligned
// DS instruction and the per-access alignment switch disappears.  BC1 then writes its two blocks' fields separately
// (the pair is only element-aligned); the other formats' accesses are element-wide already.
template <int FMT, int VARIANT, bool SA, bool SC, bool NAT>
__device__ __forceinline__ void scatter_shifted(uint8_t* lds, int u, u32x4 q, const int (&base)[6])
{
    using F = FieldStreams<FMT, SA, SC>;
    if constexpr (FMT == kBc1) {
        const uint32_t ca = decorrelate2<VARIANT>(q.x);
        const uint32_t cb = decorrelate2<VARIANT>(q.z);
        if constexpr (NAT) {
            if constexpr (SC) {
                lds_put<2, true>(lds, base[F::col] + 4 * u, ca & 0xFFFFu);
                lds_put<2, true>(lds, base[F::col] + 4 * u + 2, cb & 0xFFFFu);
                lds_put<2, true>(lds, base[F::c1] + 4 * u, ca >> 16);
                lds_put<2, true>(lds, base[F::c1] + 4 * u + 2, cb >> 16);
            } else {
                lds_put<4, true>(lds, base[F::col] + 8 * u, ca);
                lds_put<4, true>(lds, base[F::col] + 8 * u + 4, cb);
            }
            lds_put<4, true>(lds, base[F::idx] + 8 * u, q.y);
            lds_put<4, true>(lds, base[F::idx] + 8 * u + 4, q.w);
        } else {
            if constexpr (SC) {
                lds_put<4>(lds, base[F::col] + 4 * u, (ca & 0xFFFFu) | (cb << 16));
                lds_put<4>(lds, base[F::c1] + 4 * u, (ca >> 16) | (cb & 0xFFFF0000u));
            } else {
                lds_put<8>(lds, base[F::col] + 8 * u, (uint64_t)ca | ((uint64_t)cb << 32));
            }
            lds_put<8>(lds, base[F::idx] + 8 * u, (uint64_t)q.y | ((uint64_t)q.w << 32));
        }
    } else {
        const uint32_t c = decorrelate2<VARIANT>(q.z);
        if constexpr (FMT == kBc2) {
            lds_put<8, NAT>(lds, base[F::alpha] + 8 * u, (uint64_t)q.x | ((uint64_t)q.y << 32));
        } else {
            if constexpr (SA) {
                lds_put<1>(lds, base[F::alpha] + u, q.x & 0xFF);
                lds_put<1>(lds, base[F::a1] + u, (q.x >> 8) & 0xFF);
            } else {
                lds_put<2, NAT>(lds, base[F::alpha] + 2 * u, q.x & 0xFFFF);
            }
            if constexpr (NAT && DXTLT_BC3_RECORD6) {
                lds_put_record6(lds, base[F::aidx] + 6 * u, q.x >> 16, q.y);   // (an even address: the shift is natural)
            } else {
                lds_put<2, NAT>(lds, base[F::aidx] + 6 * u + 0, q.x >> 16);
                lds_put<2, NAT>(lds, base[F::aidx] + 6 * u + 2, q.y & 0xFFFF);
                lds_put<2, NAT>(lds, base[F::aidx] + 6 * u + 4, q.y >> 16);
            }
        }
        if constexpr (SC) {
            lds_put<2, NAT>(lds, base[F::col] + 2 * u, c & 0xFFFF);
            lds_put<2, NAT>(lds, base[F::c1] + 2 * u, c >> 16);
        } else {
            lds_put<4, NAT>(lds, base[F::col] + 4 * u, c);
        }
        lds_put<4, NAT>(lds, base[F::idx] + 4 * u, q.w);
    }
}

template <int FMT, int VARIANT, bool SA, bool SC, bool NAT>
__device__ __forceinline__ u32x4 gather_shifted(uint8_t* lds, int u, const int (&base)[6])
{
    using F = FieldStreams<FMT, SA, SC>;
    u32x4 q;
    if constexpr (FMT == kBc1) {
        uint32_t ca, cb;
        uint64_t idx;
        if constexpr (NAT) {
            if constexpr (SC) {
                ca = (uint32_t)lds_get<2, true>(lds, base[F::col] + 4 * u) | ((uint32_t)lds_get<2, true>(lds, base[F::c1] + 4 * u) << 16);
                cb = (uint32_t)lds_get<2, true>(lds, base[F::col] + 4 * u + 2) |
                     ((uint32_t)lds_get<2, true>(lds, base[F::c1] + 4 * u + 2) << 16);
            } else {
                ca = (uint32_t)lds_get<4, true>(lds, base[F::col] + 8 * u);
                cb = (uint32_t)lds_get<4, true>(lds, base[F::col] + 8 * u + 4);
            }
            idx = lds_get<4, true>(lds, base[F::idx] + 8 * u) | (lds_get<4, true>(lds, base[F::idx] + 8 * u + 4) << 32);
        } else {
            if constexpr (SC) {
                const uint32_t c0 = (uint32_t)lds_get<4>(lds, base[F::col] + 4 * u);
                const uint32_t c1 = (uint32_t)lds_get<4>(lds, base[F::c1] + 4 * u);
                ca = (c0 & 0xFFFFu) | (c1 << 16);
                cb = (c0 >> 16) | (c1 & 0xFFFF0000u);
            } else {
                const uint64_t p = lds_get<8>(lds, base[F::col] + 8 * u);
                ca = (uint32_t)p;
                cb = (uint32_t)(p >> 32);
            }
            idx = lds_get<8>(lds, base[F::idx] + 8 * u);
        }
        q.x = recorrelate2<VARIANT>(ca);
        q.y = (uint32_t)idx;
        q.z = recorrelate2<VARIANT>(cb);
        q.w = (uint32_t)(idx >> 32);
    } else {
        if constexpr (FMT == kBc2) {
            const uint64_t a = lds_get<8, NAT>(lds, base[F::alpha] + 8 * u);
            q.x = (uint32_t)a;
            q.y = (uint32_t)(a >> 32);
        } else {
            uint32_t a01;
            if constexpr (SA)
                a01 = (uint32_t)lds_get<1>(lds, base[F::alpha] + u) | ((uint32_t)lds_get<1>(lds, base[F::a1] + u) << 8);
            else
                a01 = (uint32_t)lds_get<2, NAT>(lds, base[F::alpha] + 2 * u);
            if constexpr (NAT && DXTLT_BC3_RECORD6) {
                uint32_t i01, i2345;
                lds_get_record6(lds, base[F::aidx] + 6 * u, i01, i2345);   // (reads up to 2 bytes past the record: the region's padding)
                q.x = a01 | (i01 << 16);
                q.y = i2345;
            } else {
                const uint32_t i01 = (uint32_t)lds_get<2, NAT>(lds, base[F::aidx] + 6 * u + 0);
                const uint32_t i23 = (uint32_t)lds_get<2, NAT>(lds, base[F::aidx] + 6 * u + 2);
                const uint32_t i45 = (uint32_t)lds_get<2, NAT>(lds, base[F::aidx] + 6 * u + 4);
                q.x = a01 | (i01 << 16);
                q.y = i23 | (i45 << 16);
            }
        }
        uint32_t c;
        if constexpr (SC)
            c = (uint32_t)lds_get<2, NAT>(lds, base[F::col] + 2 * u) | ((uint32_t)lds_get<2, NAT>(lds, base[F::c1] + 2 * u) << 16);
        else
            c = (uint32_t)lds_get<4, NAT>(lds, base[F::col] + 4 * u);
        q.z = recorrelate2<VARIANT>(c);
        q.w = (uint32_t)lds_get<4, NAT>(lds, base[F::idx] + 4 * u);
    }
    return q;
}

// Copy bytes [lo, hi) of one 16-byte segment between two pointers that are both 16-byte aligned at byte 0 of
// the segment; either lo == 0 (a slice's tail) or hi == 16 (a slice's head).  Typed 1/2/4/8-byte moves.
template <bool TO_GLOBAL>
__device__ __forceinline__ void copy_partial_segment(uint8_t* dst, const uint8_t* src, int lo, int hi)
{
    auto mv = [&](int p, int w) {
        if (w == 1) dst[p] = src[p];
        if (w == 2) *reinterpret_cast<uint16_t*>(dst + p) = *reinterpret_cast<const uint16_t*>(src + p);
        if (w == 4) *reinterpret_cast<uint32_t*>(dst + p) = *reinterpret_cast<const uint32_t*>(src + p);
        if (w == 8) *reinterpret_cast<u32x2*>(dst + p) = *reinterpret_cast<const u32x2*>(src + p);
    };
    if (hi == 16) {  // head: [lo, 16)
        int p = lo;
        if (p & 1) { mv(p, 1); p += 1; }
        if (p & 2) { mv(p, 2); p += 2; }
        if (p & 4) { mv(p, 4); p += 4; }
        if (p & 8) { mv(p, 8); }
    } else {  // tail: [0, hi)
        int p = 0;
        if (hi & 8) { mv(p, 8); p += 8; }
        if (hi & 4) { mv(p, 4); p += 4; }
        if (hi & 2) { mv(p, 2); p += 2; }
        if (hi & 1) { mv(p, 1); }
    }
}

// 64-bit value that is the same in every lane, kept in SGPRs (the compiler otherwise sinks the per-stream base
// computations into the per-lane branches and does them with quarter-rate v_mad_u64_u32)
__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// global offset (from the SoA pointer) of segment 0 of every stream's slice of tile `tile` of the range (T blocks per
// tile): aligned, d[s] bytes before the slice
template <int FMT, bool SA, bool SC, int T>
__device__ __forceinline__ void slice_bases(uint64_t tile, const Shifts& sh, uint64_t (&gb)[6])
{
    constexpr Streams S = make_streams(FMT, SA, SC);
#pragma unroll
    for (int s = 0; s < 6; ++s)
        gb[s] = s < S.n ? uniform64(sh.gbase[s] + tile * (uint64_t)(S.width[s] * T)) : 0;
}

// LDS bytes of a shifted tile of R x 256 lanes' worth of blocks: the image plus 16 bytes of padding per stream
constexpr int shift_lds_bytes(int r, int threads = 256) { return r * threads * 16 + 16 * 6; }
constexpr int kShiftLdsBytes = shift_lds_bytes(1);

// Lanes of the halo (forward), shifted (inverse) and edge tiles and of the batch kernel: 256 -- except the FORWARD halo tiles of BC1
// WITHOUT the colour split, which take 128.  That setting's 256-lane tile writes two runs of exactly 2 KiB (colours, indices), the one
// shape the memory side's write path dislikes -- the ALIGNED kernel shows it as soon as it is run with 256 lanes (forward 0.810
// against 0.861 with 128 lanes; with the split 0.839 / 0.846; +50 % TCC_EA0_WRREQ_STALL at identical request counts) -- and it is
// what round 3 reported as "BC1 no-split halo forward -0.026, cause not found" (tools/bc1_nosplit_probe.py,
// profiles/r05_bc1_nosplit.txt).  128 lanes for EVERY BC1 tile was measured too (-DDXTLT_BC1_SHIFT_THREADS=128: the size the
// aligned BC1 tiles have had since round 1) and is not the default: single call +0.004 / -0.005, corpus batch -0.033 / -0.008
// (twice the workgroups, twice the halo share, twice the lookups per byte).
#ifndef DXTLT_BC1_SHIFT_THREADS
#define DXTLT_BC1_SHIFT_THREADS 256
#endif
__host__ __device__ constexpr int shift_tile_threads(int fmt) { return fmt == kBc1 ? DXTLT_BC1_SHIFT_THREADS : 256; }
__host__ __device__ constexpr int halo_tile_threads(int fmt, bool split_colour)
{
    return fmt == kBc1 && !split_colour ? 128 : shift_tile_threads(fmt);
}
// Lanes of every tile of a BATCH launch, planned per (format, colour split, direction) like the single-buffer call's: the forward
// launch of BC1 without the colour split takes the 128-lane tiles too (aligned, halo and edge tiles alike: the aligned tile shows the
// 2 KiB + 2 KiB store shape just as the halo tile does).  Same-box A/B against -DDXTLT_BATCH_BC1_NOSPLIT_FWD_THREADS=256, the round-5
// shape (tools/batch_nosplit_probe.py, profiles/r06_batch_bc1_nosplit.txt): 64 x (16 MiB - 1 block) 0.72 -> 0.79, 1024 x (1 MiB - 1
// block) 0.69 -> 0.77, mixed aligned sizes 0.76 -> 0.78, the mip-chained corpus level (-0.015 with YCoCg-R, +0.013 without).  The same
// record holds the negative for the SPLIT setting: 128 lanes lose 0.015-0.045 on every size class of the corpus, so 256 stay there.
#ifndef DXTLT_BATCH_BC1_NOSPLIT_FWD_THREADS
#define DXTLT_BATCH_BC1_NOSPLIT_FWD_THREADS 128
#endif
__host__ __device__ constexpr int batch_tile_threads(int fmt, bool split_colour, bool inverse)
{
    return !inverse && fmt == kBc1 && !split_colour ? DXTLT_BATCH_BC1_NOSPLIT_FWD_THREADS : shift_tile_threads(fmt);
}
// R = sub-tiles of 256 lanes per workgroup.  Only R = 1 is instantiated: with misaligned stream bases every slice shares
// its first and last 128-byte line with the neighbouring tiles (BC3, 256 blocks per tile: 12 of 38 lines), and R = 4
// cuts that to 12 of 134 -- but it measured slower, not faster (BC3 odd count 0.710 / 0.773 forward / inverse against
// 0.702 / 0.797, BC1 0.755 / 0.777 against 0.775 / 0.818; profiles/r01_z/shift_probe_with_big_tiles.txt).  L2 merges the
// shared lines either way: HBM traffic is 1.003 x the algorithmic bytes on odd counts (PMC).


// ------------------------------------------------------------------------------------------------
// Forward HALO tiles: the tiled structure for transformed buffers whose stream bases are off their lines, forward direction.
// The first form (bcn_experiments.h) kept each tile's slices where they fall and moved the partial first / last 16-byte segment
// of every slice with typed narrow stores; the counters on odd block counts (profiles/r02_a_shift_pmc.txt, BC3 default settings,
// 2^26 + 1 blocks against 2^26) showed HBM requests identical and all of them full 64-byte ones, memory-side write stalls 50 x
// LOWER -- but 5.5 x the vector-memory store instructions per wave, 2 x the VALU and 3.6 x the SALU instructions, 2.65 x the
// issue-stall cycles: bound by its own instruction stream, not by memory.  So the partial segments went: a tile's window on
// stream s is moved BACK by the stream's misalignment d_s, to the aligned range [G_s - d_s, G_s - d_s + w_s * T).  Its first d_s
// bytes are records of blocks in front of the tile: the workgroup loads that halo too (Shifts::halo_vecs vectors, one more load
// instruction for the first lanes of wave 0; the lines were fetched moments before by the previous tile and come out of the
// memory-side cache), every stream region of the LDS image is kHaloBlocks longer at the front, and what leaves the workgroup is
// exactly what leaves an aligned tile: one full, aligned 16-byte store per lane.  Bytes no whole tile's window covers -- the head
// of every stream of the RANGE (tile 0 has no halo: the blocks before it may not exist or belong to another call) and everything
// behind the last window -- are written by the EDGE tiles further down (rounds 1-3: by the element-granular kernel).
// ------------------------------------------------------------------------------------------------
// Windows are moved back to a 64-BYTE boundary, not just a 16-byte one (d_s = stream base mod 64): two tiles that meet
// inside a 128-byte line then each write whole 64-byte sectors of it, which is what the memory side writes without a
// read-modify-write -- with 16-byte boundaries such lines had to meet in one L2 (XCD-contiguous tile order, itself worth
// -0.04) or cost 0.08 (profiles/r02_b_shift_probe.txt).  The halo grows to at most 63 bytes per stream = at most 63
// blocks for a 1-byte stream; only as many vectors as some stream needs are fetched (Shifts::halo_vecs).
#ifndef DXTLT_HALO_ALIGN
#define DXTLT_HALO_ALIGN 64   // experiment: 128 = windows on 128-byte lines (no line shared between tiles, halo up to 127 bytes per stream)
#endif
constexpr int kHaloAlign = DXTLT_HALO_ALIGN;
constexpr int kHaloBlocks = kHaloAlign;
constexpr int kHaloPad = kHaloAlign;   // bytes between the stream regions of the LDS image: room for d_s
// THREADS: lanes of a halo tile -- 256; 128 for BC1 without the colour split (launch_transform); 512 in the experiments build
// (round 5's attempt to halve the halo share per tile: BC3 +-0.005, BC1 -0.025, profiles/r05_halo_512.txt)
template <int FMT, int THREADS = 256>
constexpr int halo_lds_bytes() { return fmt_block(FMT) * (tile_blocks(FMT, THREADS) + kHaloBlocks) + kHaloPad * 6; }

// bytes [lo, hi) of a 16-byte segment (both pointers 16-byte aligned at byte 0) as the fewest naturally aligned 1/2/4/8-byte
// pieces -- at most six.  A first version moved them one byte at a time: fifteen dependent narrow stores into one 64-byte
// sector cost an edge tile more than all its vector stores together (4096 x (256 KiB - 1 block) BC3 forward: 0.61 with byte
// loops, 0.67 without them, 0.71 with no edge tiles at all; profiles/r04_batch_edge_tiles.txt).
__device__ __forceinline__ void copy_segment_bytes(uint8_t* dst, const uint8_t* src, int lo, int hi)
{
    int p = lo;
#pragma clang loop vectorize(disable) unroll(disable)
    while (p < hi) {
        const int left = hi - p;
        if ((p & 7) == 0 && left >= 8) {
            *reinterpret_cast<u32x2*>(dst + p) = *reinterpret_cast<const u32x2*>(src + p);
            p += 8;
        } else if ((p & 3) == 0 && left >= 4) {
            *reinterpret_cast<uint32_t*>(dst + p) = *reinterpret_cast<const uint32_t*>(src + p);
            p += 4;
        } else if ((p & 1) == 0 && left >= 2) {
            *reinterpret_cast<uint16_t*>(dst + p) = *reinterpret_cast<const uint16_t*>(src + p);
            p += 2;
        } else {
            dst[p] = src[p];
            p += 1;
        }
    }
}

// Copy-out of wave W of a halo tile: lane t moves image byte 16 t of the tile's windows.  A wave's 1 KiB of the image
// meets at most three streams (BC3 with split alphas, wave 0) and usually one, and which ones is known at compile time:
// the per-lane select over the streams -- a third of the first version's vector instructions -- shrinks to the streams
// the wave can meet.
// EDGE: the tile is the first or the last of its range: vlo[s] .. vhi[s] (window coordinates, uniform) are the bytes of stream
// s that are this tile's to write.  A segment they cut leaves as a few narrow pieces, a segment outside them not at all, a
// segment inside them whole -- with a plain nt store: the lines at both ends of an edge tile's windows are completed by other
// workgroups (the neighbouring tile, the tile at the other end of the buffer where one stream ends and the next begins, a
// neighbouring buffer's), some of them with narrow stores, and write-through next to that is the collapse round 1 met on shared
// lines.  Measured on 4096 x (256 KiB - 1 block) BC3, forward: plain nt throughout 0.66; write-through for every sector that is
// the tile's alone 0.49; write-through except in the line where the stream begins or ends 0.55 (profiles/r04_batch_edge_tiles.txt).
template <int FMT, bool SA, bool SC, int W, bool EDGE = false, int THREADS = 256>
__device__ __forceinline__ void halo_copy_out_wave(uint8_t* __restrict__ soa, const uint8_t* lds, int t, const uint64_t (&gb)[6],
                                                   const Shifts& sh, const int (&vlo)[6], const int (&vhi)[6])
{
    constexpr Streams S = make_streams(FMT, SA, SC);
    constexpr int T = tile_blocks(FMT, THREADS);
    constexpr int H = kHaloBlocks;
    constexpr int wave_lo = W * 1024, wave_hi = wave_lo + 1024;
    const int o = t * 16;
    int la = 0, wo = 0, lo_s = 0, hi_s = 0;
    uint64_t g = 0;
    static_for<0, S.n>([&](auto si) {
        constexpr int s = decltype(si)::value;
        constexpr int lo = S.off[s] * T;
        constexpr int hi = lo + S.width[s] * T;
        if constexpr (lo < wave_hi && hi > wave_lo) {
            constexpr bool only = lo <= wave_lo && hi >= wave_hi;   // the whole wave sits in this stream
            if (only || (o >= lo && o < hi)) {
                la = S.off[s] * (T + H) + kHaloPad * s + S.width[s] * H + (o - lo);
                g = gb[s] + (uint64_t)(o - lo);
                if constexpr (EDGE) {
                    wo = o - lo;
                    lo_s = vlo[s];
                    hi_s = vhi[s];
                }
            }
        }
    });
    if constexpr (EDGE) {
        const int lo = lo_s - wo > 0 ? lo_s - wo : 0;
        const int hi = hi_s - wo < 16 ? hi_s - wo : 16;
        if (hi <= lo)
            return;
        if (lo != 0 || hi != 16)
            copy_segment_bytes(soa + g, lds + la, lo, hi);
        else
            __builtin_nontemporal_store(lds_at<u32x4>(const_cast<uint8_t*>(lds), la), reinterpret_cast<u32x4*>(soa + g));
        return;
    }
    // Every window starts on a 64-byte sector, so no sector is shared between tiles: write-through streaming stores as in the
    // aligned tiles (the experiments build can ask for plain nt, which lets L2 merge the parts of a shared line)
    if (shifts_halo_write_through(sh))
        gstore16(soa + g, lds_at<u32x4>(const_cast<uint8_t*>(lds), la));
    else
        __builtin_nontemporal_store(lds_at<u32x4>(const_cast<uint8_t*>(lds), la), reinterpret_cast<u32x4*>(soa + g));
}

// one halo tile; `lds`: halo_lds_bytes<FMT, THREADS>() bytes
template <int FMT, int VARIANT, bool SA, bool SC, int NORM, bool NAT, int THREADS = 256>
__device__ __forceinline__ void fwd_halo_tile(const uint8_t* __restrict__ aos, uint8_t* __restrict__ soa,
                                              uint64_t /*total_blocks: in sh.gbase*/, uint64_t /*first_block: in sh.gbase*/, const Shifts& sh, uint64_t tile,
                                              uint8_t* lds)
{
    constexpr Streams S = make_streams(FMT, SA, SC);
    constexpr int T = tile_blocks(FMT, THREADS);
    constexpr int H = kHaloBlocks;
    constexpr int HV = H * fmt_block(FMT) / 16;  // halo vectors at most: 64 (BC2 / BC3) or 32 (BC1)
    const int t = threadIdx.x;
    // region of stream s: starts at off_s * (T + H) + 64 * s (16-byte aligned), holds the records of blocks
    // [blk0 - H, blk0 + T) from byte d_s on (d_s = 0..63); base[s] = address of the record of the tile's block 0
    int base[6];
#pragma unroll
    for (int s = 0; s < 6; ++s)
        base[s] = s < S.n ? S.off[s] * (T + H) + kHaloPad * s + sh.d[s] + S.width[s] * H : 0;

    const uint8_t* tile_aos = aos + tile * (THREADS * 16);
    // TEMPORAL load (no `nt`): the tile's last blocks are read a second time, as the next tile's halo, by a workgroup on
    // another XCD; a line fetched with `nt` is gone by then and comes from HBM again (PMC: 1.25 x the algorithmic read with
    // a 63-block halo), a line fetched temporally is still in the memory-side cache.  Found by accident -- the compiler
    // merged an experiment's two loads and dropped the hint -- and worth 0.05-0.07 of peak on large halos
    // (profiles/r02_b_shift_probe.txt); on small halos it costs nothing.
    WG_MARK(1);
    const u32x4 q = *reinterpret_cast<const u32x4*>(tile_aos + t * 16);
    // Only the blocks that have bytes inside a window are fetched: max over the streams of ceil(d_s / w_s) blocks, at
    // most 16 (the whole halo costs 0.02 of peak on BC3 -- 6 % more bytes read -- profiles/r02_b_shift_probe.txt).
    const int hv = sh.halo_vecs;
    const bool has_halo = tile > 0 && t < hv && !shifts_skip_partial(sh);
    if (has_halo) {
        // plain load: the previous tile has just fetched these lines
        const u32x4 qh = *reinterpret_cast<const u32x4*>(tile_aos - hv * 16 + t * 16);
        scatter_shifted<FMT, VARIANT, SA, SC, NAT>(lds, t - hv, normalize_vector<FMT, NORM>(qh), base);
    }
    static_assert(HV <= THREADS, "the halo is loaded by the first lanes of the workgroup");
    WG_MARK_LOADS_DONE(2);
    scatter_shifted<FMT, VARIANT, SA, SC, NAT>(lds, t, normalize_vector<FMT, NORM>(q), base);
    __syncthreads();
    WG_MARK(3);

    uint64_t gb[6];
    slice_bases<FMT, SA, SC, T>(tile, sh, gb);
    const int none[6] = {0, 0, 0, 0, 0, 0};   // (tile 0 and the last tile of a range run fwd_halo_edge_tile)
    if constexpr (THREADS == 256) {
        switch (__builtin_amdgcn_readfirstlane(t >> 6)) {
        case 0: halo_copy_out_wave<FMT, SA, SC, 0>(soa, lds, t, gb, sh, none, none); break;
        case 1: halo_copy_out_wave<FMT, SA, SC, 1>(soa, lds, t, gb, sh, none, none); break;
        case 2: halo_copy_out_wave<FMT, SA, SC, 2>(soa, lds, t, gb, sh, none, none); break;
        default: halo_copy_out_wave<FMT, SA, SC, 3>(soa, lds, t, gb, sh, none, none); break;
        }
    } else {
        for_this_wave<THREADS / 64>(t, [&](auto wi) {
            halo_copy_out_wave<FMT, SA, SC, decltype(wi)::value, false, THREADS>(soa, lds, t, gb, sh, none, none);
        });
    }
    WG_MARK(4);
}

// ------------------------------------------------------------------------------------------------
// Edge tiles.  The first and the last tile of a range are the same tiles with masks: lanes whose blocks do not exist load
// nothing, and a 16-byte segment of a window leaves (arrives) whole when it lies inside the stream's bytes of the range,
// byte by byte where the stream begins or ends inside it, not at all outside.  Round 3 sent these blocks -- the records of a
// range's first 64 blocks, everything behind the last whole tile -- to the element path (one lane per block, 8-11 narrow
// stores each), as separate launches in the single-buffer call and as 256-block workgroups in the batch kernel: for a
// texture of a few hundred KiB with a mip chain (an odd block count: every stream base off its line) those workgroups were
// the slowest part of the launch (4096 x (256 KiB - 1 block) BC3: 0.60 / 0.57 of peak against 0.79 / 0.74 for 4096 x 256 KiB,
// profiles/r03_batch_spacing.txt).  Here every full segment of an edge still moves as one 16-byte vector.
// ------------------------------------------------------------------------------------------------
// bytes [lo, hi) of a 16-byte segment, one at a time (a few lanes per stream and range)
// The window segment a lane looks after, selected by data (no control flow: the memory instructions that follow are then one
// per lane for the whole workgroup, issued together -- a first version branched per stream and ran a stream's load, wait and
// LDS store after the other: 9-18 us per edge tile against the 2.5 us of a whole one).
struct EdgeSlot {
    int s;         // stream
    int wo;        // byte offset of the segment in its stream's window
    int width;     // bytes per block of the stream
    int d;         // the stream's shift
    int lds_base;  // LDS address of window byte 0
    uint64_t g;    // offset from the transformed-side pointer of window byte 0 (may wrap)
};

// LDS address of window byte 0 of stream s: off_s * LA + LB * s + w_s * LC (halo image: T + H, kHaloPad, H; shifted image: T, 16, 0)
template <int FMT, bool SA, bool SC, int T, int LA, int LB, int LC>
__device__ __forceinline__ EdgeSlot edge_slot_of_image_byte(int o, const Shifts& sh, const uint64_t (&gb)[6])
{
    constexpr Streams S = make_streams(FMT, SA, SC);
    EdgeSlot e{0, 0, 1, 0, 0, 0};
    static_for<0, S.n>([&](auto si) {
        constexpr int s = decltype(si)::value;
        constexpr int lo = S.off[s] * T, hi = lo + S.width[s] * T;
        if (o >= lo && o < hi) {
            e.s = s;
            e.wo = o - lo;
            e.width = S.width[s];
            e.d = sh.d[s];
            e.lds_base = S.off[s] * LA + LB * s + S.width[s] * LC;
            e.g = gb[s];
        }
    });
    return e;
}

// the k-th extra segment behind stream s_sel's window
template <int FMT, bool SA, bool SC, int T, int LA, int LB, int LC>
__device__ __forceinline__ EdgeSlot edge_slot_behind_window(int s_sel, int k, const Shifts& sh, const uint64_t (&gb)[6])
{
    constexpr Streams S = make_streams(FMT, SA, SC);
    EdgeSlot e{0, 0, 1, 0, 0, 0};
    static_for<0, S.n>([&](auto si) {
        constexpr int s = decltype(si)::value;
        if (s == s_sel) {
            e.s = s;
            e.wo = S.width[s] * T + 16 * k;
            e.width = S.width[s];
            e.d = sh.d[s];
            e.lds_base = S.off[s] * LA + LB * s + S.width[s] * LC;
            e.g = gb[s];
        }
    });
    return e;
}

// The forward edge tile: halo tile `tile` of a range of sh.range_blocks blocks, of which it owns min(T, what is left) -- none
// at all when the range is a whole number of tiles and only the last d_s bytes of every stream are still to be written.
// Tile 0 (no halo: the blocks in front of it may not exist) writes every stream from its first byte; a tile that is followed
// by another one stops at the end of its window, the last one at the end of the stream -- up to 63 bytes past its window,
// the extra segments lanes 0 .. 4 n - 1 look after.
template <int FMT, int VARIANT, bool SA, bool SC, int NORM, bool NAT, int THREADS = 256>
__device__ __forceinline__ void fwd_halo_edge_tile(const uint8_t* __restrict__ aos, uint8_t* __restrict__ soa, const Shifts& sh,
                                                   uint64_t tile, uint8_t* lds)
{
    constexpr Streams S = make_streams(FMT, SA, SC);
    constexpr int T = tile_blocks(FMT, THREADS);
    constexpr int H = kHaloBlocks;
    constexpr int PV = 16 / fmt_block(FMT);   // blocks per 16-byte vector
    const int t = threadIdx.x;
    int base[6];
#pragma unroll
    for (int s = 0; s < 6; ++s)
        base[s] = s < S.n ? S.off[s] * (T + H) + kHaloPad * s + sh.d[s] + S.width[s] * H : 0;

    const uint64_t left = sh.range_blocks - tile * (uint64_t)T;       // the host launches no tile behind the range
    const int own = left < (uint64_t)T ? (int)left : T;
    const uint8_t* tile_aos = aos + tile * (THREADS * 16);
    const int hv = sh.halo_vecs;
    const bool has_halo = tile > 0 && t < hv;
    const bool whole_vec = (t + 1) * PV <= own;
    const bool half_vec = PV == 2 && !whole_vec && t * PV < own;   // BC1, odd count: the vector's second block does not exist
    u32x4 q = {0u, 0u, 0u, 0u}, qh = {0u, 0u, 0u, 0u};
    if (whole_vec)
        q = *reinterpret_cast<const u32x4*>(tile_aos + t * 16);
    if (half_vec) {   // (its 8 bytes may lie outside the caller's buffer)
        const u32x2 h = *reinterpret_cast<const u32x2*>(tile_aos + t * 16);
        q = u32x4{h.x, h.y, 0u, 0u};
    }
    if (has_halo)
        qh = *reinterpret_cast<const u32x4*>(tile_aos - hv * 16 + t * 16);
    if (has_halo)
        scatter_shifted<FMT, VARIANT, SA, SC, NAT>(lds, t - hv, normalize_vector<FMT, NORM>(qh), base);
    if (whole_vec || half_vec)
        scatter_shifted<FMT, VARIANT, SA, SC, NAT>(lds, t, normalize_vector<FMT, NORM>(q), base);
    __syncthreads();

    uint64_t gb[6];
    slice_bases<FMT, SA, SC, T>(tile, sh, gb);
    // the stream's bytes of the range that are this tile's, in window coordinates: from d_s on in tile 0 (nothing in front of the
    // range is ours), up to the end of the window when another tile follows, else to the stream's last byte
    int vlo[6], vhi[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int d = s < S.n ? sh.d[s] : 0, w = s < S.n ? S.width[s] : 0;
        vlo[s] = tile == 0 ? d : 0;
        vhi[s] = own == T ? w * T : d + w * own;
    }
    if constexpr (THREADS == 256) {
        switch (__builtin_amdgcn_readfirstlane(t >> 6)) {
        case 0: halo_copy_out_wave<FMT, SA, SC, 0, true>(soa, lds, t, gb, sh, vlo, vhi); break;
        case 1: halo_copy_out_wave<FMT, SA, SC, 1, true>(soa, lds, t, gb, sh, vlo, vhi); break;
        case 2: halo_copy_out_wave<FMT, SA, SC, 2, true>(soa, lds, t, gb, sh, vlo, vhi); break;
        default: halo_copy_out_wave<FMT, SA, SC, 3, true>(soa, lds, t, gb, sh, vlo, vhi); break;
        }
    } else {
        for_this_wave<THREADS / 64>(t, [&](auto wi) {
            halo_copy_out_wave<FMT, SA, SC, decltype(wi)::value, true, THREADS>(soa, lds, t, gb, sh, vlo, vhi);
        });
    }
    constexpr int XS = kHaloAlign / 16;   // up to kHaloAlign - 1 bytes of a stream lie behind its window: XS more segments per stream
    if (t < XS * S.n) {
        const EdgeSlot e = edge_slot_behind_window<FMT, SA, SC, T, T + H, kHaloPad, H>(t / XS, t % XS, sh, gb);
        int hi_s = 0;
#pragma unroll
        for (int s = 0; s < 6; ++s)
            hi_s = s == e.s ? vhi[s] : hi_s;
        const int hi = hi_s - e.wo < 16 ? hi_s - e.wo : 16;
        if (hi > 0) {
            uint8_t* la = lds + e.lds_base + e.wo;
            uint8_t* g = soa + e.g + (uint64_t)(int64_t)e.wo;
            if (hi != 16)
                copy_segment_bytes(g, la, 0, hi);
            else
                __builtin_nontemporal_store(*reinterpret_cast<u32x4*>(la), reinterpret_cast<u32x4*>(g));
        }
    }
}

// Workgroups [0, sh.full_tiles) are whole tiles (tile 0 through the edge body: it has a head to write); a workgroup behind
// them, when the launch has one, is the edge tile at the end of the range.
template <int FMT, int VARIANT, bool SA, bool SC, int NORM, bool NAT, int THREADS = 256>
__global__ void __launch_bounds__(THREADS)
fwd_tiled_halo(const uint8_t* __restrict__ aos_arg, uint8_t* __restrict__ soa_arg, uint64_t total_blocks, uint64_t first_block,
               Shifts sh_arg)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[halo_lds_bytes<FMT, THREADS>()];
    WG_MARK(0);
    const uint32_t wg = blockIdx.x;
    const Shifts sh = shifts_fetched_at_once(sh_arg);
    const uint8_t* __restrict__ aos = fetched_now(aos_arg);
    uint8_t* __restrict__ soa = fetched_now(soa_arg);
    const bool whole = wg < sh.full_tiles;
    // identity tile order: the halo is read again by the next tile, on another XCD, out of the memory-side cache (temporal loads)
    const uint64_t tile = !whole ? (uint64_t)sh.full_tiles : shifts_xcd_contiguous(sh, false) ? xcd_contiguous_tile(wg, sh.full_tiles) : (uint64_t)wg;
    const bool edge = !whole || tile == 0;
    if (edge)
        fwd_halo_edge_tile<FMT, VARIANT, SA, SC, NORM, NAT, THREADS>(aos, soa, sh, tile, lds);
    else
        fwd_halo_tile<FMT, VARIANT, SA, SC, NORM, NAT, THREADS>(aos, soa, total_blocks, first_block, sh, tile, lds);
}

// Loads of wave W of an inverse shifted tile: lane t fetches the aligned 16-byte segment that holds image byte 16 t.
// Which streams a wave's 1 KiB of the image can meet is known at compile time (BC3 with both splits: wave 0 meets the two
// alpha endpoint streams and the start of the alpha indices, wave 1 the indices only, wave 2 the two colour streams, wave 3
// the colour indices), so the per-lane select over the streams shrinks to those, and the extra partial last segment of
// every stream -- lanes 0 .. n-1 -- is wave 0's business alone.  The kernel is bound by its instruction stream (a wave
// instruction takes the SIMD four cycles: the first form ran 146 VALU + 114 SALU per wave against 67 + 36 in the aligned
// tile, PMC, profiles/r02_a_shift_pmc.txt; with the per-wave loads 69 + 107, r02_b_shift_probe.txt), which is why this
// matters -- and why what is left of the distance to the aligned tiles is not instructions any more.
// A load may fetch the whole aligned segment even when only part of it belongs to this tile's slice: the other bytes land
// in the stream's LDS padding.  Only a segment that pokes outside the transformed buffer itself (first tile of the first
// stream, last tile of the last stream) is fetched piecewise.  All loads of the wave are issued before the first wait.
template <int FMT, bool SA, bool SC, int W, int THREADS = 256>
__device__ __forceinline__ void inv_shift_load_wave(const uint8_t* __restrict__ soa, uint8_t* lds, int t,
                                                    const uint64_t (&gb)[6], const Shifts& sh, uint64_t total_bytes)
{
    constexpr Streams S = make_streams(FMT, SA, SC);
    constexpr int T = tile_blocks(FMT, THREADS);
    constexpr int wave_lo = W * 1024, wave_hi = wave_lo + 1024;
    const int o = t * 16;
    int la = 0, k = 0, shift = 0;
    uint64_t g = 0;
    static_for<0, S.n>([&](auto si) {
        constexpr int s = decltype(si)::value;
        constexpr int lo = S.off[s] * T;
        constexpr int hi = lo + S.width[s] * T;
        if constexpr (lo < wave_hi && hi > wave_lo) {
            constexpr bool only = lo <= wave_lo && hi >= wave_hi;   // the whole wave sits in this stream
            if (only || (o >= lo && o < hi)) {
                la = lo + 16 * s + (o - lo);
                g = gb[s] + (uint64_t)(o - lo);
                k = (o - lo) >> 4;
                shift = sh.d[s];
            }
        }
    });
    // g is an offset from soa; a head segment of stream 0 may start before the buffer (wraps to a huge value)
    const bool main_inside = g + 16 <= total_bytes;
    // Everything both loads need is computed first and both destinations are cleared before the first load is issued: with
    // the clearing of v_tail between the two loads the compiler (28 VGPRs instead of 48 once the kernel around this tile got
    // smaller) put an s_waitcnt vmcnt(0) between them -- wave 0 of every workgroup then waited out a whole memory round
    // trip before asking for its tail segments: BC3 inverse on odd counts 0.77 -> 0.70 (profiles/r04_batch_edge_tiles.txt).
    int la_t = 0, shift_t = 0;
    uint64_t g_t = 0;
    bool has_tail = false, tail_inside = false;
    if constexpr (W == 0) {
        // the extra, partial last segment of stream t (lanes 0..n-1), selected by data, not by control flow
#pragma unroll
        for (int ss = 0; ss < S.n; ++ss) {
            if (ss == t) {
                const int bytes = S.width[ss] * T;
                la_t = S.off[ss] * T + 16 * ss + bytes;
                shift_t = sh.d[ss];
                g_t = gb[ss] + bytes;
            }
        }
        has_tail = t < S.n && shift_t > 0;
        tail_inside = g_t + 16 <= total_bytes;
    }
    u32x4 v_main = {0, 0, 0, 0}, v_tail = {0, 0, 0, 0};
    // both destinations cleared HERE, not between the loads -- and the tail's address and predicates made values of this point too
    // (with 128-lane tiles the compiler sank their computation behind the first load and put the wait in front of it)
    int tail_flags = (has_tail ? 1 : 0) | (tail_inside ? 2 : 0);
    asm volatile("" : "+v"(v_main), "+v"(v_tail), "+v"(g_t), "+v"(tail_flags));
    has_tail = (tail_flags & 1) != 0;
    tail_inside = (tail_flags & 2) != 0;
    if (main_inside)
        v_main = gload16(soa + g);
    if constexpr (W == 0) {
        // (Making these two loads branch-free -- every lane also loading a "tail", lanes without one re-reading their main
        // segment -- cost 0.08 of peak: the second load instruction is not free even when it hits L1.)
        if (has_tail && tail_inside)
            v_tail = gload16(soa + g_t);
        if (has_tail) {
            if (tail_inside)
                lds_at<u32x4>(lds, la_t) = v_tail;
            else
                copy_partial_segment<false>(lds + la_t, soa + g_t, 0, shift_t);
        }
    }
    if (main_inside)
        lds_at<u32x4>(lds, la) = v_main;
    else
        copy_partial_segment<false>(lds + la, soa + g, (k == 0) ? shift : 0, (k == 0) ? 16 : shift);
}

// one shifted tile, inverse; `lds`: shift_lds_bytes(1, THREADS)
template <int FMT, int VARIANT, bool SA, bool SC, int THREADS = 256>
__device__ __forceinline__ void inv_shift_tile(const uint8_t* __restrict__ soa, uint8_t* __restrict__ aos,
                                               uint64_t total_blocks, uint64_t /*first_block: in sh.gbase*/, const Shifts& sh, uint64_t tile,
                                               uint8_t* lds)
{
    constexpr Streams S = make_streams(FMT, SA, SC);
    constexpr int T = tile_blocks(FMT, THREADS);
    const int t = threadIdx.x;
    int base[6];
#pragma unroll
    for (int s = 0; s < 6; ++s)
        base[s] = s < S.n ? S.off[s] * T + 16 * s + sh.d[s] : 0;

    const uint64_t total_bytes = total_blocks * (uint64_t)fmt_block(FMT);
    uint64_t gb[6];
    slice_bases<FMT, SA, SC, T>(tile, sh, gb);
    if constexpr (THREADS == 256) {
        switch (__builtin_amdgcn_readfirstlane(t >> 6)) {
        case 0: inv_shift_load_wave<FMT, SA, SC, 0>(soa, lds, t, gb, sh, total_bytes); break;
        case 1: inv_shift_load_wave<FMT, SA, SC, 1>(soa, lds, t, gb, sh, total_bytes); break;
        case 2: inv_shift_load_wave<FMT, SA, SC, 2>(soa, lds, t, gb, sh, total_bytes); break;
        default: inv_shift_load_wave<FMT, SA, SC, 3>(soa, lds, t, gb, sh, total_bytes); break;
        }
    } else {
        for_this_wave<THREADS / 64>(t, [&](auto wi) {
            inv_shift_load_wave<FMT, SA, SC, decltype(wi)::value, THREADS>(soa, lds, t, gb, sh, total_bytes);
        });
    }
    __syncthreads();
    const u32x4 q = sh.natural ? gather_shifted<FMT, VARIANT, SA, SC, true>(lds, t, base)
                               : gather_shifted<FMT, VARIANT, SA, SC, false>(lds, t, base);
    gstore16_aos(aos, aos + tile * (THREADS * 16) + t * 16, q);
}

// The inverse edge tile: shifted tile `tile` (= sh.full_tiles) of a range whose last 1 .. T - 1 blocks it owns.  A segment is
// fetched when it holds bytes of those blocks' records: whole when it lies inside the transformed buffer (the bytes that
// belong to the next stream land in LDS nobody reads), byte by byte where it pokes out of the buffer.
template <int FMT, int VARIANT, bool SA, bool SC, int THREADS = 256>
__device__ __forceinline__ void inv_shift_edge_tile(const uint8_t* __restrict__ soa, uint8_t* __restrict__ aos,
                                                    uint64_t total_blocks, const Shifts& sh, uint64_t tile, uint8_t* lds)
{
    constexpr Streams S = make_streams(FMT, SA, SC);
    constexpr int T = tile_blocks(FMT, THREADS);
    constexpr int PV = 16 / fmt_block(FMT);
    const int t = threadIdx.x;
    int base[6];
#pragma unroll
    for (int s = 0; s < 6; ++s)
        base[s] = s < S.n ? S.off[s] * T + 16 * s + sh.d[s] : 0;
    const uint64_t left = sh.range_blocks - tile * (uint64_t)T;
    const int own = left < (uint64_t)T ? (int)left : T;
    const uint64_t total_bytes = total_blocks * (uint64_t)fmt_block(FMT);
    uint64_t gb[6];
    slice_bases<FMT, SA, SC, T>(tile, sh, gb);
    // which bytes of a segment belong to the records of this tile's blocks, and whether the segment can be fetched whole
    auto plan = [&](const EdgeSlot& e, int& lo, int& hi, bool& whole) {
        const int vhi = e.d + e.width * own;
        lo = e.d - e.wo > 0 ? e.d - e.wo : 0;
        hi = vhi - e.wo < 16 ? vhi - e.wo : 16;
        const uint64_t g = e.g + (uint64_t)(int64_t)e.wo;   // wraps for a head segment in front of the buffer
        whole = total_bytes >= 16 && g <= total_bytes - 16;
    };
    const EdgeSlot em = edge_slot_of_image_byte<FMT, SA, SC, T, T, 16, 0>(t * 16, sh, gb);
    // the extra segment behind stream t's slice (a slice shifted by d_s reaches up to 15 bytes into it)
    const EdgeSlot ex = edge_slot_behind_window<FMT, SA, SC, T, T, 16, 0>(t, 0, sh, gb);
    int mlo, mhi, xlo = 0, xhi = 0;
    bool mwhole, xwhole = false;
    plan(em, mlo, mhi, mwhole);
    if (t < S.n)
        plan(ex, xlo, xhi, xwhole);
    const uint8_t* gm = soa + em.g + (uint64_t)(int64_t)em.wo;
    const uint8_t* gx = soa + ex.g + (uint64_t)(int64_t)ex.wo;
    u32x4 vm = {0u, 0u, 0u, 0u}, vx = {0u, 0u, 0u, 0u};
    if (mhi > mlo && mwhole)
        vm = *reinterpret_cast<const u32x4*>(gm);
    if (xhi > xlo && xwhole)
        vx = *reinterpret_cast<const u32x4*>(gx);
    if (mhi > mlo) {
        if (mwhole)
            *reinterpret_cast<u32x4*>(lds + em.lds_base + em.wo) = vm;
        else
            copy_segment_bytes(lds + em.lds_base + em.wo, gm, mlo, mhi);
    }
    if (xhi > xlo) {
        if (xwhole)
            *reinterpret_cast<u32x4*>(lds + ex.lds_base + ex.wo) = vx;
        else
            copy_segment_bytes(lds + ex.lds_base + ex.wo, gx, xlo, xhi);
    }
    // vmcnt(0), spelled out.  Every load above has been waited for on the path that issued it, but not on every path the compiler
    // sees, and in a kernel that holds this tile AND a whole tile (the two are alternatives, yet the structurised control flow falls
    // through one into the other) the whole tile inherited "loads outstanding" and got a wait between ITS two loads
    // (tests/test_isa_invariants.py; BC3 inverse 0.77 -> 0.70 when that happened in round 4).
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (t * PV < own) {
        const u32x4 q = sh.natural ? gather_shifted<FMT, VARIANT, SA, SC, true>(lds, t, base)
                                   : gather_shifted<FMT, VARIANT, SA, SC, false>(lds, t, base);
        uint8_t* out = aos + tile * (THREADS * 16) + t * 16;
        if ((t + 1) * PV <= own)
            __builtin_nontemporal_store(q, reinterpret_cast<u32x4*>(out));
        else   // BC1, odd count: only the vector's first block exists
            __builtin_nontemporal_store(u32x2{q.x, q.y}, reinterpret_cast<u32x2*>(out));
    }
}

// Workgroups [0, sh.full_tiles) are whole tiles; a workgroup behind them, when the launch has one, is the edge tile.
template <int FMT, int VARIANT, bool SA, bool SC, int THREADS = 256>
__global__ void __launch_bounds__(THREADS)
inv_tiled_shift(const uint8_t* __restrict__ soa_arg, uint8_t* __restrict__ aos_arg, uint64_t total_blocks, uint64_t first_block,
                Shifts sh_arg)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[shift_lds_bytes(1, THREADS)];
    const uint32_t wg = blockIdx.x;
    const Shifts sh = shifts_fetched_at_once(sh_arg);
    const uint8_t* __restrict__ soa = fetched_now(soa_arg);
    uint8_t* __restrict__ aos = fetched_now(aos_arg);
    asm("" : "+s"(total_blocks));
    const bool whole = wg < sh.full_tiles;
    // XCD-contiguous tile order: neighbouring tiles share their first and last line
    const uint64_t tile = !whole ? (uint64_t)sh.full_tiles : shifts_xcd_contiguous(sh, true) ? xcd_contiguous_tile(wg, sh.full_tiles) : (uint64_t)wg;
    const bool edge = !whole;
    if (edge)
        inv_shift_edge_tile<FMT, VARIANT, SA, SC, THREADS>(soa, aos, total_blocks, sh, tile, lds);
    else
        inv_shift_tile<FMT, VARIANT, SA, SC, THREADS>(soa, aos, total_blocks, first_block, sh, tile, lds);
}

// ------------------------------------------------------------------------------------------------
// Byte-granular accessors (the synthetic-data fill kernel; the element-granular kernel of the experiments build).
// ------------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void store_bytes(uint8_t* p, uint64_t v, bool natural)
{
    // W in {1,2,4,8}; `natural` = p is W-aligned (uniform per stream)
    if (natural) {
        if constexpr (W == 1) *p = (uint8_t)v;
        if constexpr (W == 2) *reinterpret_cast<uint16_t*>(p) = (uint16_t)v;
        if constexpr (W == 4) *reinterpret_cast<uint32_t*>(p) = (uint32_t)v;
        if constexpr (W == 8) *reinterpret_cast<uint64_t*>(p) = v;
    } else {
#pragma unroll
        for (int i = 0; i < W; ++i)
            p[i] = (uint8_t)(v >> (8 * i));
    }
}

template <int W>
__device__ __forceinline__ uint64_t load_bytes(const uint8_t* p, bool natural)
{
    if (natural) {
        if constexpr (W == 1) return *p;
        if constexpr (W == 2) return *reinterpret_cast<const uint16_t*>(p);
        if constexpr (W == 4) return *reinterpret_cast<const uint32_t*>(p);
        if constexpr (W == 8) return *reinterpret_cast<const uint64_t*>(p);
    }
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < W; ++i)
        v |= (uint64_t)p[i] << (8 * i);
    return v;
}

__device__ __forceinline__ bool aligned_to(const void* p, int a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }


}  // namespace dxtlt

#ifdef DXTLT_EXPERIMENTS
#include "bcn_experiments.h"
#endif
