// bc7_kernels.hip -- gfx950 kernels of the BC7 granule-sorted field split, version 1 (docs/BC7_FORMAT.md).
//
// A format of this build's own: the reference has no BC7 transform (core/dxt-lossless-transform-bc7/src/lib.rs:1-13);
// it documents the modes' bit fields (assets/research/dds-bc7-blocks.hexpat:286-654), which bc7_fields.h follows.
//
// What is computed.  The block array is cut into granules of 1024 blocks.  Inside a granule the blocks are ordered by
// class (mode 0..7, then the reserved byte-0 == 0 encoding), blocks of one class keeping their order; a block's bit
// fields are regrouped into a 16-byte record (marker + header | p-bits + index bits | low parts of the endpoints | high
// nibbles of the endpoints), and the records leave as eight streams: record bytes 1..8, 9..10, 11, 12, 13, 14, 15 at
// the block's SORTED position, byte 0 (which carries the mode marker) in block order.  The first N - N % 1024 blocks
// form the main part, whose streams start at multiples of the granule size (so every slice is 128-byte aligned
// whatever N is); the last N % 1024 blocks form a tail part with the same streams over its own block count.
//
// How it maps to the machine.
//   * ONE pass: 16 bytes in, 16 bytes out per block, no workspace, no grand totals, no second read of the input
//     (version 0 placed blocks by global per-mode prefix sums: histogram pass + scatter pass = 3 x len of traffic).
//   * One workgroup of 1024 lanes = one granule = one block per lane.  Forward: coalesced 16-byte loads; class by
//     trailing zeros; rank inside the class by a 4-ballot wave match + mbcnt, per-wave class counts through a 9 x 16
//     table in LDS, one scan per wave (lanes 0..8, DPP row shifts) -> sorted position; the raw blocks go to LDS at their
//     sorted positions ("per-mode wavefront dispatch": after the barrier lane j holds sorted block j, so a wave's 64
//     blocks are of one mode except where two classes meet, and the mode switch below is wave-uniform -- the eight
//     field permutations, 40-120 vector instructions each, are not executed eight times per wave); records are written
//     into an LDS image laid out like the output; the image leaves as one aligned 16-byte streaming store per lane,
//     every wave writing 1 KiB of ONE stream (8 waves Q8, 2 waves Q2, one wave per byte stream): no per-lane stream
//     select at all.
//   * Inverse: the mirror -- slices in (1 KiB per wave), classes from the F stream, the same ranks, F bytes to their
//     sorted positions, records -> blocks in the sorted domain (wave-uniform modes again), blocks back to block order
//     through LDS, coalesced 16-byte stores.
//   * HBM-bound by design: 32 bytes of traffic per block; four workgroup barriers per granule.
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "bc7_fields.h"
#include "bc7_launch.h"
#include "streaming_store.h"

namespace dxtlt {
namespace bc7 {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kT = kGranule;          // blocks per granule == lanes per workgroup
constexpr int kWaves = kT / 64;       // 16
constexpr int kClasses = 9;
static_assert(kT == 1024, "one block per lane, the copy-out assigns whole waves to streams");

// stream s of a part of n blocks starts at byte off[s] * n and holds width[s] bytes per block:
//   s      0 (Q8)  1 (Q2)  2 (B0)  3 (B1)  4 (B2)  5 (B3)  6 (B4)  7 (F)
//   off    0       8       10      11      12      13      14      15
//   width  8       2       1       1       1       1       1       1

// LDS: raw blocks at sorted positions | image of the output | per-class per-wave counts | per-wave class bases | sorted F
constexpr int kLdsRaw = 0;
constexpr int kLdsImage = kLdsRaw + kT * 16;
constexpr int kLdsCounts = kLdsImage + kT * 16;               // uint16_t [9][16]
constexpr int kLdsBases = kLdsCounts + kClasses * kWaves * 2 + 32;  // uint16_t [16 waves][16]
constexpr int kLdsSortedF = kLdsBases + kWaves * 16 * 2;      // uint8_t [1024] (inverse)
constexpr int kLdsBytes = kLdsSortedF + kT;

__device__ __forceinline__ u32x4 gload16(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }

template <typename T>
__device__ __forceinline__ T& lds_at(uint8_t* lds, int byte_off)
{
    return *reinterpret_cast<T*>(lds + byte_off);
}

// Sorted position of this lane's block inside the granule.  cls: 0..8, or 9 for lanes beyond a tail part's blocks (they
// sort behind everything and are never stored).  Two barriers inside; the counts table must have been zeroed and a
// barrier passed before the call.
__device__ __forceinline__ int sorted_position(uint8_t* lds, int cls, int lane, int wave)
{
    // lanes of this wave with the same class: AND over the four class bits of (bit set ? ballot : ~ballot)
    uint64_t same = ~0ull;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool bit = (cls >> k) & 1;
        const uint64_t b = __ballot(bit);
        same &= bit ? b : ~b;
    }
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0));
    const int in_wave = __popcll(same);
    uint16_t* counts = reinterpret_cast<uint16_t*>(lds + kLdsCounts);
    if (rank == in_wave - 1 && cls < kClasses)
        counts[cls * kWaves + wave] = (uint16_t)in_wave;   // the class's last lane in the wave reports its count
    __syncthreads();

    // lanes 0..8 of every wave, lane = class c: blocks of class c in the waves before this one, and in all waves;
    // exclusive scan of the totals over the classes; the wave's base for class c
    {
        const int c = lane < kClasses ? lane : kClasses - 1;
        const u32x4 lo = lds_at<u32x4>(lds, kLdsCounts + c * (kWaves * 2));
        const u32x4 hi = lds_at<u32x4>(lds, kLdsCounts + c * (kWaves * 2) + 16);
        const uint32_t d[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};   // d[k] = counts of waves 2k, 2k + 1
        uint32_t all = 0, before = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            all += d[k];   // two 16-bit sums side by side; at most 1024 each, no carry between them
            const uint32_t mask = (2 * k < wave ? 0xFFFFu : 0u) | (2 * k + 1 < wave ? 0xFFFF0000u : 0u);
            before += d[k] & mask;
        }
        const int total = (int)((all & 0xFFFFu) + (all >> 16));
        const int prior = (int)((before & 0xFFFFu) + (before >> 16));
        // inclusive scan of `total` over lanes 0..15 of each row (classes sit in lanes 0..8): DPP row_shr 1, 2, 4, 8
        int x = lane < kClasses ? total : 0;
        x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
        if (lane < kClasses)
            lds_at<uint16_t>(lds, kLdsBases + wave * 32 + lane * 2) = (uint16_t)(x - total + prior);
    }
    // same wave, LDS operations complete in order: no barrier between the store above and this load
    const int base = cls < kClasses ? (int)lds_at<uint16_t>(lds, kLdsBases + wave * 32 + cls * 2) : 0;
    return base + rank;
}

// offset of image byte 16 * t inside a full granule's slices: wave -> stream (8 waves Q8, 2 waves Q2, 6 byte streams)
__device__ __forceinline__ uint64_t slice_offset_of_lane(int t, int wave, uint64_t part_blocks, uint64_t first_block_of_granule)
{
    const int s = wave < 8 ? 0 : wave < 10 ? 1 : wave - 8;   // wave-uniform
    const int off = s == 0 ? 0 : s == 1 ? 8 : s + 8;          // kOff[s]
    const int width = s == 0 ? 8 : s == 1 ? 2 : 1;            // kWidth[s]
    return (uint64_t)off * part_blocks + (uint64_t)width * first_block_of_granule + (uint64_t)(16 * t - off * kT);
}

// Forward.  aos: the range's first block.  Full granules (TAIL = false): soa = byte 0 of the part's streams, part_blocks
// = blocks of the part (a multiple of 1024), first_block = the range's first block inside the part (a multiple of 1024),
// gridDim.x = granules of the range.  TAIL: one workgroup, n = blocks of the tail part (< 1024), soa = its first byte.
template <bool TAIL>
__global__ void __launch_bounds__(kT)
bc7_forward(const uint8_t* __restrict__ aos, uint8_t* __restrict__ soa, uint64_t part_blocks, uint64_t first_block, int n_tail)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n = TAIL ? n_tail : kT;
    const uint64_t granule = blockIdx.x;
    const bool live = !TAIL || t < n;

    u32x4 q = {0, 0, 0, 0};
    if (live)
        q = gload16(aos + (granule * kT + (uint64_t)t) * 16);
    if (t < kClasses * kWaves)
        lds_at<uint16_t>(lds, kLdsCounts + 2 * t) = 0;
    __syncthreads();

    const B128 b = {{q.x, q.y, q.z, q.w}};
    const int cls = live ? block_class(q.x) : kClasses;
    const int pos = sorted_position(lds, cls, lane, wave);
    if (live) {
        lds_at<u32x4>(lds, kLdsRaw + 16 * pos) = q;
        lds_at<uint8_t>(lds, kLdsImage + 15 * n + t) = (uint8_t)record_byte0(b, cls);   // F: block order
    }
    __syncthreads();

    // sorted domain: lane j holds sorted block j; the class is the same across the wave except where two classes meet
    if (live) {
        const u32x4 s = lds_at<u32x4>(lds, kLdsRaw + 16 * t);
        const B128 sb = {{s.x, s.y, s.z, s.w}};
        const B128 r = record_of_block_any(sb, block_class(s.x));
        // record bytes 1..8 -> Q8, 9..10 -> Q2, 11..15 -> B0..B4
        lds_at<u32x2>(lds, kLdsImage + 8 * t) = u32x2{__builtin_amdgcn_alignbyte(r.d[1], r.d[0], 1), __builtin_amdgcn_alignbyte(r.d[2], r.d[1], 1)};
        lds_at<uint16_t>(lds, kLdsImage + 8 * n + 2 * t) = (uint16_t)(r.d[2] >> 8);
        lds_at<uint8_t>(lds, kLdsImage + 10 * n + t) = (uint8_t)(r.d[2] >> 24);
        lds_at<uint8_t>(lds, kLdsImage + 11 * n + t) = (uint8_t)r.d[3];
        lds_at<uint8_t>(lds, kLdsImage + 12 * n + t) = (uint8_t)(r.d[3] >> 8);
        lds_at<uint8_t>(lds, kLdsImage + 13 * n + t) = (uint8_t)(r.d[3] >> 16);
        lds_at<uint8_t>(lds, kLdsImage + 14 * n + t) = (uint8_t)(r.d[3] >> 24);
    }
    __syncthreads();

    if constexpr (TAIL) {
        // the tail part is one contiguous run of 16 n bytes with the image's own layout
        if (live)
            *reinterpret_cast<u32x4*>(soa + 16 * t) = lds_at<u32x4>(lds, kLdsImage + 16 * t);
    } else {
        const uint64_t o = slice_offset_of_lane(t, wave, part_blocks, first_block + granule * kT);
        store_streaming16(soa + o, lds_at<u32x4>(lds, kLdsImage + 16 * t));
    }
}

template <bool TAIL>
__global__ void __launch_bounds__(kT)
bc7_inverse(const uint8_t* __restrict__ soa, uint8_t* __restrict__ aos, uint64_t part_blocks, uint64_t first_block, int n_tail)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n = TAIL ? n_tail : kT;
    const uint64_t granule = blockIdx.x;
    const bool live = !TAIL || t < n;

    if constexpr (TAIL) {
        if (live)
            lds_at<u32x4>(lds, kLdsImage + 16 * t) = *reinterpret_cast<const u32x4*>(soa + 16 * t);
    } else {
        const uint64_t o = slice_offset_of_lane(t, wave, part_blocks, first_block + granule * kT);
        lds_at<u32x4>(lds, kLdsImage + 16 * t) = gload16(soa + o);
    }
    if (t < kClasses * kWaves)
        lds_at<uint16_t>(lds, kLdsCounts + 2 * t) = 0;
    __syncthreads();

    const uint32_t f = live ? lds_at<uint8_t>(lds, kLdsImage + 15 * n + t) : 0u;
    const int cls = live ? block_class(f) : kClasses;
    const int pos = sorted_position(lds, cls, lane, wave);
    if (live)
        lds_at<uint8_t>(lds, kLdsSortedF + pos) = (uint8_t)f;
    __syncthreads();

    if (live) {
        const uint32_t f2 = lds_at<uint8_t>(lds, kLdsSortedF + t);
        const u32x2 q8 = lds_at<u32x2>(lds, kLdsImage + 8 * t);
        const uint32_t q2 = lds_at<uint16_t>(lds, kLdsImage + 8 * n + 2 * t);
        const uint32_t b0 = lds_at<uint8_t>(lds, kLdsImage + 10 * n + t);
        const uint32_t b1 = lds_at<uint8_t>(lds, kLdsImage + 11 * n + t);
        const uint32_t b2 = lds_at<uint8_t>(lds, kLdsImage + 12 * n + t);
        const uint32_t b3 = lds_at<uint8_t>(lds, kLdsImage + 13 * n + t);
        const uint32_t b4 = lds_at<uint8_t>(lds, kLdsImage + 14 * n + t);
        B128 r;
        r.d[0] = f2 | (q8.x << 8);
        r.d[1] = (q8.x >> 24) | (q8.y << 8);
        r.d[2] = (q8.y >> 24) | (q2 << 8) | (b0 << 24);
        r.d[3] = b1 | (b2 << 8) | (b3 << 16) | (b4 << 24);
        const B128 blk = block_of_record_any(r, block_class(f2));
        lds_at<u32x4>(lds, kLdsRaw + 16 * t) = u32x4{blk.d[0], blk.d[1], blk.d[2], blk.d[3]};
    }
    __syncthreads();

    if (live)
        store_streaming16(aos + (granule * kT + (uint64_t)t) * 16, lds_at<u32x4>(lds, kLdsRaw + 16 * pos));
}

// ------------------------------------------------------------------------------------------------
// Host-side dispatch
// ------------------------------------------------------------------------------------------------
hipError_t launch_range(bool inverse, const void* src, void* dst, uint64_t total_blocks, uint64_t first_block,
                        uint64_t num_blocks, hipStream_t stream)
{
    if (num_blocks == 0)
        return hipSuccess;
    const uint64_t main_blocks = total_blocks - total_blocks % kT;
    const uint64_t tail = total_blocks - main_blocks;
    // a range starts on a granule and ends on one or at the end of the array
    if (first_block % kT != 0 || first_block > total_blocks || num_blocks > total_blocks - first_block ||
        ((first_block + num_blocks) % kT != 0 && first_block + num_blocks != total_blocks))
        return hipErrorInvalidValue;
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) != 0)
        return hipErrorInvalidValue;
    const uint8_t* aos = static_cast<const uint8_t*>(inverse ? dst : src);     // the range's first block
    const uint8_t* soa = static_cast<const uint8_t*>(inverse ? src : dst);     // byte 0 of the whole transformed buffer
    const uint64_t range_main = first_block >= main_blocks ? 0 : (first_block + num_blocks > main_blocks ? main_blocks : first_block + num_blocks) - first_block;
    // a launch of 2^32 or more threads is refused: at most 2^21 granules (32 GiB of blocks) per launch
    constexpr uint64_t kMaxGranules = 1ull << 21;
    for (uint64_t g0 = 0; g0 < range_main / kT; g0 += kMaxGranules) {
        const uint64_t ng = range_main / kT - g0 < kMaxGranules ? range_main / kT - g0 : kMaxGranules;
        const uint8_t* a = aos + g0 * kT * 16;
        if (inverse)
            hipLaunchKernelGGL(bc7_inverse<false>, dim3((unsigned)ng), dim3(kT), 0, stream, soa, const_cast<uint8_t*>(a),
                               main_blocks, first_block + g0 * kT, 0);
        else
            hipLaunchKernelGGL(bc7_forward<false>, dim3((unsigned)ng), dim3(kT), 0, stream, a, const_cast<uint8_t*>(soa),
                               main_blocks, first_block + g0 * kT, 0);
        if (hipError_t e = hipGetLastError(); e != hipSuccess)
            return e;
    }
    if (tail != 0 && first_block + num_blocks == total_blocks) {
        const uint8_t* a = aos + (main_blocks - first_block) * 16;   // first_block <= main_blocks here
        const uint8_t* s = soa + main_blocks * 16;
        if (inverse)
            hipLaunchKernelGGL(bc7_inverse<true>, dim3(1), dim3(kT), 0, stream, s, const_cast<uint8_t*>(a), tail, 0, (int)tail);
        else
            hipLaunchKernelGGL(bc7_forward<true>, dim3(1), dim3(kT), 0, stream, a, const_cast<uint8_t*>(s), tail, 0, (int)tail);
        if (hipError_t e = hipGetLastError(); e != hipSuccess)
            return e;
    }
    return hipSuccess;
}

hipError_t launch(bool inverse, const void* src, void* dst, uint64_t n_blocks, hipStream_t stream)
{
    return launch_range(inverse, src, dst, n_blocks, 0, n_blocks, stream);
}

}  // namespace bc7
}  // namespace dxtlt
