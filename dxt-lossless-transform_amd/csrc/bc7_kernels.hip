// bc7_kernels.hip -- gfx950 kernels of the BC7 granule-sorted field split, version 2 (docs/BC7_FORMAT.md).
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
//   * One workgroup = one granule: 256 lanes x 4 blocks per lane (lane t owns blocks t, t + 256, ...: coalesced).
//     Forward: 16-byte loads; class by trailing zeros; rank inside the class by a 3-ballot wave match + mbcnt per 64-block
//     segment, per-segment class counts through a 9 x 16 table in LDS, one scan per wave for its four segments (16-lane
//     rows, DPP row shifts) -> sorted position; the raw blocks go to LDS at their sorted positions ("per-mode wavefront
//     dispatch": after the barrier lane j holds sorted block j, so a wave's 64 blocks are of one mode except where two
//     classes meet, and the mode switch below is wave-uniform -- the eight field permutations, 40-90 vector instructions
//     each, are not executed eight times per wave); records are written into an LDS image laid out like the output; the
//     image leaves as one aligned 16-byte streaming store per lane, every wave-instruction writing 1 KiB of ONE stream
//     (8 segments Q8, 2 Q2, one per byte stream): no per-lane stream select at all.
//   * Inverse: the mirror -- slices in (1 KiB per wave instruction), classes from the F stream, the same ranks, F bytes
//     to their sorted positions, records -> blocks in the sorted domain (wave-uniform modes again), blocks back to block
//     order through LDS, coalesced 16-byte stores.
//   * 19 KiB of LDS per workgroup (raw blocks and stream image share one region), so eight workgroups = eight granules
//     in different phases per CU: the kernel's time does not depend on the mode mix any more --
//     what bounds it is how much of a granule's life (load, five barriers, store) overlaps with its neighbours'.
//     Persistent workgroups that prefetch the next granule were built three times (last: hand-kept vmcnt through inline
//     asm, LDS-only barriers) and lost every time: 0.58 against 0.73-0.76.
//   * 32 bytes of traffic per block, exactly (PMC).
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdlib>

#include "bc7_fields.h"
#include "bc7_launch.h"
#include "streaming_store.h"

namespace dxtlt {
namespace bc7 {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kT = kGranule;          // blocks per granule
constexpr int kSegments = kT / 64;    // 16 runs of 64 consecutive blocks: one wave instruction's worth each
constexpr int kClasses = 9;
static_assert(kT == 1024, "the copy-out assigns whole 64-block segments to streams");

// stream s of a part of n blocks starts at byte off[s] * n and holds width[s] bytes per block:
//   s      0 (Q8)  1 (Q2)  2 (B0)  3 (B1)  4 (B2)  5 (B3)  6 (B4)  7 (F)
//   off    0       8       10      11      12      13      14      15
//   width  8       2       1       1       1       1       1       1

// LDS.  Full granules: the raw blocks at their sorted positions and the image of the output's sorted streams (15 bytes per
// block) take turns in ONE 16 KiB region -- every lane has its blocks / records in registers before the region changes
// hands (one more barrier) -- and the F stream, which is written in block order while the raw blocks are being placed,
// has a region of its own: 19 KiB per workgroup instead of 35, eight workgroups of 256 lanes per CU instead of four.
//   data 16 KiB | F 1 KiB | per-class per-segment counts | per-segment class bases | sorted F 1 KiB (inverse)
// Tail parts (one workgroup per call, n < 1024 blocks): the image is one contiguous run of 16 n bytes, F at byte 15 n,
// in a region of its own behind the rest.
constexpr int kLdsRaw = 0;
constexpr int kLdsF = kLdsRaw + kT * 16;                               // uint8_t [1024], block order
constexpr int kLdsCounts = kLdsF + kT;                                // uint16_t [9][16]
constexpr int kLdsBases = kLdsCounts + kClasses * kSegments * 2 + 32;  // uint16_t [16 segments][16]
constexpr int kLdsSortedF = kLdsBases + kSegments * 16 * 2;           // uint8_t [1024] (inverse)
constexpr int kLdsTailImage = kLdsSortedF + kT;                       // 16 KiB, tail parts only
template <bool TAIL>
constexpr int lds_bytes() { return TAIL ? kLdsTailImage + kT * 16 : kLdsTailImage; }

__device__ __forceinline__ u32x4 gload16(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }

template <typename T>
__device__ __forceinline__ T& lds_at(uint8_t* lds, int byte_off)
{
    return *reinterpret_cast<T*>(lds + byte_off);
}

// Rank of this lane's block among the blocks of its class in its 64-block segment (= wave instruction), and the class's
// count in the segment: lanes with the same class = AND over the class bits of (bit set ? ballot : ~ballot).
// cls: 0..8, or 9 for lanes beyond a tail part's blocks.  Classes 8 and 9 are rare: the fourth class bit is only
// matched when some lane of the wave has it set (a scalar branch).
__device__ __forceinline__ void rank_in_segment(int cls, int& rank, int& count)
{
    uint32_t lo = 0xFFFFFFFFu, hi = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int sext = __builtin_amdgcn_sbfe(cls, k, 1);   // -1 when bit k is set, else 0
        const uint64_t b = __ballot(sext != 0);
        lo &= ~((uint32_t)b ^ (uint32_t)sext);               // bit set: b, else ~b
        hi &= ~((uint32_t)(b >> 32) ^ (uint32_t)sext);
    }
    const uint64_t high = __ballot(cls >= 8);
    if (high != 0) {
        const uint32_t m = cls >= 8 ? 0xFFFFFFFFu : 0u;
        lo &= ~((uint32_t)high ^ m);
        hi &= ~((uint32_t)(high >> 32) ^ m);
    }
    rank = (int)__builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0));
    count = __popc(lo) + __popc(hi);
}

// The same for a segment whose 64 blocks are all of classes 0..7 (the caller checks), with a quarter fewer vector
// instructions: a lane's class as a one-hot byte counter -- classes 0..3 in one dword, 4..7 in a second; at most 64 per
// byte, no carry -- and one inclusive wave scan per dword (four row shifts and two row broadcasts, each fused into its
// add).  The rank is the lane's own byte of its scan value minus one; lane 63's scan value holds every class's count, which
// lanes 0..7 write to the counts table (so no class's "last lane" has to be found).
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);    // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);    // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);    // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);    // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return x;
}

__device__ __forceinline__ void rank_by_scan(uint8_t* lds, int cls, int lane, int segment, int& rank)
{
    const uint64_t one = 1ull << (8 * cls);
    const uint32_t a = wave_scan_add((uint32_t)one), b = wave_scan_add((uint32_t)(one >> 32));
    rank = (int)__builtin_amdgcn_ubfe(cls < 4 ? a : b, 8 * cls, 8) - 1;   // the offset operand is taken modulo 32
    const uint32_t ta = __builtin_amdgcn_readlane(a, 63), tb = __builtin_amdgcn_readlane(b, 63);
    if (lane < kClasses)   // lanes 0..7: the segment's count of class `lane`; lane 8: class 8 is absent here (the caller checked)
        lds_at<uint16_t>(lds, kLdsCounts + lane * (kSegments * 2) + segment * 2) =
            lane < 8 ? (uint16_t)__builtin_amdgcn_ubfe(lane < 4 ? ta : tb, 8 * lane, 8) : (uint16_t)0;
}

// Rank of the lane's block inside its class in this segment, and the segment's class counts into the table.
template <bool TAIL>
__device__ __forceinline__ void rank_and_count(uint8_t* lds, int cls, int lane, int segment, int& rank)
{
    if (!TAIL && __ballot(cls >= 8) == 0) {
        rank_by_scan(lds, cls, lane, segment, rank);
    } else {
        // the segment's column of the table is written whole by this wave -- zeros first, then the counts that exist (LDS
        // operations of one wave complete in order) -- so the table needs no zero fill and no barrier in front of the ranks
        if (lane < kClasses)
            lds_at<uint16_t>(lds, kLdsCounts + lane * (kSegments * 2) + segment * 2) = 0;
        int count;
        rank_in_segment(cls, rank, count);
        if (rank == count - 1 && cls < kClasses)   // the class's last lane in the segment reports its count
            lds_at<uint16_t>(lds, kLdsCounts + cls * (kSegments * 2) + segment * 2) = (uint16_t)count;
    }
}

// Every wave turns the counts table into the class bases of ITS segments: 16-lane row r of the wave works on the wave's
// r-th segment (segment number r * WAVES + wave), lane c of the row on class c: blocks of class c in earlier segments
// and in all segments; exclusive scan of the totals over the classes (DPP row shifts stay inside a row).
// "Earlier segments" of row r = all of the segment groups 0..r-1 (WAVES segments each) plus the segments of group r
// below `wave`; the latter is the same masked sum for every group, with masks that depend on the wave number only
// (scalar registers) -- built per lane from the segment number it cost more vector instructions than everything else
// in this function.
template <int WAVES, int V>
__device__ __forceinline__ void segment_bases(uint8_t* lds, int lane, int wave)
{
    constexpr int P = WAVES / 2;   // pairs of 16-bit counts (dwords) per group
    static_assert(P * 2 == WAVES && P * V == 8, "a group's counts fill whole dwords");
    const int row = lane >> 4, c = (lane & 15) < kClasses ? (lane & 15) : kClasses - 1;
    const u32x4 lo = lds_at<u32x4>(lds, kLdsCounts + c * (kSegments * 2));
    const u32x4 hi = lds_at<u32x4>(lds, kLdsCounts + c * (kSegments * 2) + 16);
    const uint32_t d[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};   // d[k] = counts of segments 2k, 2k + 1
    uint32_t mask[P];
#pragma unroll
    for (int j = 0; j < P; ++j)
        mask[j] = (2 * j < wave ? 0xFFFFu : 0u) | (2 * j + 1 < wave ? 0xFFFF0000u : 0u);   // scalar
    // two 16-bit sums side by side; at most 1024 each, no carry between them
    uint32_t group_sum[V], group_below[V];
#pragma unroll
    for (int g = 0; g < V; ++g) {
        group_sum[g] = 0;
        group_below[g] = 0;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            group_sum[g] += d[g * P + j];
            group_below[g] += d[g * P + j] & mask[j];
        }
    }
    uint32_t all = 0, before = 0, running = 0;
#pragma unroll
    for (int g = 0; g < V; ++g) {
        if (row == g)
            before = running + group_below[g];
        running += group_sum[g];
    }
    all = running;
    const int total = (int)((all & 0xFFFFu) + (all >> 16));
    const int prior = (int)((before & 0xFFFFu) + (before >> 16));
    int x = (lane & 15) < kClasses ? total : 0;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);   // row_shr:1, 2, 4, 8: inclusive scan inside the row
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
    if (row < V && (lane & 15) < kClasses)
        lds_at<uint16_t>(lds, kLdsBases + (row * WAVES + wave) * 32 + (lane & 15) * 2) = (uint16_t)(x - total + prior);
}

// byte offset, from the part's first byte, of image byte 16 * j of a full granule (j = 0..1023): the image's 64-block
// segment -> stream (8 segments Q8, 2 segments Q2, one per byte stream); wave-uniform
__device__ __forceinline__ uint64_t slice_offset(int j, int segment, uint64_t part_blocks, uint64_t first_block_of_granule)
{
    const int s = segment < 8 ? 0 : segment < 10 ? 1 : segment - 8;
    const int off = s == 0 ? 0 : s == 1 ? 8 : s + 8;
    const int width = s == 0 ? 8 : s == 1 ? 2 : 1;
    return (uint64_t)off * part_blocks + (uint64_t)width * first_block_of_granule + (uint64_t)(16 * j - off * kT);
}

// One granule, forward.  src = the granule's first block; soa, part_blocks as below; granule_first = the granule's first
// block inside the part (a multiple of 1024).
template <int LANES, bool TAIL>
__device__ __forceinline__ void bc7_forward_granule(const uint8_t* __restrict__ src, uint8_t* __restrict__ soa,
                                                    uint64_t part_blocks, uint64_t granule_first, int n_tail)
{
    constexpr int V = kT / LANES, WAVES = LANES / 64;
    static_assert(V >= 1 && V <= 4 && V * LANES == kT, "four 16-lane rows per wave: at most four segments per wave");
    __shared__ __attribute__((aligned(16))) uint8_t lds[lds_bytes<TAIL>()];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n = TAIL ? n_tail : kT;
    const int image = TAIL ? kLdsTailImage : kLdsRaw;            // sorted streams: record bytes 1..15
    const int image_f = TAIL ? kLdsTailImage + 15 * n : kLdsF;   // F stream, block order

    u32x4 q[V];
    int cls[V], rank[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        q[v] = u32x4{0, 0, 0, 0};
        if (!TAIL || v * LANES + t < n)
            q[v] = gload16(src + (v * LANES + t) * 16);
    }
    // no barrier between the loads and the ranks: every wave writes the whole table column of each of its segments itself
    // (rank_and_count), and the rank of the blocks that have arrived is computed under the loads still in flight
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const bool live = !TAIL || v * LANES + t < n;
        cls[v] = live ? block_class(q[v].x) : kClasses;
        rank_and_count<TAIL>(lds, cls[v], lane, v * WAVES + wave, rank[v]);
    }
    __syncthreads();

    segment_bases<WAVES, V>(lds, lane, wave);
#pragma unroll
    for (int v = 0; v < V; ++v) {
        if (cls[v] < kClasses) {
            // same wave, LDS operations complete in order: no barrier between segment_bases' stores and this load
            const int pos = (int)lds_at<uint16_t>(lds, kLdsBases + (v * WAVES + wave) * 32 + cls[v] * 2) + rank[v];
            const B128 b = {{q[v].x, q[v].y, q[v].z, q[v].w}};
            lds_at<u32x4>(lds, kLdsRaw + 16 * pos) = q[v];
            lds_at<uint8_t>(lds, image_f + v * LANES + t) = (uint8_t)record_byte0(b, cls[v]);   // F: block order
        }
    }
    __syncthreads();

    // sorted domain: lane t holds sorted blocks t, t + LANES, ...; the class is the same across a wave's 64 blocks except
    // where two classes meet
    u32x4 sorted[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int j = v * LANES + t;
        sorted[v] = u32x4{0, 0, 0, 0};
        if (!TAIL || j < n)
            sorted[v] = lds_at<u32x4>(lds, kLdsRaw + 16 * j);
    }
    if constexpr (!TAIL)
        __syncthreads();   // the raw blocks are in registers: their region becomes the image
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int j = v * LANES + t;
        if (!TAIL || j < n) {
            const B128 sb = {{sorted[v].x, sorted[v].y, sorted[v].z, sorted[v].w}};
            const B128 r = record_of_block_any(sb, block_class(sorted[v].x));
            // record bytes 1..8 -> Q8, 9..10 -> Q2, 11..15 -> B0..B4
            lds_at<u32x2>(lds, image + 8 * j) = u32x2{__builtin_amdgcn_alignbyte(r.d[1], r.d[0], 1), __builtin_amdgcn_alignbyte(r.d[2], r.d[1], 1)};
            lds_at<uint16_t>(lds, image + 8 * n + 2 * j) = (uint16_t)(r.d[2] >> 8);
            lds_at<uint8_t>(lds, image + 10 * n + j) = (uint8_t)(r.d[2] >> 24);
            lds_at<uint8_t>(lds, image + 11 * n + j) = (uint8_t)r.d[3];
            lds_at<uint8_t>(lds, image + 12 * n + j) = (uint8_t)(r.d[3] >> 8);
            lds_at<uint8_t>(lds, image + 13 * n + j) = (uint8_t)(r.d[3] >> 16);
            lds_at<uint8_t>(lds, image + 14 * n + j) = (uint8_t)(r.d[3] >> 24);
        }
    }
    __syncthreads();

#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int j = v * LANES + t;
        if constexpr (TAIL) {
            // the tail part is one contiguous run of 16 n bytes with the image's own layout
            if (j < n)
                *reinterpret_cast<u32x4*>(soa + 16 * j) = lds_at<u32x4>(lds, image + 16 * j);
        } else {
            const int segment = v * WAVES + wave;   // segment 15 is the F stream
            const uint64_t o = slice_offset(j, segment, part_blocks, granule_first);
            store_streaming16(soa + o, lds_at<u32x4>(lds, segment == 15 ? kLdsF + 16 * (j - 15 * 64) : image + 16 * j));
        }
    }
}

// One granule, inverse: dst = where the granule's first block goes.
template <int LANES, bool TAIL>
__device__ __forceinline__ void bc7_inverse_granule(const uint8_t* __restrict__ soa, uint8_t* __restrict__ dst,
                                                    uint64_t part_blocks, uint64_t granule_first, int n_tail)
{
    constexpr int V = kT / LANES, WAVES = LANES / 64;
    static_assert(V >= 1 && V <= 4 && V * LANES == kT, "four 16-lane rows per wave: at most four segments per wave");
    __shared__ __attribute__((aligned(16))) uint8_t lds[lds_bytes<TAIL>()];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n = TAIL ? n_tail : kT;
    const int image = TAIL ? kLdsTailImage : kLdsRaw;
    const int image_f = TAIL ? kLdsTailImage + 15 * n : kLdsF;

    u32x4 in[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int j = v * LANES + t;
        in[v] = u32x4{0, 0, 0, 0};
        if constexpr (TAIL) {
            if (j < n)
                in[v] = *reinterpret_cast<const u32x4*>(soa + 16 * j);
        } else {
            in[v] = gload16(soa + slice_offset(j, v * WAVES + wave, part_blocks, granule_first));
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int j = v * LANES + t;
        if constexpr (TAIL) {
            if (j < n)
                lds_at<u32x4>(lds, image + 16 * j) = in[v];
        } else {
            lds_at<u32x4>(lds, v * WAVES + wave == 15 ? kLdsF + 16 * (j - 15 * 64) : image + 16 * j) = in[v];
        }
    }
    __syncthreads();

    int cls[V], rank[V];
    uint32_t f[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const bool live = !TAIL || v * LANES + t < n;
        f[v] = live ? lds_at<uint8_t>(lds, image_f + v * LANES + t) : 0u;
        cls[v] = live ? block_class(f[v]) : kClasses;
        rank_and_count<TAIL>(lds, cls[v], lane, v * WAVES + wave, rank[v]);
    }
    __syncthreads();

    segment_bases<WAVES, V>(lds, lane, wave);
    int pos[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        pos[v] = 0;
        if (cls[v] < kClasses) {
            pos[v] = (int)lds_at<uint16_t>(lds, kLdsBases + (v * WAVES + wave) * 32 + cls[v] * 2) + rank[v];
            lds_at<uint8_t>(lds, kLdsSortedF + pos[v]) = (uint8_t)f[v];
        }
    }
    __syncthreads();

    // sorted domain: the record of sorted block j from the streams' image and the sorted F bytes
    B128 rec[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int j = v * LANES + t;
        rec[v] = B128{{0, 0, 0, 0}};
        if (!TAIL || j < n) {
            const uint32_t f2 = lds_at<uint8_t>(lds, kLdsSortedF + j);
            const u32x2 q8 = lds_at<u32x2>(lds, image + 8 * j);
            const uint32_t q2 = lds_at<uint16_t>(lds, image + 8 * n + 2 * j);
            const uint32_t b0 = lds_at<uint8_t>(lds, image + 10 * n + j);
            const uint32_t b1 = lds_at<uint8_t>(lds, image + 11 * n + j);
            const uint32_t b2 = lds_at<uint8_t>(lds, image + 12 * n + j);
            const uint32_t b3 = lds_at<uint8_t>(lds, image + 13 * n + j);
            const uint32_t b4 = lds_at<uint8_t>(lds, image + 14 * n + j);
            rec[v].d[0] = f2 | (q8.x << 8);
            rec[v].d[1] = (q8.x >> 24) | (q8.y << 8);
            rec[v].d[2] = (q8.y >> 24) | (q2 << 8) | (b0 << 24);
            rec[v].d[3] = b1 | (b2 << 8) | (b3 << 16) | (b4 << 24);
        }
    }
    if constexpr (!TAIL)
        __syncthreads();   // the records are in registers: the image's region takes the blocks
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int j = v * LANES + t;
        if (!TAIL || j < n) {
            const B128 blk = block_of_record_any(rec[v], block_class(rec[v].d[0]));
            lds_at<u32x4>(lds, kLdsRaw + 16 * j) = u32x4{blk.d[0], blk.d[1], blk.d[2], blk.d[3]};
        }
    }
    __syncthreads();

#pragma unroll
    for (int v = 0; v < V; ++v)
        if (cls[v] < kClasses)
            store_streaming16(dst + (uint64_t)(v * LANES + t) * 16, lds_at<u32x4>(lds, kLdsRaw + 16 * pos[v]));
}

// Forward.  LANES lanes per workgroup, V = 1024 / LANES blocks per lane.  aos: the range's first block.  Full granules
// (TAIL = false): soa = byte 0 of the part's streams, part_blocks = blocks of the part (a multiple of 1024), first_block =
// the range's first block inside the part (a multiple of 1024), gridDim.x = granules of the range.  TAIL: one workgroup,
// n = blocks of the tail part (< 1024), soa = its first byte.
template <int LANES, bool TAIL>
__global__ void __launch_bounds__(LANES)
bc7_forward(const uint8_t* __restrict__ aos, uint8_t* __restrict__ soa, uint64_t part_blocks, uint64_t first_block, int n_tail)
{
    const uint64_t granule = blockIdx.x;
    bc7_forward_granule<LANES, TAIL>(aos + granule * (kT * 16), soa, part_blocks, first_block + granule * kT, n_tail);
}

template <int LANES, bool TAIL>
__global__ void __launch_bounds__(LANES)
bc7_inverse(const uint8_t* __restrict__ soa, uint8_t* __restrict__ aos, uint64_t part_blocks, uint64_t first_block, int n_tail)
{
    const uint64_t granule = blockIdx.x;
    bc7_inverse_granule<LANES, TAIL>(soa, aos + granule * (kT * 16), part_blocks, first_block + granule * kT, n_tail);
}

// Many buffers per launch (dxtlt_transform_batch_device / _host with format 7): workgroup b finds its buffer in the
// table -- `coarse[b / 64]` is the entry of workgroup 64 * (b / 64), a short scan from there -- and runs one of its
// granules; the tail parts of all buffers (one workgroup each) go in a second launch over `tails`.
template <bool INVERSE>
__global__ void __launch_bounds__(256)
bc7_batch_granules(const BatchEntry* __restrict__ entries, const uint32_t* __restrict__ coarse, uint32_t n_entries)
{
    const uint32_t b = blockIdx.x;
    uint32_t i = coarse[b >> 6];
    while (i + 1 < n_entries && entries[i + 1].first_wg <= b)
        ++i;
    const BatchEntry e = entries[i];
    const uint64_t granule = b - e.first_wg;
    if constexpr (INVERSE)
        bc7_inverse_granule<256, false>(e.src, e.dst + granule * (kT * 16), e.main_blocks, granule * kT, 0);
    else
        bc7_forward_granule<256, false>(e.src + granule * (kT * 16), e.dst, e.main_blocks, granule * kT, 0);
}

template <bool INVERSE>
__global__ void __launch_bounds__(256)
bc7_batch_tails(const BatchEntry* __restrict__ tails)
{
    const BatchEntry e = tails[blockIdx.x];   // src / dst: the tail part's first byte on both sides
    if constexpr (INVERSE)
        bc7_inverse_granule<256, true>(e.src, e.dst, e.tail, 0, (int)e.tail);
    else
        bc7_forward_granule<256, true>(e.src, e.dst, e.tail, 0, (int)e.tail);
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
    // Any pointer alignment: 16-byte vector accesses at unaligned addresses are exact on gfx950 (tools/unaligned_lab.hip);
    // 16-byte aligned buffers are the fast case.
    const uint8_t* aos = static_cast<const uint8_t*>(inverse ? dst : src);     // the range's first block
    const uint8_t* soa = static_cast<const uint8_t*>(inverse ? src : dst);     // byte 0 of the whole transformed buffer
    const uint64_t range_main = first_block >= main_blocks ? 0 : (first_block + num_blocks > main_blocks ? main_blocks : first_block + num_blocks) - first_block;
    // Workgroup size: 256 lanes x 4 blocks per lane (DESIGN.md section 8 has the measurements; the experiments side build,
    // -DDXTLT_EXPERIMENTS, also carries the 512- and 1024-lane kernels behind DXTLT_BC7_LANES).  A launch of 2^32 or more
    // threads is refused: at most 2^21 granules per launch.
    using Kernel = void (*)(const uint8_t*, uint8_t*, uint64_t, uint64_t, int);
#ifdef DXTLT_EXPERIMENTS
    static const int lanes = [] { const char* v = std::getenv("DXTLT_BC7_LANES"); const int x = v ? std::atoi(v) : 0; return x == 512 || x == 1024 ? x : 256; }();
    const Kernel fwd = lanes == 1024 ? bc7_forward<1024, false> : lanes == 512 ? bc7_forward<512, false> : bc7_forward<256, false>;
    const Kernel inv = lanes == 1024 ? bc7_inverse<1024, false> : lanes == 512 ? bc7_inverse<512, false> : bc7_inverse<256, false>;
#else
    constexpr int lanes = 256;
    const Kernel fwd = bc7_forward<256, false>;
    const Kernel inv = bc7_inverse<256, false>;
#endif
    constexpr uint64_t kMaxGranules = 1ull << 21;
    for (uint64_t g0 = 0; g0 < range_main / kT; g0 += kMaxGranules) {
        const uint64_t ng = range_main / kT - g0 < kMaxGranules ? range_main / kT - g0 : kMaxGranules;
        const uint8_t* a = aos + g0 * kT * 16;
        if (inverse)
            hipLaunchKernelGGL(inv, dim3((unsigned)ng), dim3(lanes), 0, stream, soa, const_cast<uint8_t*>(a), main_blocks,
                               first_block + g0 * kT, 0);
        else
            hipLaunchKernelGGL(fwd, dim3((unsigned)ng), dim3(lanes), 0, stream, a, const_cast<uint8_t*>(soa), main_blocks,
                               first_block + g0 * kT, 0);
        if (hipError_t e = hipGetLastError(); e != hipSuccess)
            return e;
    }
    if (tail != 0 && first_block + num_blocks == total_blocks) {
        const uint8_t* a = aos + (main_blocks - first_block) * 16;   // first_block <= main_blocks here
        const uint8_t* s = soa + main_blocks * 16;
        if (inverse)
            hipLaunchKernelGGL((bc7_inverse<256, true>), dim3(1), dim3(256), 0, stream, s, const_cast<uint8_t*>(a), tail, 0, (int)tail);
        else
            hipLaunchKernelGGL((bc7_forward<256, true>), dim3(1), dim3(256), 0, stream, a, const_cast<uint8_t*>(s), tail, 0, (int)tail);
        if (hipError_t e = hipGetLastError(); e != hipSuccess)
            return e;
    }
    return hipSuccess;
}

hipError_t launch(bool inverse, const void* src, void* dst, uint64_t n_blocks, hipStream_t stream)
{
    return launch_range(inverse, src, dst, n_blocks, 0, n_blocks, stream);
}

hipError_t launch_batch(bool inverse, const BatchEntry* d_entries, const uint32_t* d_coarse, uint32_t n_entries,
                        uint32_t granule_wgs, const BatchEntry* d_tails, uint32_t n_tails, hipStream_t stream)
{
    if (granule_wgs > 0) {
        if (inverse)
            hipLaunchKernelGGL(bc7_batch_granules<true>, dim3(granule_wgs), dim3(256), 0, stream, d_entries, d_coarse, n_entries);
        else
            hipLaunchKernelGGL(bc7_batch_granules<false>, dim3(granule_wgs), dim3(256), 0, stream, d_entries, d_coarse, n_entries);
        if (hipError_t e = hipGetLastError(); e != hipSuccess)
            return e;
    }
    if (n_tails > 0) {
        if (inverse)
            hipLaunchKernelGGL(bc7_batch_tails<true>, dim3(n_tails), dim3(256), 0, stream, d_tails);
        else
            hipLaunchKernelGGL(bc7_batch_tails<false>, dim3(n_tails), dim3(256), 0, stream, d_tails);
        return hipGetLastError();
    }
    return hipSuccess;
}

}  // namespace bc7
}  // namespace dxtlt
