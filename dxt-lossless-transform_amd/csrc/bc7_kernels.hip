// bc7_kernels.hip -- gfx950 kernels for the BC7 mode-split transform, version 0 (docs/BC7_FORMAT.md).
//
// PARITY UNPINNED: the reference has no BC7 transform (/root/reference/src/core/dxt-lossless-transform-bc7/src/
// lib.rs:1-13); the format is this build's own and is checked against oracle/dxtlt_oracle_bc7.c + exact round trips.
//
// Layout: [first: byte 0 of every block][for m = 0..8: head_m records, tail_m records], head/tail = block bytes
// 1..H[m] / H[m]+1..15 of the blocks of mode m in block order.  Output placement depends on the data, so the
// transform is a small pipeline on one stream:
//   1. bc7_hist_*      per tile of 1024 blocks: mode histogram (9 counters); the forward pass also writes `first`
//   2. bc7_group_sums  per group of 1024 tiles: sum of the tile histograms
//   3. bc7_scan        per group: exclusive prefix of every mode's counts over all earlier tiles; the last group
//                      also records the nine grand totals, from which the 18 stream bases follow
//   4. bc7_scatter_fwd / bc7_gather_inv   per tile: rank every block inside its mode (wave ballots + a 16x9 LDS
//                      table), build the tile's 18 stream pieces in LDS at offsets congruent to their global
//                      addresses modulo 16, and move every piece with aligned 16-byte accesses (partial first/last
//                      segments bytewise) -- the shifted-tile scheme of bcn_kernels.hip applied to 18 variable pieces.
// HBM traffic: forward reads the blocks twice (histogram, then scatter) and writes them once = 3*len against an
// algorithmic 2*len; inverse reads `first` twice and everything else once = 2.06*len.
#include <hip/hip_runtime.h>

#include "bc7_launch.h"

namespace dxtlt {
namespace bc7 {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kTileBlocks = 1024;  // 16 KiB of blocks per workgroup
constexpr int kVecs = kTileBlocks / kThreads;
constexpr int kGroupTiles = 1024;  // tiles per scan group
constexpr int kImageBytes = 15 * kTileBlocks + 18 * 32;  // pieces + (31 bytes of slack each, rounded up)

__device__ __forceinline__ int head_bytes(int m)
{
    // H[m] = {9, 9, 11, 11, 5, 7, 7, 11, 15}, packed 4 bits each
    return (int)((0xFB775BB99ull >> (4 * m)) & 0xF);
}

__device__ __forceinline__ int mode_of(uint32_t b0)
{
    b0 &= 0xFF;
    return b0 ? __builtin_ctz(b0) : 8;
}

// ---------------------------------------------------------------------------------------------------------
// 1. histograms.  Every lane counts the modes of its own blocks in two packed 64-bit words (modes 0-3: 16 bits each,
//    modes 4-8: 12 bits each; a wave holds at most 1024 blocks), the wave adds them up with xor-shuffles.
// ---------------------------------------------------------------------------------------------------------
struct PackedCounts {
    uint64_t lo, hi;
};

__device__ __forceinline__ void count_mode(PackedCounts& c, int m)
{
    if (m < 4) c.lo += 1ull << (16 * m);
    else if (m < 9) c.hi += 1ull << (12 * (m - 4));
}

__device__ __forceinline__ PackedCounts wave_sum(PackedCounts c)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c.lo += __shfl_xor(c.lo, o);
        c.hi += __shfl_xor(c.hi, o);
    }
    return c;
}

__device__ __forceinline__ uint32_t unpack_count(const PackedCounts& c, int m)
{
    return m < 4 ? (uint32_t)(c.lo >> (16 * m)) & 0xFFFFu : (uint32_t)(c.hi >> (12 * (m - 4))) & 0xFFFu;
}

__global__ void __launch_bounds__(kThreads)
bc7_hist_fwd(const uint8_t* __restrict__ aos, uint8_t* __restrict__ first_out, uint32_t* __restrict__ hist,
             uint64_t n_blocks, uint64_t num_tiles)
{
    // one tile per workgroup; lane t reads byte 0 (the low dword) of blocks j*256 + t
    __shared__ uint32_t part[kThreads / 64][9];
    const uint64_t tile = blockIdx.x;
    uint32_t w[kVecs];
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        const uint64_t b = tile * kTileBlocks + (uint64_t)j * kThreads + threadIdx.x;
        w[j] = b < n_blocks ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(aos + b * 16)) : 0x100u;
    }
    PackedCounts c{0, 0};
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        const uint64_t b = tile * kTileBlocks + (uint64_t)j * kThreads + threadIdx.x;
        if (b < n_blocks) {
            first_out[b] = (uint8_t)w[j];
            count_mode(c, mode_of(w[j]));
        }
    }
    c = wave_sum(c);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 9)
        part[wave][lane] = unpack_count(c, lane);
    __syncthreads();
    if (threadIdx.x < 9)
        hist[(uint64_t)threadIdx.x * num_tiles + tile] =
            part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

__global__ void __launch_bounds__(kThreads)
bc7_hist_inv(const uint8_t* __restrict__ first_in, uint32_t* __restrict__ hist, uint64_t n_blocks, uint64_t num_tiles)
{
    // one tile per WAVE: 64 lanes x 16 first-bytes = 1024 blocks; 4 tiles per workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t tile = (uint64_t)blockIdx.x * 4 + wave;
    if (tile >= num_tiles)
        return;
    const uint64_t b0 = tile * kTileBlocks + (uint64_t)lane * 16;
    PackedCounts c{0, 0};
    if (b0 + 16 <= n_blocks) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(first_in + b0));
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; ++k)
            count_mode(c, mode_of(w[k >> 2] >> (8 * (k & 3))));
    } else {
        for (uint64_t b = b0; b < n_blocks && b < b0 + 16; ++b)
            count_mode(c, mode_of(first_in[b]));
    }
    c = wave_sum(c);
    if (lane < 9)
        hist[(uint64_t)lane * num_tiles + tile] = unpack_count(c, lane);
}

// ---------------------------------------------------------------------------------------------------------
// 2./3. prefix sums over tiles (two levels)
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
bc7_group_sums(const uint32_t* __restrict__ hist, uint32_t* __restrict__ gsum, uint64_t num_tiles, uint32_t groups)
{
    // grid = (groups, 9): sum of one mode's counts over one group of tiles
    const uint32_t g = blockIdx.x, m = blockIdx.y;
    __shared__ uint32_t part[kThreads / 64];
    uint32_t s = 0;
    for (uint32_t i = threadIdx.x; i < kGroupTiles; i += kThreads) {
        const uint64_t tile = (uint64_t)g * kGroupTiles + i;
        if (tile < num_tiles)
            s += hist[(uint64_t)m * num_tiles + tile];
    }
    for (int o = 32; o > 0; o >>= 1)
        s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        gsum[(uint64_t)m * groups + g] = part[0] + part[1] + part[2] + part[3];
}

__global__ void __launch_bounds__(1024)
bc7_scan(const uint32_t* __restrict__ hist, const uint32_t* __restrict__ gsum, uint32_t* __restrict__ prefix,
         uint64_t* __restrict__ totals, uint64_t num_tiles, uint32_t groups)
{
    // grid = (groups, 9), 1024 threads = one tile each
    const uint32_t g = blockIdx.x, m = blockIdx.y, t = threadIdx.x;
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t gbase;
    // (a) blocks of mode m in all earlier groups
    uint32_t acc = 0;
    for (uint32_t i = t; i < g; i += 1024)
        acc += gsum[(uint64_t)m * groups + i];
    for (int o = 32; o > 0; o >>= 1)
        acc += __shfl_down(acc, o);
    if ((t & 63) == 0)
        wsum[t >> 6] = acc;
    __syncthreads();
    if (t == 0) {
        uint32_t s = 0;
        for (int i = 0; i < 16; ++i) s += wsum[i];
        gbase = s;
    }
    __syncthreads();
    // (b) exclusive scan inside the group
    const uint64_t tile = (uint64_t)g * kGroupTiles + t;
    const uint32_t v = tile < num_tiles ? hist[(uint64_t)m * num_tiles + tile] : 0;
    uint32_t incl = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if ((int)(t & 63) >= o) incl += up;
    }
    __syncthreads();
    if ((t & 63) == 63)
        wsum[t >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t i = 0; i < (t >> 6); ++i)
        wbase += wsum[i];
    if (tile < num_tiles)
        prefix[(uint64_t)m * num_tiles + tile] = gbase + wbase + incl - v;

    // (c) the last group knows the grand total of its mode
    if (g == groups - 1 && t == 1023)
        totals[m] = (uint64_t)gbase + wbase + incl;
}

// ---------------------------------------------------------------------------------------------------------
// 4. scatter / gather
// ---------------------------------------------------------------------------------------------------------
struct TileTables {
    uint32_t raw[16][9];    // per (vector j, wave w) slot: blocks of mode m in that slot
    uint32_t slot[16][9];   // ... exclusive prefix over the slots (= blocks of mode m earlier in this tile)
    int lds_off[18];        // LDS offset of piece r (congruent to its global address modulo 16)
    int bytes[18];
    int seg_prefix[19];     // 16-byte segments of the pieces, flattened
    uint64_t g_off[18];     // global byte offset of piece r inside the transformed buffer
};

// Ranks of this lane's blocks inside their modes and the tile's piece table, with two workgroup barriers:
//   ballots -> raw slot counts -> barrier -> 144 lanes turn them into exclusive slot prefixes while lanes 0..17 (wave 0)
//   derive their piece's size from the same raw counts and lay the 18 pieces out with wave-level prefix sums -> barrier.
// LDS layout rule: piece r starts at align16(P_r) + (g_off[r] & 15) with P_r = sum over earlier pieces of (bytes + 31),
// so every piece has room for its own misalignment and its 16-byte segments never touch a neighbour's.
__device__ __forceinline__ void rank_and_layout(const int (&mode)[kVecs], uint32_t (&rank_in_wave)[kVecs], TileTables& tb,
                                                uint64_t origin)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        rank_in_wave[j] = 0;
#pragma unroll
        for (int m = 0; m < 9; ++m) {
            const uint64_t mask = __ballot(mode[j] == m);
            if (mode[j] == m)
                rank_in_wave[j] = (uint32_t)__popcll(mask & lt);
            if (lane == 0)
                tb.raw[j * 4 + wave][m] = (uint32_t)__popcll(mask);
        }
    }
    __syncthreads();
    if (threadIdx.x < 144) {
        const int sidx = threadIdx.x / 9, m = threadIdx.x - sidx * 9;
        uint32_t excl = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            excl += i < sidx ? tb.raw[i][m] : 0;
        tb.slot[sidx][m] = excl;
    }
    if (wave == 0) {
        int bytes = 0, a0 = 0, nseg = 0;
        if (lane < 18) {
            const int m = lane < 9 ? lane : lane - 9;
            const int w = lane < 9 ? head_bytes(m) : 15 - head_bytes(m);
            uint32_t count = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                count += tb.raw[i][m];
            bytes = (int)count * w;
            a0 = (int)(origin & 15);
            nseg = bytes ? (a0 + bytes + 15) >> 4 : 0;
        }
        // inclusive prefix sums over lanes 0..17 of (bytes + 31) and of nseg
        int p = lane < 18 ? bytes + 31 : 0, q = nseg;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const int pu = __shfl_up(p, o), qu = __shfl_up(q, o);
            if (lane >= o) {
                p += pu;
                q += qu;
            }
        }
        if (lane < 18) {
            const int excl_p = p - (bytes + 31);
            tb.lds_off[lane] = ((excl_p + 15) & ~15) + a0;
            tb.bytes[lane] = bytes;
            tb.g_off[lane] = origin;
            tb.seg_prefix[lane] = q - nseg;
            if (lane == 17)
                tb.seg_prefix[18] = q;
        }
    }
    __syncthreads();
}

// piece table: 9 head pieces then 9 tail pieces.  Lanes 0..17 fetch their tile prefix and stream base at kernel entry
// (fetch_piece_origin), so that this dependent global read overlaps the tile's block loads instead of following them.
__device__ __forceinline__ uint64_t fetch_piece_origin(const uint32_t* prefix, const uint64_t* totals, uint64_t num_tiles,
                                                       uint64_t tile, uint64_t n_blocks)
{
    if (threadIdx.x >= 18)
        return 0;
    const int r = threadIdx.x, m = r < 9 ? r : r - 9;
    const uint64_t w = r < 9 ? head_bytes(m) : 15 - head_bytes(m);
    uint64_t base = n_blocks;
    for (int mm = 0; mm < m; ++mm)
        base += totals[mm] * 15;
    if (r >= 9)
        base += totals[m] * (uint64_t)head_bytes(m);
    return base + (uint64_t)prefix[(uint64_t)m * num_tiles + tile] * w;
}

// flattened copy of the 18 pieces between LDS and global memory; TO_GLOBAL selects the direction
template <bool TO_GLOBAL>
__device__ __forceinline__ void move_pieces(uint8_t* img, const TileTables& tb, uint8_t* soa, uint64_t total_bytes)
{
    const int total = tb.seg_prefix[18];
    for (int s = threadIdx.x; s < total; s += kThreads) {
        // last piece r whose first segment is <= s (seg_prefix is non-decreasing; empty pieces repeat a value)
        int r = 0;
#pragma unroll
        for (int step = 16; step > 0; step >>= 1) {
            const int cand = r + step;
            if (cand < 18 && tb.seg_prefix[cand] <= s)
                r = cand;
        }
        const int k = s - tb.seg_prefix[r];
        const uint64_t g = tb.g_off[r];
        const int a0 = (int)(g & 15);
        const uint64_t gseg = g - a0 + (uint64_t)16 * k;
        const int lseg = tb.lds_off[r] - a0 + 16 * k;
        const int lo = k == 0 ? a0 : 0;
        const int end = a0 + tb.bytes[r] - 16 * k;
        const int hi = end < 16 ? end : 16;
        if (TO_GLOBAL) {
            if (lo == 0 && hi == 16) {
                __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(img + lseg), reinterpret_cast<u32x4*>(soa + gseg));
            } else {
                for (int p = lo; p < hi; ++p)
                    soa[gseg + p] = img[lseg + p];
            }
        } else {
            // the whole aligned segment may be fetched when it lies inside the buffer: spare bytes land in padding
            if (gseg + 16 <= total_bytes && lseg >= 0) {
                *reinterpret_cast<u32x4*>(img + lseg) = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(soa + gseg));
            } else {
                for (int p = lo; p < hi; ++p)
                    img[lseg + p] = soa[gseg + p];
            }
        }
    }
}

__global__ void __launch_bounds__(kThreads)
bc7_scatter_fwd(const uint8_t* __restrict__ aos, uint8_t* __restrict__ soa, const uint32_t* __restrict__ prefix,
                const uint64_t* __restrict__ totals, uint64_t n_blocks, uint64_t num_tiles)
{
    __shared__ __attribute__((aligned(16))) uint8_t img[kImageBytes];
    __shared__ TileTables tb;
    const uint64_t tile = blockIdx.x;
    const int wave = threadIdx.x >> 6;
    const uint64_t origin = fetch_piece_origin(prefix, totals, num_tiles, tile, n_blocks);

    u32x4 q[kVecs];
    int mode[kVecs];
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        const uint64_t b = tile * kTileBlocks + (uint64_t)j * kThreads + threadIdx.x;
        mode[j] = 9;
        q[j] = u32x4{0, 0, 0, 0};
        if (b < n_blocks) {
            q[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(aos + b * 16));
            mode[j] = mode_of(q[j].x);
        }
    }
    uint32_t rank_in_wave[kVecs];
    rank_and_layout(mode, rank_in_wave, tb, origin);

#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        if (mode[j] < 9) {
            const int m = mode[j], h = head_bytes(m);
            const int rank = (int)(tb.slot[j * 4 + wave][m] + rank_in_wave[j]);
            const int ho = tb.lds_off[m] + rank * h - 1;            // byte k of the block goes to ho + k (k <= h)
            const int to = tb.lds_off[9 + m] + rank * (15 - h) - 1 - h;  // ... or to + k (k > h)
            const uint32_t w[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
            for (int k = 1; k < 16; ++k) {
                const uint8_t byte = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
                img[(k <= h ? ho : to) + k] = byte;
            }
        }
    }
    __syncthreads();
    move_pieces<true>(img, tb, soa, n_blocks * 16);
}

__global__ void __launch_bounds__(kThreads)
bc7_gather_inv(const uint8_t* __restrict__ soa, uint8_t* __restrict__ aos, const uint32_t* __restrict__ prefix,
               const uint64_t* __restrict__ totals, uint64_t n_blocks, uint64_t num_tiles)
{
    __shared__ __attribute__((aligned(16))) uint8_t img[kImageBytes];
    __shared__ TileTables tb;
    const uint64_t tile = blockIdx.x;
    const int wave = threadIdx.x >> 6;
    const uint64_t origin = fetch_piece_origin(prefix, totals, num_tiles, tile, n_blocks);

    int mode[kVecs];
    uint32_t b0[kVecs];
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        const uint64_t b = tile * kTileBlocks + (uint64_t)j * kThreads + threadIdx.x;
        mode[j] = 9;
        b0[j] = 0;
        if (b < n_blocks) {
            b0[j] = soa[b];
            mode[j] = mode_of(b0[j]);
        }
    }
    uint32_t rank_in_wave[kVecs];
    rank_and_layout(mode, rank_in_wave, tb, origin);
    move_pieces<false>(img, tb, const_cast<uint8_t*>(soa), n_blocks * 16);
    __syncthreads();

#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        if (mode[j] < 9) {
            const uint64_t b = tile * kTileBlocks + (uint64_t)j * kThreads + threadIdx.x;
            const int m = mode[j], h = head_bytes(m);
            const int rank = (int)(tb.slot[j * 4 + wave][m] + rank_in_wave[j]);
            const int ho = tb.lds_off[m] + rank * h - 1;
            const int to = tb.lds_off[9 + m] + rank * (15 - h) - 1 - h;
            uint32_t w[4] = {b0[j] & 0xFF, 0, 0, 0};
#pragma unroll
            for (int k = 1; k < 16; ++k)
                w[k >> 2] |= (uint32_t)img[(k <= h ? ho : to) + k] << (8 * (k & 3));
            __builtin_nontemporal_store(u32x4{w[0], w[1], w[2], w[3]}, reinterpret_cast<u32x4*>(aos + b * 16));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static inline uint64_t tiles_for(uint64_t n_blocks) { return (n_blocks + kTileBlocks - 1) / kTileBlocks; }
static inline uint64_t groups_for(uint64_t num_tiles) { return (num_tiles + kGroupTiles - 1) / kGroupTiles; }

size_t workspace_bytes(uint64_t n_blocks)
{
    const uint64_t tiles = tiles_for(n_blocks), groups = groups_for(tiles);
    // hist[9][tiles] + prefix[9][tiles] + gsum[9][groups] as u32, bases[18] as u64
    return (size_t)((2 * 9 * tiles + 9 * groups) * sizeof(uint32_t) + 256 + 18 * sizeof(uint64_t));
}

hipError_t launch(bool inverse, const void* src, void* dst, uint64_t n_blocks, void* workspace, size_t ws_bytes,
                  hipStream_t stream)
{
    if (n_blocks == 0)
        return hipSuccess;
    if (ws_bytes < workspace_bytes(n_blocks) || workspace == nullptr)
        return hipErrorInvalidValue;
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(workspace)) & 15) != 0)
        return hipErrorInvalidValue;  // BC7 v0 takes 16-byte aligned device buffers only
    const uint64_t tiles = tiles_for(n_blocks), groups = groups_for(tiles);
    if (tiles > 0x7FFFFFFFull || groups > 65535ull * 1024)
        return hipErrorInvalidValue;
    uint8_t* ws = static_cast<uint8_t*>(workspace);
    uint64_t* totals = reinterpret_cast<uint64_t*>(ws);                    // 9 x u64: blocks per mode
    uint32_t* hist = reinterpret_cast<uint32_t*>(ws + 256);
    uint32_t* prefix = hist + 9 * tiles;
    uint32_t* gsum = prefix + 9 * tiles;
    const uint8_t* s8 = static_cast<const uint8_t*>(src);
    uint8_t* d8 = static_cast<uint8_t*>(dst);

    if (!inverse)
        hipLaunchKernelGGL(bc7_hist_fwd, dim3((unsigned)tiles), dim3(kThreads), 0, stream, s8, d8, hist, n_blocks, tiles);
    else
        hipLaunchKernelGGL(bc7_hist_inv, dim3((unsigned)((tiles + 3) / 4)), dim3(kThreads), 0, stream, s8, hist, n_blocks,
                           tiles);
    hipLaunchKernelGGL(bc7_group_sums, dim3((unsigned)groups, 9), dim3(kThreads), 0, stream, hist, gsum, tiles,
                       (uint32_t)groups);
    hipLaunchKernelGGL(bc7_scan, dim3((unsigned)groups, 9), dim3(1024), 0, stream, hist, gsum, prefix, totals, tiles,
                       (uint32_t)groups);
    if (!inverse)
        hipLaunchKernelGGL(bc7_scatter_fwd, dim3((unsigned)tiles), dim3(kThreads), 0, stream, s8, d8, prefix, totals,
                           n_blocks, tiles);
    else
        hipLaunchKernelGGL(bc7_gather_inv, dim3((unsigned)tiles), dim3(kThreads), 0, stream, s8, d8, prefix, totals,
                           n_blocks, tiles);
    return hipGetLastError();
}

}  // namespace bc7
}  // namespace dxtlt
