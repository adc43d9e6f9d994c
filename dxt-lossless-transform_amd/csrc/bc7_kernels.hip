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
//   3b. bc7_stream_bases  one wave: the 18 stream bases from the nine totals
//   4. bc7_scatter_fwd / bc7_gather_inv   per tile: rank every block inside its mode (wave match + a 16x9 LDS
//                      table), build the tile's 18 stream pieces in LDS at offsets congruent to their global
//                      addresses modulo 16, and move every piece with aligned 16-byte accesses.
// HBM traffic: forward reads the blocks twice (histogram, then scatter) and writes them once = 3*len against an
// algorithmic 2*len; inverse reads `first` twice and everything else once = 2.06*len.
#include <hip/hip_runtime.h>

#include "bc7_launch.h"
#include "streaming_store.h"

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
    // H[m] = 5 + 2 * {2, 2, 3, 3, 0, 1, 1, 3, 5}[m], three bits per mode
    return 5 + 2 * (int)((0x56486d2u >> (3 * m)) & 7u);
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
    // One tile per WAVE: lane l reads byte 0 (the low dword) of blocks i*64 + l, i = 0..15, no workgroup barrier.  The
    // tile's 1024 first bytes leave through the wave's 1 KiB of LDS as 64 lanes x 16 bytes.  History (4 GiB, tools/read_lab.hip for the ceiling): one tile per workgroup with a byte store per
    // block 0.776 ms; the same with the first bytes through LDS 0.744 ms; a kernel that only reads these dwords 0.593 ms.
    // hipcc gives each conditional load below its own branch and `s_waitcnt vmcnt(0)`, so a wave has ONE load in flight
    // at a time -- and that is the fast form here: branch-free (clamped) loads with sixteen or with four in flight
    // both took 0.80 ms, two in flight 0.77 ms, one branch-free load at a time 0.69 ms like the compiled form
    // (profiles/r01_z/bc7_hist_per_wave.txt): it is the depth, not the branches.
    __shared__ __attribute__((aligned(16))) uint8_t firsts[kThreads / 64][kTileBlocks];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t tile = (uint64_t)blockIdx.x * (kThreads / 64) + wave;
    if (tile >= num_tiles)
        return;
    const uint64_t tile_first = tile * kTileBlocks;
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint64_t b = tile_first + (uint64_t)(i * 64 + lane);
        w[i] = b < n_blocks ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(aos + b * 16)) : 0x100u;
    }
    PackedCounts c{0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        firsts[wave][i * 64 + lane] = (uint8_t)w[i];
        if (tile_first + (uint64_t)(i * 64 + lane) < n_blocks)
            count_mode(c, mode_of(w[i]));
    }
    c = wave_sum(c);
    if (lane < 9)
        hist[(uint64_t)lane * num_tiles + tile] = unpack_count(c, lane);
    __builtin_amdgcn_wave_barrier();   // same wave wrote the bytes it now reads; DS operations of a wave stay in order
    const uint64_t b0 = tile_first + (uint64_t)lane * 16;
    if (b0 + 16 <= n_blocks) {
        __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(&firsts[wave][16 * lane]), reinterpret_cast<u32x4*>(first_out + b0));
    } else {
        for (int i = 0; i < 16; ++i)
            if (b0 + i < n_blocks)
                first_out[b0 + i] = firsts[wave][16 * lane + i];
    }
}

constexpr int kInvTilesPerWave = 4;

__global__ void __launch_bounds__(kThreads)
bc7_hist_inv(const uint8_t* __restrict__ first_in, uint32_t* __restrict__ hist, uint64_t n_blocks, uint64_t num_tiles)
{
    // one tile per wave and step: 64 lanes x 16 first-bytes = 1024 blocks; every wave takes kInvTilesPerWave
    // consecutive tiles with all loads issued up front (a wave with a single 1 KiB load is pure latency)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t tile0 = ((uint64_t)blockIdx.x * (kThreads / 64) + wave) * kInvTilesPerWave;
    u32x4 v[kInvTilesPerWave];
#pragma unroll
    for (int i = 0; i < kInvTilesPerWave; ++i) {
        const uint64_t b0 = (tile0 + i) * kTileBlocks + (uint64_t)lane * 16;
        v[i] = u32x4{0, 0, 0, 0};
        if (b0 + 16 <= n_blocks)
            v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(first_in + b0));
    }
#pragma unroll
    for (int i = 0; i < kInvTilesPerWave; ++i) {
        const uint64_t tile = tile0 + i;
        if (tile >= num_tiles)
            break;
        const uint64_t b0 = tile * kTileBlocks + (uint64_t)lane * 16;
        PackedCounts c{0, 0};
        if (b0 + 16 <= n_blocks) {
            const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
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

// Inputs of up to 1024 tiles (16 MiB) are launch bound: five dependent launches cost ~27 us whatever the size.  For
// them one workgroup does steps 2, 3 and 3b at once -- thread t owns tile t, the nine modes are scanned side by side --
// and the pipeline is three launches.
__global__ void __launch_bounds__(1024)
bc7_scan_small(const uint32_t* __restrict__ hist, uint32_t* __restrict__ prefix, uint64_t* __restrict__ totals,
               uint64_t* __restrict__ bases, uint64_t num_tiles, uint64_t n_blocks)
{
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    __shared__ uint32_t wsum[9][16];
    __shared__ uint64_t tot[9];
    uint32_t v[9], incl[9];
#pragma unroll
    for (int m = 0; m < 9; ++m)
        v[m] = t < num_tiles ? hist[(uint64_t)m * num_tiles + t] : 0;
#pragma unroll
    for (int m = 0; m < 9; ++m) {
        incl[m] = v[m];
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl[m], o);
            if ((int)lane >= o) incl[m] += up;
        }
        if (lane == 63)
            wsum[m][wave] = incl[m];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 9; ++m) {
        uint32_t wbase = 0;
        for (uint32_t i = 0; i < wave; ++i)
            wbase += wsum[m][i];
        if (t < num_tiles)
            prefix[(uint64_t)m * num_tiles + t] = wbase + incl[m] - v[m];
        if (t == 1023) {
            tot[m] = (uint64_t)wbase + incl[m];
            totals[m] = tot[m];
        }
    }
    __syncthreads();
    if (t < 18) {
        const int m = t < 9 ? (int)t : (int)t - 9;
        uint64_t base = n_blocks;
        for (int mm = 0; mm < m; ++mm)
            base += tot[mm] * 15;
        if (t >= 9)
            base += tot[m] * (uint64_t)head_bytes(m);
        bases[t] = base;
    }
}

// ---- piece movement ---------------------------------------------------------------------------------------
// Bytes [lo, hi) of one 16-byte segment, both pointers 16-byte aligned at byte 0 of the segment: an ascending
// ladder of naturally aligned 1/2/4/8-byte moves from lo, then a descending one for what is left.
template <typename Mover>
__device__ __forceinline__ void partial_segment(int lo, int hi, Mover mv)
{
    int p = lo;
    if ((p & 1) && p + 1 <= hi) { mv(p, 1); p += 1; }
    if ((p & 2) && p + 2 <= hi) { mv(p, 2); p += 2; }
    if ((p & 4) && p + 4 <= hi) { mv(p, 4); p += 4; }
    if ((p & 8) && p + 8 <= hi) { mv(p, 8); p += 8; }
    const int rem = hi - p;
    if (rem & 8) { mv(p, 8); p += 8; }
    if (rem & 4) { mv(p, 4); p += 4; }
    if (rem & 2) { mv(p, 2); p += 2; }
    if (rem & 1) { mv(p, 1); }
}

__device__ __forceinline__ void typed_move(uint8_t* dst, const uint8_t* src, int p, int w)
{
    if (w == 1) dst[p] = src[p];
    if (w == 2) *reinterpret_cast<uint16_t*>(dst + p) = *reinterpret_cast<const uint16_t*>(src + p);
    if (w == 4) *reinterpret_cast<uint32_t*>(dst + p) = *reinterpret_cast<const uint32_t*>(src + p);
    if (w == 8) *reinterpret_cast<uint64_t*>(dst + p) = *reinterpret_cast<const uint64_t*>(src + p);
}

__device__ __forceinline__ void store_block(uint8_t* p, u32x4 v)
{
    // AoS output: whole 1 KiB runs per wave instruction, nobody else touches these lines -> write-through streaming
    store_streaming16(p, v);   // streaming_store.h
}

// ---------------------------------------------------------------------------------------------------------
// 4. scatter / gather.  One tile of 1024 blocks per workgroup; a wave owns 256 consecutive blocks (vector j = blocks
//    wave*256 + j*64 + lane).  The tile's 18 stream pieces are built in an LDS image at offsets congruent to their
//    global addresses modulo 16 and moved as aligned 16-byte rows (partial first/last rows with typed 1/2/4/8-byte
//    moves) -- the shifted-tile scheme of bcn_kernels.hip applied to 18 variable pieces.  What the first version
//    (profiles/r01_l: 1120 VALU instructions per wave, VALU-issue bound) taught, profiles/r01_s:
//      * ranks from a 4-bit match (4 ballots + 2 mbcnt per vector) instead of 9 ballots with 9 selects; the first
//        lane of every mode class stores the class size, every wave pre-zeroes its own four table rows;
//      * record bytes move with constant instruction offsets from five base pointers (bytes 1-5 always head; 6-7,
//        8-9, 10-11, 12-15 head or tail by one compare each) -- and as single bytes: DS instructions at addresses
//        that are not multiples of their width run several times slower, so the byte pointers are volatile to stop
//        clang from fusing them;
//      * every LDS row is stamped once with its piece number by the record that owns its first byte, so the mover
//        maps row -> piece with one byte read instead of a search over the piece table;
//      * no branch around a global load: all loads of a lane are in flight together (lanes past the end re-read
//        something harmless);
//      * the 18 stream bases come from a one-wave kernel after the scan instead of a loop in every tile.
// ---------------------------------------------------------------------------------------------------------
// volatile byte view of an LDS array, in the LDS address space (a plain volatile pointer would turn into flat_* ops)
typedef volatile uint8_t __attribute__((address_space(3))) lds_byte;
__device__ __forceinline__ lds_byte* lds_bytes(uint8_t* p) { return (lds_byte*)p; }

constexpr int kRows = kImageBytes / 16;            // 996 LDS rows of 16 bytes
constexpr int kStampBytes = 1024;
static_assert(kRows <= kStampBytes && kImageBytes % 16 == 0, "stamp table covers the image");
constexpr int kRowIters = (kRows + kThreads - 1) / kThreads;

struct PieceRef {           // 16 bytes, read with one ds_read_b128
    uint64_t g_row0;        // global byte offset of the piece's first row (16-byte aligned)
    int32_t lds_first;      // LDS byte offset of the piece's first byte; row0 = lds_first >> 4, a0 = lds_first & 15
    int32_t bytes;
};

struct TileTables {
    uint32_t raw[16][9];
    uint32_t slot[16][9];
    __attribute__((aligned(16))) PieceRef piece[18];
    __attribute__((aligned(16))) uint8_t stamp[kStampBytes];   // row -> piece + 1; 0 = padding row
};

__global__ void __launch_bounds__(64)
bc7_stream_bases(const uint64_t* __restrict__ totals, uint64_t* __restrict__ bases, uint64_t n_blocks)
{
    const int r = threadIdx.x;
    if (r >= 18)
        return;
    const int m = r < 9 ? r : r - 9;
    uint64_t base = n_blocks;
    for (int mm = 0; mm < m; ++mm)
        base += totals[mm] * 15;
    if (r >= 9)
        base += totals[m] * (uint64_t)head_bytes(m);
    bases[r] = base;
}

// rank of this lane among the earlier lanes of the wave holding the same key (0..15), and the class size
__device__ __forceinline__ void match_rank(int key, uint32_t& rank, uint32_t& count)
{
    uint32_t lo = ~0u, hi = ~0u;
#pragma unroll
    for (int bit = 0; bit < 4; ++bit) {
        const bool mine = (key >> bit) & 1;
        const uint64_t b = __ballot(mine);
        const uint64_t sel = mine ? b : ~b;
        lo &= (uint32_t)sel;
        hi &= (uint32_t)(sel >> 32);
    }
    rank = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
    count = (uint32_t)__popc(lo) + (uint32_t)__popc(hi);
}

__device__ __forceinline__ void rank_and_layout(const int (&mode)[kVecs], uint32_t (&rank_in_wave)[kVecs], TileTables& tb,
                                                 uint64_t origin)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < kStampBytes / 16)
        reinterpret_cast<u32x4*>(tb.stamp)[threadIdx.x] = u32x4{0, 0, 0, 0};
    if (lane < 36)
        (&tb.raw[wave * 4][0])[lane] = 0;   // this wave's four rows; LDS operations of one wave stay in order
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        uint32_t count;
        match_rank(mode[j], rank_in_wave[j], count);
        if (rank_in_wave[j] == 0 && mode[j] < 9)
            tb.raw[wave * 4 + j][mode[j]] = count;
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
    if (wave == 3) {   // the piece table: lanes 0..17 of the last wave (the first three carry the slot prefixes)
        int bytes = 0, a0 = 0;
        if (lane < 18) {
            const int m = lane < 9 ? lane : lane - 9;
            const int w = lane < 9 ? head_bytes(m) : 15 - head_bytes(m);
            uint32_t count = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                count += tb.raw[i][m];
            bytes = (int)count * w;
            a0 = (int)(origin & 15);
        }
        int p = lane < 18 ? bytes + 31 : 0;   // inclusive prefix of (bytes + 31) over the pieces
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const int pu = __shfl_up(p, o);
            if (lane >= o)
                p += pu;
        }
        if (lane < 18) {
            const int excl_p = p - (bytes + 31);
            PieceRef ref;
            ref.lds_first = ((excl_p + 15) & ~15) + a0;
            ref.bytes = bytes;
            ref.g_row0 = origin - (uint64_t)a0;
            tb.piece[lane] = ref;
        }
    }
    __syncthreads();
}

// lanes 0..17 of wave 3: global byte offset of this tile's piece r.  Branch-free (the other lanes fetch piece 0 or 17
// and ignore it) so that the two loads are issued together with the tile's block loads instead of ahead of them.
__device__ __forceinline__ uint64_t fetch_piece_origin(const uint32_t* prefix, const uint64_t* bases, uint64_t num_tiles,
                                                        uint64_t tile)
{
    int r = (int)threadIdx.x - 192;
    r = r < 0 ? 0 : (r > 17 ? 17 : r);
    const int m = r < 9 ? r : r - 9;
    const uint64_t w = r < 9 ? head_bytes(m) : 15 - head_bytes(m);
    return bases[r] + (uint64_t)prefix[(uint64_t)m * num_tiles + tile] * w;
}

struct RecordPlace {
    int head, tail, h;   // LDS byte offsets of the block's head and tail records
    int head0, tail0;    // ... and of the first byte of their pieces
};

__device__ __forceinline__ RecordPlace place_records(const TileTables& tb, int m, int slot_row, uint32_t rank_in_wave)
{
    RecordPlace rp;
    rp.h = head_bytes(m);
    const int rank = (int)(tb.slot[slot_row][m] + rank_in_wave);
    rp.head0 = tb.piece[m].lds_first;
    rp.tail0 = tb.piece[9 + m].lds_first;
    rp.head = rp.head0 + rank * rp.h;
    rp.tail = rp.tail0 + rank * (15 - rp.h);
    return rp;
}

// Every LDS row of a piece is stamped exactly once, by the record that owns the row's first piece byte: a record
// stamps its first row when it starts the piece or starts exactly on the row boundary (otherwise its predecessor
// already reaches into that row), and its last row when it crosses into it.  (Stamping from every record makes
// dozens of lanes write the same byte, which the LDS serialises: measured 3000 wait cycles per wave.)
__device__ __forceinline__ void stamp_record(TileTables& tb, int first, int bytes, int piece_first, uint8_t mark)
{
    const int row_s = first >> 4, row_e = (first + bytes - 1) >> 4;
    if (first == piece_first || (first & 15) == 0)
        tb.stamp[row_s] = mark;
    if (row_e != row_s)
        tb.stamp[row_e] = mark;
}

struct RowRef {
    bool live, whole;
    int lo, hi;
    uint64_t gseg;
};

__device__ __forceinline__ RowRef locate_row(const TileTables& tb, int row)
{
    RowRef ref{};
    const int st = row < kRows ? tb.stamp[row] : 0;
    ref.live = st != 0;
    if (!ref.live)
        return ref;
    const PieceRef pc = tb.piece[st - 1];
    const int a0 = pc.lds_first & 15;
    const int k = row - (pc.lds_first >> 4);
    ref.gseg = pc.g_row0 + (uint64_t)(16 * k);
    ref.lo = k == 0 ? a0 : 0;
    const int end = a0 + pc.bytes - 16 * k;
    ref.hi = end < 16 ? end : 16;
    ref.whole = ref.lo == 0 && ref.hi == 16;
    return ref;
}

__global__ void __launch_bounds__(kThreads)
bc7_scatter_fwd(const uint8_t* __restrict__ aos, uint8_t* __restrict__ soa, const uint32_t* __restrict__ prefix,
                 const uint64_t* __restrict__ bases, uint64_t n_blocks, uint64_t num_tiles)
{
    __shared__ __attribute__((aligned(16))) uint8_t img[kImageBytes];
    __shared__ TileTables tb;
    const uint64_t tile = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t origin = fetch_piece_origin(prefix, bases, num_tiles, tile);

    u32x4 q[kVecs];
    int mode[kVecs];
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        const uint64_t b = tile * kTileBlocks + (uint64_t)(wave * 256 + j * 64 + lane);
        // no branch around the load: all four are in flight together (lanes past the end re-read the last block)
        q[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(aos + (b < n_blocks ? b : n_blocks - 1) * 16));
        mode[j] = b < n_blocks ? mode_of(q[j].x) : 9;
    }
    uint32_t rank_in_wave[kVecs];
    rank_and_layout(mode, rank_in_wave, tb, origin);

#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        if (mode[j] < 9) {
            const RecordPlace rp = place_records(tb, mode[j], wave * 4 + j, rank_in_wave[j]);
            stamp_record(tb, rp.head, rp.h, rp.head0, (uint8_t)(mode[j] + 1));
            if (rp.h < 15)
                stamp_record(tb, rp.tail, 15 - rp.h, rp.tail0, (uint8_t)(mode[j] + 10));
            // volatile: the compiler would otherwise fuse neighbouring byte accesses into ds_write_b32/b16 at odd
            // addresses, which the LDS executes far slower than the separate byte writes (measured, r01_s)
            lds_byte* ho = lds_bytes(img) + rp.head - 1;             // block byte k (1 <= k <= h) -> ho[k]
            lds_byte* to = lds_bytes(img) + rp.tail - 1 - rp.h;      // block byte k (k > h)       -> to[k]
            lds_byte* p67 = rp.h >= 7 ? ho : to;
            lds_byte* p89 = rp.h >= 9 ? ho : to;
            lds_byte* pab = rp.h >= 11 ? ho : to;
            lds_byte* pcf = rp.h >= 15 ? ho : to;
            const uint32_t w0 = q[j].x, w1 = q[j].y, w2 = q[j].z, w3 = q[j].w;
            ho[1] = (uint8_t)(w0 >> 8);
            ho[2] = (uint8_t)(w0 >> 16);
            ho[3] = (uint8_t)(w0 >> 24);
            ho[4] = (uint8_t)w1;
            ho[5] = (uint8_t)(w1 >> 8);
            p67[6] = (uint8_t)(w1 >> 16);
            p67[7] = (uint8_t)(w1 >> 24);
            p89[8] = (uint8_t)w2;
            p89[9] = (uint8_t)(w2 >> 8);
            pab[10] = (uint8_t)(w2 >> 16);
            pab[11] = (uint8_t)(w2 >> 24);
            pcf[12] = (uint8_t)w3;
            pcf[13] = (uint8_t)(w3 >> 8);
            pcf[14] = (uint8_t)(w3 >> 16);
            pcf[15] = (uint8_t)(w3 >> 24);
        }
    }
    __syncthreads();

#pragma unroll
    for (int it = 0; it < kRowIters; ++it) {
        const int row = it * kThreads + (int)threadIdx.x;
        const RowRef ref = locate_row(tb, row);
        if (!ref.live)
            continue;
        if (ref.whole) {
            // neighbouring tiles share 128-byte lines here: plain streaming store, L2 merges the halves
            __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(img + 16 * row),
                                        reinterpret_cast<u32x4*>(soa + ref.gseg));
        } else {
            uint8_t* dst = soa + ref.gseg;
            const uint8_t* src = img + 16 * row;
            partial_segment(ref.lo, ref.hi, [&](int p, int w) { typed_move(dst, src, p, w); });
        }
    }
}

__global__ void __launch_bounds__(kThreads)
bc7_gather_inv(const uint8_t* __restrict__ soa, uint8_t* __restrict__ aos, const uint32_t* __restrict__ prefix,
                const uint64_t* __restrict__ bases, uint64_t n_blocks, uint64_t num_tiles)
{
    __shared__ __attribute__((aligned(16))) uint8_t img[kImageBytes];
    __shared__ TileTables tb;
    const uint64_t tile = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t origin = fetch_piece_origin(prefix, bases, num_tiles, tile);
    const uint64_t total_bytes = n_blocks * 16;

    int mode[kVecs];
    uint32_t b0[kVecs];
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        const uint64_t b = tile * kTileBlocks + (uint64_t)(wave * 256 + j * 64 + lane);
        b0[j] = soa[b < n_blocks ? b : n_blocks - 1];   // branch-free: the four loads overlap
        mode[j] = b < n_blocks ? mode_of(b0[j]) : 9;
    }
    uint32_t rank_in_wave[kVecs];
    rank_and_layout(mode, rank_in_wave, tb, origin);

    RecordPlace rp[kVecs];
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        if (mode[j] < 9) {
            rp[j] = place_records(tb, mode[j], wave * 4 + j, rank_in_wave[j]);
            stamp_record(tb, rp[j].head, rp[j].h, rp[j].head0, (uint8_t)(mode[j] + 1));
            if (rp[j].h < 15)
                stamp_record(tb, rp[j].tail, 15 - rp[j].h, rp[j].tail0, (uint8_t)(mode[j] + 10));
        }
    }
    __syncthreads();

    // rows in: all loads of the lane are in flight before the first LDS write.  A whole aligned row is fetched
    // whenever it lies inside the buffer: the bytes of neighbouring pieces land in this piece's LDS padding.
    RowRef ref[kRowIters];
    u32x4 v[kRowIters];
#pragma unroll
    for (int it = 0; it < kRowIters; ++it) {
        ref[it] = locate_row(tb, it * kThreads + (int)threadIdx.x);
        ref[it].whole = ref[it].live && ref[it].gseg + 16 <= total_bytes;
        // unconditional (rows that are not fetched whole read offset 0 and drop the result): no branch, no wait
        v[it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(soa + (ref[it].whole ? ref[it].gseg : 0)));
    }
#pragma unroll
    for (int it = 0; it < kRowIters; ++it) {
        const int row = it * kThreads + (int)threadIdx.x;
        if (ref[it].whole) {
            *reinterpret_cast<u32x4*>(img + 16 * row) = v[it];
        } else if (ref[it].live) {
            uint8_t* dst = img + 16 * row;
            const uint8_t* src = soa + ref[it].gseg;
            partial_segment(ref[it].lo, ref[it].hi, [&](int p, int w) { typed_move(dst, src, p, w); });
        }
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        if (mode[j] < 9) {
            const uint64_t b = tile * kTileBlocks + (uint64_t)(wave * 256 + j * 64 + lane);
            // volatile: keeps 15 ds_read_u8; fused unaligned ds_read_u16/b32 are far slower on this LDS
            const lds_byte* ho = lds_bytes(img) + rp[j].head - 1;
            const lds_byte* to = lds_bytes(img) + rp[j].tail - 1 - rp[j].h;
            const lds_byte* p67 = rp[j].h >= 7 ? ho : to;
            const lds_byte* p89 = rp[j].h >= 9 ? ho : to;
            const lds_byte* pab = rp[j].h >= 11 ? ho : to;
            const lds_byte* pcf = rp[j].h >= 15 ? ho : to;
            u32x4 o;
            o.x = (b0[j] & 0xFF) | ((uint32_t)ho[1] << 8) | ((uint32_t)ho[2] << 16) | ((uint32_t)ho[3] << 24);
            o.y = (uint32_t)ho[4] | ((uint32_t)ho[5] << 8) | ((uint32_t)p67[6] << 16) | ((uint32_t)p67[7] << 24);
            o.z = (uint32_t)p89[8] | ((uint32_t)p89[9] << 8) | ((uint32_t)pab[10] << 16) | ((uint32_t)pab[11] << 24);
            o.w = (uint32_t)pcf[12] | ((uint32_t)pcf[13] << 8) | ((uint32_t)pcf[14] << 16) | ((uint32_t)pcf[15] << 24);
            store_block(aos + b * 16, o);
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
    if (tiles * kThreads > 0xFFFFFFFFull || groups > 65535ull * 1024)   // a launch holds fewer than 2^32 threads: 256 GiB of blocks
        return hipErrorInvalidValue;
    uint8_t* ws = static_cast<uint8_t*>(workspace);
    uint64_t* totals = reinterpret_cast<uint64_t*>(ws);                    // 9 x u64: blocks per mode
    uint64_t* bases = reinterpret_cast<uint64_t*>(ws + 96);                // 18 x u64: stream bases
    uint32_t* hist = reinterpret_cast<uint32_t*>(ws + 256);
    uint32_t* prefix = hist + 9 * tiles;
    uint32_t* gsum = prefix + 9 * tiles;
    const uint8_t* s8 = static_cast<const uint8_t*>(src);
    uint8_t* d8 = static_cast<uint8_t*>(dst);

    if (!inverse)
        hipLaunchKernelGGL(bc7_hist_fwd, dim3((unsigned)((tiles + kThreads / 64 - 1) / (kThreads / 64))), dim3(kThreads), 0, stream, s8, d8,
                           hist, n_blocks, tiles);
    else
        hipLaunchKernelGGL(bc7_hist_inv, dim3((unsigned)((tiles + kInvTilesPerWave * 4 - 1) / (kInvTilesPerWave * 4))), dim3(kThreads), 0,
                           stream, s8, hist, n_blocks, tiles);
    if (tiles <= 1024) {
        hipLaunchKernelGGL(bc7_scan_small, dim3(1), dim3(1024), 0, stream, hist, prefix, totals, bases, tiles, n_blocks);
    } else {
        hipLaunchKernelGGL(bc7_group_sums, dim3((unsigned)groups, 9), dim3(kThreads), 0, stream, hist, gsum, tiles,
                           (uint32_t)groups);
        hipLaunchKernelGGL(bc7_scan, dim3((unsigned)groups, 9), dim3(1024), 0, stream, hist, gsum, prefix, totals, tiles,
                           (uint32_t)groups);
        hipLaunchKernelGGL(bc7_stream_bases, dim3(1), dim3(64), 0, stream, totals, bases, n_blocks);
    }
    if (!inverse)
        hipLaunchKernelGGL(bc7_scatter_fwd, dim3((unsigned)tiles), dim3(kThreads), 0, stream, s8, d8, prefix, bases, n_blocks,
                           tiles);
    else
        hipLaunchKernelGGL(bc7_gather_inv, dim3((unsigned)tiles), dim3(kThreads), 0, stream, s8, d8, prefix, bases, n_blocks,
                           tiles);
    return hipGetLastError();
}

// Only the counting part of the inverse pipeline: from the `first` stream of n_blocks blocks to the nine per-mode
// totals at the start of the workspace (multi-GPU sharding needs every shard's counts before it can place pieces).
hipError_t launch_counts(const void* first, uint64_t n_blocks, void* workspace, size_t ws_bytes, hipStream_t stream)
{
    if (n_blocks == 0)
        return hipSuccess;
    if (ws_bytes < workspace_bytes(n_blocks) || workspace == nullptr)
        return hipErrorInvalidValue;
    if (((reinterpret_cast<uintptr_t>(first) | reinterpret_cast<uintptr_t>(workspace)) & 15) != 0)
        return hipErrorInvalidValue;
    const uint64_t tiles = tiles_for(n_blocks), groups = groups_for(tiles);
    if (tiles * kThreads > 0xFFFFFFFFull || groups > 65535ull * 1024)   // a launch holds fewer than 2^32 threads: 256 GiB of blocks
        return hipErrorInvalidValue;
    uint8_t* ws = static_cast<uint8_t*>(workspace);
    uint64_t* totals = reinterpret_cast<uint64_t*>(ws);
    uint32_t* hist = reinterpret_cast<uint32_t*>(ws + 256);
    uint32_t* prefix = hist + 9 * tiles;
    uint32_t* gsum = prefix + 9 * tiles;
    hipLaunchKernelGGL(bc7_hist_inv, dim3((unsigned)((tiles + kInvTilesPerWave * 4 - 1) / (kInvTilesPerWave * 4))), dim3(kThreads), 0,
                       stream, static_cast<const uint8_t*>(first), hist, n_blocks, tiles);
    hipLaunchKernelGGL(bc7_group_sums, dim3((unsigned)groups, 9), dim3(kThreads), 0, stream, hist, gsum, tiles,
                       (uint32_t)groups);
    hipLaunchKernelGGL(bc7_scan, dim3((unsigned)groups, 9), dim3(1024), 0, stream, hist, gsum, prefix, totals, tiles,
                       (uint32_t)groups);
    return hipGetLastError();
}

}  // namespace bc7
}  // namespace dxtlt
