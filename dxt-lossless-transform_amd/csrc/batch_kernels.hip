// batch_kernels.hip -- many buffers of one format, direction and settings in ONE launch (dxtlt_transform_batch_device).
//
// A texture of a few MiB cannot fill 256 CUs, and launching textures one by one is bound by the ~5 us a launch costs the
// host.  Here every workgroup looks up which buffer it belongs to and runs one tile of it: the tile the single-buffer call
// would pick for that buffer (bcn_device.h) -- aligned, forward halo or inverse shifted -- or the buffer's edge tile.
//
// What round 4 changed, and why (profiles/r03_batch_spacing.txt, profiles/r04_batch_*): the reference's own benchmark is
// a corpus of ~4 MiB DDS textures with mip chains (bc1-api README.MD:286-311), i.e. odd block counts, and on such buffers the
// round-3 kernel sat at 0.57-0.72 of the HBM peak where the single call does 0.82:
//   * heads and tails went through 256-block workgroups of the element path (one lane per block, 8-11 narrow stores per
//     lane) -- now ONE edge tile per buffer moves them as 16-byte vectors like every other tile, and a buffer owns exactly
//     ceil(blocks / T) (+1 forward when only stream tails are left) workgroups: no element path, no padding to multiples of 8;
//   * every workgroup decoded its entry and switched over variant / splits / tile form at run time: 58 scalar instructions per
//     wave on the one scalar unit a CU's four SIMDs share -- now the kernel is instantiated per settings (the host launches
//     one kernel per settings combination present in the batch; a corpus usually has one) and the entry carries the
//     per-stream bases (Shifts::gbase) ready-made instead of six 64-bit products per workgroup.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bcn_device.h"

namespace dxtlt {

// A table entry as the workgroup sees it: everything arrives through scalar (dword) loads -- byte fields read one by
// one would go through the vector memory path and add a second round trip before the tile's own load can start --
// and the buffer pointers are tagged as global memory again (a pointer that was loaded from memory is a generic one
// to the compiler, which would turn every access of the tile into a flat_* instruction).
struct BatchView {
    const uint8_t* src;
    uint8_t* dst;
    uint64_t blocks;
    uint32_t first_wg, end_wg, full_tiles;
    uint32_t flags;       // form | halo_vecs << 8 | natural << 16
    uint32_t shifts[2];   // shift[0..3], shift[4..5]
    uint64_t gbase[6];
};

__device__ __forceinline__ BatchView load_batch_entry(const BatchEntry* entry)
{
    const uint64_t* q = reinterpret_cast<const uint64_t*>(entry);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(entry);
    BatchView v;
    v.src = (const uint8_t*)(global_cptr)q[0];
    v.dst = (uint8_t*)(global_ptr)q[1];
    v.blocks = q[2];
    v.first_wg = w[6];
    v.end_wg = w[7];
    v.full_tiles = w[8];
    v.flags = w[9];
    v.shifts[0] = w[10];
    v.shifts[1] = w[11];
#pragma unroll
    for (int i = 0; i < 6; ++i)
        v.gbase[i] = q[6 + i];
    return v;
}
__device__ __forceinline__ void pin_batch_view(BatchView& v)
{
#ifdef DXTLT_WG_TIMING
    (void)v;   // (the experiment build's stores to its timing array make the table loads vector loads: nothing to pin)
    return;
#endif
    uint64_t s = reinterpret_cast<uintptr_t>(v.src), d = reinterpret_cast<uintptr_t>(v.dst);
    asm("" : "+s"(s), "+s"(d), "+s"(v.blocks));
    v.src = (const uint8_t*)(global_cptr)s;
    v.dst = (uint8_t*)(global_ptr)d;
    asm("" : "+s"(v.first_wg), "+s"(v.end_wg), "+s"(v.full_tiles), "+s"(v.flags), "+s"(v.shifts[0]), "+s"(v.shifts[1]));
#pragma unroll
    for (int i = 0; i < 6; ++i)
        asm("" : "+s"(v.gbase[i]));
}
static_assert(offsetof(BatchEntry, first_wg) == 24 && offsetof(BatchEntry, end_wg) == 28 && offsetof(BatchEntry, full_tiles) == 32 && offsetof(BatchEntry, form) == 36 &&
                  offsetof(BatchEntry, shift) == 40 && offsetof(BatchEntry, gbase) == 48,
              "load_batch_entry reads BatchEntry by dword offsets");

constexpr uint32_t kBatchAligned = 1;   // BatchEntry::form

// uniform_wgs != 0: every buffer of the launch owns exactly that many workgroups (buffers of one size -- the texture
// sets a batch is made for), so the owning entry is wg / uniform_wgs and ONE scalar load -- the entry, a line every
// workgroup of the buffer shares -- stands between the start of the workgroup and its tile's load instead of two
// dependent ones (coarse index, then entry).  The quotient comes from a multiply-high with magic = floor(2^32 /
// uniform_wgs): exact or one short for wg < 2^24, put right by one compare.
//
// strided.on: the batch is a regular array of buffers -- one size, sources and destinations each a constant stride apart (an
// array texture, the mip level of a texture set that a decompressor wrote into one allocation).  Its first entry then
// travels in the kernel arguments with the two strides, and a workgroup reaches its tile without any table load at all:
// entry = first entry with both pointers advanced by (wg / uniform_wgs) strides.
struct StridedBatch {
    BatchEntry first;
    int64_t src_stride, dst_stride;
    uint32_t on;
};

#ifdef DXTLT_WG_TIMING
// EXPERIMENT build only (tools/wg_timing_probe.py): per workgroup {100 MHz ticks from its first instruction to the acknowledgement of
// its last store, kind of tile, start tick (low 32 bits), XCC id}
__device__ uint32_t g_wg_timing[4 << 20];
extern "C" int dxtlt_debug_read_wg_timing(uint32_t* out, size_t count)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_timing), count * 4);
}
extern "C" int dxtlt_debug_read_wg_marks(uint32_t* out, size_t count)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dxtlt::g_wg_marks), count * 4);
}
#define WG_TIMING_BEGIN const uint64_t t_begin = __builtin_amdgcn_s_memrealtime();
#define WG_TIMING_END(kind)                                                                         \
    do {                                                                                            \
        __builtin_amdgcn_s_waitcnt(0);                                                              \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
        if (threadIdx.x == 0 && blockIdx.x < (1u << 20)) {                                          \
            uint32_t xcc;                                                                           \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                       \
            g_wg_timing[4 * blockIdx.x] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - t_begin);   \
            g_wg_timing[4 * blockIdx.x + 1] = (kind);                                               \
            g_wg_timing[4 * blockIdx.x + 2] = (uint32_t)t_begin;                                    \
            g_wg_timing[4 * blockIdx.x + 3] = xcc & 0xF;                                            \
        }                                                                                           \
    } while (0)
#else
#define WG_TIMING_BEGIN
#define WG_TIMING_END(kind)
#endif

// THREADS = batch_tile_threads(FMT, SC, INVERSE): the lanes of every tile of the launch (256; 128 for the forward launch of BC1
// without the colour split -- bcn_device.h has both measurements)
template <int FMT, int VARIANT, bool SA, bool SC, bool INVERSE, int THREADS = batch_tile_threads(FMT, SC, INVERSE)>
__global__ void __launch_bounds__(THREADS)
batch_kernel(const BatchEntry* __restrict__ entries_arg, const uint8_t* __restrict__ index_arg, uint32_t n_base, uint32_t uniform_wgs,
             uint32_t magic, StridedBatch strided)
{
    constexpr int kLds = INVERSE ? shift_lds_bytes(1, THREADS) : halo_lds_bytes<FMT, THREADS>();
    __shared__ __attribute__((aligned(16))) uint8_t lds[kLds];
    const uint32_t wg = blockIdx.x;
    WG_TIMING_BEGIN
    BatchView en;
    uint32_t e;   // index of the buffer in its launch
    // both table pointers in the FIRST scalar round trip (the compiler otherwise fetches `index` on the path that uses it, behind
    // the first wait); through integers: see fetched_now
    uint64_t entries_at = reinterpret_cast<uintptr_t>(entries_arg), index_at = reinterpret_cast<uintptr_t>(index_arg);
#ifndef DXTLT_WG_TIMING
    asm("" : "+s"(entries_at), "+s"(index_at), "+s"(n_base), "+s"(uniform_wgs), "+s"(magic));
#endif
    const BatchEntry* entries = (const BatchEntry*)(const __attribute__((address_space(1))) BatchEntry*)entries_at;
    const uint8_t* index = (const uint8_t*)(global_cptr)index_at;
    if (uniform_wgs != 0) {
        e = __umulhi(wg, magic);
        if ((e + 1) * uniform_wgs <= wg)
            ++e;
        if (strided.on != 0) {
            // kernel arguments: scalar loads off the kernarg pointer, nothing depends on a table
            const BatchEntry& f = strided.first;
            en.src = (const uint8_t*)(global_cptr)(reinterpret_cast<uintptr_t>(f.src) + (uint64_t)((int64_t)e * strided.src_stride));
            en.dst = (uint8_t*)(global_ptr)(reinterpret_cast<uintptr_t>(f.dst) + (uint64_t)((int64_t)e * strided.dst_stride));
            en.blocks = f.blocks;
            en.first_wg = e * uniform_wgs;
            en.end_wg = en.first_wg + uniform_wgs;
            en.full_tiles = f.full_tiles;
            en.flags = (uint32_t)f.form | ((uint32_t)f.halo_vecs << 8) | ((uint32_t)f.natural << 16);
            en.shifts[0] = (uint32_t)f.shift[0] | ((uint32_t)f.shift[1] << 8) | ((uint32_t)f.shift[2] << 16) | ((uint32_t)f.shift[3] << 24);
            en.shifts[1] = (uint32_t)f.shift[4] | ((uint32_t)f.shift[5] << 8);
#pragma unroll
            for (int i = 0; i < 6; ++i)
                en.gbase[i] = f.gbase[i];
        } else {
            en = load_batch_entry(entries + e);
        }
    } else {
        // base[wg / 4096] + delta[wg / 64] = the entry that owns workgroup 64 * (wg / 64) (bcn_launch.h); an entry carries its own
        // end, so a workgroup of a buffer of 64 workgroups or more is two dependent table loads away from its tile (three scalar
        // round trips with the kernel arguments).  What was measured on the way here (profiles/r04_batch_edge_tiles.txt; one
        // 2 GiB odd-count buffer or the corpus, forward): no table load 0.80, ONE load of an entry that thousands of workgroups
        // share 0.80 -- a scalar-cache hit costs next to nothing -- but an index record that every workgroup of a CU sees for the
        // first time 0.72-0.76 whatever it saves in round trips (a 144-byte record per 256 workgroups with the entry inline:
        // 0.72; a 16-byte bit mask per 64: 0.76; 4 bytes per 64: 0.77).  So the index is as small as it can be -- one byte per 64
        // workgroups, a cache line per 4096 -- and the entry, shared by all workgroups of its buffer, is what is fetched behind it.
        // Wide form (bit 31 of n_base; build_batch_index): 16-bit deltas, for launches in which more than 255 entries begin inside
        // one 4096-workgroup span -- thousands of buffers of one to three tiles -- where a byte would saturate and leave a walk of
        // up to ~3800 entries.  Branch-free on purpose: both forms issue the same two index loads.
        const uint32_t* base = reinterpret_cast<const uint32_t*>(index);
        const uint32_t wide = n_base >> 31;
        const uint8_t* delta = index + (n_base & 0x7FFFFFFFu) * 4;
        e = base[wg >> 12];
        const uint32_t dword = reinterpret_cast<const uint32_t*>(delta)[wg >> (8u - wide)];   // (a scalar load is a dword load)
        e += (dword >> (((wg >> 6) & (3u >> wide)) << (3u + wide))) & (0xFFu | (wide * 0xFF00u));
        en = load_batch_entry(entries + e);
        // every field is needed HERE (empty non-volatile asm: the value becomes opaque, memory is untouched, the loads stay
        // scalar): left alone the compiler fetches end_wg, runs the search and only then asks for the rest of the entry
        pin_batch_view(en);
        if (en.end_wg <= wg) {
            // Only buffers of fewer than 64 workgroups take this: `e` owns workgroup 64 * (wg / 64), so the owner of `wg` is one of
            // the next (wg & 63) entries.  Bisection over their end_wg fields -- at most six dependent dword loads, where walking
            // on entry by entry took up to 63 loads of a whole 96-byte entry (`magic` carries the entry count in this mode).
            uint32_t lo = e + 1, hi = e + (wg & 63u);
            hi = hi < magic - 1u ? hi : magic - 1u;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint32_t end = reinterpret_cast<const uint32_t*>(entries + mid)[7];   // BatchEntry::end_wg
                if (end <= wg)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            e = lo;
            en = load_batch_entry(entries + e);
        }
    }
    // Which of the buffer's tiles this workgroup takes.  Workgroups go to the eight XCDs round robin (wg % 8), and an edge tile
    // takes 1.2-1.9 x a whole tile's time (tools/wg_timing_probe.py): with buffers of 8 k workgroups each, launch order would put
    // EVERY buffer's last tile on one and the same XCD, which then runs 10 % behind the other seven -- the tail that made 4096
    // x (256 KiB - 1 block) run at 0.60 where 4096 x 16407 blocks (65 workgroups per buffer) ran at 0.71, and that round 3
    // built in by padding every buffer to a multiple of 8 workgroups (profiles/r04_batch_edge_tiles.txt section 7).  So the
    // buffer's tiles are rotated by r, chosen so that buffer e's last tile runs on XCD e % 8 (and its tile 0 on (e + 1) % 8).
    const uint32_t n_wgs = en.end_wg - en.first_wg;
    uint32_t local = wg - en.first_wg;
    if (n_wgs >= 8) {
        local += (en.first_wg + n_wgs - 1u - e) & 7u;
        local = local >= n_wgs ? local - n_wgs : local;
    }
    const bool aligned = (en.flags & 0xFF) == kBatchAligned;
    if (aligned && local < en.full_tiles) {
        // every stream base on a 128-byte line: the aligned tile, tiles in launch order (as the single-buffer call runs it)
        if constexpr (INVERSE)
            inv_aligned_tile<FMT, VARIANT, SA, SC, THREADS>(en.src, en.dst, en.blocks, 0, local, lds);
        else
            fwd_aligned_tile<FMT, VARIANT, SA, SC, THREADS>(en.src, en.dst, en.blocks, 0, local, lds);
        WG_TIMING_END(1);
        return;
    }
    Shifts sh;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        sh.d[i] = (int)((en.shifts[i >> 2] >> (8 * (i & 3))) & 127u);
        sh.gbase[i] = en.gbase[i];
    }
#ifdef DXTLT_EXPERIMENTS
    sh.xcd_remap = 0;
    sh.line_policy = INVERSE ? 1 : 3;
    sh.skip_partial = 0;
#endif
    sh.natural = 1;             // plan_batch_entry hands buffers with other shifts back to the host
    sh.halo_vecs = (int)(en.flags >> 8) & 0xFF;
    sh.full_tiles = en.full_tiles;
    sh.range_blocks = en.blocks;
    // The whole tile FIRST, with a return behind it: laid out behind the edge tile's code, the whole tile's block inherits the edge
    // tile's outstanding loads in the compiler's wait-count analysis (the two are alternatives, but the structurised control flow
    // falls through one into the other) and gets an s_waitcnt vmcnt(0) between its own two loads (tests/test_isa_invariants.py).
    if constexpr (INVERSE) {
        if (local < en.full_tiles) {
            // Neighbouring tiles share 128-byte lines: consecutive tiles stay on one XCD (xcd_contiguous_tile).  Workgroup
            // residues mod 8 are XCDs whatever the buffer's first workgroup and rotation are: equal residues of `local` meet on
            // one XCD (but for the few workgroups the rotation wraps around).
            const uint64_t tile = xcd_contiguous_tile(local, en.full_tiles);
            inv_shift_tile<FMT, VARIANT, SA, SC, THREADS>(en.src, en.dst, en.blocks, 0, sh, tile, lds);
            WG_TIMING_END(2);
            return;
        }
        inv_shift_edge_tile<FMT, VARIANT, SA, SC, THREADS>(en.src, en.dst, en.blocks, sh, en.full_tiles, lds);
        WG_TIMING_END(4);
    } else {
        if (local != 0 && local < en.full_tiles) {
            fwd_halo_tile<FMT, VARIANT, SA, SC, kNormNone, true, THREADS>(en.src, en.dst, en.blocks, 0, sh, local, lds);
            WG_TIMING_END(2);
            return;
        }
        fwd_halo_edge_tile<FMT, VARIANT, SA, SC, kNormNone, true, THREADS>(en.src, en.dst, sh, local, lds);
        WG_TIMING_END(local == 0 ? 3 : 4);
    }
}

uint32_t plan_batch_entry(Format fmt, bool inverse, const Settings& s, BatchEntry& e)
{
    if (e.blocks == 0)
        return 0;
    const bool sa = fmt == kBc3 && s.split_alpha;
    const Streams S = make_streams(fmt, sa, s.split_colour);
    const void* soa = inverse ? (const void*)e.src : (const void*)e.dst;
    // any AoS alignment: unaligned 16-byte vector accesses are exact and cheap on gfx950 (launch_transform)
    const uint64_t T = (uint64_t)tile_blocks(fmt, batch_tile_threads(fmt, s.split_colour, inverse));
    const uint64_t tiles = e.blocks / T, rest = e.blocks % T;
    // The tile forms of launch_transform: aligned tiles when every stream base is on a 128-byte line; otherwise forward halo
    // tiles (windows moved back to a 64-byte boundary) and inverse shifted tiles (slices displaced by the base modulo 16).
    const uint64_t mask = inverse ? 15 : (uint64_t)(kHaloAlign - 1);
    bool on_lines = true, stream_tails = false;
    int d[6] = {0, 0, 0, 0, 0, 0}, halo_blocks = 0;
    for (int i = 0; i < S.n; ++i) {
        const uint64_t base = reinterpret_cast<uintptr_t>(soa) + (uint64_t)S.off[i] * e.blocks;
        d[i] = (int)(base & mask);
        on_lines = on_lines && (base & 127) == 0;
        stream_tails = stream_tails || d[i] != 0;
        halo_blocks = std::max(halo_blocks, (d[i] + S.width[i] - 1) / S.width[i]);
    }
    if (!shifts_are_natural(S, d))
        return 0xFFFFFFFFu;   // a transformed-side pointer that is not even 8-byte aligned: the single-buffer call handles it
    e.form = on_lines ? 1 : 0;
    e.natural = 1;
    e.reserved = 0;
    e.reserved2[0] = e.reserved2[1] = 0;
    const int per_vec = 16 / fmt_block(fmt);
    e.halo_vecs = inverse ? 0 : (uint8_t)((halo_blocks + per_vec - 1) / per_vec);
    for (int i = 0; i < 6; ++i) {
        e.shift[i] = (uint8_t)d[i];
        e.gbase[i] = i < S.n ? (uint64_t)S.off[i] * e.blocks - (uint64_t)d[i] : 0;
    }
    e.full_tiles = (uint32_t)tiles;
    // the edge tile: the blocks behind the last whole tile and, forward, the last d_s bytes of every stream, which the whole
    // tiles' moved-back windows leave out
    const bool edge = rest != 0 || (!inverse && stream_tails);
    const uint32_t wgs = (uint32_t)tiles + (edge ? 1u : 0u);
    e.end_wg = e.first_wg + wgs;
    return wgs;
}

bool build_batch_index(const BatchEntry* entries, size_t n_entries, uint32_t total_wgs, uint8_t* index)
{
    const size_t n_base = batch_index_base_count(total_wgs), n_delta = batch_index_delta_count(total_wgs);
    uint32_t* base = reinterpret_cast<uint32_t*>(index);
    std::memset(index, 0, batch_index_bytes(total_wgs));
    // owner of workgroup 64 j for every j, relative to the owner of the span's first workgroup
    std::vector<uint32_t> d(n_delta);
    uint32_t largest = 0;
    size_t cur = 0;
    for (size_t j = 0; j < n_delta; ++j) {
        const uint32_t wg = (uint32_t)(j * kBatchIndexWgs);
        while (cur + 1 < n_entries && entries[cur].end_wg <= wg)
            ++cur;
        if (wg % kBatchBaseWgs == 0)
            base[wg / kBatchBaseWgs] = (uint32_t)cur;
        d[j] = (uint32_t)(cur - base[wg / kBatchBaseWgs]);   // < 4096: every entry owns at least one workgroup
        largest = std::max(largest, d[j]);
    }
    const bool wide = largest > 255;
    uint8_t* delta = index + n_base * 4;
    for (size_t j = 0; j < n_delta; ++j) {
        if (wide) {
            delta[2 * j] = (uint8_t)d[j];
            delta[2 * j + 1] = (uint8_t)(d[j] >> 8);
        } else {
            delta[j] = (uint8_t)d[j];
        }
    }
    return wide;
}

namespace {

using BatchFn = void (*)(const BatchEntry*, const uint8_t*, uint32_t, uint32_t, uint32_t, StridedBatch);

template <int FMT, int VARIANT, bool SA, bool SC>
BatchFn batch_fn(bool inverse) { return inverse ? batch_kernel<FMT, VARIANT, SA, SC, true> : batch_kernel<FMT, VARIANT, SA, SC, false>; }

template <int FMT, int VARIANT>
BatchFn batch_splits(bool sa, bool sc, bool inverse)
{
    if constexpr (FMT == kBc3) {
        if (sa)
            return sc ? batch_fn<FMT, VARIANT, true, true>(inverse) : batch_fn<FMT, VARIANT, true, false>(inverse);
    }
    return sc ? batch_fn<FMT, VARIANT, false, true>(inverse) : batch_fn<FMT, VARIANT, false, false>(inverse);
}

template <int FMT>
BatchFn batch_variant(int variant, bool sa, bool sc, bool inverse)
{
    switch (variant) {
    case kNone: return batch_splits<FMT, kNone>(sa, sc, inverse);
    case kVar1: return batch_splits<FMT, kVar1>(sa, sc, inverse);
    case kVar2: return batch_splits<FMT, kVar2>(sa, sc, inverse);
    default: return batch_splits<FMT, kVar3>(sa, sc, inverse);
    }
}

}  // namespace

hipError_t launch_batch(Format fmt, bool inverse, const Settings& s, const BatchEntry* d_entries, const uint8_t* d_index,
                        uint32_t n_entries, uint32_t total_wgs, uint32_t uniform_wgs, bool wide_index, hipStream_t stream,
                        const BatchEntry* strided_first, int64_t src_stride, int64_t dst_stride)
{
    if (n_entries == 0 || total_wgs == 0)
        return hipSuccess;
    if (s.variant < 0 || s.variant > 3 || total_wgs > 0xFFFFFFu)
        return hipErrorInvalidValue;
    if (uniform_wgs != 0 && (uint64_t)uniform_wgs * n_entries != total_wgs)
        return hipErrorInvalidValue;
    // uniform: floor(2^32 / uniform_wgs), the kernel's multiply-high is exact or one short and one compare puts it right.  One
    // workgroup per buffer (thousands of textures of a tile or less): 2^32 - 1 stands in for 2^32 -- the product's high word is
    // wg - 1 (0 for wg 0), the same compare makes it wg: e == wg without a table lookup.  General lookup: the entry count (the
    // bound of the kernel's bisection).
    const uint32_t magic = uniform_wgs > 1 ? (uint32_t)((1ull << 32) / uniform_wgs) : uniform_wgs == 1 ? 0xFFFFFFFFu : n_entries;
    static const bool no_array = experiment_env("DXTLT_BATCH_NO_ARRAY") != nullptr;   // A/B switch (experiments build)
    if (strided_first != nullptr && uniform_wgs != 0 && !no_array && strided_first->form == 1) {
        const hipError_t e = launch_tiled_array(fmt, inverse, s, strided_first->src, strided_first->dst, strided_first->blocks, n_entries,
                                                src_stride, dst_stride, stream);
        if (e != hipErrorNotSupported)
            return e;
    }
    StridedBatch strided{};
    if (strided_first != nullptr && uniform_wgs != 0) {
        strided.first = *strided_first;
        strided.src_stride = src_stride;
        strided.dst_stride = dst_stride;
        strided.on = 1;
    }
    const bool sa = fmt == kBc3 && s.split_alpha, sc = s.split_colour;
    BatchFn k = nullptr;
    switch (fmt) {
    case kBc1: k = batch_variant<kBc1>(s.variant, false, sc, inverse); break;
    case kBc2: k = batch_variant<kBc2>(s.variant, false, sc, inverse); break;
    case kBc3: k = batch_variant<kBc3>(s.variant, sa, sc, inverse); break;
    default: return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(k, dim3(total_wgs), dim3(batch_tile_threads(fmt, sc, inverse)), 0, stream, d_entries, d_index,
                       (uint32_t)batch_index_base_count(total_wgs) | (wide_index ? 0x80000000u : 0u), uniform_wgs, magic, strided);
    return hipGetLastError();
}

}  // namespace dxtlt
