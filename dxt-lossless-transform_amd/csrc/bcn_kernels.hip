// bcn_kernels.hip -- gfx950 (MI355X, CDNA4) kernels for the BCn block transform hot path.
//
// What is computed (reference, paths under /root/reference/src/core/):
//   BC1  dxt-lossless-transform-bc1/src/transform/transform_with_settings.rs:31-72 / 92-135
//   BC2  dxt-lossless-transform-bc2/src/transform/transform_with_settings.rs:30-73 / 93-138
//   BC3  dxt-lossless-transform-bc3/src/transform/transform_with_settings.rs:32-142 / 162-272
//   YCoCg-R  dxt-lossless-transform-common/src/color_565/decorrelate.rs:101-344 (ycocg_swar.h)
// i.e. an AoS -> SoA field split of 8-/16-byte blocks into 2..6 adjacent streams, optional endpoint
// split, optional YCoCg-R of the two RGB565 endpoints; and the inverse.  Output length == input length.
//
// How it is mapped to the machine (this is NOT how the reference does it; the reference is x86 SIMD):
//   * HBM-bound byte shuffling: 2 bytes of traffic per byte of input, ~20 integer ops per colour pair.  No MFMA.  What matters
//     is that every HBM access is a full-width, fully coalesced 16 B/lane vector access and that the memory system always has
//     thousands of independent, short-lived workgroups to pull from.
//   * One workgroup owns ONE contiguous tile of blocks: THREADS x 16 bytes (BC1 128 lanes = 256 blocks, BC2 / BC3 256 lanes = 256
//     blocks; profiles/r01_q_*).  Forward: one global_load_dwordx4 per lane -> YCoCg-R on packed colour pairs -> each field goes
//     to its place in an LDS image laid out exactly like the output (stream after stream) -> barrier -> the image is read back
//     linearly (ds_read_b128) and leaves with one global_store_dwordx4 per lane; every stream slice of a tile is one contiguous
//     run.  Inverse: the mirror.  One tile per workgroup, no loop: 2^21-2^23 workgroups for 8 GiB.  A persistent grid-stride
//     loop reached 0.58-0.60 of the 8 TB/s peak where this reaches 0.84-0.86 and a plain copy 0.82-0.85 (profiles/r01_b_*,
//     r01_c_*, r04_pipelined_tiles.txt): the hardware dispatcher is the better scheduler for a pure stream.
//   * `nt` loads for bytes touched once; write-through streaming stores (`sc1 nt`, streaming_store.h) for 128-byte lines a
//     workgroup writes whole; plain `nt` stores where a line is shared with another workgroup (profiles/r01_p, r01_q).
//   * 64-bit block indices and byte offsets everywhere (8 GiB of BC1 = 2^30 blocks); launches of at most 2^31 blocks.
//
// Which tile a range gets (launch_transform; the batch kernel picks the same per buffer) -- a property of the addresses only:
//   * ALIGNED tiles (fwd_tiled / inv_tiled): every stream base soa + off_s * N + w_s * first_block on a 128-byte line.
//   * Stream bases anywhere (odd block counts -- every DDS payload with a mip chain -- or ranges starting at odd blocks):
//       forward HALO tiles (fwd_tiled_halo): each stream's window is moved back to a 64-byte boundary; the up-to-63 bytes in
//         front are records of blocks before the tile, which the workgroup loads too (temporal loads: the next tile re-reads
//         them out of the memory-side cache), and what leaves is one whole aligned 16-byte store per lane;
//       inverse SHIFTED tiles (inv_tiled_shift): each slice of the LDS image sits `base & 15` bytes further in, so LDS and global
//         addresses agree modulo 16 and whole aligned segments are loaded; XCD-contiguous tile order, because neighbouring
//         tiles share their first and last line (profiles/r03_inv_shift_pmc.txt).
//     The AoS side may sit at any byte address (a DDS file resident in HBM has its payload at byte 128 or 148): 16-byte vector
//     accesses at unaligned addresses are exact on gfx950 and cost 0.02-0.06 (tools/unaligned_lab.hip).
//   * EDGE tiles (fwd_halo_edge_tile / inv_shift_edge_tile): the first tile of a halo range and the ragged last tile of any
//     range are the same tiles with masks -- whole 16-byte vectors wherever a segment lies inside the stream, a few narrow
//     pieces where a stream begins or ends inside one, nothing outside.  They run as one more workgroup of the same launch
//     (halo / shifted) or as one more launch of one workgroup (behind aligned tiles).  There is no element-granular path.
//
// Experiments (the element-granular kernel, the first form of the forward shifted tiles, run-time store policies / tile orders
// and the wrong-output timing switch the measurements in profiles/ were taken with) are compiled only with -DDXTLT_EXPERIMENTS,
// a side build (bcn_device.h); this file then also honours the `force_path` bits documented at launch_transform.
#include "bcn_device.h"

namespace dxtlt {

// ------------------------------------------------------------------------------------------------
// Synthetic data: counter-based splitmix64, one qword per lane-iteration (bench / test plumbing).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(kThreads)
fill_splitmix64_kernel(uint8_t* __restrict__ dst, uint64_t len_bytes, uint64_t seed, uint64_t first_qword)
{
    const uint64_t qwords = len_bytes / 8;
    const uint64_t stride = (uint64_t)gridDim.x * kThreads;
    const bool al8 = aligned_to(dst, 8);
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < qwords; i += stride)
        store_bytes<8>(dst + 8 * i, splitmix64_at(seed, first_qword + i), al8);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint64_t rem = len_bytes - 8 * qwords;
        const uint64_t v = splitmix64_at(seed, first_qword + qwords);
        for (uint64_t k = 0; k < rem; ++k)
            dst[8 * qwords + k] = (uint8_t)(v >> (8 * k));
    }
}

// ------------------------------------------------------------------------------------------------
// Host-side dispatch
// ------------------------------------------------------------------------------------------------
namespace {

using TiledFn = void (*)(const uint8_t*, uint8_t*, uint64_t, uint64_t, int, int64_t, int64_t);
using ShiftFn = void (*)(const uint8_t*, uint8_t*, uint64_t, uint64_t, Shifts);
#ifdef DXTLT_EXPERIMENTS
using GenericFn = void (*)(const uint8_t*, uint8_t*, uint64_t, uint64_t, uint64_t, uint64_t);
#endif

struct KernelSet {
    TiledFn tiled[4];  // aligned tiles of 64, 128, 256, 512 threads
    ShiftFn shifted;   // inverse: shifted tiles + edge tile (shift_tile_threads(fmt) lanes); forward: nullptr (experiments build: the first form, 256)
    ShiftFn halo[2];   // forward: halo tiles + edge tiles, halo_tile_threads(fmt, split_colour) lanes ([1]: natural shifts); inverse: nullptr
#ifdef DXTLT_EXPERIMENTS
    ShiftFn halo512;   // natural shifts: 512-lane halo tiles, dxtlt_set_tuning(512, ...) -- measured and not adopted (profiles/r05_halo_512.txt)
#endif
#ifdef DXTLT_EXPERIMENTS
    GenericFn generic;
#endif
};

inline int threads_slot(int threads) { return threads == 64 ? 0 : threads == 128 ? 1 : threads == 512 ? 3 : 2; }

template <int FMT, int VARIANT, bool SA, bool SC>
KernelSet kernels_for(bool inverse)
{
    KernelSet ks{};
    if (inverse) {
        ks.tiled[0] = inv_tiled<FMT, VARIANT, SA, SC, 64>;
        ks.tiled[1] = inv_tiled<FMT, VARIANT, SA, SC, 128>;
        ks.tiled[2] = inv_tiled<FMT, VARIANT, SA, SC, 256>;
        ks.tiled[3] = inv_tiled<FMT, VARIANT, SA, SC, 512>;
        ks.shifted = inv_tiled_shift<FMT, VARIANT, SA, SC, shift_tile_threads(FMT)>;
    } else {
        ks.tiled[0] = fwd_tiled<FMT, VARIANT, SA, SC, 64>;
        ks.tiled[1] = fwd_tiled<FMT, VARIANT, SA, SC, 128>;
        ks.tiled[2] = fwd_tiled<FMT, VARIANT, SA, SC, 256>;
        ks.tiled[3] = fwd_tiled<FMT, VARIANT, SA, SC, 512>;
        ks.halo[0] = fwd_tiled_halo<FMT, VARIANT, SA, SC, kNormNone, false, halo_tile_threads(FMT, SC)>;
        ks.halo[1] = fwd_tiled_halo<FMT, VARIANT, SA, SC, kNormNone, true, halo_tile_threads(FMT, SC)>;
#ifdef DXTLT_EXPERIMENTS
        ks.halo512 = fwd_tiled_halo<FMT, VARIANT, SA, SC, kNormNone, true, 512>;
#endif
    }
#ifdef DXTLT_EXPERIMENTS
    if (inverse) {
        ks.generic = generic_kernel<FMT, VARIANT, SA, SC, true>;
    } else {
        ks.shifted = fwd_tiled_shift<FMT, VARIANT, SA, SC>;
        ks.generic = generic_kernel<FMT, VARIANT, SA, SC, false>;
    }
#endif
    return ks;
}

template <int FMT, int VARIANT>
KernelSet pick_splits(bool sa, bool sc, bool inverse)
{
    if constexpr (FMT == kBc3) {
        if (sa)
            return sc ? kernels_for<FMT, VARIANT, true, true>(inverse) : kernels_for<FMT, VARIANT, true, false>(inverse);
        return sc ? kernels_for<FMT, VARIANT, false, true>(inverse) : kernels_for<FMT, VARIANT, false, false>(inverse);
    } else {
        return sc ? kernels_for<FMT, VARIANT, false, true>(inverse) : kernels_for<FMT, VARIANT, false, false>(inverse);
    }
}

template <int FMT>
KernelSet pick_variant(int variant, bool sa, bool sc, bool inverse)
{
    switch (variant) {
    case kNone: return pick_splits<FMT, kNone>(sa, sc, inverse);
    case kVar1: return pick_splits<FMT, kVar1>(sa, sc, inverse);
    case kVar2: return pick_splits<FMT, kVar2>(sa, sc, inverse);
    default: return pick_splits<FMT, kVar3>(sa, sc, inverse);
    }
}

// BC1 forward with block normalisation fused in: one tile size (the BC1 default), halo / edge tiles
template <int VARIANT, bool SC, int NORM>
KernelSet bc1_norm_kernels()
{
    constexpr int TH = default_tile_threads(kBc1, false);
    TiledFn tiled = fwd_tiled<kBc1, VARIANT, false, SC, TH, NORM>;
    KernelSet ks{};
    ks.tiled[0] = ks.tiled[1] = ks.tiled[2] = ks.tiled[3] = tiled;
    ks.halo[0] = fwd_tiled_halo<kBc1, VARIANT, false, SC, NORM, false, halo_tile_threads(kBc1, SC)>;
    ks.halo[1] = fwd_tiled_halo<kBc1, VARIANT, false, SC, NORM, true, halo_tile_threads(kBc1, SC)>;
#ifdef DXTLT_EXPERIMENTS
    ks.shifted = fwd_tiled_shift<kBc1, VARIANT, false, SC, NORM>;
    ks.generic = generic_kernel<kBc1, VARIANT, false, SC, false, NORM>;
#endif
    return ks;
}

template <int NORM>
KernelSet pick_bc1_norm(int variant, bool sc)
{
    switch (variant) {
    case kNone: return sc ? bc1_norm_kernels<kNone, true, NORM>() : bc1_norm_kernels<kNone, false, NORM>();
    case kVar1: return sc ? bc1_norm_kernels<kVar1, true, NORM>() : bc1_norm_kernels<kVar1, false, NORM>();
    case kVar2: return sc ? bc1_norm_kernels<kVar2, true, NORM>() : bc1_norm_kernels<kVar2, false, NORM>();
    default: return sc ? bc1_norm_kernels<kVar3, true, NORM>() : bc1_norm_kernels<kVar3, false, NORM>();
    }
}

int cached_cu_count()
{
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return 256;
    if (cus[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        cus[dev] = v;
    }
    return cus[dev];
}

}  // namespace

// force_path bits of dxtlt_set_tuning as this build honours them (include/dxtlt_gfx950.h, dxtlt_tuning_mask()).
//   product:      2 = halo / shifted tiles even for aligned stream bases; 0x20 = the generic (switch per access) LDS scatter /
//                 gather of the halo / shifted tiles even for natural shifts.  Both are test levers: every path they select is
//                 a path some address pattern selects by itself, and the results are exact.
//   experiments:  1 = element-granular kernel; 0x10 = leave out partial segments / the halo load (TIMING ONLY, WRONG OUTPUT; also
//                 needs DXTLT_TIMING_EXPERIMENTS in the environment); 0x40 / 0x80 = store policies of the first-form shifted
//                 tiles; 0x100 / 0x200 = XCD-contiguous tile order off / on; 0x400 = first form of the forward shifted tiles;
//                 0x800 = plain nt stores in the halo tiles; 0x1000 = round-1 rule (tiles only for a 16-byte aligned AoS
//                 pointer); 0x2000 = round-3 routing (heads and tails through the element kernel).
#ifdef DXTLT_EXPERIMENTS
constexpr int kForceMask = 0x3CFF;
#else
constexpr int kForceMask = 2 | 0x20;
#endif
int launch_force_mask() { return kForceMask; }

// ---- planning: which launches a range takes (pure host arithmetic on addresses; dxtlt_debug_plan_transform exposes it to the
// CPU tests, launch_transform executes it) ---------------------------------------------------------------------------------------
namespace {

struct PlannedLaunch {
    int kind;             // 0: aligned tiles; 1: halo tiles + edge tiles (forward); 2: shifted tiles + edge tile (inverse)
    int threads;          // lanes per workgroup
    uint32_t grid;        // workgroups
    uint64_t aos_offset;  // bytes from the range's AoS pointer to the launch's (an edge tile behind aligned tiles starts further in)
    Shifts sh;            // kinds 1 and 2
};

// One range of at most 2^31 blocks.  `aligned_threads`: lanes of the aligned tiles (default or tuning); `halo_threads_override`:
// 0, or the experiments build's 512.  emit(const PlannedLaunch&) -> hipError_t; the first error ends the walk.
template <typename Emit>
hipError_t plan_launches(Format fmt, bool inverse, bool sa, bool sc, const void* soa, const Range& r, int force_bits,
                         int aligned_threads, int halo_threads_override, Emit&& emit)
{
    // Which tiles take the range?  (the table at the top of this file)  The AoS side may sit at any byte address: 16-byte vector
    // loads / stores at unaligned addresses are exact on gfx950 under ROCm's default memory mode and cost little
    // (tools/unaligned_lab.hip: a streaming copy at 0.845 of peak drops to 0.81-0.83 with misaligned loads, 0.78-0.79 with
    // misaligned stores).
    const Streams S = make_streams(fmt, sa, sc);
    auto stream_base = [&](int i, uint64_t first_block) {
        return reinterpret_cast<uintptr_t>(soa) + (uint64_t)S.off[i] * r.total_blocks + (uint64_t)S.width[i] * first_block;
    };
    // The aligned tiles need 16-byte aligned stream bases to be correct and 128-byte aligned ones to be fast: with bases that
    // are only 16-byte aligned every slice shares its first and last line with the neighbouring tiles, which the halo / shifted
    // tiles handle and the aligned ones do not -- BC3 forward 0.65 against 0.77 of peak (profiles/r01_z/shift_probe.txt).
    bool any_shift = false;
    for (int i = 0; i < S.n; ++i)
        any_shift = any_shift || (stream_base(i, r.first_block) & 127) != 0;
    const bool use_shift = any_shift || (force_bits & 3) == 2;

    // Shifts of the range [first_block + local_first, ...) for the halo tiles (stream base mod 64) or the shifted tiles (mod 16)
    auto shifts_of = [&](uint64_t local_first, bool halo) {
        Shifts sh{};
        const uint64_t mask = halo ? (uint64_t)(kHaloAlign - 1) : 15;
        int halo_blocks = 0;
        for (int i = 0; i < S.n; ++i) {
            sh.d[i] = (int)(stream_base(i, r.first_block + local_first) & mask);
            halo_blocks = std::max(halo_blocks, (sh.d[i] + S.width[i] - 1) / S.width[i]);
        }
        fill_gbase(sh, S, r.total_blocks, r.first_block + local_first);
        sh.natural = (force_bits & 0x20) ? 0 : shifts_are_natural(S, sh.d);
        // only the blocks that have bytes inside a window are fetched as halo: max over the streams of ceil(d_s / w_s) blocks
        const int per_vec = 16 / fmt_block(fmt);
        sh.halo_vecs = halo ? (halo_blocks + per_vec - 1) / per_vec : 0;
        return sh;
    };
    // Lanes of the halo / shifted / edge tiles (bcn_device.h): 256, and 128 for the forward tiles of BC1 without the colour split.
    const int edge_threads = inverse ? shift_tile_threads(fmt) : halo_tile_threads(fmt, sc);
    if (use_shift) {
        Shifts sh = shifts_of(0, !inverse);
        const int threads = (!inverse && halo_threads_override != 0 && sh.natural) ? halo_threads_override : edge_threads;
        const uint64_t T = (uint64_t)tile_blocks(fmt, threads);
        const uint64_t num_tiles = r.num_blocks / T, rest = r.num_blocks - num_tiles * T;
        // Forward: halo tiles + edge tiles in ONE launch -- tile 0 writes the head of every stream itself, and one more workgroup
        // takes the blocks behind the last whole tile and the last d_s bytes of every stream.  Identity tile order, write-through
        // stores: every window starts on a 64-byte sector, and the halo -- re-read by the next tile, on another XCD -- comes out of
        // the memory-side cache (temporal loads; bcn_device.h).  Inverse: shifted tiles + the edge tile in one launch;
        // XCD-contiguous tile order (neighbouring tiles share lines).
        bool tail = rest > 0;
        if (!inverse)
            for (int i = 0; i < S.n; ++i)
                tail = tail || sh.d[i] > 0;
        sh.full_tiles = (uint32_t)num_tiles;
        sh.range_blocks = r.num_blocks;
        return emit(PlannedLaunch{inverse ? 2 : 1, threads, (uint32_t)(num_tiles + (tail ? 1 : 0)), 0, sh});
    }
    const uint64_t T = (uint64_t)tile_blocks(fmt, aligned_threads);
    const uint64_t num_tiles = r.num_blocks / T;
    if (num_tiles > 0)
        if (hipError_t e = emit(PlannedLaunch{0, aligned_threads, (uint32_t)num_tiles, 0, Shifts{}}); e != hipSuccess)
            return e;
    // Behind aligned tiles (or a range smaller than a tile): the rest as edge tiles -- a halo tile 0 forward (no halo; writes
    // every stream from its first byte to its last), a shifted tile 0 inverse -- of up to edge_threads lanes' worth of blocks
    // each (one, unless a tuning size made the aligned tiles larger than that).
    const uint64_t T_edge = (uint64_t)tile_blocks(fmt, edge_threads);
    for (uint64_t at = num_tiles * T; at < r.num_blocks; at += T_edge) {
        Shifts e = shifts_of(at, !inverse);
        e.full_tiles = 0;
        e.range_blocks = std::min(T_edge, r.num_blocks - at);
        if (hipError_t err = emit(PlannedLaunch{inverse ? 2 : 1, edge_threads, 1, at * (uint64_t)fmt_block(fmt), e}); err != hipSuccess)
            return err;
    }
    return hipSuccess;
}

// HIP refuses a launch of 2^32 or more threads (grid x workgroup), which a 64 GiB buffer reaches at 16 bytes per lane.  Larger
// ranges go out as consecutive sub-ranges of 2^31 blocks (a multiple of every tile size, so stream alignment and tile boundaries
// are the same as in one launch): the AoS side advances, the SoA side is addressed through first_block as always.
constexpr uint64_t kMaxBlocksPerLaunch = 1ull << 31;

int aligned_tile_threads(Format fmt, bool inverse, bool normalizing, const LaunchTuning* tuning)
{
    int threads = default_tile_threads(fmt, inverse);
    if (tuning && !normalizing && (tuning->tile_threads == 64 || tuning->tile_threads == 128 || tuning->tile_threads == 256 ||
                                   tuning->tile_threads == 512))
        threads = tuning->tile_threads;   // (normalisation: the default is the only tile size instantiated)
    return threads;
}

}  // namespace

int debug_plan_transform(Format fmt, bool inverse, const Settings& s, uint64_t src_address, uint64_t dst_address, const Range& r,
                         const LaunchTuning* tuning, DebugPlannedLaunch* out, int cap)
{
    if (s.variant < 0 || s.variant > 3 || r.first_block + r.num_blocks > r.total_blocks || (fmt != kBc1 && fmt != kBc2 && fmt != kBc3))
        return -1;
    const bool sa = fmt == kBc3 && s.split_alpha, sc = s.split_colour;
    const int force_bits = (tuning ? tuning->force_generic : 0) & kForceMask & (2 | 0x20);   // (the product's levers)
    const void* soa = reinterpret_cast<const void*>(static_cast<uintptr_t>(inverse ? src_address : dst_address));
    int n = 0;
    for (uint64_t off = 0; off < r.num_blocks; off += kMaxBlocksPerLaunch) {
        const Range sub{r.total_blocks, r.first_block + off, std::min(kMaxBlocksPerLaunch, r.num_blocks - off)};
        const uint64_t sub_aos = off * (uint64_t)fmt_block(fmt);
        const hipError_t e = plan_launches(fmt, inverse, sa, sc, soa, sub, force_bits, aligned_tile_threads(fmt, inverse, false, tuning), 0,
                                           [&](const PlannedLaunch& l) {
                                               if (n < cap) {
                                                   DebugPlannedLaunch& o = out[n];
                                                   o.kind = l.kind;
                                                   o.threads = l.threads;
                                                   o.workgroups = l.grid;
                                                   o.full_tiles = l.kind ? l.sh.full_tiles : l.grid;
                                                   o.range_blocks = l.kind ? l.sh.range_blocks : (uint64_t)l.grid * (uint64_t)tile_blocks(fmt, l.threads);
                                                   o.aos_offset = sub_aos + l.aos_offset;
                                                   for (int i = 0; i < 6; ++i) {
                                                       o.shift[i] = (uint8_t)l.sh.d[i];
                                                       o.gbase[i] = l.sh.gbase[i];
                                                   }
                                                   o.halo_vecs = (uint8_t)l.sh.halo_vecs;
                                                   o.natural = (uint8_t)l.sh.natural;
                                               }
                                               ++n;
                                               return hipSuccess;
                                           });
        if (e != hipSuccess)
            return -1;
    }
    return n;
}

hipError_t launch_transform(Format fmt, bool inverse, const Settings& s, const void* src, void* dst,
                            const Range& r, hipStream_t stream, const LaunchTuning* tuning)
{
    if (r.num_blocks == 0)
        return hipSuccess;
    if (s.variant < 0 || s.variant > 3 || r.first_block + r.num_blocks > r.total_blocks)
        return hipErrorInvalidValue;
    if (s.normalize != kNormNone && (fmt != kBc1 || inverse || s.normalize < 0 || s.normalize > kNormTransparentOnly))
        return hipErrorInvalidValue;  // normalisation exists for the BC1 forward transform only

    if (r.num_blocks > kMaxBlocksPerLaunch) {
        const uint64_t block_bytes = fmt_block(fmt);
        for (uint64_t off = 0; off < r.num_blocks; off += kMaxBlocksPerLaunch) {
            const uint64_t n = r.num_blocks - off < kMaxBlocksPerLaunch ? r.num_blocks - off : kMaxBlocksPerLaunch;
            const Range sub{r.total_blocks, r.first_block + off, n};
            const void* sub_src = inverse ? src : static_cast<const void*>(static_cast<const uint8_t*>(src) + off * block_bytes);
            void* sub_dst = inverse ? static_cast<void*>(static_cast<uint8_t*>(dst) + off * block_bytes) : dst;
            if (hipError_t e = launch_transform(fmt, inverse, s, sub_src, sub_dst, sub, stream, tuning); e != hipSuccess)
                return e;
        }
        return hipSuccess;
    }

    const bool sa = (fmt == kBc3) && s.split_alpha;
    const bool sc = s.split_colour;
    const bool normalizing = s.normalize != kNormNone;
    KernelSet ks;
    switch (fmt) {
    case kBc1:
        if (s.normalize == kNormColor0Only) ks = pick_bc1_norm<kNormColor0Only>(s.variant, sc);
        else if (s.normalize == kNormReplicateColor) ks = pick_bc1_norm<kNormReplicateColor>(s.variant, sc);
        else if (s.normalize == kNormTransparentOnly) ks = pick_bc1_norm<kNormTransparentOnly>(s.variant, sc);
        else ks = pick_variant<kBc1>(s.variant, false, sc, inverse);
        break;
    case kBc2: ks = pick_variant<kBc2>(s.variant, false, sc, inverse); break;
    case kBc3: ks = pick_variant<kBc3>(s.variant, sa, sc, inverse); break;
    default: return hipErrorInvalidValue;
    }

    const uint8_t* src8 = static_cast<const uint8_t*>(src);
    uint8_t* dst8 = static_cast<uint8_t*>(dst);
    const void* soa = inverse ? src : (const void*)dst;
    const int force_bits = (tuning ? tuning->force_generic : 0) & kForceMask;
    const int threads = aligned_tile_threads(fmt, inverse, normalizing, tuning);
    int halo_threads_override = 0;
    ShiftFn halo_alt = nullptr;

#ifdef DXTLT_EXPERIMENTS
    // ---- experiments build: 512-lane halo tiles and the routes of earlier rounds ------------------------------------------------
    if (!inverse && threads == 512 && ks.halo512 != nullptr) {
        halo_threads_override = 512;
        halo_alt = ks.halo512;
    }
    const int remap_override = tuning ? tuning->xcd_remap : -1;
    auto element_range = [&](uint64_t local_first, uint64_t count) -> hipError_t {
        if (count == 0)
            return hipSuccess;
        const uint64_t grid = (count + kThreads - 1) / kThreads;   // count <= 2^31 blocks, one per thread
        hipLaunchKernelGGL(ks.generic, dim3((unsigned)grid), dim3(kThreads), 0, stream, src8, dst8, r.total_blocks,
                           r.first_block, local_first, count);
        return hipGetLastError();
    };
    {
        const Streams S = make_streams(fmt, sa, sc);
        bool any_shift = false;
        for (int i = 0; i < S.n; ++i)
            any_shift = any_shift || ((reinterpret_cast<uintptr_t>(soa) + (uint64_t)S.off[i] * r.total_blocks +
                                       (uint64_t)S.width[i] * r.first_block) & 127) != 0;
        const bool use_shift = any_shift || (force_bits & 3) == 2;
        const void* aos = inverse ? (const void*)dst : src;
        const bool aos_ok = (reinterpret_cast<uintptr_t>(aos) & 15) == 0 || !(force_bits & 0x1000);
        if (!aos_ok || (force_bits & 3) == 1)
            return element_range(0, r.num_blocks);
        const bool first_form = use_shift && !inverse && (force_bits & 0x400);
        const bool old_routing = (force_bits & 0x2000) != 0;
        if (first_form || (old_routing && use_shift)) {
            const int route_threads = first_form ? 256 : inverse ? shift_tile_threads(fmt) : halo_tile_threads(fmt, sc);   // (the first form: 256 lanes only)
            const uint64_t T = (uint64_t)tile_blocks(fmt, route_threads);
            const uint64_t tiles = r.num_blocks / T;
            const bool halo = !inverse && !first_form;
            Shifts sh{};
            int halo_blocks = 0;
            for (int i = 0; i < S.n; ++i) {
                const uint64_t base = reinterpret_cast<uintptr_t>(soa) + (uint64_t)S.off[i] * r.total_blocks + (uint64_t)S.width[i] * r.first_block;
                sh.d[i] = (int)(base & (uint64_t)(halo ? kHaloAlign - 1 : 15));
                halo_blocks = std::max(halo_blocks, (sh.d[i] + S.width[i] - 1) / S.width[i]);
            }
            fill_gbase(sh, S, r.total_blocks, r.first_block);
            sh.natural = (force_bits & 0x20) ? 0 : shifts_are_natural(S, sh.d);
            sh.halo_vecs = halo ? (halo_blocks + 16 / fmt_block(fmt) - 1) / (16 / fmt_block(fmt)) : 0;
            sh.xcd_remap = remap_override >= 0 ? remap_override : halo ? 0 : 1;
            sh.skip_partial = (force_bits & 0x10) ? 1 : 0;
            sh.line_policy = !first_form ? ((force_bits & 0x800) ? 1 : 3) : (force_bits & 0x40) ? 0 : (force_bits & 0x80) ? 2 : 1;
            sh.full_tiles = (uint32_t)tiles;
            sh.range_blocks = tiles * T;
            if (tiles > 0) {
                hipLaunchKernelGGL(first_form || inverse ? ks.shifted : ks.halo[sh.natural ? 1 : 0], dim3((unsigned)tiles),
                                   dim3(route_threads), 0, stream, src8, dst8, r.total_blocks, r.first_block, sh);
                if (hipError_t e = hipGetLastError(); e != hipSuccess)
                    return e;
            }
            uint64_t done = tiles * T;
            if (halo && tiles > 0) {
                // what the halo tiles' windows leave out: the head of every stream of the range (records of its first 64 blocks)
                // and everything behind the last window
                if (hipError_t e = element_range(0, kHaloBlocks); e != hipSuccess)
                    return e;
                done -= kHaloBlocks;
            }
            return element_range(done, r.num_blocks - done);
        }
        if (old_routing) {
            const uint64_t T = (uint64_t)tile_blocks(fmt, threads);
            const uint64_t tiles = r.num_blocks / T;
            if (tiles > 0) {
                hipLaunchKernelGGL(ks.tiled[threads_slot(threads)], dim3((unsigned)tiles), dim3(threads), 0, stream, src8, dst8,
                                   r.total_blocks, r.first_block, remap_override >= 0 ? remap_override : 0, (int64_t)0, (int64_t)0);
                if (hipError_t e = hipGetLastError(); e != hipSuccess)
                    return e;
            }
            return element_range(tiles * T, r.num_blocks - tiles * T);
        }
    }
    const int aligned_flags = remap_override >= 0 ? remap_override : 0;
#else
    const int aligned_flags = 0;
#endif

    return plan_launches(fmt, inverse, sa, sc, soa, r, force_bits, threads, halo_threads_override, [&](const PlannedLaunch& l) {
        if (l.kind == 0) {
            hipLaunchKernelGGL(ks.tiled[threads_slot(l.threads)], dim3(l.grid), dim3(l.threads), 0, stream, src8, dst8, r.total_blocks,
                               r.first_block, aligned_flags, (int64_t)0, (int64_t)0);
            return hipGetLastError();
        }
        Shifts sh = l.sh;
#ifdef DXTLT_EXPERIMENTS
        sh.xcd_remap = remap_override >= 0 ? remap_override : l.kind == 1 ? 0 : 1;
        sh.skip_partial = (force_bits & 0x10) ? 1 : 0;
        sh.line_policy = l.kind == 1 ? ((force_bits & 0x800) ? 1 : 3) : 1;
#endif
        if (l.kind == 2) {   // the AoS side is the destination
            hipLaunchKernelGGL(ks.shifted, dim3(l.grid), dim3(l.threads), 0, stream, src8, dst8 + l.aos_offset, r.total_blocks,
                               r.first_block, sh);
        } else {
            const ShiftFn k = (halo_alt != nullptr && l.threads == halo_threads_override) ? halo_alt : ks.halo[sh.natural ? 1 : 0];
            hipLaunchKernelGGL(k, dim3(l.grid), dim3(l.threads), 0, stream, src8 + l.aos_offset, dst8, r.total_blocks, r.first_block, sh);
        }
        return hipGetLastError();
    });
}


// A regular array of buffers whose stream bases all sit on 128-byte lines and whose block count is a whole number of tiles IS
// the single-buffer aligned kernel with one more grid dimension: no table, no lookup (the batch kernel's lookup and entry
// decode are scalar instructions every wave of a workgroup executes on the one scalar unit a CU's four SIMDs share: 0.80
// against 0.835 of peak on one 1 GiB BC3 buffer, profiles/r03_batch_spacing.txt).  Returns hipErrorNotSupported when the
// array does not fit that shape (the caller then takes the batch kernel).
hipError_t launch_tiled_array(Format fmt, bool inverse, const Settings& s, const void* first_src, void* first_dst,
                              uint64_t blocks, uint32_t n_buffers, int64_t src_stride, int64_t dst_stride, hipStream_t stream)
{
    const int threads = default_tile_threads(fmt, inverse);
    const uint64_t T = (uint64_t)tile_blocks(fmt, threads);
    // (HIP refuses a launch of 2^32 threads or more)
    if (n_buffers == 0 || n_buffers > 65535 || blocks == 0 || blocks % T != 0 || blocks / T > 0x7FFFFFFFull ||
        (blocks / T) * n_buffers * (uint64_t)threads >= (1ull << 32))
        return hipErrorNotSupported;
    const bool sa = fmt == kBc3 && s.split_alpha, sc = s.split_colour;
    KernelSet ks;
    switch (fmt) {
    case kBc1: ks = pick_variant<kBc1>(s.variant, false, sc, inverse); break;
    case kBc2: ks = pick_variant<kBc2>(s.variant, false, sc, inverse); break;
    case kBc3: ks = pick_variant<kBc3>(s.variant, sa, sc, inverse); break;
    default: return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(ks.tiled[threads_slot(threads)], dim3((unsigned)(blocks / T), n_buffers), dim3(threads), 0, stream,
                       static_cast<const uint8_t*>(first_src), static_cast<uint8_t*>(first_dst), blocks, (uint64_t)0, kTiledArray,
                       src_stride, dst_stride);
    return hipGetLastError();
}

#ifdef DXTLT_WG_TIMING
extern "C" int dxtlt_debug_read_wg_marks_single(uint32_t* out, size_t count)   // experiment build: bcn_device.h, WG_MARK
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_marks), count * 4);
}
#endif

// Table upload by a kernel: the lanes read the pinned host table over PCIe and write the device twin.  A copy engine
// transfer in front of the batch kernel costs the stream two queue hand-overs (20-90 us measured per call, more than a
// quarter of a 1 GiB batch's kernel time); kernel after kernel on one queue is a barrier bit.
__global__ void __launch_bounds__(256)
table_upload_kernel(const u32x4* __restrict__ host, u32x4* __restrict__ dev, uint32_t vectors)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < vectors; i += gridDim.x * 256)
        dev[i] = host[i];
}

hipError_t launch_table_upload(const void* host_mapped, void* dev, size_t bytes, hipStream_t stream)
{
    if (bytes == 0)
        return hipSuccess;
    if ((bytes & 15) != 0 || bytes > (size_t(1) << 31))
        return hipErrorInvalidValue;
    const uint32_t vectors = (uint32_t)(bytes / 16);
    const uint32_t grid = std::min<uint32_t>((vectors + 255) / 256, 64);
    hipLaunchKernelGGL(table_upload_kernel, dim3(grid), dim3(256), 0, stream, static_cast<const u32x4*>(host_mapped),
                       static_cast<u32x4*>(dev), vectors);
    return hipGetLastError();
}

hipError_t launch_fill_splitmix64(void* dst, size_t len_bytes, uint64_t seed, uint64_t first_qword,
                                  hipStream_t stream)
{
    if (len_bytes == 0)
        return hipSuccess;
    uint64_t qwords = len_bytes / 8 + 1;
    uint64_t grid = (qwords + kThreads - 1) / kThreads;
    const uint64_t cap = (uint64_t)cached_cu_count() * 16;
    if (grid > cap)
        grid = cap;
    hipLaunchKernelGGL(fill_splitmix64_kernel, dim3((unsigned)grid), dim3(kThreads), 0, stream,
                       static_cast<uint8_t*>(dst), (uint64_t)len_bytes, seed, first_qword);
    return hipGetLastError();
}

}  // namespace dxtlt
