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
//   * HBM-bound byte shuffling: 2 bytes of traffic per byte of input, ~20 integer ops per colour pair.
//     No MFMA.  What matters is that every HBM access is a full-width, fully coalesced 16 B/lane vector
//     access and that the memory system always has many independent workgroups to pull from.
//   * One workgroup owns ONE contiguous tile of blocks: THREADS x 16 bytes (256 threads: 4 KiB = 512 BC1 or
//     256 BC2/BC3 blocks).  Forward: one global_load_dwordx4 per lane (1 KiB contiguous per wave
//     instruction) -> YCoCg-R in registers -> each field is written to its place in an LDS image that is
//     laid out exactly like the output (stream after stream) -> barrier -> the image is read back linearly
//     with ds_read_b128 and leaves with one global_store_dwordx4 per lane; every stream slice of the tile
//     is a contiguous run of >= 256 bytes.  Inverse: the mirror (linear 16-B stream loads -> LDS image ->
//     per-block gather + inverse YCoCg-R -> AoS dwordx4 store).
//   * One tile per workgroup, grid = number of tiles (2^21 workgroups for 8 GiB).  Measured on MI355X
//     (profiles/r01_b_*, r01_c_*): a persistent grid-stride loop with 4 vectors per lane and software
//     prefetch reached 0.58-0.60 of the 8 TB/s peak, the same structure as one small tile per workgroup
//     0.79-0.80, against 0.81-0.83 for a plain dwordx4 copy launched the same way.  The hardware
//     dispatcher is the better scheduler for a pure stream: short-lived workgroups de-synchronise reads
//     and writes and keep every channel busy.
//   * Non-temporal loads (every byte is touched exactly once): +2-3 % over default policy.  Stores of aligned tiles
//     are write-through streaming stores (`sc1 nt`): another +1-3 % (profiles/r01_p, r01_q).  Shifted tiles share
//     128-B lines with their neighbours and keep plain `nt` stores, so L2 can merge the two halves of a line.
//   * 64-bit block indices and byte offsets everywhere (8 GiB of BC1 = 2^30 blocks).
//   * Stream bases that are not 16-byte aligned (odd block counts, ranges starting at odd blocks) take the
//     "shifted tile" kernels further down: same structure, each stream's LDS slice displaced by its
//     misalignment so the body still moves as aligned 16-byte vectors (0.72-0.79 of peak).
//   * What no tile path can take (the < 1 tile tail, an AoS pointer that is itself misaligned) goes to an
//     element-granular kernel: one lane per block, natural-width or byte accesses.
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
using GenericFn = void (*)(const uint8_t*, uint8_t*, uint64_t, uint64_t, uint64_t, uint64_t);
using ShiftFn = void (*)(const uint8_t*, uint8_t*, uint64_t, uint64_t, Shifts);

struct KernelSet {
    TiledFn tiled[4];  // 64, 128, 256, 512 threads
    ShiftFn shifted;   // 256 threads, misaligned stream bases
    GenericFn generic;
    ShiftFn halo[2];   // forward only: shifted tiles with a halo, whole segments only ([1]: natural shifts); nullptr for the inverse
};

inline int threads_slot(int threads) { return threads == 64 ? 0 : threads == 128 ? 1 : threads == 512 ? 3 : 2; }

template <int FMT, int VARIANT, bool SA, bool SC>
KernelSet kernels_for(bool inverse)
{
    if (inverse)
        return {{inv_tiled<FMT, VARIANT, SA, SC, 64>, inv_tiled<FMT, VARIANT, SA, SC, 128>,
                 inv_tiled<FMT, VARIANT, SA, SC, 256>, inv_tiled<FMT, VARIANT, SA, SC, 512>},
                inv_tiled_shift<FMT, VARIANT, SA, SC>, generic_kernel<FMT, VARIANT, SA, SC, true>, {nullptr, nullptr}};
    return {{fwd_tiled<FMT, VARIANT, SA, SC, 64>, fwd_tiled<FMT, VARIANT, SA, SC, 128>,
             fwd_tiled<FMT, VARIANT, SA, SC, 256>, fwd_tiled<FMT, VARIANT, SA, SC, 512>},
            fwd_tiled_shift<FMT, VARIANT, SA, SC>, generic_kernel<FMT, VARIANT, SA, SC, false>,
            {fwd_tiled_halo<FMT, VARIANT, SA, SC, kNormNone, false>, fwd_tiled_halo<FMT, VARIANT, SA, SC, kNormNone, true>}};
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

// BC1 forward with block normalisation fused in: one tile size (the BC1 default), shifted tiles, element kernel
template <int VARIANT, bool SC, int NORM>
KernelSet bc1_norm_kernels()
{
    constexpr int TH = default_tile_threads(kBc1, false);
    TiledFn tiled = fwd_tiled<kBc1, VARIANT, false, SC, TH, NORM>;
    return {{tiled, tiled, tiled, tiled}, fwd_tiled_shift<kBc1, VARIANT, false, SC, NORM>,
            generic_kernel<kBc1, VARIANT, false, SC, false, NORM>,
            {fwd_tiled_halo<kBc1, VARIANT, false, SC, NORM, false>, fwd_tiled_halo<kBc1, VARIANT, false, SC, NORM, true>}};
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

static inline int force_bits_early(const LaunchTuning* tuning) { return tuning ? tuning->force_generic : 0; }

hipError_t launch_transform(Format fmt, bool inverse, const Settings& s, const void* src, void* dst,
                            const Range& r, hipStream_t stream, const LaunchTuning* tuning)
{
    if (r.num_blocks == 0)
        return hipSuccess;
    if (s.variant < 0 || s.variant > 3 || r.first_block + r.num_blocks > r.total_blocks)
        return hipErrorInvalidValue;
    if (s.normalize != kNormNone && (fmt != kBc1 || inverse || s.normalize < 0 || s.normalize > kNormTransparentOnly))
        return hipErrorInvalidValue;  // normalisation exists for the BC1 forward transform only

    // HIP refuses a launch of 2^32 or more threads (grid x workgroup), which a 64 GiB buffer reaches at 16 bytes per
    // lane.  Larger ranges go out as consecutive sub-ranges of 2^31 blocks (a multiple of every tile size, so stream
    // alignment and tile boundaries are the same as in one launch): the AoS side advances, the SoA side is addressed
    // through first_block as always.
    constexpr uint64_t kMaxBlocksPerLaunch = 1ull << 31;
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
    const void* aos = inverse ? (const void*)dst : src;
    const void* soa = inverse ? src : (const void*)dst;

    // Which path can take the range?
    //   aligned tiles: both pointers and every stream base 16-byte aligned
    //   shifted tiles: AoS pointer 16-byte aligned, stream bases anywhere
    //   element kernel: everything else, and the tail that does not fill a tile
    const Streams S = make_streams(fmt, sa, sc);
    // The AoS side may sit at any byte address: 16-byte vector loads / stores at unaligned addresses are exact on gfx950
    // under ROCm's default memory mode and cost little (tools/unaligned_lab.hip: a streaming copy at 0.845 of peak drops
    // to 0.81-0.83 with misaligned loads, 0.78-0.79 with misaligned stores) -- far less than the element kernel, which
    // round 1 sent such buffers to (0.4-0.7).  A DDS file that lies whole in HBM has its payload at byte 128 or 148.
    // Experiment switch 0x1000 restores the round-1 rule (tiles only for a 16-byte aligned AoS pointer).
    const bool aos_ok = (reinterpret_cast<uintptr_t>(aos) & 15) == 0 || !(force_bits_early(tuning) & 0x1000);
    Shifts sh{};
    bool any_shift = false;
    for (int i = 0; i < S.n; ++i) {
        const uint64_t base = reinterpret_cast<uintptr_t>(soa) + (uint64_t)S.off[i] * r.total_blocks +
                              (uint64_t)S.width[i] * r.first_block;
        sh.d[i] = (int)(base & 15);
        // The aligned tiles need 16-byte aligned stream bases to be correct and 128-byte aligned ones to be fast: with
        // bases that are only 16-byte aligned every slice shares its first and last line with the neighbouring tiles,
        // which the shifted tiles handle (XCD-contiguous tile order, no write-through on shared lines) and the aligned
        // ones do not -- BC3 forward 0.65 against 0.77 of peak (tools/shift_probe.py, profiles/r01_z/shift_probe.txt).
        any_shift = any_shift || (base & 127) != 0;
    }
    fill_gbase(sh, S, r.total_blocks, r.first_block);
    sh.natural = (force_bits_early(tuning) & 0x20) ? 0 : shifts_are_natural(S, sh.d);   // experiment switch 0x20: generic LDS accesses
    // XCD-contiguous tile order: measured +2..+9 % on shifted tiles (neighbouring tiles share 128-byte lines), -1..-3 %
    // on aligned tiles (profiles/r01_i_*) -- so on for the former, off for the latter unless an experiment says so
    const int remap_override = tuning ? tuning->xcd_remap : -1;
    sh.xcd_remap = remap_override >= 0 ? remap_override : 1;
    const int aligned_remap = remap_override >= 0 ? remap_override : 0;
    const int force_bits = tuning ? tuning->force_generic : 0;
    const int force = force_bits & 3;  // 1 = element kernel, 2 = shifted tiles
    sh.skip_partial = (force_bits & 0x10) ? 1 : 0;
    sh.line_policy = (force_bits & 0x40) ? 0 : (force_bits & 0x80) ? 2 : 1;  // 0x80: shared lines with plain (temporal) stores  // experiment switch: 0x40 = plain nt stores on every shifted line
    const bool use_tiles = aos_ok && force != 1;
    const bool use_shift = use_tiles && (any_shift || force == 2);

    int threads = default_tile_threads(fmt, inverse);
    if (tuning && (tuning->tile_threads == 64 || tuning->tile_threads == 128 || tuning->tile_threads == 256 ||
                   tuning->tile_threads == 512))
        threads = tuning->tile_threads;
    if (normalizing)
        threads = default_tile_threads(fmt, inverse);  // the only tile size instantiated with normalisation
    if (use_shift)
        threads = 256;
    const uint64_t T = (uint64_t)tile_blocks(fmt, threads);
    const uint64_t num_tiles = use_tiles ? r.num_blocks / T : 0;
    // forward shifted tiles take the halo form (whole segments only) unless experiment switch 0x400 asks for the first form
    const bool use_halo = use_shift && !inverse && ks.halo[0] != nullptr && !(force_bits & 0x400);
    Shifts shh = sh;   // the halo tiles' own shifts: stream base mod 64
    if (use_halo) {
        int halo_blocks = 0;
        for (int i = 0; i < S.n; ++i) {
            const uint64_t base = reinterpret_cast<uintptr_t>(soa) + (uint64_t)S.off[i] * r.total_blocks +
                                  (uint64_t)S.width[i] * r.first_block;
            shh.d[i] = (int)(base & (uint64_t)(kHaloAlign - 1));
            halo_blocks = std::max(halo_blocks, (shh.d[i] + S.width[i] - 1) / S.width[i]);
        }
        fill_gbase(shh, S, r.total_blocks, r.first_block);
        shh.natural = (force_bits & 0x20) ? 0 : shifts_are_natural(S, shh.d);
        // every window starts on a 64-byte sector: no line needs to meet its other half in one L2 -- identity tile order
        // and, unless experiment switch 0x800 asks for plain nt, the write-through streaming stores of the aligned tiles
        shh.line_policy = (force_bits & 0x800) ? 1 : 3;
        const int per_vec = 16 / fmt_block(fmt);
        shh.halo_vecs = (halo_blocks + per_vec - 1) / per_vec;
        // The halo is read again by the next tile, which runs on another XCD in the identity order: the kernel's temporal
        // loads keep those lines in the memory-side cache for it, so the identity order (0.035 faster by itself than the
        // XCD-contiguous one) is right for every halo size.
        shh.xcd_remap = remap_override >= 0 ? remap_override : 0;
    }
    auto element_range = [&](uint64_t local_first, uint64_t count) -> hipError_t {
        if (count == 0)
            return hipSuccess;
        const uint64_t grid = (count + kThreads - 1) / kThreads;   // count <= 2^31 blocks, one per thread
        hipLaunchKernelGGL(ks.generic, dim3((unsigned)grid), dim3(kThreads), 0, stream, src8, dst8, r.total_blocks,
                           r.first_block, local_first, count);
        return hipGetLastError();
    };
    // Shifts of the sub-range [local_first, local_first + count) of the range as ONE edge tile (count < T): what is left
    // behind the aligned tiles, or a range smaller than a tile.  Forward: a halo tile 0 (no halo, writes every stream from its
    // first byte to its last); inverse: a shifted tile 0.
    auto edge_tile_of = [&](uint64_t local_first, uint64_t count) -> hipError_t {
        if (count == 0)
            return hipSuccess;
        Shifts e{};
        const int mask = inverse ? 15 : kHaloAlign - 1;
        for (int i = 0; i < S.n; ++i) {
            const uint64_t base = reinterpret_cast<uintptr_t>(soa) + (uint64_t)S.off[i] * r.total_blocks +
                                  (uint64_t)S.width[i] * (r.first_block + local_first);
            e.d[i] = (int)(base & (uint64_t)mask);
        }
        fill_gbase(e, S, r.total_blocks, r.first_block + local_first);
        e.natural = (force_bits & 0x20) ? 0 : shifts_are_natural(S, e.d);
        e.line_policy = 1;
        e.full_tiles = 0;
        e.range_blocks = count;
        const uint64_t aos_off = local_first * (uint64_t)fmt_block(fmt);
        if (inverse)
            hipLaunchKernelGGL(ks.shifted, dim3(1), dim3(256), 0, stream, src8, dst8 + aos_off, r.total_blocks, r.first_block, e);
        else
            hipLaunchKernelGGL(ks.halo[e.natural ? 1 : 0], dim3(1), dim3(256), 0, stream, src8 + aos_off, dst8, r.total_blocks,
                               r.first_block, e);
        return hipGetLastError();
    };
    const uint64_t rest = r.num_blocks - num_tiles * T;
    // experiment switch 0x2000: the round-3 routing (heads and tails through the element kernel)
    const bool edge_tiles = use_tiles && !(force_bits & 0x2000) && (inverse || ks.halo[0] != nullptr);
    if (use_halo && edge_tiles) {
        // halo tiles + edge tiles in ONE launch: tile 0 writes the head of every stream itself, and one more workgroup takes the
        // blocks behind the last whole tile and the last d_s bytes of every stream
        bool tail = rest > 0;
        for (int i = 0; i < S.n; ++i)
            tail = tail || shh.d[i] > 0;
        shh.full_tiles = (uint32_t)num_tiles;
        shh.range_blocks = r.num_blocks;
        hipLaunchKernelGGL(ks.halo[shh.natural ? 1 : 0], dim3((unsigned)(num_tiles + (tail ? 1 : 0))), dim3(256), 0, stream, src8, dst8,
                           r.total_blocks, r.first_block, shh);
        return hipGetLastError();
    }
    if (use_shift && inverse && edge_tiles) {
        sh.full_tiles = (uint32_t)num_tiles;
        sh.range_blocks = r.num_blocks;
        hipLaunchKernelGGL(ks.shifted, dim3((unsigned)(num_tiles + (rest > 0 ? 1 : 0))), dim3(256), 0, stream, src8, dst8,
                           r.total_blocks, r.first_block, sh);
        return hipGetLastError();
    }
    if (num_tiles > 0) {
        if (use_halo) {   // (switch 0x2000)
            shh.full_tiles = (uint32_t)num_tiles;
            shh.range_blocks = num_tiles * T;
            shh.skip_partial = 0;
            hipLaunchKernelGGL(ks.halo[shh.natural ? 1 : 0], dim3((unsigned)num_tiles), dim3(256), 0, stream, src8, dst8,
                               r.total_blocks, r.first_block, shh);
        } else if (use_shift) {
            sh.full_tiles = (uint32_t)num_tiles;
            sh.range_blocks = num_tiles * T;
            hipLaunchKernelGGL(ks.shifted, dim3((unsigned)num_tiles), dim3(256), 0, stream, src8, dst8, r.total_blocks,
                               r.first_block, sh);
        } else {
            hipLaunchKernelGGL(ks.tiled[threads_slot(threads)], dim3((unsigned)num_tiles), dim3(threads), 0, stream,
                               src8, dst8, r.total_blocks, r.first_block, aligned_remap, (int64_t)0, (int64_t)0);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return e;
    }
    uint64_t done = num_tiles * T;
    if (use_halo && num_tiles > 0) {
        // (switch 0x2000) what the halo tiles' windows leave out: the head of every stream of the range (records of its first 64
        // blocks) and everything behind the last window (records of the last 64 blocks of the tiles, and the rest)
        if (hipError_t e = element_range(0, kHaloBlocks); e != hipSuccess)
            return e;
        done -= kHaloBlocks;
        return element_range(done, r.num_blocks - done);
    }
    // behind aligned tiles (or a range smaller than a tile): one edge tile -- 256 threads' worth of blocks at most
    if (edge_tiles && rest <= (uint64_t)tile_blocks(fmt, 256))
        return edge_tile_of(done, rest);
    if (edge_tiles) {   // BC1 with 512-thread tiles (an experiment size): up to 1023 blocks left
        const uint64_t T256 = (uint64_t)tile_blocks(fmt, 256);
        for (uint64_t at = done; at < r.num_blocks; at += T256)
            if (hipError_t e = edge_tile_of(at, std::min(T256, r.num_blocks - at)); e != hipSuccess)
                return e;
        return hipSuccess;
    }
    return element_range(done, r.num_blocks - done);
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
