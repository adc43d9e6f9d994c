// bcn_launch.h -- internal launch interface between the C ABI layer and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include <cstdlib>

namespace dxtlt {

// A/B switches read from the environment exist in the experiments side build only (-DDXTLT_EXPERIMENTS, bcn_device.h): in the
// shipped library the variable is never looked at.
inline const char* experiment_env(const char* name)
{
#ifdef DXTLT_EXPERIMENTS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

enum Format : int { kBc1 = 1, kBc2 = 2, kBc3 = 3 };

struct Settings {
    int variant;        // core YCoCgVariant numbering: 0 None, 1..3
    bool split_alpha;   // BC3 only
    bool split_colour;
    int normalize = 0;  // BC1 forward only: ColorNormalizationMode 0 None, 1 Color0Only, 2 ReplicateColor (fused)
};

// One contiguous range of blocks of a larger block array (the whole array when first_block == 0 and
// num_blocks == total_blocks).  `aos` points at the AoS data of the RANGE's first block; `soa` points
// at byte 0 of the WHOLE transformed buffer (stream bases are functions of total_blocks).
struct Range {
    uint64_t total_blocks;
    uint64_t first_block;
    uint64_t num_blocks;
};

struct LaunchTuning {
    int tile_threads;   // aligned tiles: 0 = per-format default; 64, 128, 256, 512
    int force_generic;  // force_path bits of dxtlt_set_tuning, see launch_transform (bcn_kernels.hip); 2 = halo / shifted tiles always
    int xcd_remap;      // experiments build only: -1 = default, 0 = off, 1 = XCD-contiguous tile order
};

// the force_path bits this build of the kernels honours (2 | 0x20 in the shipped library)
int launch_force_mask();

// What launch_transform would enqueue for a range, without enqueueing anything (addresses are numbers: nothing is dereferenced,
// no device is needed): the host logic of the single-buffer call for the CPU tests (dxtlt_debug_plan_transform).  Returns the
// number of launches (records beyond `cap` are counted, not written), -1 for arguments launch_transform refuses.
struct DebugPlannedLaunch {
    int32_t kind;           // 0: aligned tiles; 1: halo tiles + edge tiles (forward); 2: shifted tiles + edge tile (inverse)
    int32_t threads;        // lanes per workgroup
    uint32_t workgroups;
    uint32_t full_tiles;    // whole tiles among them (the others are edge tiles)
    uint64_t range_blocks;  // blocks the launch covers
    uint64_t aos_offset;    // bytes from the call's AoS pointer to the launch's first block
    uint8_t shift[6];       // kinds 1, 2: stream base modulo 64 (forward) / 16 (inverse)
    uint8_t halo_vecs;
    uint8_t natural;
    uint64_t gbase[6];
};
int debug_plan_transform(Format fmt, bool inverse, const Settings& s, uint64_t src_address, uint64_t dst_address, const Range& r,
                         const LaunchTuning* tuning, DebugPlannedLaunch* out, int cap);

// Forward: aos (input) -> soa (output).  Inverse: soa (input) -> aos (output).
// Enqueues on `stream`; returns the first HIP error.
hipError_t launch_transform(Format fmt, bool inverse, const Settings& s, const void* src, void* dst,
                            const Range& r, hipStream_t stream, const LaunchTuning* tuning = nullptr);

// Deterministic synthetic data: qword i = splitmix64(seed, first_qword + i) (matches oracle_fill_splitmix64).
hipError_t launch_fill_splitmix64(void* dst, size_t len_bytes, uint64_t seed, uint64_t first_qword,
                                  hipStream_t stream);

// BC1 block normalisation (bc1_normalize.hip; reference experimental/normalize_blocks/normalize.rs).  `mode` is the
// ColorNormalizationMode (0 None, 1 Color0Only, 2 ReplicateColor).  All enqueue on `stream`.
//   blocks:     in -> out, AoS blocks; in == out allowed (in place), partial overlap is not
//   split:      colours (4 bytes per block) and indices (4 bytes per block) in two separate arrays, in place
//   all_modes:  in -> out[0..2], one output per mode in enum order; *d_any (device uint32) is set to 1 when any block
//               was normalised (the caller zeroes it first)
//   any:        only the flag
hipError_t launch_normalize_bc1_blocks(const void* in, void* out, uint64_t num_blocks, int mode, hipStream_t stream);
hipError_t launch_normalize_bc1_split(void* colours, void* indices, uint64_t num_blocks, int mode, hipStream_t stream);
hipError_t launch_normalize_bc1_all_modes(const void* in, void* const out[3], uint64_t num_blocks, uint32_t* d_any,
                                          hipStream_t stream);
hipError_t launch_bc1_any_normalizable(const void* in, uint64_t num_blocks, uint32_t* d_any, hipStream_t stream);

// bcn_decode.hip: fmt = 1, 2, 3.  `out` = num_blocks * 64 bytes (sixteen r, g, b, a per block, row-major);
// `d_count` = one device uint64_t, zeroed by the call
hipError_t launch_decode_blocks(int fmt, const void* in, void* out, uint64_t num_blocks, hipStream_t stream);
hipError_t launch_count_pixel_differences(int fmt, const void* a, const void* b, uint64_t num_blocks, uint64_t* d_count,
                                          hipStream_t stream);

// Array-level colour operations of the reference's common crate (color565_ops.hip): YCoCg-R over `num_items` RGB565
// colours (in == out allowed), recorrelation with interleave of two half arrays, and the (c0, c1) endpoint split.
hipError_t launch_color565_ycocg(bool inverse, const void* in, void* out, uint64_t num_items, int variant, hipStream_t stream);
hipError_t launch_color565_recorrelate_split(const void* src0, const void* src1, void* dst, uint64_t num_items, int variant,
                                             hipStream_t stream);
hipError_t launch_split_565_color_endpoints(const void* in, void* out, uint64_t len_bytes, hipStream_t stream);

// BC2 / BC3 block normalisation (bc23_normalize.hip; reference bc2/bc3 experimental/normalize_blocks/normalize.rs).
// fmt 2 or 3; alpha_mode = AlphaNormalizationMode (BC3 only, 0 for BC2), color_mode = ColorNormalizationMode.
//   all_modes: outs[3] for BC2 (colour modes), outs[12] for BC3 ([alpha_mode * 3 + colour_mode])
//   split:     the reference's section layout -- BC2 colours / indices (4 + 4 bytes per block; alpha untouched),
//              BC3 alpha endpoints (2), alpha indices (6), colour endpoints (4), colour indices (4) -- in place
hipError_t launch_normalize_bc23_blocks(int fmt, const void* in, void* out, uint64_t num_blocks, int alpha_mode, int color_mode,
                                        hipStream_t stream);
hipError_t launch_normalize_bc23_all_modes(int fmt, const void* in, void* const* outs, uint64_t num_blocks, hipStream_t stream);
hipError_t launch_normalize_bc2_split(void* colours, void* indices, uint64_t num_blocks, int color_mode, hipStream_t stream);
hipError_t launch_normalize_bc3_split(void* alpha_endpoints, void* alpha_indices, void* color_endpoints, void* color_indices,
                                      uint64_t num_blocks, int alpha_mode, int color_mode, hipStream_t stream);

// ---- batch launch: many buffers of one format, direction and settings in one kernel (batch_kernels.hip) ------------
// One entry per buffer, in workgroup order.  Workgroups [first_wg, first_wg + full_tiles) run one whole 256-lane tile each, in
// the form `form` names (the one launch_transform would pick for the buffer); workgroup first_wg + full_tiles, when
// end_wg says it exists, is the buffer's edge tile: the blocks behind the last whole tile and, forward, the last bytes of every
// stream (bcn_device.h, "Edge tiles").  No element path, no padding between buffers.
struct BatchEntry {
    const uint8_t* src;
    uint8_t* dst;
    uint64_t blocks;
    uint32_t first_wg;
    uint32_t end_wg;        // first_wg + workgroups of this buffer = the next buffer's first_wg
    uint32_t full_tiles;
    uint8_t form;           // 1: every stream base on a 128-byte line (aligned tiles); 0: halo tiles forward, shifted tiles inverse
    uint8_t halo_vecs;      // halo tiles: 16-byte vectors of blocks in front of a tile that have bytes in its windows
    uint8_t natural;        // every shift a multiple of its stream's element width (the kernel handles nothing else: plan_batch_entry)
    uint8_t reserved;
    uint8_t shift[6];       // misalignment of every stream base: mod 64 forward, mod 16 inverse
    uint8_t reserved2[2];
    uint64_t gbase[6];      // Shifts::gbase: off_s * blocks - shift[s], from the transformed-side pointer
};
static_assert(sizeof(BatchEntry) == 96, "BatchEntry layout is shared between host and device");

// The workgroup -> entry index of a batch launch, in two levels so that the bytes a workgroup reads are shared with as many other
// workgroups of its CU as possible (what a lookup costs is its scalar-cache MISSES, not its round trips: batch_kernels.hip):
//   base[k]   (uint32) for workgroups [4096 k, 4096 k + 4096): the entry that owns workgroup 4096 k
//   delta[j]  for workgroups [64 j, 64 j + 64): entry that owns workgroup 64 j, minus base[j / 64] -- a byte each; 16 bits each
//             (the WIDE form) when more than 255 entries begin inside some 4096-workgroup span, i.e. for launches of thousands of
//             buffers of a few tiles, so that the delta never saturates
// base[wg / 4096] + delta[wg / 64] is the owner of workgroup 64 * (wg / 64); the owner of wg is that entry or one of the next
// wg % 64, found by bisection over the entries' end_wg.  One 64-byte line of byte deltas serves 4096 workgroups (sixteen per
// CU), one of `base` 65536.
constexpr uint32_t kBatchIndexWgs = 64, kBatchBaseWgs = 4096;
inline size_t batch_index_base_count(uint32_t total_wgs) { return ((size_t)total_wgs + kBatchBaseWgs - 1) / kBatchBaseWgs; }
inline size_t batch_index_delta_count(uint32_t total_wgs) { return ((size_t)total_wgs + kBatchIndexWgs - 1) / kBatchIndexWgs; }
// bytes of the index as build_batch_index lays it out -- base[] then delta[] -- with room for the wide form, padded to 16
inline size_t batch_index_bytes(uint32_t total_wgs)
{
    return (batch_index_base_count(total_wgs) * 4 + 2 * batch_index_delta_count(total_wgs) + 15) & ~(size_t)15;
}

// Fills first_wg-relative planning fields of `e` (src, dst, blocks set by the caller; first_wg too) for settings `s` and
// returns the number of workgroups the buffer needs (0 for an empty buffer), or 0xFFFFFFFF when the batch kernel cannot take the
// buffer (a transformed-side pointer whose stream shifts are not multiples of the element widths: the caller launches it alone).
uint32_t plan_batch_entry(Format fmt, bool inverse, const Settings& s, BatchEntry& e);

// Copies a table of `bytes` (a multiple of 16) from mapped pinned host memory (its device-side address) to device memory
// with a small kernel on `stream` -- no copy-engine hand-over in front of the batch kernel.
hipError_t launch_table_upload(const void* host_mapped, void* dev, size_t bytes, hipStream_t stream);

// d_entries / d_index: device copies of the entry table and of the workgroup index (build_batch_index: base[] then delta[];
// wide_index = its return value), total_wgs = end_wg of the last entry.
// uniform_wgs: 0, or the number of workgroups EVERY entry owns (first_wg == index * uniform_wgs): the kernel then finds a
// workgroup's entry by division instead of through the index (1: the entry IS the workgroup number).
// strided_first != nullptr (needs uniform_wgs != 0): the batch is a regular array -- every entry equals *strided_first but for
// its pointers, which advance by src_stride / dst_stride bytes per entry; the kernel then reads no table at all.
hipError_t launch_batch(Format fmt, bool inverse, const Settings& s, const BatchEntry* d_entries, const uint8_t* d_index,
                        uint32_t n_entries, uint32_t total_wgs, uint32_t uniform_wgs, bool wide_index, hipStream_t stream,
                        const BatchEntry* strided_first = nullptr, int64_t src_stride = 0, int64_t dst_stride = 0);

// the index (batch_index_bytes(total_wgs) bytes) from the entries (sorted by first_wg, each owning at least one workgroup, no
// gaps); returns true when it chose the wide (16-bit delta) form
bool build_batch_index(const BatchEntry* entries, size_t n_entries, uint32_t total_wgs, uint8_t* index);

// A regular array of aligned buffers as the single-buffer aligned kernel with blockIdx.y = buffer (bcn_kernels.hip);
// hipErrorNotSupported when the array does not have that shape.
hipError_t launch_tiled_array(Format fmt, bool inverse, const Settings& s, const void* first_src, void* first_dst,
                              uint64_t blocks, uint32_t n_buffers, int64_t src_stride, int64_t dst_stride, hipStream_t stream);

inline int block_bytes(Format f) { return f == kBc1 ? 8 : 16; }

// ------------------------------------------------------------------------------------------------
// Stream table.  A transformed buffer is a concatenation of streams; stream s holds `width` bytes per
// block and starts at byte `off * N` (N = total blocks).  `off` also equals the field's byte offset
// inside the AoS block, and the LDS image of a T-block tile uses the same table with N := T.
//   BC1  split: c0 2@0, c1 2@2, idx 4@4          no split: colours 4@0, idx 4@4
//   BC2  alpha 8@0, then colours at 8 (2+2 or 4), idx 4@12
//   BC3  alpha endpoints at 0 (1+1 or 2), alpha indices 6@2, colours at 8 (2+2 or 4), idx 4@12
// (reference: bc1 transform_with_settings.rs:43-58, bc2 :43-46, bc3 :54-56,76-80)
// ------------------------------------------------------------------------------------------------
struct Streams {
    int n;
    int width[6];
    int off[6];
};

constexpr Streams make_streams(int fmt, bool split_alpha, bool split_colour)
{
    Streams s{};
    int n = 0, off = 0;
    if (fmt == kBc3) {
        if (split_alpha) {
            s.width[n] = 1; s.off[n] = off; off += 1; ++n;
            s.width[n] = 1; s.off[n] = off; off += 1; ++n;
        } else {
            s.width[n] = 2; s.off[n] = off; off += 2; ++n;
        }
        s.width[n] = 6; s.off[n] = off; off += 6; ++n;
    }
    if (fmt == kBc2) {
        s.width[n] = 8; s.off[n] = off; off += 8; ++n;
    }
    if (split_colour) {
        s.width[n] = 2; s.off[n] = off; off += 2; ++n;
        s.width[n] = 2; s.off[n] = off; off += 2; ++n;
    } else {
        s.width[n] = 4; s.off[n] = off; off += 4; ++n;
    }
    s.width[n] = 4; s.off[n] = off; off += 4; ++n;
    s.n = n;
    return s;
}

}  // namespace dxtlt
