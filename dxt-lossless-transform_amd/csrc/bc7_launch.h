// bc7_launch.h -- internal launch interface of the BC7 granule-sorted field split, version 2 (docs/BC7_FORMAT.md).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace dxtlt {
namespace bc7 {

// Whole buffer.  Forward: src = blocks, dst = transformed; inverse: the other way round.  16-byte aligned device
// pointers (hipErrorInvalidValue otherwise).  One or two kernels on `stream`, no workspace, no synchronisation.
hipError_t launch(bool inverse, const void* src, void* dst, uint64_t n_blocks, hipStream_t stream);

// One block range of an array of `total_blocks` blocks.  The AoS-side pointer is the range's first block, the SoA-side
// pointer byte 0 of the WHOLE transformed buffer.  first_block must be a multiple of the sort granule (1024) and the
// range must end on one or at the end of the array.
hipError_t launch_range(bool inverse, const void* src, void* dst, uint64_t total_blocks, uint64_t first_block,
                        uint64_t num_blocks, hipStream_t stream);

// Many buffers per launch.  One entry per buffer with full granules (src / dst: the buffer's first byte on both sides;
// first_wg: its first workgroup in the granule launch, entries in ascending order) and one per buffer with a tail part
// (src / dst: the tail part's first byte on both sides).  `d_coarse[k]` = the entry that owns workgroup 64 k.
struct BatchEntry {
    const uint8_t* src;
    uint8_t* dst;
    uint64_t main_blocks;   // blocks of the main part (a multiple of 1024)
    uint32_t first_wg;
    uint32_t tail;          // blocks of the tail part (tail entries)
};
hipError_t launch_batch(bool inverse, const BatchEntry* d_entries, const uint32_t* d_coarse, uint32_t n_entries,
                        uint32_t granule_wgs, const BatchEntry* d_tails, uint32_t n_tails, hipStream_t stream);

}  // namespace bc7
}  // namespace dxtlt
