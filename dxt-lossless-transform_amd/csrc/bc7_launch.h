// bc7_launch.h -- internal launch interface of the BC7 mode-split transform (docs/BC7_FORMAT.md).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace dxtlt {
namespace bc7 {

// Device scratch needed for `n_blocks` blocks (tile histograms, their prefix sums, stream bases).
size_t workspace_bytes(uint64_t n_blocks);

// Enqueue the whole pipeline on `stream`.  src/dst/workspace must be 16-byte aligned device pointers.
// Afterwards the first nine uint64_t of the workspace hold the number of blocks of every mode class (0..8).
hipError_t launch(bool inverse, const void* src, void* dst, uint64_t n_blocks, void* workspace, size_t ws_bytes,
                  hipStream_t stream);

// Only the counting part: `first` is the first stream of a transformed buffer (byte 0 of n_blocks blocks); the nine
// per-mode totals land at the start of the workspace.
hipError_t launch_counts(const void* first, uint64_t n_blocks, void* workspace, size_t ws_bytes, hipStream_t stream);

}  // namespace bc7
}  // namespace dxtlt
