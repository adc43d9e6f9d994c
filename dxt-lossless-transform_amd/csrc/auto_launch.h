// auto_launch.h -- internal interface of the fused candidate kernel of transform_bcN_auto (auto_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "bcn_launch.h"

namespace dxtlt {

// Bytes of the candidate arena for `blocks` blocks: BC3's two alpha-endpoint sections (2 bytes per block each), then per
// YCoCg-R variant (None, Variant1; with all_variants also Variant2, Variant3) the colour section as pairs and split
// (4 bytes per block each).
uint64_t auto_arena_bytes(Format fmt, bool all_variants, uint64_t blocks);
// byte offset of a section inside the arena
uint64_t auto_section_offset(Format fmt, uint64_t blocks, int variant, bool split_colour);
uint64_t auto_alpha_section_offset(uint64_t blocks, bool split_alpha);   // BC3
// One read of d_in (the AoS blocks, 16-byte aligned) -> every section.  Enqueues on `stream`.
hipError_t launch_auto_candidates(Format fmt, bool all_variants, const void* d_in, void* d_arena, uint64_t blocks,
                                  hipStream_t stream);

}  // namespace dxtlt
