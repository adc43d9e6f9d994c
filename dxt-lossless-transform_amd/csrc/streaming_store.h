// streaming_store.h -- the 16-byte write-through streaming store of the write-once streams.
//
// Cache policy measured on the 8 GiB BC1 forward kernel (tools/kernel_lab.hip, profiles/r01_p_kernel_lab_cache_policies.txt),
// loads `nt` in every row:
//   store plain 0.805 | nt 0.827 | sc1 0.838 | sc0 sc1 0.829 | sc1 nt 0.841 | sc0 sc1 nt 0.842   (fraction of 8 TB/s)
// `sc1` makes the store write-through and drops the line from the XCD's L2 (MI355X_MICROARCH.md, store flavours), which is
// what a write-once stream wants; `nt` on top marks it streaming.  There is no builtin for that combination, so the
// instruction is spelled out; it has no result, and the compiler still waits for the operands it produced.
//
// The trailing `s_nop 1` is not optional: a 16-byte store reads its data registers a few cycles after issue, hipcc pads
// that hazard for its own stores only, and without the two wait states the instruction after the asm may overwrite
// them -- seen as stale dwords in lanes 12..15 of every 16 once a store sat inside an unrolled loop
// (cdna_hip_programming.md, inline-asm rules: "an asm ..._store_dwordx3/x4 ends with s_nop 1 inside the string").
//
// Use it only for 128-byte lines that one wave instruction writes completely; a line that another wave or workgroup
// completes must stay in L2 until then (plain or `nt` store), see fwd_shift_tile in bcn_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dxtlt {

typedef uint32_t streaming_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_streaming16(void* p, streaming_u32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

}  // namespace dxtlt
