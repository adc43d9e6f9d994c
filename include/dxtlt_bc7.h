/*
 * dxtlt_bc7.h -- BC7 granule-sorted field split, version 2: C ABI (libdxtlt_gfx950.so).
 *
 * A FORMAT DEFINED BY THIS BUILD (docs/BC7_FORMAT.md).  The reference has no BC7 transform to be a drop-in for:
 * /root/reference/src/core/dxt-lossless-transform-bc7/src/lib.rs:1-13 holds two dead-code bit helpers, the BC7 API
 * crate is one line and `TransformBundle` has a placeholder for it.  What it does fix is
 * used: the bit fields of the eight modes (src/assets/research/dds-bc7-blocks.hexpat:286-654).  The entry points follow
 * the shape of the BC1-3 ones so that a future `transform_bc7_with_settings` could bind here; the format has no settings.
 * Parity: exact round trip and GPU == oracle/dxtlt_oracle_bc7.c only.
 *
 * Contract: len is a multiple of 16; output length == input length; buffers must not overlap; returns DXTLT_* status
 * codes of dxtlt_gfx950.h.  Device-pointer calls take any pointer alignment (16-byte aligned buffers are the fast case);
 * they enqueue one kernel (two when the
 * block count is not a multiple of 1024) on the stream, use no scratch memory and do not synchronise, so they can be
 * captured into a HIP graph.
 */
#ifndef DXTLT_BC7_H
#define DXTLT_BC7_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int32_t dxtlt_transform_bc7(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len);
int32_t dxtlt_untransform_bc7(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len);

/* Since version 1 the transform needs no device scratch: dxtlt_bc7_workspace_bytes returns 0 and the workspace arguments are ignored
 * (NULL / 0 are fine); both are kept so that callers written against version 0 keep compiling. */
size_t dxtlt_bc7_workspace_bytes(size_t len);
int32_t dxtlt_transform_bc7_device(const void *d_input, void *d_output, size_t len, void *d_workspace,
                                   size_t workspace_bytes, void *hip_stream);
int32_t dxtlt_untransform_bc7_device(const void *d_input, void *d_output, size_t len, void *d_workspace,
                                     size_t workspace_bytes, void *hip_stream);

/* One block range of an array of total_blocks blocks (multi-GPU shards, chunked staging), as
 * dxtlt_transform_range_device: the AoS-side pointer is the range's first block, the SoA-side pointer byte 0 of the
 * WHOLE transformed buffer.  first_block must be a multiple of dxtlt_bc7_sort_granule() and the range must end on one
 * or at the end of the array.  Ranges are independent of one another: no counters to exchange. */
int32_t dxtlt_transform_bc7_range_device(bool inverse, const void *d_src, void *d_dst, uint64_t total_blocks,
                                         uint64_t first_block, uint64_t num_blocks, void *hip_stream);
uint32_t dxtlt_bc7_sort_granule(void); /* 1024 */

/* ---- single-process multi-GPU: contiguous granule-aligned block ranges over the node's GPUs, no collective ------
 * num_shards <= 0: one shard per visible device; more shards than devices are spread round robin (that is how a
 * single-GPU machine exercises the placement).  At most 64 shards.  Same result as the unsharded call. */
int32_t dxtlt_transform_bc7_sharded(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len, int32_t num_shards);
int32_t dxtlt_untransform_bc7_sharded(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len, int32_t num_shards);

/* The placement of one shard on its own (host code, no device): the shard [first_block, first_block + num_blocks),
 * transformed as a stand-alone buffer, consists of nine pieces -- p = 0..7: its slice of main stream p (Q8, Q2, B0..B4,
 * F), p = 8: the tail part (only the shard that reaches the end of the array has one).  global_off[p] = byte offset in
 * the whole transformed buffer, local_off[p] = byte offset in the shard's own transformed buffer, bytes[p] = length.
 * Arrays of 9. */
int32_t dxtlt_bc7_shard_pieces(uint64_t total_blocks, uint64_t first_block, uint64_t num_blocks, uint64_t *global_off,
                               uint64_t *local_off, uint64_t *bytes);

#ifdef __cplusplus
}
#endif
#endif /* DXTLT_BC7_H */
