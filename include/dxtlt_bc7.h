/*
 * dxtlt_bc7.h -- BC7 mode-split transform, version 0: C ABI (libdxtlt_gfx950.so).
 *
 * A FORMAT DEFINED BY THIS BUILD (docs/BC7_FORMAT.md).  The reference has no BC7 transform to be a drop-in for:
 * /root/reference/src/core/dxt-lossless-transform-bc7/src/lib.rs:1-13 holds two dead-code bit helpers, the BC7 API
 * crate is one line and `TransformBundle` has a placeholder for it.  The entry points
 * follow the shape of the BC1-3 ones so that a future `transform_bc7_with_settings` could bind here; version 0 has no
 * settings.  Parity: exact round trip and GPU == oracle/dxtlt_oracle_bc7.c only.
 *
 * Contract: len is a multiple of 16; output length == input length; buffers must not overlap; returns DXTLT_* status
 * codes of dxtlt_gfx950.h.  Device-pointer calls need 16-byte aligned buffers and a scratch buffer of
 * dxtlt_bc7_workspace_bytes(len) bytes; they enqueue 3 kernels (inputs up to 16 MiB) or 5 on the stream and do not synchronise.
 */
#ifndef DXTLT_BC7_H
#define DXTLT_BC7_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int32_t dxtlt_transform_bc7(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len);
int32_t dxtlt_untransform_bc7(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len);

size_t dxtlt_bc7_workspace_bytes(size_t len);
int32_t dxtlt_transform_bc7_device(const void *d_input, void *d_output, size_t len, void *d_workspace,
                                   size_t workspace_bytes, void *hip_stream);
int32_t dxtlt_untransform_bc7_device(const void *d_input, void *d_output, size_t len, void *d_workspace,
                                     size_t workspace_bytes, void *hip_stream);

/* ---- single-process multi-GPU: contiguous block ranges over the node's GPUs, no collective -----
 * Output placement depends on the data of earlier shards, so every shard first reports its nine per-mode block counts;
 * the host turns them into the placement table below; then each shard's 19 stream pieces are copied to / from their
 * final places.  num_shards <= 0: one shard per visible device; more shards than devices are spread round robin (that
 * is how a single-GPU machine exercises the placement).  At most 64 shards.  Same result as the unsharded call. */
int32_t dxtlt_transform_bc7_sharded(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len, int32_t num_shards);
int32_t dxtlt_untransform_bc7_sharded(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len, int32_t num_shards);

/* The placement table on its own (host code, no device).  counts[s * 9 + m] = blocks of mode class m (0..7, 8 = the
 * reserved byte-0 == 0 encoding) in shard s; the shards are the contiguous ranges [shard_first_block[s],
 * + shard_num_blocks[s]) covering [0, total_blocks).  For piece p of shard `shard` (p = 0: `first`, 1 + m: head_m,
 * 10 + m: tail_m): global_off[p] = byte offset in the whole transformed buffer, local_off[p] = byte offset in the
 * shard's own transformed buffer, bytes[p] = length.  Arrays of 19. */
int32_t dxtlt_bc7_shard_pieces(const uint64_t *counts, int32_t num_shards, int32_t shard,
                               const uint64_t *shard_first_block, const uint64_t *shard_num_blocks,
                               uint64_t total_blocks, uint64_t *global_off, uint64_t *local_off, uint64_t *bytes);

#ifdef __cplusplus
}
#endif
#endif /* DXTLT_BC7_H */
