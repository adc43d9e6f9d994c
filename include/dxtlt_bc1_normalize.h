/*
 * dxtlt_bc1_normalize.h -- C ABI of the BC1 block-normalisation entry points of libdxtlt_gfx950.so: the MI355X
 * implementation of the reference's *experimental* module
 *   /root/reference/src/core/dxt-lossless-transform-bc1/src/experimental/normalize_blocks/
 * The reference exposes these as Rust fns only (no cbindgen export); this is what an `extern "C"` shim in that crate
 * binds.  Reference signatures replaced:
 *
 *   normalize_blocks                         normalize.rs:38    (in -> out, or in place)
 *   normalize_split_blocks_in_place          normalize.rs:286   (colours / indices already split)
 *   normalize_blocks_all_modes               normalize.rs:417   (one pass, one output per mode, returns "any changed")
 *   transform_bc1_with_normalize_blocks      transform.rs:65    (normalise + transform_bc1_with_settings, fused here)
 *   transform_bc1_auto_with_normalization    transform.rs:222   (brute force over mode x decorrelation x split)
 *
 * ColorNormalizationMode (normalize.rs:487-500), passed as a byte: None = 0, Color0Only = 1, ReplicateColor = 2.
 * A block whose 16 decoded pixels are all transparent becomes eight 0xFF bytes; a block whose pixels are one opaque
 * colour that survives RGBA8888 -> RGB565 -> RGBA8888 becomes (colour, 0, indices 0) [Color0Only] or
 * (colour, colour, indices 0) [ReplicateColor]; every other block is kept.  The operation is not invertible (the
 * decoded pixels are unchanged, the bytes are not); untransform_bc1_with_settings undoes only the transform part.
 *
 * `len` is in bytes, a multiple of 8; any pointer alignment; any block count including 0.  Status codes and
 * dxtlt_last_error() as in dxtlt_gfx950.h.
 */
#ifndef DXTLT_BC1_NORMALIZE_H
#define DXTLT_BC1_NORMALIZE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include "dlt_size_estimator.h"

#ifdef __cplusplus
extern "C" {
#endif

#define DXTLT_NORMALIZE_NONE 0
#define DXTLT_NORMALIZE_COLOR0_ONLY 1
#define DXTLT_NORMALIZE_REPLICATE_COLOR 2

/* ---- host pointers ---------------------------------------------------------------------------------- */
/* input_ptr == output_ptr is allowed (in place); partially overlapping buffers are not. */
int32_t dxtlt_bc1_normalize_blocks(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len, uint8_t color_mode);
/* colours: 4 bytes per block, indices: 4 bytes per block, both modified in place */
int32_t dxtlt_bc1_normalize_split_blocks_in_place(uint8_t *colors_ptr, uint8_t *indices_ptr, size_t num_blocks,
                                                  uint8_t color_mode);
/* output_ptrs[m] receives the blocks normalised with mode m.  As in the reference (normalize.rs:447-454) fully
 * transparent blocks become 0xFF in EVERY output, the mode-None one included; solid blocks are kept as they are in the
 * mode-None output.  *out_any_normalized: whether any block was transparent or a normalisable solid colour. */
int32_t dxtlt_bc1_normalize_blocks_all_modes(const uint8_t *input_ptr, uint8_t *const output_ptrs[3], size_t len,
                                             bool *out_any_normalized);
/* == transform_bc1_with_settings(normalize_blocks(input, color_mode), {decorrelation_mode, split}).  `work_ptr` is the
 * reference's len/2 scratch buffer: unused here, may be NULL. */
int32_t dxtlt_transform_bc1_with_normalize_blocks(const uint8_t *input_ptr, uint8_t *output_ptr, uint8_t *work_ptr,
                                                  size_t len, uint8_t color_mode, uint8_t decorrelation_mode,
                                                  bool split_colour_endpoints);
/* Candidate order, estimated section (the first len/2 bytes), strict `<`, skipped-on-estimator-error candidates and
 * the "nothing to normalise -> plain transform_bc1_auto" shortcut as in the reference.  DXTLT_E_ESTIMATOR is
 * returned only when max_compressed_size fails (or from the plain auto path). */
int32_t dxtlt_transform_bc1_auto_with_normalization(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len,
                                                    const DltSizeEstimator *estimator,
                                                    bool use_all_decorrelation_modes, uint8_t *out_color_mode,
                                                    uint8_t *out_decorrelation_mode, bool *out_split_colour_endpoints,
                                                    uint32_t *out_estimator_error);

/* ---- device pointers, asynchronous on `hip_stream` (a hipStream_t; NULL = default stream) ------------- */
int32_t dxtlt_bc1_normalize_blocks_device(const void *d_input, void *d_output, size_t len, uint8_t color_mode,
                                          void *hip_stream);
int32_t dxtlt_bc1_normalize_split_blocks_in_place_device(void *d_colors, void *d_indices, size_t num_blocks,
                                                         uint8_t color_mode, void *hip_stream);
/* d_any_normalized: a device uint32_t the caller has zeroed; set to 1 when any block changed */
int32_t dxtlt_bc1_normalize_blocks_all_modes_device(const void *d_input, void *const d_outputs[3], size_t len,
                                                    uint32_t *d_any_normalized, void *hip_stream);
int32_t dxtlt_transform_bc1_with_normalize_blocks_device(const void *d_input, void *d_output, size_t len,
                                                         uint8_t color_mode, uint8_t decorrelation_mode,
                                                         bool split_colour_endpoints, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
