/*
 * dxtlt_bc23_normalize.h -- C ABI of the BC2 / BC3 block-normalisation entry points of libdxtlt_gfx950.so: the MI355X
 * implementation of the reference's experimental modules
 *   /root/reference/src/core/dxt-lossless-transform-bc2/src/experimental/normalize_blocks/normalize.rs
 *   /root/reference/src/core/dxt-lossless-transform-bc3/src/experimental/normalize_blocks/normalize.rs
 * (Rust-only upstream, like the BC1 module: see dxtlt_bc1_normalize.h and INTEGRATION.md section 6).
 *
 *   normalize_blocks                    bc2 normalize.rs:35     bc3 normalize.rs:36
 *   normalize_blocks_all_modes          bc2 normalize.rs:193    bc3 normalize.rs:419
 *   normalize_split_blocks_in_place     bc2 normalize.rs:382    bc3 normalize.rs:539
 *
 * ColorNormalizationMode: None = 0, Color0Only = 1, ReplicateColor = 2 (bc2 normalize.rs:339-352).  The colour half of a
 * block (bytes 8..15; always decoded in four-colour mode, alpha ignored) whose 16 pixels are one colour that survives
 * RGBA8888 -> RGB565 -> RGBA8888 becomes (colour, 0, indices 0) or (colour, colour, indices 0).  BC2's explicit alpha
 * (bytes 0..7) is never changed.
 * AlphaNormalizationMode (BC3, bc3 normalize.rs:117-139): None = 0, UniformAlphaZeroIndices = 1, OpaqueFillAll = 2,
 * OpaqueZeroAlphaMaxIndices = 3.  An alpha half (bytes 0..7) whose 16 decoded alpha values are equal becomes
 * (alpha, 0, indices 0); for alpha == 255, mode 2 writes eight 0xFF bytes and mode 3 writes (0, 0, indices 0xFF).
 *
 * `len` in bytes, a multiple of 16; any pointer alignment; any block count including 0; input == output is allowed for
 * normalize_blocks.  Status codes and dxtlt_last_error() as in dxtlt_gfx950.h.
 */
#ifndef DXTLT_BC23_NORMALIZE_H
#define DXTLT_BC23_NORMALIZE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DXTLT_ALPHA_NORMALIZE_NONE 0
#define DXTLT_ALPHA_NORMALIZE_UNIFORM_ALPHA_ZERO_INDICES 1
#define DXTLT_ALPHA_NORMALIZE_OPAQUE_FILL_ALL 2
#define DXTLT_ALPHA_NORMALIZE_OPAQUE_ZERO_ALPHA_MAX_INDICES 3

/* ---- host pointers ---------------------------------------------------------------------------------- */
int32_t dxtlt_bc2_normalize_blocks(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len, uint8_t color_mode);
/* alpha_ptr (8 bytes per block) is part of the reference signature; the decision does not depend on it: may be NULL */
int32_t dxtlt_bc2_normalize_split_blocks_in_place(const uint8_t *alpha_ptr, uint8_t *colors_ptr, uint8_t *indices_ptr,
                                                  size_t num_blocks, uint8_t color_mode);
/* output_ptrs[color_mode] */
int32_t dxtlt_bc2_normalize_blocks_all_modes(const uint8_t *input_ptr, uint8_t *const output_ptrs[3], size_t len);

int32_t dxtlt_bc3_normalize_blocks(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len, uint8_t alpha_mode,
                                   uint8_t color_mode);
/* the four sections of the unsplit-endpoint BC3 layout: 2, 6, 4 and 4 bytes per block */
int32_t dxtlt_bc3_normalize_split_blocks_in_place(uint8_t *alpha_endpoints_ptr, uint8_t *alpha_indices_ptr,
                                                  uint8_t *color_endpoints_ptr, uint8_t *color_indices_ptr,
                                                  size_t num_blocks, uint8_t alpha_mode, uint8_t color_mode);
/* output_ptrs[alpha_mode * 3 + color_mode] (the reference's [alpha_mode][color_mode] array, flattened) */
int32_t dxtlt_bc3_normalize_blocks_all_modes(const uint8_t *input_ptr, uint8_t *const output_ptrs[12], size_t len);

/* ---- device pointers, asynchronous on `hip_stream` --------------------------------------------------- */
int32_t dxtlt_bc2_normalize_blocks_device(const void *d_input, void *d_output, size_t len, uint8_t color_mode,
                                          void *hip_stream);
int32_t dxtlt_bc2_normalize_split_blocks_in_place_device(void *d_colors, void *d_indices, size_t num_blocks,
                                                         uint8_t color_mode, void *hip_stream);
int32_t dxtlt_bc2_normalize_blocks_all_modes_device(const void *d_input, void *const d_outputs[3], size_t len,
                                                    void *hip_stream);
int32_t dxtlt_bc3_normalize_blocks_device(const void *d_input, void *d_output, size_t len, uint8_t alpha_mode,
                                          uint8_t color_mode, void *hip_stream);
int32_t dxtlt_bc3_normalize_split_blocks_in_place_device(void *d_alpha_endpoints, void *d_alpha_indices,
                                                         void *d_color_endpoints, void *d_color_indices,
                                                         size_t num_blocks, uint8_t alpha_mode, uint8_t color_mode,
                                                         void *hip_stream);
int32_t dxtlt_bc3_normalize_blocks_all_modes_device(const void *d_input, void *const d_outputs[12], size_t len,
                                                    void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
