/*
 * dxtlt_decode.h -- C ABI of the block decoders of libdxtlt_gfx950.so: the MI355X implementation of the reference's
 * util modules as array operations
 *
 *   decode_bc1_block   core/dxt-lossless-transform-bc1/src/util/bc1_decode.rs:42   (+ _from_slice :109)
 *   decode_bc2_block   core/dxt-lossless-transform-bc2/src/util/bc2_decode.rs:44   (+ _from_slice :130)
 *   decode_bc3_block   core/dxt-lossless-transform-bc3/src/util/bc3_decode.rs:43   (+ _from_slice :181)
 *   Decoded4x4Block    core/dxt-lossless-transform-common/src/decoded_4x4_block.rs:56
 *
 * The reference decodes one block per call; here one call decodes `len / block_size` blocks into that many
 * Decoded4x4Block records: 64 bytes each, sixteen pixels in row-major order, every pixel the bytes r, g, b, a
 * (Color8888, color_8888.rs:30).  BC1 uses the three-colour + transparent mode when c0 <= c1; BC2 / BC3 colours are
 * always four-colour; interpolation is the "ideal" DX9 rounding the reference documents (integer / 3, / 2, / 7, / 5
 * on the 8-bit expansions).
 *
 * dxtlt_count_pixel_differences_*: the number of blocks whose sixteen decoded pixels differ between two block arrays
 * of equal length -- what the reference's normalisation tests assert to be zero ("decode before == decode after").
 *
 * Status codes and dxtlt_last_error() as in dxtlt_gfx950.h: 1 = len not a multiple of the block size, 2 = NULL
 * pointer, bad format or pixels_len < 64 * blocks.
 */
#ifndef DXTLT_DECODE_H
#define DXTLT_DECODE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DXTLT_DECODED_BLOCK_BYTES 64

/* ---- host pointers ---------------------------------------------------------------------------------- */
int32_t dxtlt_decode_bc1_blocks(const uint8_t *blocks, size_t len, uint8_t *pixels, size_t pixels_len);
int32_t dxtlt_decode_bc2_blocks(const uint8_t *blocks, size_t len, uint8_t *pixels, size_t pixels_len);
int32_t dxtlt_decode_bc3_blocks(const uint8_t *blocks, size_t len, uint8_t *pixels, size_t pixels_len);
/* format = 1, 2, 3 */
int32_t dxtlt_count_pixel_differences(int32_t format, const uint8_t *blocks_a, const uint8_t *blocks_b, size_t len,
                                      uint64_t *out_count);

/* ---- device pointers, asynchronous on `hip_stream` --------------------------------------------------- */
int32_t dxtlt_decode_bc1_blocks_device(const void *d_blocks, size_t len, void *d_pixels, size_t pixels_len, void *hip_stream);
int32_t dxtlt_decode_bc2_blocks_device(const void *d_blocks, size_t len, void *d_pixels, size_t pixels_len, void *hip_stream);
int32_t dxtlt_decode_bc3_blocks_device(const void *d_blocks, size_t len, void *d_pixels, size_t pixels_len, void *hip_stream);
/* d_count = one uint64_t in device memory; zeroed, then accumulated, on the stream */
int32_t dxtlt_count_pixel_differences_device(int32_t format, const void *d_blocks_a, const void *d_blocks_b, size_t len,
                                             uint64_t *d_count, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
