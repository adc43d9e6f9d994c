/*
 * dxtlt_color565.h -- C ABI of the array-level RGB565 colour operations of libdxtlt_gfx950.so: the MI355X implementation
 * of the reference's common-crate entry points (rows a2 / a3 of SURVEY.md section 8(a) as stand-alone operations)
 *
 *   Color565::decorrelate_ycocg_r_ptr      core/dxt-lossless-transform-common/src/color_565/decorrelate_batch_ptr.rs:336
 *   Color565::recorrelate_ycocg_r_ptr      core/dxt-lossless-transform-common/src/color_565/decorrelate_batch_ptr.rs:378
 *   Color565::recorrelate_ycocg_r_ptr_split  .../color_565/decorrelate_batch_split_ptr.rs:324
 *   split_color_endpoints                  core/dxt-lossless-transform-common/src/transforms/split_565_color_endpoints/mod.rs:110
 *
 * `variant` = core YCoCgVariant numbering (None = 0, Variant1..3); colours are little-endian uint16_t; `num_items`
 * counts colours.  src == dst is allowed for the first two.  Status codes and dxtlt_last_error() as in dxtlt_gfx950.h.
 * The transform kernels do all of this fused; these exist for callers that hold colour arrays on their own (the
 * reference's experimental transform builds its result from exactly these three steps).
 */
#ifndef DXTLT_COLOR565_H
#define DXTLT_COLOR565_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- host pointers ---------------------------------------------------------------------------------- */
int32_t dxtlt_color565_decorrelate_ycocg_r(const uint16_t *src_ptr, uint16_t *dst_ptr, size_t num_items, uint8_t variant);
int32_t dxtlt_color565_recorrelate_ycocg_r(const uint16_t *src_ptr, uint16_t *dst_ptr, size_t num_items, uint8_t variant);
/* dst[2k] = recorrelate(src0[k]), dst[2k + 1] = recorrelate(src1[k]); num_items (even) counts the colours written */
int32_t dxtlt_color565_recorrelate_ycocg_r_split(const uint16_t *src_ptr_0, const uint16_t *src_ptr_1, uint16_t *dst_ptr,
                                                 size_t num_items, uint8_t variant);
/* (c0, c1) pairs -> all c0, then all c1; colors_len_bytes is a multiple of 4 */
int32_t dxtlt_split_565_color_endpoints(const uint16_t *colors, uint16_t *colors_out, size_t colors_len_bytes);

/* ---- device pointers, asynchronous on `hip_stream` --------------------------------------------------- */
int32_t dxtlt_color565_decorrelate_ycocg_r_device(const void *d_src, void *d_dst, size_t num_items, uint8_t variant,
                                                  void *hip_stream);
int32_t dxtlt_color565_recorrelate_ycocg_r_device(const void *d_src, void *d_dst, size_t num_items, uint8_t variant,
                                                  void *hip_stream);
int32_t dxtlt_color565_recorrelate_ycocg_r_split_device(const void *d_src_0, const void *d_src_1, void *d_dst,
                                                        size_t num_items, uint8_t variant, void *hip_stream);
int32_t dxtlt_split_565_color_endpoints_device(const void *d_colors, void *d_colors_out, size_t colors_len_bytes,
                                               void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
