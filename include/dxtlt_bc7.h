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
 * dxtlt_bc7_workspace_bytes(len) bytes; they enqueue 4 kernels on the stream and do not synchronise.
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

#ifdef __cplusplus
}
#endif
#endif /* DXTLT_BC7_H */
