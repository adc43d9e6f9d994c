/*
 * dltbc7.h -- ADDITIVE stable-style C API for BC7, served by libdxtlt_gfx950.so.
 *
 * The reference has NO BC7 transform and no BC7 API: core/dxt-lossless-transform-bc7/src/lib.rs:1-13 holds two bit
 * helpers, api/dxt-lossless-transform-bc7-api is empty.  This header gives the
 * mode-split transform this build defines (docs/BC7_FORMAT.md, include/dxtlt_bc7.h) the shape of the BC1/BC2 stable C
 * APIs (/root/reference/src/api/dxt-lossless-transform-bc1-api/src/c_api/transform/manual_transform_builder.rs:71-323):
 * an opaque manual builder and Transform / Untransform with the same argument order, error codes and check order
 * (input NULL, output NULL, builder NULL, then length, then size).  Version 0 of the format has no settings, so the
 * builder has no setters and there is no auto builder: nothing to choose.  Parity: unpinned, as the format is.
 */
#ifndef DLTBC7_H
#define DLTBC7_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* c_api/error.rs:10-40 (the codes that can occur here keep their values) */
typedef enum Dltbc7ErrorCode {
  Dltbc7Success = 0,
  Dltbc7InvalidLength = 1,
  Dltbc7OutputBufferTooSmall = 2,
  Dltbc7AllocationFailed = 3,
  Dltbc7NullDataPointer = 5,
  Dltbc7NullOutputBufferPointer = 9,
  Dltbc7NullManualTransformBuilderPointer = 10,
  /* additive (above the reference's range): no usable HIP device / HIP runtime failure */
  Dltbc7DeviceUnavailable = 100,
  Dltbc7DeviceError = 101,
} Dltbc7ErrorCode;

/* c_api/error.rs:43-47 */
typedef struct Dltbc7Result {
  Dltbc7ErrorCode ErrorCode;
} Dltbc7Result;

typedef struct Dltbc7ManualTransformBuilder Dltbc7ManualTransformBuilder;

Dltbc7ManualTransformBuilder *dltbc7_new_ManualTransformBuilder(void);
void dltbc7_free_ManualTransformBuilder(Dltbc7ManualTransformBuilder *builder);            /* NULL ok */
Dltbc7ManualTransformBuilder *dltbc7_clone_ManualTransformBuilder(const Dltbc7ManualTransformBuilder *builder); /* NULL -> NULL */
void dltbc7_ManualTransformBuilder_ResetToDefaults(Dltbc7ManualTransformBuilder *builder);

struct Dltbc7Result dltbc7_ManualTransformBuilder_Transform(const uint8_t *input, size_t input_len, uint8_t *output,
                                                            size_t output_len, Dltbc7ManualTransformBuilder *builder);
struct Dltbc7Result dltbc7_ManualTransformBuilder_Untransform(const uint8_t *input, size_t input_len, uint8_t *output,
                                                              size_t output_len, Dltbc7ManualTransformBuilder *builder);

/* static strings, never NULL */
const char *dltbc7_error_message(Dltbc7ErrorCode code);

#ifdef __cplusplus
}
#endif
#endif /* DLTBC7_H */
