/*
 * dltbc2.h -- STABLE C API of dxt-lossless-transform-bc2-api, served by libdxtlt_gfx950.so.
 *
 * Mirrors /root/reference/src/api/dxt-lossless-transform-bc2-api/src/c_api/ (13 symbols; cbindgen naming:
 * PascalCase fields, camelCase arguments):
 *   transform/manual_transform_builder.rs:71-329   builder lifecycle, setters, Transform / Untransform
 *   transform/auto_transform_builder.rs:63-190     auto builder
 *   error.rs:10-47,131                             error codes, Dltbc2Result, dltbc2_error_message
 *   mod.rs:188-217                                 Dltbc2TransformSettings helper struct
 *
 * Host pointers in and out; the transform runs on the current HIP device.  The reference's enum values keep their
 * numbers; two additive codes above its range report what a CPU library cannot fail with: DeviceUnavailable (100, no
 * usable HIP device) and DeviceError (101, HIP runtime failure).
 *
 * NOTE: this header and dltbc2core.h define different types under the same names (they are different cdylibs
 * upstream): include only one of them per translation unit.
 */
#ifndef DLTBC2_H
#define DLTBC2_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include "dlt_size_estimator.h"

#ifdef __cplusplus
extern "C" {
#endif

/* YCoCgVariant, STABLE numbering, #[repr(u8)] -- NOT the core numbering
 * (dxt-lossless-transform-api-common/src/reexports/color_565.rs:65-91) */
enum YCoCgVariant
#ifdef __cplusplus
  : uint8_t
#endif
{
  Variant1 = 0,
  Variant2 = 1,
  Variant3 = 2,
  None = 3,
};
#ifndef __cplusplus
typedef uint8_t YCoCgVariant;
#endif

/* c_api/error.rs:10-40 */
typedef enum Dltbc2ErrorCode {
  Success = 0,
  InvalidLength = 1,
  OutputBufferTooSmall = 2,
  AllocationFailed = 3,
  SizeEstimationFailed = 4,
  NullDataPointer = 5,
  NullEstimatorPointer = 6,
  NullTransformSettingsPointer = 7,
  NullInputPointer = 8,
  NullOutputBufferPointer = 9,
  NullManualTransformBuilderPointer = 10,
  NullBuilderPointer = 11,
  NullManualBuilderOutputPointer = 12,
  /* ADDITIVE, above the reference's range (error.rs:10-40 ends at 12): the transform runs on a HIP device, and a caller
   * must be able to tell a missing or failing device from an out-of-memory condition.  dxtlt_last_error()
   * (dxtlt_gfx950.h) has the runtime's text. */
  DeviceUnavailable = 100,
  DeviceError = 101,
} Dltbc2ErrorCode;

/* c_api/error.rs:43-47 */
typedef struct Dltbc2Result {
  Dltbc2ErrorCode ErrorCode;
} Dltbc2Result;

/* c_api/mod.rs:188-195 (defaults Variant1 / true, :210-217) */
typedef struct Dltbc2TransformSettings {
  YCoCgVariant DecorrelationMode;
  bool SplitColourEndpoints;
} Dltbc2TransformSettings;

typedef struct Dltbc2UntransformSettings {
  YCoCgVariant DecorrelationMode;
  bool SplitColourEndpoints;
} Dltbc2UntransformSettings;

/* opaque builders (manual_transform_builder.rs:41-44, auto_transform_builder.rs) */
typedef struct Dltbc2ManualTransformBuilder Dltbc2ManualTransformBuilder;
typedef struct Dltbc2AutoTransformBuilder Dltbc2AutoTransformBuilder;

/* manual_transform_builder.rs:71 -- defaults: Variant1, split = true */
Dltbc2ManualTransformBuilder *dltbc2_new_ManualTransformBuilder(void);
/* :86 -- NULL is accepted */
void dltbc2_free_ManualTransformBuilder(Dltbc2ManualTransformBuilder *builder);
/* :107 -- NULL in, NULL out */
Dltbc2ManualTransformBuilder *dltbc2_clone_ManualTransformBuilder(const Dltbc2ManualTransformBuilder *builder);
/* :150 -- NULL builder is ignored */
void dltbc2_ManualTransformBuilder_SetDecorrelationMode(Dltbc2ManualTransformBuilder *builder, YCoCgVariant mode);
/* :183 */
void dltbc2_ManualTransformBuilder_SetSplitColourEndpoints(Dltbc2ManualTransformBuilder *builder, bool split);
/* :203 */
void dltbc2_ManualTransformBuilder_ResetToDefaults(Dltbc2ManualTransformBuilder *builder);
/* :256 -- check order: input NULL -> NullDataPointer, output NULL -> NullOutputBufferPointer, builder NULL ->
 * NullManualTransformBuilderPointer, then InvalidLength, then OutputBufferTooSmall */
Dltbc2Result dltbc2_ManualTransformBuilder_Transform(const uint8_t *input, size_t inputLen, uint8_t *output,
                                                   size_t outputLen, Dltbc2ManualTransformBuilder *builder);
/* :323 */
Dltbc2Result dltbc2_ManualTransformBuilder_Untransform(const uint8_t *input, size_t inputLen, uint8_t *output,
                                                     size_t outputLen, Dltbc2ManualTransformBuilder *builder);

/* auto_transform_builder.rs:63 -- copies *estimator; NULL in, NULL out */
Dltbc2AutoTransformBuilder *dltbc2_new_AutoTransformBuilder(const DltSizeEstimator *estimator);
/* :88 */
void dltbc2_free_AutoTransformBuilder(Dltbc2AutoTransformBuilder *builder);
/* :121 -- NULL builder -> NullBuilderPointer */
Dltbc2Result dltbc2_AutoTransformBuilder_SetUseAllDecorrelationModes(Dltbc2AutoTransformBuilder *builder, bool useAll);
/* :190 -- check order: builder, data, output, outManualBuilder; on success *outManualBuilder is a new manual
 * builder holding the chosen settings (caller frees); on failure it is set to NULL */
Dltbc2Result dltbc2_AutoTransformBuilder_Transform(Dltbc2AutoTransformBuilder *builder, const uint8_t *data,
                                                 size_t dataLen, uint8_t *output, size_t outputLen,
                                                 Dltbc2ManualTransformBuilder **outManualBuilder);

/* error.rs:131 -- static strings */
const char *dltbc2_error_message(Dltbc2ErrorCode errorCode);

#ifdef __cplusplus
}
#endif
#endif /* DLTBC2_H */
