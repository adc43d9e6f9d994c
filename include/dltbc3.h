/*
 * dltbc3.h -- ADDITIVE stable-style C API for BC3, served by libdxtlt_gfx950.so.
 *
 * The reference has NO stable API for BC3: api/dxt-lossless-transform-bc3-api/src/lib.rs is one line (SURVEY.md 0.4,
 * 8(f)-3).  This header gives BC3 the shape of the BC1/BC2 stable C APIs
 * (/root/reference/src/api/dxt-lossless-transform-bc1-api/src/c_api/) over the core functions
 * transform_bc3_with_settings / untransform_bc3_with_settings / transform_bc3_auto
 * (/root/reference/src/core/dxt-lossless-transform-bc3/src/transform/transform_with_settings.rs:32,162,
 * transform_auto.rs:196), plus one extra setter for Bc3TransformSettings::split_alpha_endpoints
 * (bc3 transform/settings.rs:16-30).  Names and layout are this build's choice; error codes, YCoCgVariant numbering
 * (stable) and check order are those of dltbc1.h.
 */
#ifndef DLTBC3_H
#define DLTBC3_H

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
typedef enum Dltbc3ErrorCode {
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
} Dltbc3ErrorCode;

/* c_api/error.rs:43-47 */
typedef struct Dltbc3Result {
  Dltbc3ErrorCode ErrorCode;
} Dltbc3Result;

/* c_api/mod.rs:188-195 (defaults Variant1 / true, :210-217) */
typedef struct Dltbc3TransformSettings {
  YCoCgVariant DecorrelationMode;
  bool SplitAlphaEndpoints;
  bool SplitColourEndpoints;
} Dltbc3TransformSettings;

typedef struct Dltbc3UntransformSettings {
  YCoCgVariant DecorrelationMode;
  bool SplitAlphaEndpoints;
  bool SplitColourEndpoints;
} Dltbc3UntransformSettings;

/* opaque builders (manual_transform_builder.rs:41-44, auto_transform_builder.rs) */
typedef struct Dltbc3ManualTransformBuilder Dltbc3ManualTransformBuilder;
typedef struct Dltbc3AutoTransformBuilder Dltbc3AutoTransformBuilder;

/* manual_transform_builder.rs:71 -- defaults: Variant1, split = true */
Dltbc3ManualTransformBuilder *dltbc3_new_ManualTransformBuilder(void);
/* :86 -- NULL is accepted */
void dltbc3_free_ManualTransformBuilder(Dltbc3ManualTransformBuilder *builder);
/* :107 -- NULL in, NULL out */
Dltbc3ManualTransformBuilder *dltbc3_clone_ManualTransformBuilder(const Dltbc3ManualTransformBuilder *builder);
/* :150 -- NULL builder is ignored */
void dltbc3_ManualTransformBuilder_SetDecorrelationMode(Dltbc3ManualTransformBuilder *builder, YCoCgVariant mode);
/* :183 */
void dltbc3_ManualTransformBuilder_SetSplitColourEndpoints(Dltbc3ManualTransformBuilder *builder, bool split);
/* additive: Bc3TransformSettings::split_alpha_endpoints (default true) */
void dltbc3_ManualTransformBuilder_SetSplitAlphaEndpoints(Dltbc3ManualTransformBuilder *builder, bool split);
/* :203 */
void dltbc3_ManualTransformBuilder_ResetToDefaults(Dltbc3ManualTransformBuilder *builder);
/* :256 -- check order: input NULL -> NullDataPointer, output NULL -> NullOutputBufferPointer, builder NULL ->
 * NullManualTransformBuilderPointer, then InvalidLength, then OutputBufferTooSmall */
Dltbc3Result dltbc3_ManualTransformBuilder_Transform(const uint8_t *input, size_t inputLen, uint8_t *output,
                                                   size_t outputLen, Dltbc3ManualTransformBuilder *builder);
/* :323 */
Dltbc3Result dltbc3_ManualTransformBuilder_Untransform(const uint8_t *input, size_t inputLen, uint8_t *output,
                                                     size_t outputLen, Dltbc3ManualTransformBuilder *builder);

/* auto_transform_builder.rs:63 -- copies *estimator; NULL in, NULL out */
Dltbc3AutoTransformBuilder *dltbc3_new_AutoTransformBuilder(const DltSizeEstimator *estimator);
/* :88 */
void dltbc3_free_AutoTransformBuilder(Dltbc3AutoTransformBuilder *builder);
/* :121 -- NULL builder -> NullBuilderPointer */
Dltbc3Result dltbc3_AutoTransformBuilder_SetUseAllDecorrelationModes(Dltbc3AutoTransformBuilder *builder, bool useAll);
/* :190 -- check order: builder, data, output, outManualBuilder; on success *outManualBuilder is a new manual
 * builder holding the chosen settings (caller frees); on failure it is set to NULL */
Dltbc3Result dltbc3_AutoTransformBuilder_Transform(Dltbc3AutoTransformBuilder *builder, const uint8_t *data,
                                                 size_t dataLen, uint8_t *output, size_t outputLen,
                                                 Dltbc3ManualTransformBuilder **outManualBuilder);

/* error.rs:131 -- static strings */
const char *dltbc3_error_message(Dltbc3ErrorCode errorCode);

#ifdef __cplusplus
}
#endif
#endif /* DLTBC3_H */
