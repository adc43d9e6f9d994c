/*
 * dltbc3core.h -- "unstable" core C API of dxt-lossless-transform-bc3, served by libdxtlt_gfx950.so.
 *
 * ADDITIVE: the reference has NO C API for BC3 (dxt-lossless-transform-bc3 has no c_api/ directory).
 * This header gives BC3 the same shape as the BC1/BC2 core C APIs over the core Rust functions
 * transform_bc3_with_settings / untransform_bc3_with_settings / transform_bc3_auto
 * (/root/reference/src/core/dxt-lossless-transform-bc3/src/transform/transform_with_settings.rs:32,162,
 * transform_auto.rs:196).  Struct layout and names are this build's choice.
 *
 * Host pointers in, host pointers out; the transform itself runs on the current HIP device (H2D + gfx950 kernel +
 * D2H), there is no CPU fallback.  Device or runtime failures are reported as TransformationError (8).
 *
 * NOTE: like the reference's generated headers, this header and dltbc3.h (stable API) define different types
 * under the same names (different cdylibs upstream): include only one of them per translation unit.
 */
#ifndef DLTBC3CORE_H
#define DLTBC3CORE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include "dlt_size_estimator.h"

#ifdef __cplusplus
extern "C" {
#endif

/* YCoCgVariant, CORE numbering, #[repr(u8)]
 * (dxt-lossless-transform-common/src/color_565/decorrelate.rs:72-84) */
enum YCoCgVariant
#ifdef __cplusplus
  : uint8_t
#endif
{
  None = 0,
  Variant1 = 1,
  Variant2 = 2,
  Variant3 = 3,
};
#ifndef __cplusplus
typedef uint8_t YCoCgVariant;
#endif

/* same codes as dltbc1core.h */
typedef enum Dltbc3ErrorCode {
  Success = 0,
  NullDataPointer = 1,
  NullOutputBufferPointer = 2,
  NullEstimatorPointer = 3,
  NullTransformSettingsPointer = 4,
  InvalidDataLength = 5,
  OutputBufferTooSmall = 6,
  SizeEstimationError = 7,
  TransformationError = 8,
} Dltbc3ErrorCode;

typedef struct Dltbc3Result {
  Dltbc3ErrorCode ErrorCode;
} Dltbc3Result;

/* fields of Bc3TransformSettings (bc3 transform/settings.rs:16-30) */
typedef struct Dltbc3TransformSettings {
  bool SplitAlphaEndpoints;
  bool SplitColourEndpoints;
  YCoCgVariant DecorrelationMode;
} Dltbc3TransformSettings;

typedef struct Dltbc3UntransformSettings {
  bool SplitAlphaEndpoints;
  bool SplitColourEndpoints;
  YCoCgVariant DecorrelationMode;
} Dltbc3UntransformSettings;

typedef struct Dltbc3AutoTransformSettings {
  bool UseAllModes;
} Dltbc3AutoTransformSettings;

/* wraps transform_bc3_with_settings_safe (bc3 transform/safe/transform_with_settings.rs:90)
 * Check order: input NULL -> NullDataPointer, output NULL -> NullOutputBufferPointer, then the safe wrapper's
 * length check (InvalidDataLength) and size check (OutputBufferTooSmall). */
Dltbc3Result dltbc3core_transform(const uint8_t *input, size_t inputLen, uint8_t *output, size_t outputLen,
                          Dltbc3TransformSettings details);

/* wraps untransform_bc3_with_settings_safe (:196) */
Dltbc3Result dltbc3core_untransform(const uint8_t *input, size_t inputLen, uint8_t *output, size_t outputLen,
                            Dltbc3UntransformSettings details);

/* wraps transform_bc3_auto (bc3 transform/transform_auto.rs:196; test orders settings.rs:91,104)
 * Brute force over the reference's test order (4 or 8 candidates; BC3: 8 or 16), estimator called on the endpoint
 * streams only, strict `<` keeps the first best; on success *outDetails holds the settings used. */
Dltbc3Result dltbc3core_transform_auto(const uint8_t *data, size_t dataLen, uint8_t *output, size_t outputLen,
                               const DltSizeEstimator *estimator, Dltbc3AutoTransformSettings settings,
                               Dltbc3TransformSettings *outDetails);

#ifdef __cplusplus
}
#endif
#endif /* DLTBC3CORE_H */
