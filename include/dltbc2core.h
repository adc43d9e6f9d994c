/*
 * dltbc2core.h -- "unstable" core C API of dxt-lossless-transform-bc2, served by libdxtlt_gfx950.so.
 *
 * Mirrors /root/reference/src/core/dxt-lossless-transform-bc2/src/c_api/ (cbindgen naming: PascalCase fields,
 * camelCase arguments): transform_with_settings.rs:64,110 and transform_auto.rs:153.
 *
 * Host pointers in, host pointers out; the transform itself runs on the current HIP device (H2D + gfx950 kernel +
 * D2H), there is no CPU fallback.  Device or runtime failures are reported as TransformationError (8).
 *
 * NOTE: like the reference's generated headers, this header and dltbc2.h (stable API) define different types
 * under the same names (different cdylibs upstream): include only one of them per translation unit.
 */
#ifndef DLTBC2CORE_H
#define DLTBC2CORE_H

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

/* bc2 c_api/transform_auto.rs (same codes as BC1) */
typedef enum Dltbc2ErrorCode {
  Success = 0,
  NullDataPointer = 1,
  NullOutputBufferPointer = 2,
  NullEstimatorPointer = 3,
  NullTransformSettingsPointer = 4,
  InvalidDataLength = 5,
  OutputBufferTooSmall = 6,
  SizeEstimationError = 7,
  TransformationError = 8,
} Dltbc2ErrorCode;

typedef struct Dltbc2Result {
  Dltbc2ErrorCode ErrorCode;
} Dltbc2Result;

/* bc2 c_api/transform_auto.rs / transform_with_settings.rs */
typedef struct Dltbc2TransformSettings {
  bool SplitColourEndpoints;
  YCoCgVariant DecorrelationMode;
} Dltbc2TransformSettings;

typedef struct Dltbc2UntransformSettings {
  bool SplitColourEndpoints;
  YCoCgVariant DecorrelationMode;
} Dltbc2UntransformSettings;

typedef struct Dltbc2AutoTransformSettings {
  bool UseAllModes;
} Dltbc2AutoTransformSettings;

/* bc2 c_api/transform_with_settings.rs:64
 * Check order: input NULL -> NullDataPointer, output NULL -> NullOutputBufferPointer, then the safe wrapper's
 * length check (InvalidDataLength) and size check (OutputBufferTooSmall). */
Dltbc2Result dltbc2core_transform(const uint8_t *input, size_t inputLen, uint8_t *output, size_t outputLen,
                          Dltbc2TransformSettings details);

/* bc2 c_api/transform_with_settings.rs:110 */
Dltbc2Result dltbc2core_untransform(const uint8_t *input, size_t inputLen, uint8_t *output, size_t outputLen,
                            Dltbc2UntransformSettings details);

/* bc2 c_api/transform_auto.rs:153
 * Brute force over the reference's test order (4 or 8 candidates), estimator called on the endpoint
 * streams only, strict `<` keeps the first best; on success *outDetails holds the settings used. */
Dltbc2Result dltbc2core_transform_auto(const uint8_t *data, size_t dataLen, uint8_t *output, size_t outputLen,
                               const DltSizeEstimator *estimator, Dltbc2AutoTransformSettings settings,
                               Dltbc2TransformSettings *outDetails);

#ifdef __cplusplus
}
#endif
#endif /* DLTBC2CORE_H */
