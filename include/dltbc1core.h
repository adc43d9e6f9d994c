/*
 * dltbc1core.h -- "unstable" core C API of dxt-lossless-transform-bc1, served by libdxtlt_gfx950.so.
 *
 * Mirrors /root/reference/src/core/dxt-lossless-transform-bc1/src/c_api/ (cbindgen naming: PascalCase fields,
 * camelCase arguments): transform_with_settings.rs:73,119 and transform_auto.rs:143.
 *
 * Host pointers in, host pointers out; the transform itself runs on the current HIP device (H2D + gfx950 kernel +
 * D2H), there is no CPU fallback.  Device or runtime failures are reported as TransformationError (8).
 *
 * NOTE: like the reference's generated headers, this header and dltbc1.h (stable API) define different types
 * under the same names (different cdylibs upstream): include only one of them per translation unit.
 */
#ifndef DLTBC1CORE_H
#define DLTBC1CORE_H

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

/* bc1 c_api/transform_auto.rs:37-58 */
typedef enum Dltbc1ErrorCode {
  Success = 0,
  NullDataPointer = 1,
  NullOutputBufferPointer = 2,
  NullEstimatorPointer = 3,
  NullTransformSettingsPointer = 4,
  InvalidDataLength = 5,
  OutputBufferTooSmall = 6,
  SizeEstimationError = 7,
  TransformationError = 8,
} Dltbc1ErrorCode;

typedef struct Dltbc1Result {
  Dltbc1ErrorCode ErrorCode;
} Dltbc1Result;

/* bc1 c_api/transform_auto.rs:27-34 (untransform twin: c_api/transform_with_settings.rs:14-21) */
typedef struct Dltbc1TransformSettings {
  bool SplitColourEndpoints;
  YCoCgVariant DecorrelationMode;
} Dltbc1TransformSettings;

typedef struct Dltbc1UntransformSettings {
  bool SplitColourEndpoints;
  YCoCgVariant DecorrelationMode;
} Dltbc1UntransformSettings;

typedef struct Dltbc1AutoTransformSettings {
  bool UseAllModes;
} Dltbc1AutoTransformSettings;

/* bc1 c_api/transform_with_settings.rs:73
 * Check order: input NULL -> NullDataPointer, output NULL -> NullOutputBufferPointer, then the safe wrapper's
 * length check (InvalidDataLength) and size check (OutputBufferTooSmall). */
Dltbc1Result dltbc1core_transform(const uint8_t *input, size_t inputLen, uint8_t *output, size_t outputLen,
                          Dltbc1TransformSettings details);

/* bc1 c_api/transform_with_settings.rs:119 */
Dltbc1Result dltbc1core_untransform(const uint8_t *input, size_t inputLen, uint8_t *output, size_t outputLen,
                            Dltbc1UntransformSettings details);

/* bc1 c_api/transform_auto.rs:143
 * Brute force over the reference's test order (4 or 8 candidates), estimator called on the endpoint
 * streams only, strict `<` keeps the first best; on success *outDetails holds the settings used. */
Dltbc1Result dltbc1core_transform_auto(const uint8_t *data, size_t dataLen, uint8_t *output, size_t outputLen,
                               const DltSizeEstimator *estimator, Dltbc1AutoTransformSettings settings,
                               Dltbc1TransformSettings *outDetails);

#ifdef __cplusplus
}
#endif
#endif /* DLTBC1CORE_H */
