/*
 * dlt_size_estimator.h -- the C size-estimator vtable passed to the *_transform_auto entry points.
 *
 * Mirrors `DltSizeEstimator` of dxt-lossless-transform-api-common
 * (/root/reference/src/api/dxt-lossless-transform-api-common/src/c_api/size_estimation.rs:18-52),
 * cbindgen field naming (PascalCase, .github/cbindgen_c.toml `rename_fields`).
 * Both callbacks return 0 on success and any other value as an error code.
 */
#ifndef DLT_SIZE_ESTIMATOR_H
#define DLT_SIZE_ESTIMATOR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* size_estimation.rs:18-19 */
typedef uint32_t (*DltMaxCompressedSizeFn)(void *context, size_t len_bytes, size_t *out_size);

/* size_estimation.rs:33-40.  `output_ptr`/`output_len` is a scratch buffer of max_compressed_size bytes
 * (NULL/0 when max_compressed_size reported 0). */
typedef uint32_t (*DltEstimateCompressedSizeFn)(void *context, const uint8_t *input_ptr, size_t len_bytes,
                                                uint8_t *output_ptr, size_t output_len, size_t *out_size);

/* size_estimation.rs:45-52 */
typedef struct DltSizeEstimator {
  void *Context;
  DltMaxCompressedSizeFn MaxCompressedSize;
  DltEstimateCompressedSizeFn EstimateCompressedSize;
} DltSizeEstimator;

#ifdef __cplusplus
}
#endif
#endif /* DLT_SIZE_ESTIMATOR_H */
