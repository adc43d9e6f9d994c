// c_api_core.cpp -- the reference's "unstable" core C API (dltbc1core_*, dltbc2core_*) and its additive BC3 twin
// (dltbc3core_*), served by the gfx950 path.  Declarations and reference citations: include/dltbc{1,2,3}core.h.
//
// Only argument checks and error-code mapping live here; check order follows
// /root/reference/src/core/dxt-lossless-transform-bc1/src/c_api/transform_with_settings.rs:73-104 and
// c_api/transform_auto.rs:143-190 (NULL checks in declaration order, then the safe wrapper's length check, then its
// output-size check).
#include <stddef.h>
#include <stdint.h>

#include "host_common.h"

namespace {

// core error codes (bc1 c_api/transform_auto.rs:37-58) -- identical for BC1, BC2 and the additive BC3
enum CoreCode : int32_t {
    kSuccess = 0,
    kNullDataPointer = 1,
    kNullOutputBufferPointer = 2,
    kNullEstimatorPointer = 3,
    kNullTransformSettingsPointer = 4,
    kInvalidDataLength = 5,
    kOutputBufferTooSmall = 6,
    kSizeEstimationError = 7,
    kTransformationError = 8,
};

struct CoreResult {
    int32_t ErrorCode;  // #[repr(C)] enum == int
};

struct Settings2 {  // Dltbc1TransformSettings / Dltbc2TransformSettings (core layout)
    bool SplitColourEndpoints;
    uint8_t DecorrelationMode;
};

struct Settings3 {  // Dltbc3TransformSettings (additive)
    bool SplitAlphaEndpoints;
    bool SplitColourEndpoints;
    uint8_t DecorrelationMode;
};

struct AutoSettings {
    bool UseAllModes;
};

int32_t map_status(int32_t st)
{
    switch (st) {
    case dxtlt_host::kOk: return kSuccess;
    case dxtlt_host::kInvalidLength: return kInvalidDataLength;
    case dxtlt_host::kEstimator: return kSizeEstimationError;
    case dxtlt_host::kAllocation: return kSizeEstimationError;  // DetermineBestTransform(AllocateError) maps here too
    default: return kTransformationError;
    }
}

CoreResult run(int32_t format, bool inverse, const uint8_t* input, size_t input_len, uint8_t* output, size_t output_len,
               uint8_t mode, bool sa, bool sc)
{
    if (input == nullptr)
        return {kNullDataPointer};
    if (output == nullptr)
        return {kNullOutputBufferPointer};
    const size_t block = format == 1 ? 8 : 16;
    if (input_len % block != 0)
        return {kInvalidDataLength};
    if (output_len < input_len)
        return {kOutputBufferTooSmall};
    if (mode > 3)
        return {kTransformationError};  // not a YCoCgVariant; the reference would be UB here
    return {map_status(dxtlt_host::transform(format, inverse, input, output, input_len, mode, sa, sc))};
}

CoreResult run_auto(int32_t format, const uint8_t* data, size_t data_len, uint8_t* output, size_t output_len,
                    const DltSizeEstimator* estimator, bool use_all, const void* out_details_nonnull,
                    dxtlt_host::AutoChoice* choice)
{
    if (data == nullptr)
        return {kNullDataPointer};
    if (output == nullptr)
        return {kNullOutputBufferPointer};
    if (estimator == nullptr)
        return {kNullEstimatorPointer};
    if (out_details_nonnull == nullptr)
        return {kNullTransformSettingsPointer};
    const size_t block = format == 1 ? 8 : 16;
    if (data_len % block != 0)
        return {kInvalidDataLength};
    if (output_len < data_len)
        return {kOutputBufferTooSmall};
    return {map_status(dxtlt_host::transform_auto(format, data, output, data_len, estimator, use_all, choice))};
}

}  // namespace

extern "C" {

CoreResult dltbc1core_transform(const uint8_t* input, size_t input_len, uint8_t* output, size_t output_len, Settings2 d)
{
    return run(1, false, input, input_len, output, output_len, d.DecorrelationMode, false, d.SplitColourEndpoints);
}

CoreResult dltbc1core_untransform(const uint8_t* input, size_t input_len, uint8_t* output, size_t output_len, Settings2 d)
{
    return run(1, true, input, input_len, output, output_len, d.DecorrelationMode, false, d.SplitColourEndpoints);
}

CoreResult dltbc1core_transform_auto(const uint8_t* data, size_t data_len, uint8_t* output, size_t output_len,
                                     const DltSizeEstimator* estimator, AutoSettings settings, Settings2* out_details)
{
    dxtlt_host::AutoChoice c{};
    CoreResult r = run_auto(1, data, data_len, output, output_len, estimator, settings.UseAllModes, out_details, &c);
    if (r.ErrorCode == kSuccess) {
        out_details->SplitColourEndpoints = c.split_colour;
        out_details->DecorrelationMode = c.mode;
    }
    return r;
}

CoreResult dltbc2core_transform(const uint8_t* input, size_t input_len, uint8_t* output, size_t output_len, Settings2 d)
{
    return run(2, false, input, input_len, output, output_len, d.DecorrelationMode, false, d.SplitColourEndpoints);
}

CoreResult dltbc2core_untransform(const uint8_t* input, size_t input_len, uint8_t* output, size_t output_len, Settings2 d)
{
    return run(2, true, input, input_len, output, output_len, d.DecorrelationMode, false, d.SplitColourEndpoints);
}

CoreResult dltbc2core_transform_auto(const uint8_t* data, size_t data_len, uint8_t* output, size_t output_len,
                                     const DltSizeEstimator* estimator, AutoSettings settings, Settings2* out_details)
{
    dxtlt_host::AutoChoice c{};
    CoreResult r = run_auto(2, data, data_len, output, output_len, estimator, settings.UseAllModes, out_details, &c);
    if (r.ErrorCode == kSuccess) {
        out_details->SplitColourEndpoints = c.split_colour;
        out_details->DecorrelationMode = c.mode;
    }
    return r;
}

CoreResult dltbc3core_transform(const uint8_t* input, size_t input_len, uint8_t* output, size_t output_len, Settings3 d)
{
    return run(3, false, input, input_len, output, output_len, d.DecorrelationMode, d.SplitAlphaEndpoints,
               d.SplitColourEndpoints);
}

CoreResult dltbc3core_untransform(const uint8_t* input, size_t input_len, uint8_t* output, size_t output_len, Settings3 d)
{
    return run(3, true, input, input_len, output, output_len, d.DecorrelationMode, d.SplitAlphaEndpoints,
               d.SplitColourEndpoints);
}

CoreResult dltbc3core_transform_auto(const uint8_t* data, size_t data_len, uint8_t* output, size_t output_len,
                                     const DltSizeEstimator* estimator, AutoSettings settings, Settings3* out_details)
{
    dxtlt_host::AutoChoice c{};
    CoreResult r = run_auto(3, data, data_len, output, output_len, estimator, settings.UseAllModes, out_details, &c);
    if (r.ErrorCode == kSuccess) {
        out_details->SplitAlphaEndpoints = c.split_alpha;
        out_details->SplitColourEndpoints = c.split_colour;
        out_details->DecorrelationMode = c.mode;
    }
    return r;
}

}  // extern "C"
