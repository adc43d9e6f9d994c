// c_api_stable.cpp -- the reference's STABLE C API for BC1 and BC2 (dltbc1_*, dltbc2_*: 13 symbols each) and an additive
// BC3 twin (dltbc3_*, 14 symbols), served by the gfx950 path.  Declarations and reference citations:
// include/dltbc1.h, include/dltbc2.h, include/dltbc3.h.
//
// Builders are plain heap objects holding settings in CORE numbering; the stable YCoCgVariant numbering
// (Variant1=0, Variant2=1, Variant3=2, None=3; api-common/src/reexports/color_565.rs:65-91) is converted at the
// setter, like YCoCgVariant::to_internal_variant does upstream.
#include <stddef.h>
#include <stdint.h>

#include <new>

#include "host_common.h"
#include "../../include/dxtlt_bc7.h"

namespace {

enum StableCode : int32_t {
    kSuccess = 0,
    kInvalidLength = 1,
    kOutputBufferTooSmall = 2,
    kAllocationFailed = 3,
    // additive, above the reference's range (error.rs:10-40 ends at 12): a caller on a box without a usable GPU must be
    // able to tell that from an out-of-memory condition
    kDeviceUnavailable = 100,
    kDeviceError = 101,
    kSizeEstimationFailed = 4,
    kNullDataPointer = 5,
    kNullEstimatorPointer = 6,
    kNullTransformSettingsPointer = 7,
    kNullInputPointer = 8,
    kNullOutputBufferPointer = 9,
    kNullManualTransformBuilderPointer = 10,
    kNullBuilderPointer = 11,
    kNullManualBuilderOutputPointer = 12,
};

struct StableResult {
    int32_t ErrorCode;
};

// Bc1ManualTransformBuilder / Bc2ManualTransformBuilder: defaults Variant1 + split
// (bc1-api transform/manual_transform_builder.rs:24-36)
struct ManualBuilder {
    uint8_t mode_core = 1;
    bool split_colour = true;
    bool split_alpha = true;  // BC3 only (Bc3TransformSettings default, bc3 settings.rs:39-48)
};

struct AutoBuilder {
    DltSizeEstimator estimator;  // copied at construction (auto_transform_builder.rs:63-80)
    bool use_all = false;
};

uint8_t stable_to_core(uint8_t v)
{
    switch (v & 3) {
    case 0: return 1;  // Variant1
    case 1: return 2;  // Variant2
    case 2: return 3;  // Variant3
    default: return 0; // None
    }
}

int32_t map_status(int32_t st)
{
    switch (st) {
    case dxtlt_host::kOk: return kSuccess;
    case dxtlt_host::kInvalidLength: return kInvalidLength;
    case dxtlt_host::kEstimator: return kSizeEstimationFailed;
    case dxtlt_host::kAllocation: return kAllocationFailed;
    case dxtlt_host::kNoDevice: return kDeviceUnavailable;
    default: return kDeviceError;  // HIP runtime failure: allocation on the device, copy, launch (dxtlt_last_error() has the text)
    }
}

StableResult manual_run(int32_t format, bool inverse, const uint8_t* input, size_t input_len, uint8_t* output,
                        size_t output_len, const ManualBuilder* b)
{
    // manual_transform_builder.rs:264-272
    if (input == nullptr)
        return {kNullDataPointer};
    if (output == nullptr)
        return {kNullOutputBufferPointer};
    if (b == nullptr)
        return {kNullManualTransformBuilderPointer};
    const size_t block = format == 1 ? 8 : 16;
    if (input_len % block != 0)
        return {kInvalidLength};
    if (output_len < input_len)
        return {kOutputBufferTooSmall};
    return {map_status(dxtlt_host::transform(format, inverse, input, output, input_len, b->mode_core,
                                             format == 3 && b->split_alpha, b->split_colour))};
}

StableResult auto_run(int32_t format, AutoBuilder* b, const uint8_t* data, size_t data_len, uint8_t* output,
                      size_t output_len, ManualBuilder** out_manual)
{
    // auto_transform_builder.rs:198-210
    if (b == nullptr)
        return {kNullBuilderPointer};
    if (data == nullptr)
        return {kNullDataPointer};
    if (output == nullptr)
        return {kNullOutputBufferPointer};
    if (out_manual == nullptr)
        return {kNullManualBuilderOutputPointer};
    *out_manual = nullptr;
    const size_t block = format == 1 ? 8 : 16;
    if (data_len % block != 0)
        return {kInvalidLength};
    if (output_len < data_len)
        return {kOutputBufferTooSmall};
    dxtlt_host::AutoChoice c{};
    int32_t st = dxtlt_host::transform_auto(format, data, output, data_len, &b->estimator, b->use_all, &c);
    if (st != dxtlt_host::kOk)
        return {map_status(st)};
    ManualBuilder* m = new (std::nothrow) ManualBuilder;
    if (m == nullptr)
        return {kAllocationFailed};
    m->mode_core = c.mode;
    m->split_colour = c.split_colour;
    m->split_alpha = c.split_alpha;
    *out_manual = m;
    return {kSuccess};
}

const char* message(int32_t code, const char* invalid_length_text, const char* settings_name,
                    const char* manual_name, const char* estimate_name)
{
    switch (code) {
    case kSuccess: return "Success";
    case kInvalidLength: return invalid_length_text;
    case kOutputBufferTooSmall: return "Output buffer too small for the operation";
    case kAllocationFailed: return "Memory allocation failed";
    case kDeviceUnavailable: return "No usable HIP device (this library has no CPU fallback)";
    case kDeviceError: return "HIP runtime failure (device allocation, copy or kernel launch); see dxtlt_last_error()";
    case kSizeEstimationFailed: return "Size estimation failed during transform optimization";
    case kNullDataPointer: return "Null pointer provided for data parameter";
    case kNullEstimatorPointer: return "Null pointer provided for DltSizeEstimator parameter";
    case kNullTransformSettingsPointer: return settings_name;
    case kNullInputPointer: return "Null pointer provided for input parameter";
    case kNullOutputBufferPointer: return "Null pointer provided for output parameter";
    case kNullManualTransformBuilderPointer: return manual_name;
    case kNullBuilderPointer: return estimate_name;
    case kNullManualBuilderOutputPointer: return "Null pointer provided for manual builder output parameter";
    default: return "Unknown error";
    }
}

}  // namespace

#define DLT_STABLE_API(N, FMT)                                                                                        \
    ManualBuilder* dltbc##N##_new_ManualTransformBuilder(void) { return new (std::nothrow) ManualBuilder; }           \
    void dltbc##N##_free_ManualTransformBuilder(ManualBuilder* b) { delete b; }                                       \
    ManualBuilder* dltbc##N##_clone_ManualTransformBuilder(const ManualBuilder* b)                                    \
    {                                                                                                                 \
        if (b == nullptr) return nullptr;                                                                             \
        return new (std::nothrow) ManualBuilder(*b);                                                                  \
    }                                                                                                                 \
    void dltbc##N##_ManualTransformBuilder_SetDecorrelationMode(ManualBuilder* b, uint8_t mode)                       \
    {                                                                                                                 \
        if (b) b->mode_core = stable_to_core(mode);                                                                   \
    }                                                                                                                 \
    void dltbc##N##_ManualTransformBuilder_SetSplitColourEndpoints(ManualBuilder* b, bool split)                      \
    {                                                                                                                 \
        if (b) b->split_colour = split;                                                                               \
    }                                                                                                                 \
    void dltbc##N##_ManualTransformBuilder_ResetToDefaults(ManualBuilder* b)                                          \
    {                                                                                                                 \
        if (b) *b = ManualBuilder{};                                                                                  \
    }                                                                                                                 \
    StableResult dltbc##N##_ManualTransformBuilder_Transform(const uint8_t* in, size_t in_len, uint8_t* out,          \
                                                             size_t out_len, ManualBuilder* b)                        \
    {                                                                                                                 \
        return manual_run(FMT, false, in, in_len, out, out_len, b);                                                   \
    }                                                                                                                 \
    StableResult dltbc##N##_ManualTransformBuilder_Untransform(const uint8_t* in, size_t in_len, uint8_t* out,        \
                                                               size_t out_len, ManualBuilder* b)                      \
    {                                                                                                                 \
        return manual_run(FMT, true, in, in_len, out, out_len, b);                                                    \
    }                                                                                                                 \
    AutoBuilder* dltbc##N##_new_AutoTransformBuilder(const DltSizeEstimator* est)                                     \
    {                                                                                                                 \
        if (est == nullptr) return nullptr;                                                                           \
        AutoBuilder* b = new (std::nothrow) AutoBuilder;                                                              \
        if (b) b->estimator = *est;                                                                                   \
        return b;                                                                                                     \
    }                                                                                                                 \
    void dltbc##N##_free_AutoTransformBuilder(AutoBuilder* b) { delete b; }                                           \
    StableResult dltbc##N##_AutoTransformBuilder_SetUseAllDecorrelationModes(AutoBuilder* b, bool use_all)            \
    {                                                                                                                 \
        if (b == nullptr) return {kNullBuilderPointer};                                                               \
        b->use_all = use_all;                                                                                         \
        return {kSuccess};                                                                                            \
    }                                                                                                                 \
    StableResult dltbc##N##_AutoTransformBuilder_Transform(AutoBuilder* b, const uint8_t* data, size_t data_len,      \
                                                           uint8_t* out, size_t out_len, ManualBuilder** out_manual)  \
    {                                                                                                                 \
        return auto_run(FMT, b, data, data_len, out, out_len, out_manual);                                            \
    }

extern "C" {

DLT_STABLE_API(1, 1)
DLT_STABLE_API(2, 2)
DLT_STABLE_API(3, 3)  // additive: the reference's bc3-api crate is empty (include/dltbc3.h)

void dltbc3_ManualTransformBuilder_SetSplitAlphaEndpoints(ManualBuilder* b, bool split)
{
    if (b) b->split_alpha = split;
}

// error.rs:131-175 (bc1) -- static strings
const char* dltbc1_error_message(int32_t code)
{
    return message(code, "Invalid input length: Length must be divisible by 8 (BC1 block size)",
                   "Null pointer provided for Dltbc1TransformSettings parameter",
                   "Null pointer provided for Dltbc1ManualTransformBuilder parameter",
                   "Null pointer provided for Dltbc1EstimateSettingsBuilder parameter");
}

const char* dltbc3_error_message(int32_t code)
{
    return message(code, "Invalid input length: Length must be divisible by 16 (BC3 block size)",
                   "Null pointer provided for Dltbc3TransformSettings parameter",
                   "Null pointer provided for Dltbc3ManualTransformBuilder parameter",
                   "Null pointer provided for Dltbc3EstimateSettingsBuilder parameter");
}

// ---- BC7 (additive, include/dltbc7.h): the mode-split format has no settings, so the builder is an empty handle ----
struct Bc7Builder {
    uint8_t format_version = 0;
};

Bc7Builder* dltbc7_new_ManualTransformBuilder(void) { return new (std::nothrow) Bc7Builder; }
void dltbc7_free_ManualTransformBuilder(Bc7Builder* b) { delete b; }
Bc7Builder* dltbc7_clone_ManualTransformBuilder(const Bc7Builder* b) { return b ? new (std::nothrow) Bc7Builder(*b) : nullptr; }
void dltbc7_ManualTransformBuilder_ResetToDefaults(Bc7Builder* b)
{
    if (b) *b = Bc7Builder{};
}

static StableResult bc7_run(bool inverse, const uint8_t* input, size_t input_len, uint8_t* output, size_t output_len,
                            const Bc7Builder* b)
{
    if (input == nullptr)
        return {kNullDataPointer};
    if (output == nullptr)
        return {kNullOutputBufferPointer};
    if (b == nullptr)
        return {kNullManualTransformBuilderPointer};
    if (input_len % 16 != 0)
        return {kInvalidLength};
    if (output_len < input_len)
        return {kOutputBufferTooSmall};
    return {map_status(inverse ? dxtlt_untransform_bc7(input, output, input_len) : dxtlt_transform_bc7(input, output, input_len))};
}

StableResult dltbc7_ManualTransformBuilder_Transform(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len, Bc7Builder* b)
{
    return bc7_run(false, in, in_len, out, out_len, b);
}
StableResult dltbc7_ManualTransformBuilder_Untransform(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len, Bc7Builder* b)
{
    return bc7_run(true, in, in_len, out, out_len, b);
}

const char* dltbc7_error_message(int32_t code)
{
    return message(code, "Invalid input length: Length must be divisible by 16 (BC7 block size)",
                   "Null pointer provided for Dltbc7TransformSettings parameter",
                   "Null pointer provided for Dltbc7ManualTransformBuilder parameter",
                   "Null pointer provided for Dltbc7EstimateSettingsBuilder parameter");
}

const char* dltbc2_error_message(int32_t code)
{
    return message(code, "Invalid input length: Length must be divisible by 16 (BC2 block size)",
                   "Null pointer provided for Dltbc2TransformSettings parameter",
                   "Null pointer provided for Dltbc2ManualTransformBuilder parameter",
                   "Null pointer provided for Dltbc2EstimateSettingsBuilder parameter");
}

}  // extern "C"
