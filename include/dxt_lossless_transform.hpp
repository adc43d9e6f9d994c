// dxt_lossless_transform.hpp -- C++17 host-side mirror of the reference's Rust API for the BCn transform path,
// header-only over the C ABI of libdxtlt_gfx950.so (dxtlt_gfx950.h).
//
// The reference is a Rust workspace and this image has no Rust toolchain, so the host layer a Rust user would see is
// restated here in C++ with the same names, argument meaning and error behaviour (paths under /root/reference/src/):
//
//   core::YCoCgVariant                 core/dxt-lossless-transform-common/src/color_565/decorrelate.rs:72-84
//   core::Bc{1,2,3}TransformSettings   core/dxt-lossless-transform-bc{1,2,3}/src/transform/settings.rs:16-48
//   core::transform_bcN_with_settings  core/.../transform/transform_with_settings.rs:31,92 (bc2 :30,93; bc3 :32,162)
//   core::*_safe + BcNValidationError  core/.../transform/safe/transform_with_settings.rs:18-31,88,192
//   core::transform_bcN_auto           core/.../transform/transform_auto.rs:200 (bc2/bc3 :196)
//   core::experimental::*              core/dxt-lossless-transform-bc1/src/experimental/normalize_blocks/{mod,normalize,transform}.rs
//   api::YCoCgVariant (renumbered)     api/dxt-lossless-transform-api-common/src/reexports/color_565.rs:65-91
//   api::Bc{1,2}ManualTransformBuilder api/dxt-lossless-transform-bc1-api/src/transform/manual_transform_builder.rs:18-150
//   api::Bc{1,2}AutoTransformBuilder   api/dxt-lossless-transform-bc1-api/src/transform/auto_transform_builder.rs:15-130
//   api::Bc{1,2}Error                  api/dxt-lossless-transform-bc1-api/src/error.rs:11-35
//
// Differences forced by the device: the reference's unsafe fns cannot fail; here a device/runtime failure throws
// dxt_lossless_transform::DeviceError (never a silent CPU fallback -- there is none in this library).
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>

#include "dxtlt_bc1_normalize.h"
#include "dxtlt_bc23_normalize.h"
#include "dxtlt_bc7.h"
#include "dxtlt_color565.h"
#include "dxtlt_decode.h"
#include "dxtlt_gfx950.h"

namespace dxt_lossless_transform {

struct DeviceError : std::runtime_error {
    int32_t code;
    DeviceError(int32_t c, const char* what) : std::runtime_error(what), code(c) {}
};

namespace detail {
inline void check_device(int32_t rc)
{
    if (rc != DXTLT_OK)
        throw DeviceError(rc, dxtlt_last_error());
}
}  // namespace detail

// =====================================================================================================
// core crates (unstable API)
// =====================================================================================================
namespace core {

enum class YCoCgVariant : uint8_t { None = 0, Variant1 = 1, Variant2 = 2, Variant3 = 3 };

struct Bc1TransformSettings {
    YCoCgVariant decorrelation_mode = YCoCgVariant::Variant1;  // Default: settings.rs:35-43
    bool split_colour_endpoints = true;
    bool operator==(const Bc1TransformSettings& o) const
    {
        return decorrelation_mode == o.decorrelation_mode && split_colour_endpoints == o.split_colour_endpoints;
    }
    bool operator!=(const Bc1TransformSettings& o) const { return !(*this == o); }
    // settings.rs:68 -- every variant x {true, false}
    static std::array<Bc1TransformSettings, 8> all_combinations()
    {
        std::array<Bc1TransformSettings, 8> a{};
        int i = 0;
        for (int v = 0; v < 4; ++v)
            for (bool s : {true, false})
                a[i++] = {static_cast<YCoCgVariant>(v), s};
        return a;
    }
};
using Bc1UntransformSettings = Bc1TransformSettings;  // settings.rs:33
using Bc2TransformSettings = Bc1TransformSettings;    // same fields and defaults (bc2 settings.rs:16)
using Bc2UntransformSettings = Bc2TransformSettings;

struct Bc3TransformSettings {
    YCoCgVariant decorrelation_mode = YCoCgVariant::Variant1;  // Default: bc3 settings.rs:39-48
    bool split_alpha_endpoints = true;
    bool split_colour_endpoints = true;
    bool operator==(const Bc3TransformSettings& o) const
    {
        return decorrelation_mode == o.decorrelation_mode && split_alpha_endpoints == o.split_alpha_endpoints &&
               split_colour_endpoints == o.split_colour_endpoints;
    }
    static std::array<Bc3TransformSettings, 16> all_combinations()  // bc3 settings.rs:74
    {
        std::array<Bc3TransformSettings, 16> a{};
        int i = 0;
        for (int v = 0; v < 4; ++v)
            for (bool sa : {true, false})
                for (bool sc : {true, false})
                    a[i++] = {static_cast<YCoCgVariant>(v), sa, sc};
        return a;
    }
};
using Bc3UntransformSettings = Bc3TransformSettings;

// ---- unsafe pointer API: len must be a multiple of the block size, buffers must not overlap ----
inline void transform_bc1_with_settings(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len, Bc1TransformSettings s)
{
    detail::check_device(dxtlt_transform_bc1_with_settings(input_ptr, output_ptr, len, (uint8_t)s.decorrelation_mode,
                                                           s.split_colour_endpoints));
}
inline void untransform_bc1_with_settings(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len, Bc1UntransformSettings s)
{
    detail::check_device(dxtlt_untransform_bc1_with_settings(input_ptr, output_ptr, len, (uint8_t)s.decorrelation_mode,
                                                             s.split_colour_endpoints));
}
inline void transform_bc2_with_settings(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len, Bc2TransformSettings s)
{
    detail::check_device(dxtlt_transform_bc2_with_settings(input_ptr, output_ptr, len, (uint8_t)s.decorrelation_mode,
                                                           s.split_colour_endpoints));
}
inline void untransform_bc2_with_settings(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len, Bc2UntransformSettings s)
{
    detail::check_device(dxtlt_untransform_bc2_with_settings(input_ptr, output_ptr, len, (uint8_t)s.decorrelation_mode,
                                                             s.split_colour_endpoints));
}
inline void transform_bc3_with_settings(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len, Bc3TransformSettings s)
{
    detail::check_device(dxtlt_transform_bc3_with_settings(input_ptr, output_ptr, len, (uint8_t)s.decorrelation_mode,
                                                           s.split_alpha_endpoints, s.split_colour_endpoints));
}
inline void untransform_bc3_with_settings(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len, Bc3UntransformSettings s)
{
    detail::check_device(dxtlt_untransform_bc3_with_settings(input_ptr, output_ptr, len, (uint8_t)s.decorrelation_mode,
                                                             s.split_alpha_endpoints, s.split_colour_endpoints));
}

// ---- safe slice API: Result<(), BcNValidationError> ----
struct ValidationError {  // Bc1ValidationError / Bc2ValidationError / Bc3ValidationError
    enum Kind { Ok, InvalidLength, OutputBufferTooSmall } kind = Ok;
    size_t length = 0;  // InvalidLength(len)
    size_t needed = 0;  // OutputBufferTooSmall { needed, actual }
    size_t actual = 0;
    bool is_ok() const { return kind == Ok; }
    bool is_err() const { return kind != Ok; }
};

namespace detail_safe {
inline ValidationError validate(size_t in_len, size_t out_len, size_t block)
{
    // order of safe/transform_with_settings.rs:93-105: length first, then output size
    ValidationError e;
    if (in_len % block != 0) {
        e.kind = ValidationError::InvalidLength;
        e.length = in_len;
    } else if (out_len < in_len) {
        e.kind = ValidationError::OutputBufferTooSmall;
        e.needed = in_len;
        e.actual = out_len;
    }
    return e;
}
}  // namespace detail_safe

#define DXTLT_SAFE_PAIR(N, BLOCK, SETTINGS)                                                                           \
    inline ValidationError transform_bc##N##_with_settings_safe(const uint8_t* input, size_t input_len, uint8_t* output, \
                                                                size_t output_len, SETTINGS s)                       \
    {                                                                                                                \
        ValidationError e = detail_safe::validate(input_len, output_len, BLOCK);                                     \
        if (e.is_ok()) transform_bc##N##_with_settings(input, output, input_len, s);                                 \
        return e;                                                                                                    \
    }                                                                                                                \
    inline ValidationError untransform_bc##N##_with_settings_safe(const uint8_t* input, size_t input_len,            \
                                                                  uint8_t* output, size_t output_len, SETTINGS s)    \
    {                                                                                                                \
        ValidationError e = detail_safe::validate(input_len, output_len, BLOCK);                                     \
        if (e.is_ok()) untransform_bc##N##_with_settings(input, output, input_len, s);                               \
        return e;                                                                                                    \
    }
DXTLT_SAFE_PAIR(1, 8, Bc1TransformSettings)
DXTLT_SAFE_PAIR(2, 16, Bc2TransformSettings)
DXTLT_SAFE_PAIR(3, 16, Bc3TransformSettings)
#undef DXTLT_SAFE_PAIR

// ---- transform_bcN_auto: any estimator type with
//        bool max_compressed_size(size_t len, size_t& out)          (false = error)
//        bool estimate_compressed_size(const uint8_t* in, size_t len, uint8_t* scratch, size_t scratch_len, size_t& out)
//      (the shape of SizeEstimationOperations, api-common/src/estimate/mod.rs:24) ----
struct DetermineBestTransformError {
    enum Kind { Ok, AllocateError, SizeEstimationError } kind = Ok;
    bool is_ok() const { return kind == Ok; }
};

template <class Estimator>
struct EstimateSettings {  // Bc1EstimateSettings<T> etc. (transform_auto.rs)
    Estimator size_estimator;
    bool use_all_decorrelation_modes = false;
};

namespace detail_auto {
template <class E>
uint32_t max_trampoline(void* ctx, size_t len, size_t* out)
{
    return static_cast<E*>(ctx)->max_compressed_size(len, *out) ? 0u : 1u;
}
template <class E>
uint32_t est_trampoline(void* ctx, const uint8_t* in, size_t len, uint8_t* scratch, size_t scratch_len, size_t* out)
{
    return static_cast<E*>(ctx)->estimate_compressed_size(in, len, scratch, scratch_len, *out) ? 0u : 1u;
}
template <class E>
DltSizeEstimator make_vtable(E& e)
{
    DltSizeEstimator v;
    v.Context = &e;
    v.MaxCompressedSize = &max_trampoline<E>;
    v.EstimateCompressedSize = &est_trampoline<E>;
    return v;
}
inline DetermineBestTransformError map(int32_t rc)
{
    DetermineBestTransformError e;
    if (rc == DXTLT_OK) return e;
    if (rc == DXTLT_E_ESTIMATOR) { e.kind = DetermineBestTransformError::SizeEstimationError; return e; }
    if (rc == DXTLT_E_ALLOCATION) { e.kind = DetermineBestTransformError::AllocateError; return e; }
    throw DeviceError(rc, dxtlt_last_error());
}
}  // namespace detail_auto

template <class E>
std::pair<Bc1TransformSettings, DetermineBestTransformError> transform_bc1_auto(const uint8_t* input_ptr, uint8_t* output_ptr,
                                                                                size_t len, EstimateSettings<E>& options)
{
    DltSizeEstimator v = detail_auto::make_vtable(options.size_estimator);
    uint8_t mode = 1;
    bool sc = true;
    int32_t rc = dxtlt_transform_bc1_auto(input_ptr, output_ptr, len, &v, options.use_all_decorrelation_modes, &mode, &sc, nullptr);
    return {Bc1TransformSettings{static_cast<YCoCgVariant>(mode), sc}, detail_auto::map(rc)};
}
template <class E>
std::pair<Bc2TransformSettings, DetermineBestTransformError> transform_bc2_auto(const uint8_t* input_ptr, uint8_t* output_ptr,
                                                                                size_t len, EstimateSettings<E>& options)
{
    DltSizeEstimator v = detail_auto::make_vtable(options.size_estimator);
    uint8_t mode = 1;
    bool sc = true;
    int32_t rc = dxtlt_transform_bc2_auto(input_ptr, output_ptr, len, &v, options.use_all_decorrelation_modes, &mode, &sc, nullptr);
    return {Bc2TransformSettings{static_cast<YCoCgVariant>(mode), sc}, detail_auto::map(rc)};
}
template <class E>
std::pair<Bc3TransformSettings, DetermineBestTransformError> transform_bc3_auto(const uint8_t* input_ptr, uint8_t* output_ptr,
                                                                                size_t len, EstimateSettings<E>& options)
{
    DltSizeEstimator v = detail_auto::make_vtable(options.size_estimator);
    uint8_t mode = 1;
    bool sa = true, sc = true;
    int32_t rc = dxtlt_transform_bc3_auto(input_ptr, output_ptr, len, &v, options.use_all_decorrelation_modes, &mode, &sa, &sc, nullptr);
    return {Bc3TransformSettings{static_cast<YCoCgVariant>(mode), sa, sc}, detail_auto::map(rc)};
}

// Additive (dxtlt_gfx950.h): evaluate every distinct candidate section of the auto transforms concurrently on `threads`
// host threads -- same settings and bytes; the estimator's max_compressed_size / estimate_compressed_size must then be
// safe to call from several threads at once.  Process-wide; 1 restores the reference's sequence of calls.
inline void set_auto_estimator_threads(int threads) { dxtlt_set_auto_estimator_threads(threads); }
inline int auto_estimator_threads() { return dxtlt_get_auto_estimator_threads(); }

// ---- dxt_lossless_transform_bc1::experimental::normalize_blocks (normalize.rs, transform.rs, mod.rs) ------------
namespace experimental {

enum class ColorNormalizationMode : uint8_t { None = 0, Color0Only = 1, ReplicateColor = 2 };  // normalize.rs:487-500

struct Bc1TransformDetailsWithNormalization {  // mod.rs:98-135
    ColorNormalizationMode color_normalization_mode = ColorNormalizationMode::None;
    YCoCgVariant decorrelation_mode = YCoCgVariant::Variant1;
    bool split_colour_endpoints = true;

    Bc1TransformDetailsWithNormalization() = default;
    Bc1TransformDetailsWithNormalization(ColorNormalizationMode m, YCoCgVariant v, bool s)
        : color_normalization_mode(m), decorrelation_mode(v), split_colour_endpoints(s) {}
    Bc1TransformDetailsWithNormalization(Bc1TransformSettings s)  // impl From<Bc1TransformSettings>
        : decorrelation_mode(s.decorrelation_mode), split_colour_endpoints(s.split_colour_endpoints) {}
    operator Bc1UntransformSettings() const { return {decorrelation_mode, split_colour_endpoints}; }  // impl From<...>
    bool operator==(const Bc1TransformDetailsWithNormalization& o) const
    {
        return color_normalization_mode == o.color_normalization_mode && decorrelation_mode == o.decorrelation_mode &&
               split_colour_endpoints == o.split_colour_endpoints;
    }
};

// normalize.rs:38; input_ptr == output_ptr is allowed
inline void normalize_blocks(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len, ColorNormalizationMode color_mode)
{
    detail::check_device(dxtlt_bc1_normalize_blocks(input_ptr, output_ptr, len, static_cast<uint8_t>(color_mode)));
}
// normalize.rs:286
inline void normalize_split_blocks_in_place(uint8_t* colors_ptr, uint8_t* indices_ptr, size_t num_blocks,
                                            ColorNormalizationMode color_mode)
{
    detail::check_device(dxtlt_bc1_normalize_split_blocks_in_place(colors_ptr, indices_ptr, num_blocks,
                                                                   static_cast<uint8_t>(color_mode)));
}
// normalize.rs:417; returns whether any block was normalised
inline bool normalize_blocks_all_modes(const uint8_t* input_ptr, const std::array<uint8_t*, 3>& output_ptrs, size_t len)
{
    bool any = false;
    detail::check_device(dxtlt_bc1_normalize_blocks_all_modes(input_ptr, output_ptrs.data(), len, &any));
    return any;
}
// transform.rs:65; work_ptr is accepted for signature parity and not used
inline void transform_bc1_with_normalize_blocks(const uint8_t* input_ptr, uint8_t* output_ptr, uint8_t* work_ptr, size_t len,
                                                Bc1TransformDetailsWithNormalization o)
{
    detail::check_device(dxtlt_transform_bc1_with_normalize_blocks(
        input_ptr, output_ptr, work_ptr, len, static_cast<uint8_t>(o.color_normalization_mode),
        static_cast<uint8_t>(o.decorrelation_mode), o.split_colour_endpoints));
}
// transform.rs:222
template <class E>
std::pair<Bc1TransformDetailsWithNormalization, DetermineBestTransformError> transform_bc1_auto_with_normalization(
    const uint8_t* input_ptr, uint8_t* output_ptr, size_t len, EstimateSettings<E>& options)
{
    DltSizeEstimator v = detail_auto::make_vtable(options.size_estimator);
    uint8_t norm = 0, mode = 1;
    bool sc = true;
    int32_t rc = dxtlt_transform_bc1_auto_with_normalization(input_ptr, output_ptr, len, &v, options.use_all_decorrelation_modes,
                                                             &norm, &mode, &sc, nullptr);
    return {Bc1TransformDetailsWithNormalization{static_cast<ColorNormalizationMode>(norm), static_cast<YCoCgVariant>(mode), sc},
            detail_auto::map(rc)};
}

// ---- dxt_lossless_transform_bc2 / _bc3 ::experimental::normalize_blocks (normalize.rs of those crates) ----------
namespace bc2 {
using ColorNormalizationMode = experimental::ColorNormalizationMode;   // bc2 normalize.rs:339-352
inline void normalize_blocks(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len, ColorNormalizationMode color_mode)
{
    detail::check_device(dxtlt_bc2_normalize_blocks(input_ptr, output_ptr, len, static_cast<uint8_t>(color_mode)));
}
inline void normalize_blocks_all_modes(const uint8_t* input_ptr, const std::array<uint8_t*, 3>& output_ptrs, size_t len)
{
    detail::check_device(dxtlt_bc2_normalize_blocks_all_modes(input_ptr, output_ptrs.data(), len));
}
inline void normalize_split_blocks_in_place(const uint8_t* alpha_ptr, uint8_t* colors_ptr, uint8_t* indices_ptr,
                                            size_t num_blocks, ColorNormalizationMode color_mode)
{
    detail::check_device(dxtlt_bc2_normalize_split_blocks_in_place(alpha_ptr, colors_ptr, indices_ptr, num_blocks,
                                                                   static_cast<uint8_t>(color_mode)));
}
}  // namespace bc2

namespace bc3 {
using ColorNormalizationMode = experimental::ColorNormalizationMode;   // bc3 normalize.rs:143-156
enum class AlphaNormalizationMode : uint8_t {                           // bc3 normalize.rs:117-139
    None = 0, UniformAlphaZeroIndices = 1, OpaqueFillAll = 2, OpaqueZeroAlphaMaxIndices = 3
};
inline void normalize_blocks(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len, AlphaNormalizationMode alpha_mode,
                             ColorNormalizationMode color_mode)
{
    detail::check_device(dxtlt_bc3_normalize_blocks(input_ptr, output_ptr, len, static_cast<uint8_t>(alpha_mode),
                                                    static_cast<uint8_t>(color_mode)));
}
// output_ptrs[alpha_mode][color_mode], like the reference's 2-D array
inline void normalize_blocks_all_modes(const uint8_t* input_ptr, const std::array<std::array<uint8_t*, 3>, 4>& output_ptrs,
                                       size_t len)
{
    uint8_t* flat[12];
    for (int a = 0; a < 4; ++a)
        for (int c = 0; c < 3; ++c)
            flat[a * 3 + c] = output_ptrs[a][c];
    detail::check_device(dxtlt_bc3_normalize_blocks_all_modes(input_ptr, flat, len));
}
inline void normalize_split_blocks_in_place(uint8_t* alpha_endpoints_ptr, uint8_t* alpha_indices_ptr,
                                            uint8_t* color_endpoints_ptr, uint8_t* color_indices_ptr, size_t num_blocks,
                                            AlphaNormalizationMode alpha_mode, ColorNormalizationMode color_mode)
{
    detail::check_device(dxtlt_bc3_normalize_split_blocks_in_place(alpha_endpoints_ptr, alpha_indices_ptr, color_endpoints_ptr,
                                                                   color_indices_ptr, num_blocks,
                                                                   static_cast<uint8_t>(alpha_mode),
                                                                   static_cast<uint8_t>(color_mode)));
}
}  // namespace bc3

}  // namespace experimental

// dxt_lossless_transform_common: array-level RGB565 operations (color_565/decorrelate_batch_ptr.rs:336,378,
// decorrelate_batch_split_ptr.rs:324, transforms/split_565_color_endpoints/mod.rs:110)
namespace common {
struct Color565 {
    static void decorrelate_ycocg_r_ptr(const uint16_t* src_ptr, uint16_t* dst_ptr, size_t num_items, YCoCgVariant variant)
    {
        detail::check_device(dxtlt_color565_decorrelate_ycocg_r(src_ptr, dst_ptr, num_items, static_cast<uint8_t>(variant)));
    }
    static void recorrelate_ycocg_r_ptr(const uint16_t* src_ptr, uint16_t* dst_ptr, size_t num_items, YCoCgVariant variant)
    {
        detail::check_device(dxtlt_color565_recorrelate_ycocg_r(src_ptr, dst_ptr, num_items, static_cast<uint8_t>(variant)));
    }
    static void recorrelate_ycocg_r_ptr_split(const uint16_t* src_ptr_0, const uint16_t* src_ptr_1, uint16_t* dst_ptr,
                                              size_t num_items, YCoCgVariant variant)
    {
        detail::check_device(
            dxtlt_color565_recorrelate_ycocg_r_split(src_ptr_0, src_ptr_1, dst_ptr, num_items, static_cast<uint8_t>(variant)));
    }
};
inline void split_color_endpoints(const uint16_t* colors, uint16_t* colors_out, size_t colors_len_bytes)
{
    detail::check_device(dxtlt_split_565_color_endpoints(colors, colors_out, colors_len_bytes));
}
}  // namespace common

// util modules of the three crates (bc1_decode.rs:42, bc2_decode.rs:44, bc3_decode.rs:43) as array operations
namespace util {
struct Color8888 {
    uint8_t r, g, b, a;   // color_8888.rs:30
    bool operator==(const Color8888& o) const { return r == o.r && g == o.g && b == o.b && a == o.a; }
};
struct Decoded4x4Block {
    Color8888 pixels[16];   // row-major (decoded_4x4_block.rs:56)
    Color8888 get_pixel_unchecked(size_t x, size_t y) const { return pixels[y * 4 + x]; }
    bool has_identical_pixels() const   // decoded_4x4_block.rs:105
    {
        for (int i = 1; i < 16; ++i)
            if (!(pixels[i] == pixels[0]))
                return false;
        return true;
    }
};
static_assert(sizeof(Decoded4x4Block) == DXTLT_DECODED_BLOCK_BYTES, "Decoded4x4Block is sixteen 4-byte pixels");

// `num_blocks` blocks at `src` -> `dst[0 .. num_blocks)`
inline void decode_bc1_blocks(const uint8_t* src, Decoded4x4Block* dst, size_t num_blocks)
{
    detail::check_device(dxtlt_decode_bc1_blocks(src, num_blocks * 8, reinterpret_cast<uint8_t*>(dst), num_blocks * sizeof(Decoded4x4Block)));
}
inline void decode_bc2_blocks(const uint8_t* src, Decoded4x4Block* dst, size_t num_blocks)
{
    detail::check_device(dxtlt_decode_bc2_blocks(src, num_blocks * 16, reinterpret_cast<uint8_t*>(dst), num_blocks * sizeof(Decoded4x4Block)));
}
inline void decode_bc3_blocks(const uint8_t* src, Decoded4x4Block* dst, size_t num_blocks)
{
    detail::check_device(dxtlt_decode_bc3_blocks(src, num_blocks * 16, reinterpret_cast<uint8_t*>(dst), num_blocks * sizeof(Decoded4x4Block)));
}
// the one-block forms of the reference, for drop-in use (a PCIe round trip each: batch where you can)
inline Decoded4x4Block decode_bc1_block(const uint8_t* src) { Decoded4x4Block d; decode_bc1_blocks(src, &d, 1); return d; }
inline Decoded4x4Block decode_bc2_block(const uint8_t* src) { Decoded4x4Block d; decode_bc2_blocks(src, &d, 1); return d; }
inline Decoded4x4Block decode_bc3_block(const uint8_t* src) { Decoded4x4Block d; decode_bc3_blocks(src, &d, 1); return d; }

// blocks whose decoded pixels differ between two arrays; format = 1, 2, 3
inline uint64_t count_pixel_differences(int32_t format, const uint8_t* a, const uint8_t* b, size_t len)
{
    uint64_t n = 0;
    detail::check_device(dxtlt_count_pixel_differences(format, a, b, len, &n));
    return n;
}
}  // namespace util

}  // namespace core

// =====================================================================================================
// api crates (stable API): builders.  NOTE the different YCoCgVariant numbering.
// =====================================================================================================
namespace api {

enum class YCoCgVariant : uint8_t { Variant1 = 0, Variant2 = 1, Variant3 = 2, None = 3 };

inline core::YCoCgVariant to_internal_variant(YCoCgVariant v)
{
    switch (v) {
    case YCoCgVariant::Variant1: return core::YCoCgVariant::Variant1;
    case YCoCgVariant::Variant2: return core::YCoCgVariant::Variant2;
    case YCoCgVariant::Variant3: return core::YCoCgVariant::Variant3;
    default: return core::YCoCgVariant::None;
    }
}
inline YCoCgVariant from_internal_variant(core::YCoCgVariant v)
{
    switch (v) {
    case core::YCoCgVariant::Variant1: return YCoCgVariant::Variant1;
    case core::YCoCgVariant::Variant2: return YCoCgVariant::Variant2;
    case core::YCoCgVariant::Variant3: return YCoCgVariant::Variant3;
    default: return YCoCgVariant::None;
    }
}

struct Error {  // Bc1Error<E> / Bc2Error<E> (error.rs:11-35)
    enum Kind { Ok, InvalidLength, OutputBufferTooSmall, AllocationFailed, SizeEstimationFailed } kind = Ok;
    size_t length = 0, needed = 0, actual = 0;
    bool is_ok() const { return kind == Ok; }
    bool is_err() const { return kind != Ok; }
    static Error from(const core::ValidationError& v)
    {
        Error e;
        if (v.kind == core::ValidationError::InvalidLength) { e.kind = InvalidLength; e.length = v.length; }
        if (v.kind == core::ValidationError::OutputBufferTooSmall) { e.kind = OutputBufferTooSmall; e.needed = v.needed; e.actual = v.actual; }
        return e;
    }
};

#define DXTLT_MANUAL_BUILDER(N)                                                                                        \
    class Bc##N##ManualTransformBuilder {                                                                              \
    public:                                                                                                            \
        Bc##N##ManualTransformBuilder() = default; /* new(): Variant1 + split */                                       \
        Bc##N##ManualTransformBuilder decorrelation_mode(YCoCgVariant mode) const                                      \
        {                                                                                                              \
            Bc##N##ManualTransformBuilder b = *this;                                                                   \
            b.settings_.decorrelation_mode = to_internal_variant(mode);                                                \
            return b;                                                                                                  \
        }                                                                                                              \
        Bc##N##ManualTransformBuilder split_colour_endpoints(bool split) const                                         \
        {                                                                                                              \
            Bc##N##ManualTransformBuilder b = *this;                                                                   \
            b.settings_.split_colour_endpoints = split;                                                                \
            return b;                                                                                                  \
        }                                                                                                              \
        Error transform(const uint8_t* input, size_t input_len, uint8_t* output, size_t output_len) const              \
        {                                                                                                              \
            return Error::from(core::transform_bc##N##_with_settings_safe(input, input_len, output, output_len, settings_)); \
        }                                                                                                              \
        Error untransform(const uint8_t* input, size_t input_len, uint8_t* output, size_t output_len) const            \
        {                                                                                                              \
            return Error::from(core::untransform_bc##N##_with_settings_safe(input, input_len, output, output_len, settings_)); \
        }                                                                                                              \
        const core::Bc##N##TransformSettings& settings() const { return settings_; }                                   \
                                                                                                                       \
    private:                                                                                                           \
        core::Bc##N##TransformSettings settings_{};                                                                    \
    };                                                                                                                 \
                                                                                                                       \
    template <class Estimator>                                                                                         \
    class Bc##N##AutoTransformBuilder {                                                                                \
    public:                                                                                                            \
        explicit Bc##N##AutoTransformBuilder(Estimator e) : options_{std::move(e), false} {}                           \
        static Bc##N##AutoTransformBuilder new_ultra(Estimator e)                                                      \
        {                                                                                                              \
            Bc##N##AutoTransformBuilder b(std::move(e));                                                               \
            b.options_.use_all_decorrelation_modes = true;                                                             \
            return b;                                                                                                  \
        }                                                                                                              \
        Bc##N##AutoTransformBuilder& use_all_decorrelation_modes(bool use_all)                                         \
        {                                                                                                              \
            options_.use_all_decorrelation_modes = use_all;                                                            \
            return *this;                                                                                              \
        }                                                                                                              \
        /* Ok -> a manual builder configured with the chosen settings (auto_transform_builder.rs:123) */               \
        std::pair<Bc##N##ManualTransformBuilder, Error> transform(const uint8_t* input, size_t input_len, uint8_t* output, \
                                                                  size_t output_len)                                   \
        {                                                                                                              \
            Error e = Error::from(core::detail_safe::validate(input_len, output_len, N == 1 ? 8 : 16));                \
            if (e.is_err()) return {Bc##N##ManualTransformBuilder(), e};                                               \
            auto r = core::transform_bc##N##_auto(input, output, input_len, options_);                                 \
            if (r.second.kind == core::DetermineBestTransformError::SizeEstimationError) e.kind = Error::SizeEstimationFailed; \
            if (r.second.kind == core::DetermineBestTransformError::AllocateError) e.kind = Error::AllocationFailed;   \
            Bc##N##ManualTransformBuilder m = Bc##N##ManualTransformBuilder()                                          \
                                                  .decorrelation_mode(from_internal_variant(r.first.decorrelation_mode)) \
                                                  .split_colour_endpoints(r.first.split_colour_endpoints);             \
            return {m, e};                                                                                             \
        }                                                                                                              \
                                                                                                                       \
    private:                                                                                                           \
        core::EstimateSettings<Estimator> options_;                                                                    \
    };
DXTLT_MANUAL_BUILDER(1)
DXTLT_MANUAL_BUILDER(2)
#undef DXTLT_MANUAL_BUILDER

// ADDITIVE: the builder shape for this build's BC7 mode-split transform (docs/BC7_FORMAT.md; the reference has no BC7
// transform).  Version 0 of the format has no settings, hence no setters and no auto builder.
class Bc7ManualTransformBuilder {
public:
    Error transform(const uint8_t* input, size_t input_len, uint8_t* output, size_t output_len) const
    {
        return run(false, input, input_len, output, output_len);
    }
    Error untransform(const uint8_t* input, size_t input_len, uint8_t* output, size_t output_len) const
    {
        return run(true, input, input_len, output, output_len);
    }

private:
    static Error run(bool inverse, const uint8_t* input, size_t input_len, uint8_t* output, size_t output_len)
    {
        Error e = Error::from(core::detail_safe::validate(input_len, output_len, 16));
        if (e.is_err())
            return e;
        detail::check_device(inverse ? dxtlt_untransform_bc7(input, output, input_len) : dxtlt_transform_bc7(input, output, input_len));
        return e;
    }
};

}  // namespace api
}  // namespace dxt_lossless_transform
