//! UNCOMPILED (see ../README.md).  New bodies for
//! core/dxt-lossless-transform-bc3/src/transform/transform_with_settings.rs (:32-142 and :162-272).
//! Doc comments and safety sections of the reference stay as they are.
//!
//! DEFAULT FEATURES: every call goes to the device, whatever its size, and a machine without a HIP device panics with the
//! library's `DXTLT_E_NO_DEVICE` text -- the shipped integration is the path this repository tests (the C ABI of
//! libdxtlt_gfx950.so, which has no CPU implementation and is never asked for one).  Corpora of small files are served by ONE
//! `dxtlt_transform_batch_host` / `dxtlt_transform_batch_device` call (sections 2 and 7), not by this per-file
//! entry point.
//!
//! Two OPT-IN cargo features keep the reference's own bodies in the crate -- renamed `transform_bc3_with_settings_cpu` /
//! `untransform_bc3_with_settings_cpu`, unchanged, compiled only under the internal feature `cpu` that both imply:
//!   * `cpu-below-threshold`: a call below `dxtlt_host_route_threshold_bytes()` (measured crossover 32 MiB: a PCIe round trip
//!     costs a small texture 2.5-7 x what one CPU core does) stays on the crate's own dispatch;
//!   * `cpu-without-device`: `DXTLT_E_NO_DEVICE`, and only that status, takes the crate's own body.
//! Calls that take either detour run reference-built code this repository neither contains nor tests.
use crate::gfx950_glue::abort_on_device_failure;
#[cfg(feature = "cpu-without-device")]
use crate::gfx950_glue::device_is_absent;
#[cfg(feature = "cpu-below-threshold")]
use crate::gfx950_glue::stays_on_cpu;
use crate::{Bc3TransformSettings, Bc3UntransformSettings};
use dxtlt_gfx950_sys::{dxtlt_transform_bc3_with_settings, dxtlt_untransform_bc3_with_settings};

#[inline]
pub unsafe fn transform_bc3_with_settings(
    input_ptr: *const u8,
    output_ptr: *mut u8,
    len: usize,
    transform_options: Bc3TransformSettings,
) {
    debug_assert!(len.is_multiple_of(16));
    #[cfg(feature = "cpu-below-threshold")]
    if stays_on_cpu(len) {
        return transform_bc3_with_settings_cpu(input_ptr, output_ptr, len, transform_options);
    }
    // YCoCgVariant is repr(u8) with the core numbering None = 0, Variant1..3 (common color_565/decorrelate.rs:72-84)
    let rc = dxtlt_transform_bc3_with_settings(
        input_ptr, output_ptr, len,
        transform_options.decorrelation_mode as u8,
        transform_options.split_alpha_endpoints,
        transform_options.split_colour_endpoints,
    );
    if rc != 0 {
        #[cfg(feature = "cpu-without-device")]
        if device_is_absent(rc) {
            return transform_bc3_with_settings_cpu(input_ptr, output_ptr, len, transform_options);
        }
        abort_on_device_failure("transform_bc3_with_settings", rc);
    }
}

#[inline]
pub unsafe fn untransform_bc3_with_settings(
    input_ptr: *const u8,
    output_ptr: *mut u8,
    len: usize,
    untransform_options: Bc3UntransformSettings,
) {
    debug_assert!(len.is_multiple_of(16));
    #[cfg(feature = "cpu-below-threshold")]
    if stays_on_cpu(len) {
        return untransform_bc3_with_settings_cpu(input_ptr, output_ptr, len, untransform_options);
    }
    let rc = dxtlt_untransform_bc3_with_settings(
        input_ptr, output_ptr, len,
        untransform_options.decorrelation_mode as u8,
        untransform_options.split_alpha_endpoints,
        untransform_options.split_colour_endpoints,
    );
    if rc != 0 {
        #[cfg(feature = "cpu-without-device")]
        if device_is_absent(rc) {
            return untransform_bc3_with_settings_cpu(input_ptr, output_ptr, len, untransform_options);
        }
        abort_on_device_failure("untransform_bc3_with_settings", rc);
    }
}

// ---- the reference's own bodies, verbatim, under their new names -----------------------------------------------------------
// (not reproduced in this repository: they are the lines :32-142 and :162-272 of the file this one replaces, with
//  `pub unsafe fn transform_bc3_with_settings` -> `#[cfg(feature = "cpu")] unsafe fn transform_bc3_with_settings_cpu` and the
//  same for the inverse; nothing inside them changes, so the crate's SIMD ladders, tests and benches keep working on them)
