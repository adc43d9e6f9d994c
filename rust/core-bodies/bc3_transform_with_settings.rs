//! UNCOMPILED (see ../README.md).  New bodies for
//! core/dxt-lossless-transform-bc3/src/transform/transform_with_settings.rs (:32-142 and :162-272).
use crate::gfx950_glue::abort_on_device_failure;
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
    let rc = dxtlt_transform_bc3_with_settings(
        input_ptr, output_ptr, len,
        transform_options.decorrelation_mode as u8,
        transform_options.split_alpha_endpoints,
        transform_options.split_colour_endpoints,
    );
    if rc != 0 {
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
    let rc = dxtlt_untransform_bc3_with_settings(
        input_ptr, output_ptr, len,
        untransform_options.decorrelation_mode as u8,
        untransform_options.split_alpha_endpoints,
        untransform_options.split_colour_endpoints,
    );
    if rc != 0 {
        abort_on_device_failure("untransform_bc3_with_settings", rc);
    }
}
