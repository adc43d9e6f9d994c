//! UNCOMPILED (see ../README.md).  New bodies for
//! core/dxt-lossless-transform-bc1/src/transform/transform_with_settings.rs (:31-72 and :92-135).
//! Doc comments and safety sections of the reference stay as they are; only the bodies change.
use crate::gfx950_glue::abort_on_device_failure;
use crate::{Bc1TransformSettings, Bc1UntransformSettings};
use dxtlt_gfx950_sys::{dxtlt_transform_bc1_with_settings, dxtlt_untransform_bc1_with_settings};

#[inline]
pub unsafe fn transform_bc1_with_settings(
    input_ptr: *const u8,
    output_ptr: *mut u8,
    len: usize,
    transform_options: Bc1TransformSettings,
) {
    debug_assert!(len.is_multiple_of(8));
    // YCoCgVariant is repr(u8) with the core numbering None = 0, Variant1..3 (common color_565/decorrelate.rs:72-84)
    let rc = dxtlt_transform_bc1_with_settings(
        input_ptr, output_ptr, len,
        transform_options.decorrelation_mode as u8,
        transform_options.split_colour_endpoints,
    );
    if rc != 0 {
        abort_on_device_failure("transform_bc1_with_settings", rc);
    }
}

#[inline]
pub unsafe fn untransform_bc1_with_settings(
    input_ptr: *const u8,
    output_ptr: *mut u8,
    len: usize,
    untransform_options: Bc1UntransformSettings,
) {
    debug_assert!(len.is_multiple_of(8));
    let rc = dxtlt_untransform_bc1_with_settings(
        input_ptr, output_ptr, len,
        untransform_options.decorrelation_mode as u8,
        untransform_options.split_colour_endpoints,
    );
    if rc != 0 {
        abort_on_device_failure("untransform_bc1_with_settings", rc);
    }
}
