//! UNCOMPILED (see ../README.md).  New bodies for
//! core/dxt-lossless-transform-bc2/src/transform/transform_with_settings.rs (:30-73 and :93-138).
//! Doc comments and safety sections of the reference stay as they are; only the bodies change.
use crate::gfx950_glue::abort_on_device_failure;
use crate::{Bc2TransformSettings, Bc2UntransformSettings};
use dxtlt_gfx950_sys::{dxtlt_transform_bc2_with_settings, dxtlt_untransform_bc2_with_settings};

#[inline]
pub unsafe fn transform_bc2_with_settings(
    input_ptr: *const u8,
    output_ptr: *mut u8,
    len: usize,
    transform_options: Bc2TransformSettings,
) {
    debug_assert!(len.is_multiple_of(16));
    // YCoCgVariant is repr(u8) with the core numbering None = 0, Variant1..3 (common color_565/decorrelate.rs:72-84)
    let rc = dxtlt_transform_bc2_with_settings(
        input_ptr, output_ptr, len,
        transform_options.decorrelation_mode as u8,
        transform_options.split_colour_endpoints,
    );
    if rc != 0 {
        abort_on_device_failure("transform_bc2_with_settings", rc);
    }
}

#[inline]
pub unsafe fn untransform_bc2_with_settings(
    input_ptr: *const u8,
    output_ptr: *mut u8,
    len: usize,
    untransform_options: Bc2UntransformSettings,
) {
    debug_assert!(len.is_multiple_of(16));
    let rc = dxtlt_untransform_bc2_with_settings(
        input_ptr, output_ptr, len,
        untransform_options.decorrelation_mode as u8,
        untransform_options.split_colour_endpoints,
    );
    if rc != 0 {
        abort_on_device_failure("untransform_bc2_with_settings", rc);
    }
}
