//! UNCOMPILED (see ../README.md).  New body for
//! core/dxt-lossless-transform-bc1/src/transform/transform_auto.rs (:200-270).
//! Candidate order, the estimated section, the strict `<` tie-break and the final transform with the winner are
//! implemented on the library side exactly as in the reference (csrc/auto_transform.cpp), so the same estimator
//! yields the same settings and the same bytes.
use crate::gfx950_glue::{abort_on_device_failure, vtable, EstimatorBridge, SerialEstimatorCalls};
#[cfg(feature = "cpu-without-device")]
use crate::gfx950_glue::device_is_absent;
#[cfg(feature = "cpu-below-threshold")]
use crate::gfx950_glue::stays_on_cpu;
use crate::transform::{Bc1EstimateSettings, DetermineBestTransformError};
use crate::Bc1TransformSettings;
use dxt_lossless_transform_api_common::estimate::SizeEstimationOperations;
use dxt_lossless_transform_common::allocate::AllocateError;
use dxt_lossless_transform_common::color_565::YCoCgVariant;
use dxtlt_gfx950_sys::{dxtlt_transform_bc1_auto, DXTLT_E_ALLOCATION, DXTLT_E_ESTIMATOR};

pub unsafe fn transform_bc1_auto<T>(
    input_ptr: *const u8,
    output_ptr: *mut u8,
    len: usize,
    transform_options: &Bc1EstimateSettings<T>,
) -> Result<Bc1TransformSettings, DetermineBestTransformError<T::Error>>
where
    T: SizeEstimationOperations,
{
    // OPT-IN (`cpu-below-threshold`, off by default): small inputs stay on the crate's own CPU path (the reference's body of this
    // function, renamed `transform_bc1_auto_cpu`, compiled only under the internal `cpu` feature; gfx950_glue.rs "size routing")
    #[cfg(feature = "cpu-below-threshold")]
    if stays_on_cpu(len) {
        return transform_bc1_auto_cpu(input_ptr, output_ptr, len, transform_options);
    }
    let bridge = EstimatorBridge::new(&transform_options.size_estimator);
    let table = vtable(&bridge);
    let _serial = SerialEstimatorCalls::new();   // `T` is not `Sync`: one callback at a time, on this thread (gfx950_glue.rs)
    let (mut mode, mut split_colour, mut estimator_error) = (0u8, false, 0u32);
    let rc = dxtlt_transform_bc1_auto(
        input_ptr, output_ptr, len, &table, transform_options.use_all_decorrelation_modes,
        &mut mode, &mut split_colour, &mut estimator_error,
    );
    match rc {
        0 => Ok(Bc1TransformSettings {
            decorrelation_mode: match mode { 1 => YCoCgVariant::Variant1, 2 => YCoCgVariant::Variant2,
                                             3 => YCoCgVariant::Variant3, _ => YCoCgVariant::None },
            split_colour_endpoints: split_colour,
        }),
        DXTLT_E_ESTIMATOR => Err(DetermineBestTransformError::SizeEstimationError(
            bridge.take_error().expect("the estimator callback failed, so it parked its error"))),
        DXTLT_E_ALLOCATION => Err(DetermineBestTransformError::AllocateError(AllocateError::default())),
        #[cfg(feature = "cpu-without-device")]
        other if device_is_absent(other) => transform_bc1_auto_cpu(input_ptr, output_ptr, len, transform_options),
        other => abort_on_device_failure("transform_bc1_auto", other),
    }
}
