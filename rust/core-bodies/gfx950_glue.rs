//! UNCOMPILED (see ../README.md).  `mod gfx950_glue;` of each core crate: what the swapped bodies share.
use core::ffi::{c_void, CStr};

use dxt_lossless_transform_api_common::estimate::SizeEstimationOperations;
use dxtlt_gfx950_sys::{dxtlt_last_error, DltSizeEstimator};

/// The reference's `transform_bcN_with_settings` cannot fail; a device can.  A failure must be loud -- never a silent
/// CPU fallback that would hide a broken deployment.
#[cold]
#[inline(never)]
pub(crate) fn abort_on_device_failure(what: &str, rc: i32) -> ! {
    let text = unsafe { CStr::from_ptr(dxtlt_last_error()) }.to_string_lossy();
    panic!("{what}: libdxtlt_gfx950 status {rc}: {text}");
}

/// `SizeEstimationOperations` behind the C vtable the library calls back through.  The estimator's error type is
/// generic, the callback's return value a `u32`: the first error is parked here and handed back to the caller.
pub(crate) struct EstimatorBridge<'a, T: SizeEstimationOperations> {
    pub estimator: &'a T,
    pub error: Option<T::Error>,
}

unsafe extern "C" fn max_compressed_size<T: SizeEstimationOperations>(
    context: *mut c_void, len_bytes: usize, out_size: *mut usize) -> u32 {
    let bridge = &mut *(context as *mut EstimatorBridge<T>);
    match bridge.estimator.max_compressed_size(len_bytes) {
        Ok(n) => { *out_size = n; 0 }
        Err(e) => { bridge.error.get_or_insert(e); 1 }
    }
}

unsafe extern "C" fn estimate_compressed_size<T: SizeEstimationOperations>(
    context: *mut c_void, input_ptr: *const u8, len_bytes: usize, output_ptr: *mut u8, output_len: usize,
    out_size: *mut usize) -> u32 {
    let bridge = &mut *(context as *mut EstimatorBridge<T>);
    match bridge.estimator.estimate_compressed_size(input_ptr, len_bytes, output_ptr, output_len) {
        Ok(n) => { *out_size = n; 0 }
        Err(e) => { bridge.error.get_or_insert(e); 1 }
    }
}

pub(crate) fn vtable<T: SizeEstimationOperations>(bridge: &mut EstimatorBridge<T>) -> DltSizeEstimator {
    DltSizeEstimator {
        context: bridge as *mut EstimatorBridge<T> as *mut c_void,
        max_compressed_size: max_compressed_size::<T>,
        estimate_compressed_size: estimate_compressed_size::<T>,
    }
}
