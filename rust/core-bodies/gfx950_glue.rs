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
///
/// The library may call `estimate_compressed_size` from several threads at once with this one context
/// (`dxtlt_set_auto_estimator_threads(n > 1)`, opt-in), so the callbacks only ever take a SHARED reference and the error
/// slot is a `Mutex`: no aliased `&mut`, no data race.  (`T: Sync` is what makes the concurrent calls into the estimator
/// itself sound; with an estimator that is not `Sync` leave the switch at 1 -- `vtable` asks for the bound.)
pub(crate) struct EstimatorBridge<'a, T: SizeEstimationOperations> {
    pub estimator: &'a T,
    pub error: std::sync::Mutex<Option<T::Error>>,
}

impl<'a, T: SizeEstimationOperations> EstimatorBridge<'a, T> {
    pub(crate) fn new(estimator: &'a T) -> Self {
        Self { estimator, error: std::sync::Mutex::new(None) }
    }
    fn park(&self, e: T::Error) {
        let mut slot = self.error.lock().unwrap_or_else(|p| p.into_inner());
        slot.get_or_insert(e);
    }
    /// The first error a callback reported, if any (call after the library has returned).
    pub(crate) fn take_error(&self) -> Option<T::Error> {
        self.error.lock().unwrap_or_else(|p| p.into_inner()).take()
    }
}

unsafe extern "C" fn max_compressed_size<T: SizeEstimationOperations>(
    context: *mut c_void, len_bytes: usize, out_size: *mut usize) -> u32 {
    let bridge = &*(context as *const EstimatorBridge<T>);
    match bridge.estimator.max_compressed_size(len_bytes) {
        Ok(n) => { *out_size = n; 0 }
        Err(e) => { bridge.park(e); 1 }
    }
}

unsafe extern "C" fn estimate_compressed_size<T: SizeEstimationOperations>(
    context: *mut c_void, input_ptr: *const u8, len_bytes: usize, output_ptr: *mut u8, output_len: usize,
    out_size: *mut usize) -> u32 {
    let bridge = &*(context as *const EstimatorBridge<T>);
    match bridge.estimator.estimate_compressed_size(input_ptr, len_bytes, output_ptr, output_len) {
        Ok(n) => { *out_size = n; 0 }
        Err(e) => { bridge.park(e); 1 }
    }
}

pub(crate) fn vtable<T: SizeEstimationOperations + Sync>(bridge: &EstimatorBridge<T>) -> DltSizeEstimator
where
    T::Error: Send,
{
    DltSizeEstimator {
        context: bridge as *const EstimatorBridge<T> as *mut c_void,
        max_compressed_size: max_compressed_size::<T>,
        estimate_compressed_size: estimate_compressed_size::<T>,
    }
}
