//! UNCOMPILED (see ../README.md).  `mod gfx950_glue;` of each core crate: what the swapped bodies share.
//!
//! `no_std` like the crates it goes into (they make `std` an optional feature): nothing here needs an allocator or a
//! lock from `std` -- the estimator's first error is parked behind a spin lock built on `core::sync::atomic`.
use core::cell::UnsafeCell;
use core::ffi::{c_void, CStr};
#[cfg(feature = "cpu-below-threshold")]
use core::sync::atomic::AtomicUsize;
use core::sync::atomic::{AtomicBool, Ordering};

use dxt_lossless_transform_api_common::estimate::SizeEstimationOperations;
#[cfg(feature = "cpu-below-threshold")]
use dxtlt_gfx950_sys::dxtlt_host_route_threshold_bytes;
use dxtlt_gfx950_sys::{dxtlt_last_error, DltSizeEstimator};

/// The reference's `transform_bcN_with_settings` cannot fail; a device can.  A failure must be loud -- never a silent
/// detour that would hide a broken deployment.  That includes a machine WITHOUT a device, unless the crate was built with the
/// opt-in feature `cpu-without-device` (see `device_is_absent`).
#[cold]
#[inline(never)]
pub(crate) fn abort_on_device_failure(what: &str, rc: i32) -> ! {
    let text = unsafe { CStr::from_ptr(dxtlt_last_error()) }.to_str().unwrap_or("(no text)");
    panic!("{what}: libdxtlt_gfx950 status {rc}: {text}");
}

// ---- opt-in detours to the crate's own CPU path (both OFF by default) -----------------------------------------------------
// With the default features every call reaches the device and a missing device is a panic: the shipped integration is the
// tested path.  What follows exists only under the opt-in features `cpu-below-threshold` / `cpu-without-device`.
//
// Why a maintainer might opt in: a host-pointer call into the library is a PCIe round trip -- at least ~17 us and at most
// 25-43 GiB/s, where ONE core of this crate's own SIMD path moves 20-50 GiB/s out of cache.  The reference's call pattern is
// one call per file from rayon workers (tools/dxt-lossless-transform-cli/src/commands/transform/mod.rs:154-199) on textures that
// average 4 MiB, its test assets are 32-64 KiB (64 KiB: 2.9 us against 20 us).  The crossover is the library's to know
// (`dxtlt_host_route_threshold_bytes()`: measured 32 MiB, overridable by $DXTLT_HOST_ROUTE_THRESHOLD_BYTES); it is read once.
// The device route for such a corpus is ONE batch call (`dxtlt_transform_batch_host`), not a CPU detour.

/// 0 = not asked yet; otherwise threshold + 1.
#[cfg(feature = "cpu-below-threshold")]
static ROUTE_THRESHOLD_PLUS_ONE: AtomicUsize = AtomicUsize::new(0);

/// `true`: this call is small enough that the crate's own CPU dispatch is the faster path.  Exists only under the opt-in
/// feature `cpu-below-threshold`.
#[cfg(feature = "cpu-below-threshold")]
#[inline]
pub(crate) fn stays_on_cpu(len: usize) -> bool {
    let mut t = ROUTE_THRESHOLD_PLUS_ONE.load(Ordering::Relaxed);
    if t == 0 {
        t = unsafe { dxtlt_host_route_threshold_bytes() }.saturating_add(1).max(1);
        ROUTE_THRESHOLD_PLUS_ONE.store(t, Ordering::Relaxed);
    }
    len < t - 1
}

/// `true`: the library found no HIP device.  Exists only under the opt-in feature `cpu-without-device`, with which the same
/// binary still works on a machine without a GPU; without it `DXTLT_E_NO_DEVICE` panics like every other status.
#[cfg(feature = "cpu-without-device")]
#[inline]
pub(crate) fn device_is_absent(rc: i32) -> bool {
    rc == dxtlt_gfx950_sys::DXTLT_E_NO_DEVICE
}

// ---- transform_bcN_auto: SizeEstimationOperations behind the C vtable ---------------------------------------------------

/// The estimator's error type is generic, the callback's return value a `u32`: the first error is parked here and handed
/// back to the caller.
///
/// The bounds are the reference's own (`T: SizeEstimationOperations`, nothing more -- a `T: Sync` bound would not compile for
/// the reference's signatures), so nothing here may let `T`'s methods run concurrently.  The library runs the callbacks on
/// several threads only when somebody called the process-wide `dxtlt_set_auto_estimator_threads(n > 1)`; this glue never does,
/// and it does not rely on nobody else having done so either: every auto body holds a `SerialEstimatorCalls` guard around its
/// FFI call, which caps the CALLING thread's auto transforms at one estimator thread
/// (`dxtlt_set_auto_estimator_threads_for_this_thread(1)`) and restores the previous cap on drop.  With it the callbacks run
/// one at a time on the calling thread: `&T` never crosses a thread, whatever the rest of the process configured.
pub(crate) struct EstimatorBridge<'a, T: SizeEstimationOperations> {
    pub estimator: &'a T,
    locked: AtomicBool,
    error: UnsafeCell<Option<T::Error>>,
}

impl<'a, T: SizeEstimationOperations> EstimatorBridge<'a, T> {
    pub(crate) fn new(estimator: &'a T) -> Self {
        Self { estimator, locked: AtomicBool::new(false), error: UnsafeCell::new(None) }
    }
    fn with_slot<R>(&self, f: impl FnOnce(&mut Option<T::Error>) -> R) -> R {
        while self.locked.compare_exchange_weak(false, true, Ordering::Acquire, Ordering::Relaxed).is_err() {
            core::hint::spin_loop();
        }
        // SAFETY: the flag above makes this the only reference to the slot until it is cleared again
        let r = f(unsafe { &mut *self.error.get() });
        self.locked.store(false, Ordering::Release);
        r
    }
    fn park(&self, e: T::Error) {
        self.with_slot(|slot| {
            slot.get_or_insert(e);
        });
    }
    /// The first error a callback reported, if any (call after the library has returned).
    pub(crate) fn take_error(&self) -> Option<T::Error> {
        self.with_slot(|slot| slot.take())
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

/// While alive: auto transforms called from this thread run their estimator callbacks one at a time on this thread (see
/// `EstimatorBridge`).  Restores the thread's previous cap on drop, so nesting is harmless.
pub(crate) struct SerialEstimatorCalls {
    previous_cap: i32,
}

impl SerialEstimatorCalls {
    pub(crate) fn new() -> Self {
        Self { previous_cap: unsafe { dxtlt_gfx950_sys::dxtlt_set_auto_estimator_threads_for_this_thread(1) } }
    }
}

impl Drop for SerialEstimatorCalls {
    fn drop(&mut self) {
        unsafe { dxtlt_gfx950_sys::dxtlt_set_auto_estimator_threads_for_this_thread(self.previous_cap) };
    }
}

pub(crate) fn vtable<T: SizeEstimationOperations>(bridge: &EstimatorBridge<T>) -> DltSizeEstimator {
    DltSizeEstimator {
        context: bridge as *const EstimatorBridge<T> as *mut c_void,
        max_compressed_size: max_compressed_size::<T>,
        estimate_compressed_size: estimate_compressed_size::<T>,
    }
}
