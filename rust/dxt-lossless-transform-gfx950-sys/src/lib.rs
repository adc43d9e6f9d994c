//! UNCOMPILED (see ../../README.md).  Raw bindings to `libdxtlt_gfx950.so` -- `include/dxtlt_gfx950.h`,
//! `include/dlt_size_estimator.h`.  Status codes are the `DXTLT_*` values of the header.
#![no_std]
#![allow(non_camel_case_types)]

use core::ffi::{c_char, c_void};

pub const DXTLT_OK: i32 = 0;
pub const DXTLT_E_INVALID_LENGTH: i32 = 1;
pub const DXTLT_E_INVALID_ARGUMENT: i32 = 2;
pub const DXTLT_E_NO_DEVICE: i32 = 3;
pub const DXTLT_E_DEVICE: i32 = 4;
pub const DXTLT_E_ESTIMATOR: i32 = 5;
pub const DXTLT_E_ALLOCATION: i32 = 6;

/// `DltSizeEstimator`, the C vtable of `SizeEstimationOperations`
/// (api-common `c_api/size_estimation.rs:18-52`; `include/dlt_size_estimator.h`).  Callbacks return 0 for success.
#[repr(C)]
pub struct DltSizeEstimator {
    pub context: *mut c_void,
    pub max_compressed_size: unsafe extern "C" fn(context: *mut c_void, len_bytes: usize, out_size: *mut usize) -> u32,
    pub estimate_compressed_size: unsafe extern "C" fn(
        context: *mut c_void,
        input_ptr: *const u8,
        len_bytes: usize,
        output_ptr: *mut u8,
        output_len: usize,
        out_size: *mut usize,
    ) -> u32,
}

/// One buffer of `dxtlt_transform_batch_host` / `dxtlt_transform_batch_device`.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct DxtltBatchItem {
    pub d_input: *const c_void,
    pub d_output: *mut c_void,
    pub len: u64,
    pub format: u8,  // 1, 2, 3 = BC1, BC2, BC3; 7 = BC7 (this build's own format, settings ignored)
    pub inverse: u8, // 0 = transform, 1 = untransform
    pub decorrelation_mode: u8,
    pub split_alpha_endpoints: u8,
    pub split_colour_endpoints: u8,
    pub reserved: [u8; 3],
}

extern "C" {
    // ---- host pointers: the bodies of the core crates' unsafe fns ------------------------------------------------
    pub fn dxtlt_transform_bc1_with_settings(input_ptr: *const u8, output_ptr: *mut u8, len: usize,
        decorrelation_mode: u8, split_colour_endpoints: bool) -> i32;
    pub fn dxtlt_untransform_bc1_with_settings(input_ptr: *const u8, output_ptr: *mut u8, len: usize,
        decorrelation_mode: u8, split_colour_endpoints: bool) -> i32;
    pub fn dxtlt_transform_bc2_with_settings(input_ptr: *const u8, output_ptr: *mut u8, len: usize,
        decorrelation_mode: u8, split_colour_endpoints: bool) -> i32;
    pub fn dxtlt_untransform_bc2_with_settings(input_ptr: *const u8, output_ptr: *mut u8, len: usize,
        decorrelation_mode: u8, split_colour_endpoints: bool) -> i32;
    pub fn dxtlt_transform_bc3_with_settings(input_ptr: *const u8, output_ptr: *mut u8, len: usize,
        decorrelation_mode: u8, split_alpha_endpoints: bool, split_colour_endpoints: bool) -> i32;
    pub fn dxtlt_untransform_bc3_with_settings(input_ptr: *const u8, output_ptr: *mut u8, len: usize,
        decorrelation_mode: u8, split_alpha_endpoints: bool, split_colour_endpoints: bool) -> i32;

    // ---- transform_bcN_auto ------------------------------------------------------------------------------------
    pub fn dxtlt_transform_bc1_auto(input_ptr: *const u8, output_ptr: *mut u8, len: usize,
        estimator: *const DltSizeEstimator, use_all_decorrelation_modes: bool,
        out_decorrelation_mode: *mut u8, out_split_colour_endpoints: *mut bool, out_estimator_error: *mut u32) -> i32;
    pub fn dxtlt_transform_bc2_auto(input_ptr: *const u8, output_ptr: *mut u8, len: usize,
        estimator: *const DltSizeEstimator, use_all_decorrelation_modes: bool,
        out_decorrelation_mode: *mut u8, out_split_colour_endpoints: *mut bool, out_estimator_error: *mut u32) -> i32;
    pub fn dxtlt_transform_bc3_auto(input_ptr: *const u8, output_ptr: *mut u8, len: usize,
        estimator: *const DltSizeEstimator, use_all_decorrelation_modes: bool,
        out_decorrelation_mode: *mut u8, out_split_alpha_endpoints: *mut bool, out_split_colour_endpoints: *mut bool,
        out_estimator_error: *mut u32) -> i32;

    /// opt-in: the estimator on `threads` host threads, every distinct section once (callbacks must be thread-safe)
    pub fn dxtlt_set_auto_estimator_threads(threads: i32);
    pub fn dxtlt_get_auto_estimator_threads() -> i32;
    /// cap for auto transforms called from THIS thread (0 = none); returns the previous cap
    pub fn dxtlt_set_auto_estimator_threads_for_this_thread(cap: i32) -> i32;

    // ---- data that already lives in HBM: device pointers, asynchronous on a HIP stream -----------------------------
    pub fn dxtlt_transform_bc1_with_settings_device(d_input: *const c_void, d_output: *mut c_void, len: usize,
        decorrelation_mode: u8, split_colour_endpoints: bool, hip_stream: *mut c_void) -> i32;
    pub fn dxtlt_untransform_bc1_with_settings_device(d_input: *const c_void, d_output: *mut c_void, len: usize,
        decorrelation_mode: u8, split_colour_endpoints: bool, hip_stream: *mut c_void) -> i32;
    pub fn dxtlt_transform_bc2_with_settings_device(d_input: *const c_void, d_output: *mut c_void, len: usize,
        decorrelation_mode: u8, split_colour_endpoints: bool, hip_stream: *mut c_void) -> i32;
    pub fn dxtlt_untransform_bc2_with_settings_device(d_input: *const c_void, d_output: *mut c_void, len: usize,
        decorrelation_mode: u8, split_colour_endpoints: bool, hip_stream: *mut c_void) -> i32;
    pub fn dxtlt_transform_bc3_with_settings_device(d_input: *const c_void, d_output: *mut c_void, len: usize,
        decorrelation_mode: u8, split_alpha_endpoints: bool, split_colour_endpoints: bool, hip_stream: *mut c_void) -> i32;
    pub fn dxtlt_untransform_bc3_with_settings_device(d_input: *const c_void, d_output: *mut c_void, len: usize,
        decorrelation_mode: u8, split_alpha_endpoints: bool, split_colour_endpoints: bool, hip_stream: *mut c_void) -> i32;
    pub fn dxtlt_transform_range_device(format: i32, inverse: bool, d_src: *const c_void, d_dst: *mut c_void,
        total_blocks: u64, first_block: u64, num_blocks: u64, decorrelation_mode: u8, split_alpha_endpoints: bool,
        split_colour_endpoints: bool, hip_stream: *mut c_void) -> i32;

    // ---- many buffers per call (the CLI's file-after-file pattern) --------------------------------------------------
    pub fn dxtlt_transform_batch_device(items: *const DxtltBatchItem, count: usize, hip_stream: *mut c_void) -> i32;
    pub fn dxtlt_transform_batch_host(items: *const DxtltBatchItem, count: usize) -> i32;

    // ---- one array over every GPU of the node ------------------------------------------------------------------------
    pub fn dxtlt_transform_sharded(format: i32, inverse: bool, input_ptr: *const u8, output_ptr: *mut u8, len: usize,
        decorrelation_mode: u8, split_alpha_endpoints: bool, split_colour_endpoints: bool, num_devices: i32) -> i32;

    /// What the last `dxtlt_transform_sharded` call on this thread did, shard by shard; returns the number of shards.
    pub fn dxtlt_sharded_last_stats(out: *mut DxtltShardStat, cap: i32) -> i32;

    // ---- NUMA placement of host threads that feed a device (the library's own shard workers use these themselves) ------
    pub fn dxtlt_pci_local_cpulist(pci_bdf: *const c_char, out: *mut c_char, cap: usize) -> i32;
    pub fn dxtlt_device_local_cpulist(device: i32, out: *mut c_char, cap: usize) -> i32;
    pub fn dxtlt_bind_thread_to_cpulist(cpulist: *const c_char) -> i32;

    /// The host-pointer crossover below which a caller with a CPU path of its own should use it (measured: 32 MiB;
    /// $DXTLT_HOST_ROUTE_THRESHOLD_BYTES overrides, 0 = everything to the device).
    pub fn dxtlt_host_route_threshold_bytes() -> usize;
    pub fn dxtlt_set_host_route_threshold_bytes(bytes: usize);

    pub fn dxtlt_last_error() -> *const c_char;
    pub fn dxtlt_device_count() -> i32;
    pub fn dxtlt_release_thread_resources();
}

/// `DxtltShardStat` of include/dxtlt_gfx950.h
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct DxtltShardStat {
    pub device: i32,
    pub cpus_bound: i32,
    pub first_block: u64,
    pub blocks: u64,
    pub seconds: f64,
}
