// host_common.h -- internal interface between the C ABI translation units (dxtlt_api.cpp, auto_transform.cpp,
// c_api_core.cpp, c_api_stable.cpp).  Not installed.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/dlt_size_estimator.h"

namespace dxtlt_host {

// status codes == DXTLT_* in include/dxtlt_gfx950.h
enum : int32_t {
    kOk = 0,
    kInvalidLength = 1,
    kInvalidArgument = 2,
    kNoDevice = 3,
    kDevice = 4,
    kEstimator = 5,
    kAllocation = 6,
};

// Records the failure text for dxtlt_last_error() on this thread and returns `code`.
int32_t fail(int32_t code, const char* what, hipError_t e = hipSuccess);

// Host pointers in/out, whole buffer, synchronous (H2D + kernel + D2H on the current device).
// `normalize` (BC1 forward only): ColorNormalizationMode fused into the transform, 0 = none.
int32_t transform(int32_t format, bool inverse, const uint8_t* in, uint8_t* out, size_t len, uint8_t mode,
                  bool split_alpha, bool split_colour, uint8_t normalize = 0);

// This thread's staging context on the current device: two device buffers of at least `bytes` and a stream.
int32_t acquire_staging(size_t bytes, void** d_in, void** d_out, hipStream_t* stream);

// Small host buffers (up to 1 MiB, DXTLT_MAPPED_MAX_BYTES): the calling thread's pair of MAPPED pinned staging buffers
// (h_*: host addresses, d_*: the same memory as the device sees it) and its stream; usable == false above the limit.
// The caller copies its input to h_in, launches on d_in -> d_out, waits for the stream and copies h_out out.
struct MappedStaging {
    bool usable;
    void* h_in;
    void* h_out;
    void* d_in;
    void* d_out;
    hipStream_t stream;
};
int32_t acquire_mapped_staging(size_t bytes, MappedStaging* out);

// Large BC7 host buffers: the main part (whole 1024-block granules) through the chunked upload | kernel | download
// pipeline of the BC1-3 host path.  Returns false when the buffer is below the pipeline's threshold (nothing done).
bool pipelined_bc7_main(bool inverse, const uint8_t* in, uint8_t* out, uint64_t main_blocks, int32_t* rc);

// Per-device shard contexts of the sharded entry points (a stream and two device buffers of at least `bytes`), kept
// across calls; the calling thread has made `dev` current.  release_shard_buffers hands the context back (the stream
// must be drained).
struct ShardBuffers {
    hipStream_t stream;
    void* a;
    void* b;
    void* handle;
};
int32_t acquire_shard_buffers(int dev, size_t bytes, ShardBuffers* out);
void release_shard_buffers(const ShardBuffers& sb);
// A sharded call, before it starts its workers: the HIP runtime brought up for devices [0, devices) on the CALLER's thread
// (workers narrow their affinity before their first HIP call; runtime threads started lazily from one would inherit the mask).
// After its workers have joined: idle contexts beyond the per-device cap are freed (never on a shard's completion path:
// hipFree synchronises the device).
void init_runtime_for_devices(int devices);
void trim_idle_shard_buffers();
// Blocks [first, first + count) of the main part (total_main blocks, whole granules) of a BC7 host array through the
// chunked pipeline on the shard's buffers; false = below the pipeline's threshold (nothing done).
bool pipelined_bc7_shard(const ShardBuffers& sb, int dev, bool inverse, const uint8_t* in, uint8_t* out, uint64_t total_main,
                         uint64_t first, uint64_t count, int32_t* rc);

// Binds the calling thread -- one this library created for `device` -- to the CPUs local to the device (numa_affinity.cpp).
// Returns the number of CPUs bound to, 0 when nothing was changed.
int bind_this_thread_near_device(int device);

// Enqueue one whole-buffer transform on device pointers.
int32_t enqueue(int32_t format, bool inverse, const void* d_src, void* d_dst, uint64_t blocks, uint8_t mode,
                bool split_alpha, bool split_colour, hipStream_t stream, uint8_t normalize = 0);

// Per-thread device resources of the other translation units, freed by dxtlt_release_thread_resources().
void release_bc7_thread_scratch();      // bc7_api.cpp
void release_normalize_thread_flag();   // normalize_api.cpp
void release_batch_thread_tables();     // batch_api.cpp
void release_auto_thread_arena();       // auto_transform.cpp

struct AutoChoice {
    uint8_t mode;  // core numbering
    bool split_alpha;
    bool split_colour;
    uint32_t estimator_error;  // the callback's non-zero return when the status is kEstimator
};

// transform_bcN_auto on host pointers (see auto_transform.cpp).
int32_t transform_auto(int32_t format, const uint8_t* in, uint8_t* out, size_t len, const DltSizeEstimator* estimator,
                       bool use_all_decorrelation_modes, AutoChoice* choice);

}  // namespace dxtlt_host
