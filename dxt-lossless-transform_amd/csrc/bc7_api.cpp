// bc7_api.cpp -- C ABI of the BC7 granule-sorted field split, version 2 (include/dxtlt_bc7.h, docs/BC7_FORMAT.md).
// A format of this build's own: the reference has no BC7 transform; parity unpinned.
#include "../../include/dxtlt_bc7.h"

#include <hip/hip_runtime_api.h>

#include <cstring>

#include "bc7_launch.h"
#include "host_common.h"

namespace {

int32_t host_call(bool inverse, const uint8_t* in, uint8_t* out, size_t len)
{
    using namespace dxtlt_host;
    if (len % 16 != 0)
        return fail(kInvalidLength, "len is not a multiple of 16 (BC7 block size)");
    if (len == 0)
        return kOk;
    if (in == nullptr || out == nullptr)
        return fail(kInvalidArgument, "NULL buffer with len > 0");
    // large buffers: the main part in chunks (upload, kernel and the eight per-stream downloads of consecutive chunks
    // overlap), then the tail part -- a BC7 buffer of its own -- through the one-shot path below
    const uint64_t blocks = len / 16, main_blocks = blocks - blocks % 1024;
    int32_t prc = kOk;
    if (pipelined_bc7_main(inverse, in, out, main_blocks, &prc)) {
        if (prc != kOk || main_blocks == blocks)
            return prc;
        return host_call(inverse, in + main_blocks * 16, out + main_blocks * 16, (size_t)((blocks - main_blocks) * 16));
    }
    // small buffers: the kernel reads and writes mapped pinned staging itself (no copy-engine hand-overs)
    MappedStaging m;
    int32_t rc = acquire_mapped_staging(len, &m);
    if (rc != kOk)
        return rc;
    if (m.usable) {
        std::memcpy(m.h_in, in, len);
        hipError_t e = dxtlt::bc7::launch(inverse, m.d_in, m.d_out, len / 16, m.stream);
        const hipError_t drained = hipStreamSynchronize(m.stream);
        if (e == hipSuccess)
            e = drained;
        if (e != hipSuccess)
            return fail(kDevice, "BC7 transform", e);
        std::memcpy(out, m.h_out, len);
        return kOk;
    }
    void *d_in = nullptr, *d_out = nullptr;
    hipStream_t st = nullptr;
    rc = acquire_staging(len, &d_in, &d_out, &st);
    if (rc != kOk)
        return rc;
    hipError_t e = hipMemcpyAsync(d_in, in, len, hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = dxtlt::bc7::launch(inverse, d_in, d_out, len / 16, st);
    if (e == hipSuccess)
        e = hipMemcpyAsync(out, d_out, len, hipMemcpyDeviceToHost, st);
    // drained on every exit: the staging buffers belong to this thread's next call
    const hipError_t drained = hipStreamSynchronize(st);
    if (e == hipSuccess)
        e = drained;
    if (e != hipSuccess)
        return fail(kDevice, "BC7 transform", e);
    return kOk;
}

int32_t device_range(bool inverse, const void* d_src, void* d_dst, uint64_t total, uint64_t first, uint64_t num, void* stream)
{
    using namespace dxtlt_host;
    if (num == 0)
        return kOk;
    if (d_src == nullptr || d_dst == nullptr)
        return fail(kInvalidArgument, "NULL device buffer");
    hipError_t e = dxtlt::bc7::launch_range(inverse, d_src, d_dst, total, first, num, (hipStream_t)stream);
    if (e == hipErrorInvalidValue)
        return fail(kInvalidArgument, "BC7: a range starts on a sort granule (1024 blocks) and ends on one or at the end of the "
                                      "array");
    if (e != hipSuccess)
        return fail(kDevice, "BC7 kernel launch", e);
    return kOk;
}

int32_t device_call(bool inverse, const void* d_in, void* d_out, size_t len, void* stream)
{
    using namespace dxtlt_host;
    if (len % 16 != 0)
        return fail(kInvalidLength, "len is not a multiple of 16 (BC7 block size)");
    return device_range(inverse, d_in, d_out, len / 16, 0, len / 16, stream);
}

}  // namespace

void dxtlt_host::release_bc7_thread_scratch() {}   // since version 1: no per-thread device scratch

extern "C" {

int32_t dxtlt_transform_bc7(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len)
{
    return host_call(false, input_ptr, output_ptr, len);
}
int32_t dxtlt_untransform_bc7(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len)
{
    return host_call(true, input_ptr, output_ptr, len);
}
size_t dxtlt_bc7_workspace_bytes(size_t len)
{
    (void)len;
    return 0;   // since version 1: a single pass with no device scratch
}
int32_t dxtlt_transform_bc7_device(const void* d_input, void* d_output, size_t len, void* d_workspace,
                                   size_t workspace_bytes, void* hip_stream)
{
    (void)d_workspace;
    (void)workspace_bytes;
    return device_call(false, d_input, d_output, len, hip_stream);
}
int32_t dxtlt_untransform_bc7_device(const void* d_input, void* d_output, size_t len, void* d_workspace,
                                     size_t workspace_bytes, void* hip_stream)
{
    (void)d_workspace;
    (void)workspace_bytes;
    return device_call(true, d_input, d_output, len, hip_stream);
}
int32_t dxtlt_transform_bc7_range_device(bool inverse, const void* d_src, void* d_dst, uint64_t total_blocks,
                                         uint64_t first_block, uint64_t num_blocks, void* hip_stream)
{
    if (first_block > total_blocks || num_blocks > total_blocks - first_block)
        return dxtlt_host::fail(dxtlt_host::kInvalidArgument, "block range exceeds total_blocks");
    return device_range(inverse, d_src, d_dst, total_blocks, first_block, num_blocks, hip_stream);
}

}  // extern "C"
