// bc7_api.cpp -- C ABI of the BC7 mode-split transform, version 0 (include/dxtlt_bc7.h, docs/BC7_FORMAT.md).
// A format of this build's own: the reference has no BC7 transform; parity unpinned.
#include "../../include/dxtlt_bc7.h"

#include <hip/hip_runtime_api.h>

#include "bc7_launch.h"
#include "host_common.h"

namespace {

// per-thread device scratch for the host-pointer entry points (grow-only)
struct Bc7Scratch {
    void* ptr = nullptr;
    size_t cap = 0;
    int device = -1;
    ~Bc7Scratch()
    {
        if (ptr) (void)hipFree(ptr);
    }
};
thread_local Bc7Scratch g_scratch;

int32_t host_call(bool inverse, const uint8_t* in, uint8_t* out, size_t len)
{
    using namespace dxtlt_host;
    if (len % 16 != 0)
        return fail(kInvalidLength, "len is not a multiple of 16 (BC7 block size)");
    if (len == 0)
        return kOk;
    if (in == nullptr || out == nullptr)
        return fail(kInvalidArgument, "NULL buffer with len > 0");
    void *d_in = nullptr, *d_out = nullptr;
    hipStream_t st = nullptr;
    int32_t rc = acquire_staging(len, &d_in, &d_out, &st);
    if (rc != kOk)
        return rc;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t need = dxtlt::bc7::workspace_bytes(len / 16);
    if (g_scratch.device != dev || g_scratch.cap < need) {
        if (g_scratch.ptr) (void)hipFree(g_scratch.ptr);
        g_scratch.ptr = nullptr;
        g_scratch.cap = 0;
        hipError_t e = hipMalloc(&g_scratch.ptr, need);
        if (e != hipSuccess)
            return fail(kDevice, "hipMalloc(BC7 workspace)", e);
        g_scratch.cap = need;
        g_scratch.device = dev;
    }
    hipError_t e = hipMemcpyAsync(d_in, in, len, hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = dxtlt::bc7::launch(inverse, d_in, d_out, len / 16, g_scratch.ptr, g_scratch.cap, st);
    if (e == hipSuccess)
        e = hipMemcpyAsync(out, d_out, len, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess)
        e = hipStreamSynchronize(st);
    if (e != hipSuccess)
        return fail(kDevice, "BC7 transform", e);
    return kOk;
}

int32_t device_call(bool inverse, const void* d_in, void* d_out, size_t len, void* ws, size_t ws_bytes, void* stream)
{
    using namespace dxtlt_host;
    if (len % 16 != 0)
        return fail(kInvalidLength, "len is not a multiple of 16 (BC7 block size)");
    if (len == 0)
        return kOk;
    if (d_in == nullptr || d_out == nullptr || ws == nullptr)
        return fail(kInvalidArgument, "NULL device buffer / workspace");
    if (ws_bytes < dxtlt::bc7::workspace_bytes(len / 16))
        return fail(kInvalidArgument, "workspace smaller than dxtlt_bc7_workspace_bytes(len)");
    hipError_t e = dxtlt::bc7::launch(inverse, d_in, d_out, len / 16, ws, ws_bytes, (hipStream_t)stream);
    if (e == hipErrorInvalidValue)
        return fail(kInvalidArgument, "BC7 v0 needs 16-byte aligned device buffers");
    if (e != hipSuccess)
        return fail(kDevice, "BC7 kernel launch", e);
    return kOk;
}

}  // namespace

void dxtlt_host::release_bc7_thread_scratch()
{
    if (g_scratch.ptr) (void)hipFree(g_scratch.ptr);
    g_scratch.ptr = nullptr;
    g_scratch.cap = 0;
    g_scratch.device = -1;
}

extern "C" {

int32_t dxtlt_transform_bc7(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len)
{
    return host_call(false, input_ptr, output_ptr, len);
}
int32_t dxtlt_untransform_bc7(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len)
{
    return host_call(true, input_ptr, output_ptr, len);
}
size_t dxtlt_bc7_workspace_bytes(size_t len) { return dxtlt::bc7::workspace_bytes(len / 16); }
int32_t dxtlt_transform_bc7_device(const void* d_input, void* d_output, size_t len, void* d_workspace,
                                   size_t workspace_bytes, void* hip_stream)
{
    return device_call(false, d_input, d_output, len, d_workspace, workspace_bytes, hip_stream);
}
int32_t dxtlt_untransform_bc7_device(const void* d_input, void* d_output, size_t len, void* d_workspace,
                                     size_t workspace_bytes, void* hip_stream)
{
    return device_call(true, d_input, d_output, len, d_workspace, workspace_bytes, hip_stream);
}

}  // extern "C"
