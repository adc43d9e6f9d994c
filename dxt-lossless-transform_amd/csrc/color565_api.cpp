// color565_api.cpp -- C ABI of the array-level RGB565 colour operations (include/dxtlt_color565.h); kernels in
// color565_ops.hip.  Host-pointer calls stage through the calling thread's device buffers.
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/dxtlt_color565.h"
#include "../../include/dxtlt_gfx950.h"
#include "bcn_launch.h"
#include "host_common.h"

using dxtlt_host::fail;
using dxtlt_host::kDevice;
using dxtlt_host::kInvalidArgument;
using dxtlt_host::kInvalidLength;
using dxtlt_host::kOk;

namespace {

#define HIP_TRY_C(expr, what)                   \
    do {                                        \
        hipError_t e_ = (expr);                 \
        if (e_ != hipSuccess)                   \
            return fail(kDevice, what, e_);     \
    } while (0)

int32_t ycocg_device(bool inverse, const void* src, void* dst, size_t n, uint8_t variant, void* stream)
{
    if (variant > 3)
        return fail(kInvalidArgument, "variant must be 0..3");
    if (n > 0 && (src == nullptr || dst == nullptr))
        return fail(kInvalidArgument, "NULL device buffer with num_items > 0");
    HIP_TRY_C(dxtlt::launch_color565_ycocg(inverse, src, dst, n, variant, static_cast<hipStream_t>(stream)), "kernel launch");
    return kOk;
}

int32_t ycocg_host(bool inverse, const uint16_t* src, uint16_t* dst, size_t n, uint8_t variant)
{
    if (variant > 3)
        return fail(kInvalidArgument, "variant must be 0..3");
    if (n == 0)
        return kOk;
    if (src == nullptr || dst == nullptr)
        return fail(kInvalidArgument, "NULL buffer with num_items > 0");
    if (variant == 0) {
        if (src != dst)
            std::memmove(dst, src, n * 2);
        return kOk;
    }
    void *d_a = nullptr, *d_b = nullptr;
    hipStream_t st = nullptr;
    if (int32_t rc = dxtlt_host::acquire_staging(n * 2, &d_a, &d_b, &st); rc != kOk)
        return rc;
    HIP_TRY_C(hipMemcpyAsync(d_a, src, n * 2, hipMemcpyHostToDevice, st), "H2D copy");
    HIP_TRY_C(dxtlt::launch_color565_ycocg(inverse, d_a, d_a, n, variant, st), "kernel launch");
    HIP_TRY_C(hipMemcpyAsync(dst, d_a, n * 2, hipMemcpyDeviceToHost, st), "D2H copy");
    HIP_TRY_C(hipStreamSynchronize(st), "stream synchronize");
    return kOk;
}

}  // namespace

extern "C" {

int32_t dxtlt_color565_decorrelate_ycocg_r(const uint16_t* s, uint16_t* d, size_t n, uint8_t v) { return ycocg_host(false, s, d, n, v); }
int32_t dxtlt_color565_recorrelate_ycocg_r(const uint16_t* s, uint16_t* d, size_t n, uint8_t v) { return ycocg_host(true, s, d, n, v); }
int32_t dxtlt_color565_decorrelate_ycocg_r_device(const void* s, void* d, size_t n, uint8_t v, void* st)
{
    return ycocg_device(false, s, d, n, v, st);
}
int32_t dxtlt_color565_recorrelate_ycocg_r_device(const void* s, void* d, size_t n, uint8_t v, void* st)
{
    return ycocg_device(true, s, d, n, v, st);
}

int32_t dxtlt_color565_recorrelate_ycocg_r_split_device(const void* s0, const void* s1, void* d, size_t n, uint8_t v, void* st)
{
    if (v > 3)
        return fail(kInvalidArgument, "variant must be 0..3");
    if (n % 2 != 0)
        return fail(kInvalidLength, "num_items must be even for split operations");
    if (n > 0 && (s0 == nullptr || s1 == nullptr || d == nullptr))
        return fail(kInvalidArgument, "NULL device buffer with num_items > 0");
    HIP_TRY_C(dxtlt::launch_color565_recorrelate_split(s0, s1, d, n, v, static_cast<hipStream_t>(st)), "kernel launch");
    return kOk;
}

int32_t dxtlt_color565_recorrelate_ycocg_r_split(const uint16_t* s0, const uint16_t* s1, uint16_t* d, size_t n, uint8_t v)
{
    if (v > 3)
        return fail(kInvalidArgument, "variant must be 0..3");
    if (n % 2 != 0)
        return fail(kInvalidLength, "num_items must be even for split operations");
    if (n == 0)
        return kOk;
    if (s0 == nullptr || s1 == nullptr || d == nullptr)
        return fail(kInvalidArgument, "NULL buffer with num_items > 0");
    const size_t half = n;   // bytes of each source half: (n / 2) colours * 2 bytes
    const size_t off1 = (half + 255) & ~(size_t)255;
    void *d_src = nullptr, *d_dst = nullptr;
    hipStream_t st = nullptr;
    if (int32_t rc = dxtlt_host::acquire_staging(off1 + half > 2 * n ? off1 + half : 2 * n, &d_src, &d_dst, &st); rc != kOk)
        return rc;
    uint8_t* p = static_cast<uint8_t*>(d_src);
    HIP_TRY_C(hipMemcpyAsync(p, s0, half, hipMemcpyHostToDevice, st), "H2D copy");
    HIP_TRY_C(hipMemcpyAsync(p + off1, s1, half, hipMemcpyHostToDevice, st), "H2D copy");
    HIP_TRY_C(dxtlt::launch_color565_recorrelate_split(p, p + off1, d_dst, n, v, st), "kernel launch");
    HIP_TRY_C(hipMemcpyAsync(d, d_dst, 2 * n, hipMemcpyDeviceToHost, st), "D2H copy");
    HIP_TRY_C(hipStreamSynchronize(st), "stream synchronize");
    return kOk;
}

int32_t dxtlt_split_565_color_endpoints_device(const void* c, void* o, size_t len, void* st)
{
    if (len % 4 != 0)
        return fail(kInvalidLength, "colors_len_bytes is not a multiple of 4");
    if (len > 0 && (c == nullptr || o == nullptr))
        return fail(kInvalidArgument, "NULL device buffer with colors_len_bytes > 0");
    HIP_TRY_C(dxtlt::launch_split_565_color_endpoints(c, o, len, static_cast<hipStream_t>(st)), "kernel launch");
    return kOk;
}

int32_t dxtlt_split_565_color_endpoints(const uint16_t* c, uint16_t* o, size_t len)
{
    if (len % 4 != 0)
        return fail(kInvalidLength, "colors_len_bytes is not a multiple of 4");
    if (len == 0)
        return kOk;
    if (c == nullptr || o == nullptr)
        return fail(kInvalidArgument, "NULL buffer with colors_len_bytes > 0");
    void *d_a = nullptr, *d_b = nullptr;
    hipStream_t st = nullptr;
    if (int32_t rc = dxtlt_host::acquire_staging(len, &d_a, &d_b, &st); rc != kOk)
        return rc;
    HIP_TRY_C(hipMemcpyAsync(d_a, c, len, hipMemcpyHostToDevice, st), "H2D copy");
    HIP_TRY_C(dxtlt::launch_split_565_color_endpoints(d_a, d_b, len, st), "kernel launch");
    HIP_TRY_C(hipMemcpyAsync(o, d_b, len, hipMemcpyDeviceToHost, st), "D2H copy");
    HIP_TRY_C(hipStreamSynchronize(st), "stream synchronize");
    return kOk;
}

}  // extern "C"
