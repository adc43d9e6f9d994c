// decode_api.cpp -- C ABI of the block decoders (include/dxtlt_decode.h); kernels in bcn_decode.hip.  Host-pointer
// calls go through the calling thread's staging buffers in slices, so that the 64-byte-per-block output never needs
// more than 2 x 256 MiB of device memory.
#include <hip/hip_runtime.h>

#include "../../include/dxtlt_decode.h"
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

constexpr size_t kSliceBlocks = size_t(4) << 20;   // 256 MiB of pixels per slice

inline size_t block_bytes(int fmt) { return fmt == 1 ? 8 : 16; }

int32_t check_decode(int fmt, const void* in, size_t len, const void* out, size_t out_len)
{
    if (len % block_bytes(fmt) != 0)
        return fail(kInvalidLength, "len is not a multiple of the block size");
    const size_t n = len / block_bytes(fmt);
    if (n > 0 && (in == nullptr || out == nullptr))
        return fail(kInvalidArgument, "NULL buffer with len > 0");
    if (out_len / DXTLT_DECODED_BLOCK_BYTES < n)
        return fail(kInvalidArgument, "pixels_len is smaller than 64 bytes per block");
    return kOk;
}

int32_t decode_device(int fmt, const void* in, size_t len, void* out, size_t out_len, void* stream)
{
    if (int32_t rc = check_decode(fmt, in, len, out, out_len); rc != kOk)
        return rc;
    HIP_TRY_C(dxtlt::launch_decode_blocks(fmt, in, out, len / block_bytes(fmt), static_cast<hipStream_t>(stream)), "kernel launch");
    return kOk;
}

int32_t decode_host(int fmt, const uint8_t* in, size_t len, uint8_t* out, size_t out_len)
{
    if (int32_t rc = check_decode(fmt, in, len, out, out_len); rc != kOk)
        return rc;
    const size_t bs = block_bytes(fmt), n = len / bs;
    if (n == 0)
        return kOk;
    const size_t slice = n < kSliceBlocks ? n : kSliceBlocks;
    void *d_in = nullptr, *d_out = nullptr;
    hipStream_t st = nullptr;
    if (int32_t rc = dxtlt_host::acquire_staging(slice * DXTLT_DECODED_BLOCK_BYTES, &d_in, &d_out, &st); rc != kOk)
        return rc;
    for (size_t first = 0; first < n; first += slice) {
        const size_t m = n - first < slice ? n - first : slice;
        HIP_TRY_C(hipMemcpyAsync(d_in, in + first * bs, m * bs, hipMemcpyHostToDevice, st), "H2D copy");
        HIP_TRY_C(dxtlt::launch_decode_blocks(fmt, d_in, d_out, m, st), "kernel launch");
        HIP_TRY_C(hipMemcpyAsync(out + first * DXTLT_DECODED_BLOCK_BYTES, d_out, m * DXTLT_DECODED_BLOCK_BYTES, hipMemcpyDeviceToHost, st),
                  "D2H copy");
    }
    HIP_TRY_C(hipStreamSynchronize(st), "stream synchronize");
    return kOk;
}

int32_t check_difference(int32_t fmt, const void* a, const void* b, size_t len, const void* count)
{
    if (fmt < 1 || fmt > 3)
        return fail(kInvalidArgument, "format must be 1 (BC1), 2 (BC2) or 3 (BC3)");
    if (count == nullptr)
        return fail(kInvalidArgument, "NULL count pointer");
    if (len % block_bytes(fmt) != 0)
        return fail(kInvalidLength, "len is not a multiple of the block size");
    if (len > 0 && (a == nullptr || b == nullptr))
        return fail(kInvalidArgument, "NULL buffer with len > 0");
    return kOk;
}

}  // namespace

extern "C" {

int32_t dxtlt_decode_bc1_blocks(const uint8_t* b, size_t len, uint8_t* px, size_t px_len) { return decode_host(1, b, len, px, px_len); }
int32_t dxtlt_decode_bc2_blocks(const uint8_t* b, size_t len, uint8_t* px, size_t px_len) { return decode_host(2, b, len, px, px_len); }
int32_t dxtlt_decode_bc3_blocks(const uint8_t* b, size_t len, uint8_t* px, size_t px_len) { return decode_host(3, b, len, px, px_len); }

int32_t dxtlt_decode_bc1_blocks_device(const void* b, size_t len, void* px, size_t px_len, void* st)
{
    return decode_device(1, b, len, px, px_len, st);
}
int32_t dxtlt_decode_bc2_blocks_device(const void* b, size_t len, void* px, size_t px_len, void* st)
{
    return decode_device(2, b, len, px, px_len, st);
}
int32_t dxtlt_decode_bc3_blocks_device(const void* b, size_t len, void* px, size_t px_len, void* st)
{
    return decode_device(3, b, len, px, px_len, st);
}

int32_t dxtlt_count_pixel_differences_device(int32_t fmt, const void* a, const void* b, size_t len, uint64_t* d_count, void* st)
{
    if (int32_t rc = check_difference(fmt, a, b, len, d_count); rc != kOk)
        return rc;
    HIP_TRY_C(dxtlt::launch_count_pixel_differences(fmt, a, b, len / block_bytes(fmt), d_count, static_cast<hipStream_t>(st)),
              "kernel launch");
    return kOk;
}

int32_t dxtlt_count_pixel_differences(int32_t fmt, const uint8_t* a, const uint8_t* b, size_t len, uint64_t* out_count)
{
    if (int32_t rc = check_difference(fmt, a, b, len, out_count); rc != kOk)
        return rc;
    *out_count = 0;
    if (len == 0)
        return kOk;
    const size_t slice = len < (size_t(256) << 20) ? len : (size_t(256) << 20);   // a multiple of both block sizes
    const size_t padded = (slice + 255) & ~size_t(255);
    void *d_a = nullptr, *d_b = nullptr;
    hipStream_t st = nullptr;
    if (int32_t rc = dxtlt_host::acquire_staging(padded + 256, &d_a, &d_b, &st); rc != kOk)
        return rc;
    auto* d_count = reinterpret_cast<uint64_t*>(static_cast<uint8_t*>(d_a) + padded);
    uint64_t total = 0;
    for (size_t off = 0; off < len; off += slice) {
        const size_t m = len - off < slice ? len - off : slice;
        uint64_t part = 0;
        HIP_TRY_C(hipMemcpyAsync(d_a, a + off, m, hipMemcpyHostToDevice, st), "H2D copy");
        HIP_TRY_C(hipMemcpyAsync(d_b, b + off, m, hipMemcpyHostToDevice, st), "H2D copy");
        HIP_TRY_C(dxtlt::launch_count_pixel_differences(fmt, d_a, d_b, m / block_bytes(fmt), d_count, st), "kernel launch");
        HIP_TRY_C(hipMemcpyAsync(&part, d_count, sizeof part, hipMemcpyDeviceToHost, st), "D2H copy");
        HIP_TRY_C(hipStreamSynchronize(st), "stream synchronize");
        total += part;
    }
    *out_count = total;
    return kOk;
}

}  // extern "C"
