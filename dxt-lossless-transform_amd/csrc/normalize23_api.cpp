// normalize23_api.cpp -- C ABI of the BC2 / BC3 block-normalisation entry points (include/dxtlt_bc23_normalize.h): the
// reference's experimental modules dxt-lossless-transform-bc{2,3}/src/experimental/normalize_blocks/normalize.rs on the
// device.  Host-pointer calls stage through the calling thread's device buffers (H2D, kernel, D2H, synchronous).
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/dxtlt_bc23_normalize.h"
#include "../../include/dxtlt_gfx950.h"
#include "bcn_launch.h"
#include "host_common.h"

using dxtlt_host::fail;
using dxtlt_host::kDevice;
using dxtlt_host::kInvalidArgument;
using dxtlt_host::kInvalidLength;
using dxtlt_host::kOk;

namespace {

#define HIP_TRY_23(expr, what)                  \
    do {                                        \
        hipError_t e_ = (expr);                 \
        if (e_ != hipSuccess)                   \
            return fail(kDevice, what, e_);     \
    } while (0)

int32_t check_modes(int fmt, uint8_t alpha_mode, uint8_t color_mode)
{
    if (color_mode > 2)
        return fail(kInvalidArgument, "color_mode must be 0 (None), 1 (Color0Only) or 2 (ReplicateColor)");
    if (fmt == 3 && alpha_mode > 3)
        return fail(kInvalidArgument, "alpha_mode must be 0..3");
    return kOk;
}

int32_t blocks_device(int fmt, const void* d_in, void* d_out, size_t len, uint8_t alpha_mode, uint8_t color_mode, void* stream)
{
    if (len % 16 != 0)
        return fail(kInvalidLength, "len is not a multiple of 16");
    if (int32_t rc = check_modes(fmt, alpha_mode, color_mode); rc != kOk)
        return rc;
    if (len > 0 && (d_in == nullptr || d_out == nullptr))
        return fail(kInvalidArgument, "NULL device buffer with len > 0");
    HIP_TRY_23(dxtlt::launch_normalize_bc23_blocks(fmt, d_in, d_out, len / 16, alpha_mode, color_mode,
                                                   static_cast<hipStream_t>(stream)),
               "kernel launch");
    return kOk;
}

int32_t blocks_host(int fmt, const uint8_t* in, uint8_t* out, size_t len, uint8_t alpha_mode, uint8_t color_mode)
{
    if (len % 16 != 0)
        return fail(kInvalidLength, "len is not a multiple of 16");
    if (int32_t rc = check_modes(fmt, alpha_mode, color_mode); rc != kOk)
        return rc;
    if (len == 0)
        return kOk;
    if (in == nullptr || out == nullptr)
        return fail(kInvalidArgument, "NULL buffer with len > 0");
    if (alpha_mode == 0 && color_mode == 0) {
        if (in != out)
            std::memcpy(out, in, len);
        return kOk;
    }
    void *d_a = nullptr, *d_b = nullptr;
    hipStream_t st = nullptr;
    if (int32_t rc = dxtlt_host::acquire_staging(len, &d_a, &d_b, &st); rc != kOk)
        return rc;
    HIP_TRY_23(hipMemcpyAsync(d_a, in, len, hipMemcpyHostToDevice, st), "H2D copy");
    HIP_TRY_23(dxtlt::launch_normalize_bc23_blocks(fmt, d_a, d_a, len / 16, alpha_mode, color_mode, st), "kernel launch");
    HIP_TRY_23(hipMemcpyAsync(out, d_a, len, hipMemcpyDeviceToHost, st), "D2H copy");
    HIP_TRY_23(hipStreamSynchronize(st), "stream synchronize");
    return kOk;
}

int32_t all_modes_device(int fmt, const void* d_in, void* const* d_outs, size_t len, void* stream)
{
    if (len % 16 != 0)
        return fail(kInvalidLength, "len is not a multiple of 16");
    if (d_outs == nullptr)
        return fail(kInvalidArgument, "NULL output pointer array");
    const int count = fmt == 2 ? 3 : 12;
    if (len > 0) {
        if (d_in == nullptr)
            return fail(kInvalidArgument, "NULL device buffer with len > 0");
        for (int i = 0; i < count; ++i)
            if (d_outs[i] == nullptr)
                return fail(kInvalidArgument, "NULL device output buffer with len > 0");
    }
    HIP_TRY_23(dxtlt::launch_normalize_bc23_all_modes(fmt, d_in, d_outs, len / 16, static_cast<hipStream_t>(stream)),
               "kernel launch");
    return kOk;
}

int32_t all_modes_host(int fmt, const uint8_t* in, uint8_t* const* outs, size_t len)
{
    if (len % 16 != 0)
        return fail(kInvalidLength, "len is not a multiple of 16");
    if (outs == nullptr)
        return fail(kInvalidArgument, "NULL output pointer array");
    if (len == 0)
        return kOk;
    const int count = fmt == 2 ? 3 : 12;
    if (in == nullptr)
        return fail(kInvalidArgument, "NULL buffer with len > 0");
    for (int i = 0; i < count; ++i)
        if (outs[i] == nullptr)
            return fail(kInvalidArgument, "NULL output buffer with len > 0");
    // staging: the input at the start of the first buffer, the outputs packed after it and in the second buffer
    const size_t padded = (len + 255) & ~(size_t)255;
    const size_t per_buffer = (size_t)(count + 2) / 2 * padded;
    void *d_a = nullptr, *d_b = nullptr;
    hipStream_t st = nullptr;
    if (int32_t rc = dxtlt_host::acquire_staging(per_buffer, &d_a, &d_b, &st); rc != kOk)
        return rc;
    void* d_outs[12] = {};
    for (int i = 0; i < count; ++i) {
        const int slot = i + 1;   // slot 0 of buffer a holds the input
        const int half = (count + 2) / 2;
        d_outs[i] = slot < half ? static_cast<uint8_t*>(d_a) + (size_t)slot * padded
                                : static_cast<uint8_t*>(d_b) + (size_t)(slot - half) * padded;
    }
    HIP_TRY_23(hipMemcpyAsync(d_a, in, len, hipMemcpyHostToDevice, st), "H2D copy");
    HIP_TRY_23(dxtlt::launch_normalize_bc23_all_modes(fmt, d_a, d_outs, len / 16, st), "kernel launch");
    for (int i = 0; i < count; ++i)
        HIP_TRY_23(hipMemcpyAsync(outs[i], d_outs[i], len, hipMemcpyDeviceToHost, st), "D2H copy");
    HIP_TRY_23(hipStreamSynchronize(st), "stream synchronize");
    return kOk;
}

}  // namespace

extern "C" {

int32_t dxtlt_bc2_normalize_blocks(const uint8_t* i, uint8_t* o, size_t len, uint8_t color_mode)
{
    return blocks_host(2, i, o, len, 0, color_mode);
}
int32_t dxtlt_bc3_normalize_blocks(const uint8_t* i, uint8_t* o, size_t len, uint8_t alpha_mode, uint8_t color_mode)
{
    return blocks_host(3, i, o, len, alpha_mode, color_mode);
}
int32_t dxtlt_bc2_normalize_blocks_device(const void* i, void* o, size_t len, uint8_t color_mode, void* st)
{
    return blocks_device(2, i, o, len, 0, color_mode, st);
}
int32_t dxtlt_bc3_normalize_blocks_device(const void* i, void* o, size_t len, uint8_t alpha_mode, uint8_t color_mode, void* st)
{
    return blocks_device(3, i, o, len, alpha_mode, color_mode, st);
}
int32_t dxtlt_bc2_normalize_blocks_all_modes(const uint8_t* i, uint8_t* const o[3], size_t len)
{
    return all_modes_host(2, i, o, len);
}
int32_t dxtlt_bc3_normalize_blocks_all_modes(const uint8_t* i, uint8_t* const o[12], size_t len)
{
    return all_modes_host(3, i, o, len);
}
int32_t dxtlt_bc2_normalize_blocks_all_modes_device(const void* i, void* const o[3], size_t len, void* st)
{
    return all_modes_device(2, i, o, len, st);
}
int32_t dxtlt_bc3_normalize_blocks_all_modes_device(const void* i, void* const o[12], size_t len, void* st)
{
    return all_modes_device(3, i, o, len, st);
}

int32_t dxtlt_bc2_normalize_split_blocks_in_place_device(void* d_colors, void* d_indices, size_t num_blocks, uint8_t color_mode,
                                                         void* st)
{
    if (int32_t rc = check_modes(2, 0, color_mode); rc != kOk)
        return rc;
    if (num_blocks > 0 && (d_colors == nullptr || d_indices == nullptr))
        return fail(kInvalidArgument, "NULL device buffer with num_blocks > 0");
    HIP_TRY_23(dxtlt::launch_normalize_bc2_split(d_colors, d_indices, num_blocks, color_mode, static_cast<hipStream_t>(st)),
               "kernel launch");
    return kOk;
}

int32_t dxtlt_bc3_normalize_split_blocks_in_place_device(void* d_aep, void* d_aidx, void* d_cep, void* d_cidx, size_t num_blocks,
                                                         uint8_t alpha_mode, uint8_t color_mode, void* st)
{
    if (int32_t rc = check_modes(3, alpha_mode, color_mode); rc != kOk)
        return rc;
    if (num_blocks > 0 && (d_aep == nullptr || d_aidx == nullptr || d_cep == nullptr || d_cidx == nullptr))
        return fail(kInvalidArgument, "NULL device buffer with num_blocks > 0");
    HIP_TRY_23(dxtlt::launch_normalize_bc3_split(d_aep, d_aidx, d_cep, d_cidx, num_blocks, alpha_mode, color_mode,
                                                 static_cast<hipStream_t>(st)),
               "kernel launch");
    return kOk;
}

int32_t dxtlt_bc2_normalize_split_blocks_in_place(const uint8_t* alpha_ptr, uint8_t* colors_ptr, uint8_t* indices_ptr,
                                                  size_t num_blocks, uint8_t color_mode)
{
    (void)alpha_ptr;   // the colour decision ignores alpha (bc2 normalize.rs:131-150)
    if (int32_t rc = check_modes(2, 0, color_mode); rc != kOk)
        return rc;
    if (num_blocks == 0 || color_mode == 0)
        return kOk;
    if (colors_ptr == nullptr || indices_ptr == nullptr)
        return fail(kInvalidArgument, "NULL buffer with num_blocks > 0");
    const size_t half = num_blocks * 4;
    void *d_c = nullptr, *d_i = nullptr;
    hipStream_t st = nullptr;
    if (int32_t rc = dxtlt_host::acquire_staging(half, &d_c, &d_i, &st); rc != kOk)
        return rc;
    HIP_TRY_23(hipMemcpyAsync(d_c, colors_ptr, half, hipMemcpyHostToDevice, st), "H2D copy");
    HIP_TRY_23(hipMemcpyAsync(d_i, indices_ptr, half, hipMemcpyHostToDevice, st), "H2D copy");
    HIP_TRY_23(dxtlt::launch_normalize_bc2_split(d_c, d_i, num_blocks, color_mode, st), "kernel launch");
    HIP_TRY_23(hipMemcpyAsync(colors_ptr, d_c, half, hipMemcpyDeviceToHost, st), "D2H copy");
    HIP_TRY_23(hipMemcpyAsync(indices_ptr, d_i, half, hipMemcpyDeviceToHost, st), "D2H copy");
    HIP_TRY_23(hipStreamSynchronize(st), "stream synchronize");
    return kOk;
}

int32_t dxtlt_bc3_normalize_split_blocks_in_place(uint8_t* aep, uint8_t* aidx, uint8_t* cep, uint8_t* cidx, size_t num_blocks,
                                                  uint8_t alpha_mode, uint8_t color_mode)
{
    if (int32_t rc = check_modes(3, alpha_mode, color_mode); rc != kOk)
        return rc;
    if (num_blocks == 0 || (alpha_mode == 0 && color_mode == 0))
        return kOk;
    if (aep == nullptr || aidx == nullptr || cep == nullptr || cidx == nullptr)
        return fail(kInvalidArgument, "NULL buffer with num_blocks > 0");
    // sections packed into the two staging buffers: [alpha endpoints | alpha indices] and [colour endpoints | indices]
    const size_t n = num_blocks, off_aidx = (2 * n + 255) & ~(size_t)255, off_cidx = (4 * n + 255) & ~(size_t)255;
    void *d_a = nullptr, *d_c = nullptr;
    hipStream_t st = nullptr;
    const size_t need_a = off_aidx + 6 * n, need_c = off_cidx + 4 * n;
    if (int32_t rc = dxtlt_host::acquire_staging(need_a > need_c ? need_a : need_c, &d_a, &d_c, &st); rc != kOk)
        return rc;
    uint8_t* da = static_cast<uint8_t*>(d_a);
    uint8_t* dc = static_cast<uint8_t*>(d_c);
    HIP_TRY_23(hipMemcpyAsync(da, aep, 2 * n, hipMemcpyHostToDevice, st), "H2D copy");
    HIP_TRY_23(hipMemcpyAsync(da + off_aidx, aidx, 6 * n, hipMemcpyHostToDevice, st), "H2D copy");
    HIP_TRY_23(hipMemcpyAsync(dc, cep, 4 * n, hipMemcpyHostToDevice, st), "H2D copy");
    HIP_TRY_23(hipMemcpyAsync(dc + off_cidx, cidx, 4 * n, hipMemcpyHostToDevice, st), "H2D copy");
    HIP_TRY_23(dxtlt::launch_normalize_bc3_split(da, da + off_aidx, dc, dc + off_cidx, n, alpha_mode, color_mode, st),
               "kernel launch");
    HIP_TRY_23(hipMemcpyAsync(aep, da, 2 * n, hipMemcpyDeviceToHost, st), "D2H copy");
    HIP_TRY_23(hipMemcpyAsync(aidx, da + off_aidx, 6 * n, hipMemcpyDeviceToHost, st), "D2H copy");
    HIP_TRY_23(hipMemcpyAsync(cep, dc, 4 * n, hipMemcpyDeviceToHost, st), "D2H copy");
    HIP_TRY_23(hipMemcpyAsync(cidx, dc + off_cidx, 4 * n, hipMemcpyDeviceToHost, st), "D2H copy");
    HIP_TRY_23(hipStreamSynchronize(st), "stream synchronize");
    return kOk;
}

}  // extern "C"
