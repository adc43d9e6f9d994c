// normalize_api.cpp -- C ABI of the BC1 block-normalisation entry points (include/dxtlt_bc1_normalize.h): the
// reference's experimental module, /root/reference/src/core/dxt-lossless-transform-bc1/src/experimental/
// normalize_blocks/{normalize.rs, transform.rs}, on the device.
//
// transform_bc1_auto_with_normalization (transform.rs:222-333) as the reference runs it:
//   * one max_compressed_size(len / 2) query up front (its failure is the only estimator error that propagates);
//   * all blocks are classified once; if no block would change, the call IS transform_bc1_auto;
//   * otherwise for mode in {None, Color0Only, ReplicateColor}, for (decorrelation, split) in the BC1 test order:
//     the estimator sees the first len / 2 bytes (the colour section) of transform(normalize(input, mode));
//     a failing estimate skips that candidate; strict `<` against the running best, which starts at
//     {None, Variant1, split} with size usize::MAX;
//   * the data is transformed once more with the winner.
// GPU shape: the input is uploaded once; the classification is a read-only kernel that sets a flag; every candidate is
// one fused normalise+transform launch on the resident copy and only its colour section travels back.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "../../include/dxtlt_bc1_normalize.h"
#include "../../include/dxtlt_gfx950.h"
#include "bcn_launch.h"
#include "host_common.h"

using dxtlt_host::fail;
using dxtlt_host::kAllocation;
using dxtlt_host::kDevice;
using dxtlt_host::kEstimator;
using dxtlt_host::kInvalidArgument;
using dxtlt_host::kInvalidLength;
using dxtlt_host::kOk;

namespace {

#define HIP_TRY_N(expr, what)                   \
    do {                                        \
        hipError_t e_ = (expr);                 \
        if (e_ != hipSuccess)                   \
            return fail(kDevice, what, e_);     \
    } while (0)

int32_t check_mode(uint8_t color_mode)
{
    if (color_mode > DXTLT_NORMALIZE_REPLICATE_COLOR)
        return fail(kInvalidArgument, "color_mode must be 0 (None), 1 (Color0Only) or 2 (ReplicateColor)");
    return kOk;
}

struct Candidate {
    uint8_t variant;
    bool split;
};
// bc1 settings.rs:81-86 (FAST_TEST_ORDER) and :89-98 (COMPREHENSIVE_TEST_ORDER)
const Candidate kFast[] = {{0, false}, {0, true}, {1, false}, {1, true}};
const Candidate kAll[] = {{2, false}, {0, false}, {0, true}, {3, false}, {3, true}, {2, true}, {1, false}, {1, true}};

// a device word for the "any block normalised" flag, next to the staging buffers of this thread
struct FlagWord {
    uint32_t* d = nullptr;
    int device = -1;
    ~FlagWord()
    {
        if (d) (void)hipFree(d);
    }
    int32_t get(uint32_t** out)
    {
        int dev = 0;
        HIP_TRY_N(hipGetDevice(&dev), "hipGetDevice");
        if (dev != device) {
            if (d) (void)hipFree(d);
            d = nullptr;
            HIP_TRY_N(hipMalloc(reinterpret_cast<void**>(&d), 256), "hipMalloc(flag)");
            device = dev;
        }
        *out = d;
        return kOk;
    }
};
thread_local FlagWord g_flag;

}  // namespace

void dxtlt_host::release_normalize_thread_flag()
{
    if (g_flag.d) (void)hipFree(g_flag.d);
    g_flag.d = nullptr;
    g_flag.device = -1;
}

extern "C" {

int32_t dxtlt_bc1_normalize_blocks_device(const void* d_input, void* d_output, size_t len, uint8_t color_mode,
                                          void* hip_stream)
{
    if (len % 8 != 0)
        return fail(kInvalidLength, "len is not a multiple of 8");
    if (int32_t rc = check_mode(color_mode); rc != kOk)
        return rc;
    if (len > 0 && (d_input == nullptr || d_output == nullptr))
        return fail(kInvalidArgument, "NULL device buffer with len > 0");
    HIP_TRY_N(dxtlt::launch_normalize_bc1_blocks(d_input, d_output, len / 8, color_mode, static_cast<hipStream_t>(hip_stream)),
              "kernel launch");
    return kOk;
}

int32_t dxtlt_bc1_normalize_split_blocks_in_place_device(void* d_colors, void* d_indices, size_t num_blocks,
                                                         uint8_t color_mode, void* hip_stream)
{
    if (int32_t rc = check_mode(color_mode); rc != kOk)
        return rc;
    if (num_blocks > 0 && (d_colors == nullptr || d_indices == nullptr))
        return fail(kInvalidArgument, "NULL device buffer with num_blocks > 0");
    HIP_TRY_N(dxtlt::launch_normalize_bc1_split(d_colors, d_indices, num_blocks, color_mode,
                                                static_cast<hipStream_t>(hip_stream)),
              "kernel launch");
    return kOk;
}

int32_t dxtlt_bc1_normalize_blocks_all_modes_device(const void* d_input, void* const d_outputs[3], size_t len,
                                                    uint32_t* d_any_normalized, void* hip_stream)
{
    if (len % 8 != 0)
        return fail(kInvalidLength, "len is not a multiple of 8");
    if (d_outputs == nullptr)
        return fail(kInvalidArgument, "NULL output pointer array");
    if (len > 0 && (d_input == nullptr || d_outputs[0] == nullptr || d_outputs[1] == nullptr || d_outputs[2] == nullptr))
        return fail(kInvalidArgument, "NULL device buffer with len > 0");
    HIP_TRY_N(dxtlt::launch_normalize_bc1_all_modes(d_input, d_outputs, len / 8, d_any_normalized,
                                                    static_cast<hipStream_t>(hip_stream)),
              "kernel launch");
    return kOk;
}

int32_t dxtlt_transform_bc1_with_normalize_blocks_device(const void* d_input, void* d_output, size_t len,
                                                         uint8_t color_mode, uint8_t decorrelation_mode,
                                                         bool split_colour_endpoints, void* hip_stream)
{
    if (len % 8 != 0)
        return fail(kInvalidLength, "len is not a multiple of 8");
    if (int32_t rc = check_mode(color_mode); rc != kOk)
        return rc;
    if (decorrelation_mode > 3)
        return fail(kInvalidArgument, "decorrelation_mode must be 0..3");
    if (len > 0 && (d_input == nullptr || d_output == nullptr))
        return fail(kInvalidArgument, "NULL device buffer with len > 0");
    return dxtlt_host::enqueue(1, false, d_input, d_output, len / 8, decorrelation_mode, false, split_colour_endpoints,
                               static_cast<hipStream_t>(hip_stream), color_mode);
}

int32_t dxtlt_bc1_normalize_blocks(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len, uint8_t color_mode)
{
    if (len % 8 != 0)
        return fail(kInvalidLength, "len is not a multiple of 8");
    if (int32_t rc = check_mode(color_mode); rc != kOk)
        return rc;
    if (len == 0)
        return kOk;
    if (input_ptr == nullptr || output_ptr == nullptr)
        return fail(kInvalidArgument, "NULL buffer with len > 0");
    if (color_mode == DXTLT_NORMALIZE_NONE) {   // normalize.rs:53-64: plain copy, nothing when in place
        if (input_ptr != output_ptr)
            std::memcpy(output_ptr, input_ptr, len);
        return kOk;
    }
    void *d_in = nullptr, *d_out = nullptr;
    hipStream_t st = nullptr;
    if (int32_t rc = dxtlt_host::acquire_staging(len, &d_in, &d_out, &st); rc != kOk)
        return rc;
    HIP_TRY_N(hipMemcpyAsync(d_in, input_ptr, len, hipMemcpyHostToDevice, st), "H2D copy");
    HIP_TRY_N(dxtlt::launch_normalize_bc1_blocks(d_in, d_in, len / 8, color_mode, st), "kernel launch");
    HIP_TRY_N(hipMemcpyAsync(output_ptr, d_in, len, hipMemcpyDeviceToHost, st), "D2H copy");
    HIP_TRY_N(hipStreamSynchronize(st), "stream synchronize");
    return kOk;
}

int32_t dxtlt_bc1_normalize_split_blocks_in_place(uint8_t* colors_ptr, uint8_t* indices_ptr, size_t num_blocks,
                                                  uint8_t color_mode)
{
    if (int32_t rc = check_mode(color_mode); rc != kOk)
        return rc;
    if (num_blocks == 0 || color_mode == DXTLT_NORMALIZE_NONE)   // normalize.rs:293-295
        return kOk;
    if (colors_ptr == nullptr || indices_ptr == nullptr)
        return fail(kInvalidArgument, "NULL buffer with num_blocks > 0");
    const size_t half = num_blocks * 4;
    void *d_col = nullptr, *d_idx = nullptr;
    hipStream_t st = nullptr;
    if (int32_t rc = dxtlt_host::acquire_staging(half, &d_col, &d_idx, &st); rc != kOk)
        return rc;
    HIP_TRY_N(hipMemcpyAsync(d_col, colors_ptr, half, hipMemcpyHostToDevice, st), "H2D copy");
    HIP_TRY_N(hipMemcpyAsync(d_idx, indices_ptr, half, hipMemcpyHostToDevice, st), "H2D copy");
    HIP_TRY_N(dxtlt::launch_normalize_bc1_split(d_col, d_idx, num_blocks, color_mode, st), "kernel launch");
    HIP_TRY_N(hipMemcpyAsync(colors_ptr, d_col, half, hipMemcpyDeviceToHost, st), "D2H copy");
    HIP_TRY_N(hipMemcpyAsync(indices_ptr, d_idx, half, hipMemcpyDeviceToHost, st), "D2H copy");
    HIP_TRY_N(hipStreamSynchronize(st), "stream synchronize");
    return kOk;
}

int32_t dxtlt_bc1_normalize_blocks_all_modes(const uint8_t* input_ptr, uint8_t* const output_ptrs[3], size_t len,
                                             bool* out_any_normalized)
{
    if (len % 8 != 0)
        return fail(kInvalidLength, "len is not a multiple of 8");
    if (output_ptrs == nullptr)
        return fail(kInvalidArgument, "NULL output pointer array");
    if (out_any_normalized)
        *out_any_normalized = false;
    if (len == 0)
        return kOk;
    if (input_ptr == nullptr || output_ptrs[0] == nullptr || output_ptrs[1] == nullptr || output_ptrs[2] == nullptr)
        return fail(kInvalidArgument, "NULL buffer with len > 0");
    // staging: the input in d_a[0, len) -- overwritten in place by the `None` output, which is not a plain copy: fully
    // transparent blocks are rewritten in every output (normalize.rs:447-454) --, the two others in d_a[len, 2 len)
    // and d_b[0, len)
    const size_t padded = (len + 255) & ~(size_t)255;
    void *d_a = nullptr, *d_b = nullptr;
    hipStream_t st = nullptr;
    if (int32_t rc = dxtlt_host::acquire_staging(2 * padded, &d_a, &d_b, &st); rc != kOk)
        return rc;
    uint32_t* d_flag = nullptr;
    if (int32_t rc = g_flag.get(&d_flag); rc != kOk)
        return rc;
    void* outs[3] = {d_a, static_cast<uint8_t*>(d_a) + padded, d_b};
    uint32_t any = 0;
    HIP_TRY_N(hipMemsetAsync(d_flag, 0, sizeof(uint32_t), st), "memset flag");
    HIP_TRY_N(hipMemcpyAsync(d_a, input_ptr, len, hipMemcpyHostToDevice, st), "H2D copy");
    HIP_TRY_N(dxtlt::launch_normalize_bc1_all_modes(d_a, outs, len / 8, d_flag, st), "kernel launch");
    for (int m = 0; m < 3; ++m)
        HIP_TRY_N(hipMemcpyAsync(output_ptrs[m], outs[m], len, hipMemcpyDeviceToHost, st), "D2H copy");
    HIP_TRY_N(hipMemcpyAsync(&any, d_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, st), "D2H flag");
    HIP_TRY_N(hipStreamSynchronize(st), "stream synchronize");
    if (out_any_normalized)
        *out_any_normalized = any != 0;
    return kOk;
}

int32_t dxtlt_transform_bc1_with_normalize_blocks(const uint8_t* input_ptr, uint8_t* output_ptr, uint8_t* work_ptr,
                                                  size_t len, uint8_t color_mode, uint8_t decorrelation_mode,
                                                  bool split_colour_endpoints)
{
    (void)work_ptr;  // the reference's CPU scratch buffer; the fused kernel needs none
    if (int32_t rc = check_mode(color_mode); rc != kOk)
        return rc;
    return dxtlt_host::transform(1, false, input_ptr, output_ptr, len, decorrelation_mode, false, split_colour_endpoints,
                                 color_mode);
}

int32_t dxtlt_transform_bc1_auto_with_normalization(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len,
                                                    const DltSizeEstimator* est, bool use_all, uint8_t* out_color_mode,
                                                    uint8_t* out_decorrelation_mode, bool* out_split_colour_endpoints,
                                                    uint32_t* out_estimator_error)
{
    if (out_estimator_error)
        *out_estimator_error = 0;
    if (len % 8 != 0)
        return fail(kInvalidLength, "len is not a multiple of 8");
    if (est == nullptr || est->MaxCompressedSize == nullptr || est->EstimateCompressedSize == nullptr)
        return fail(kInvalidArgument, "NULL estimator");
    if (len > 0 && (input_ptr == nullptr || output_ptr == nullptr))
        return fail(kInvalidArgument, "NULL buffer with len > 0");

    size_t max_comp = 0;
    if (uint32_t rc_est = est->MaxCompressedSize(est->Context, len / 2, &max_comp); rc_est != 0) {
        if (out_estimator_error)
            *out_estimator_error = rc_est;
        return fail(kEstimator, "size estimator: max_compressed_size failed");
    }

    // classification pass: would any block change?
    uint32_t any = 0;
    void *d_in = nullptr, *d_out = nullptr;
    hipStream_t st = nullptr;
    if (len > 0) {
        if (int32_t rc = dxtlt_host::acquire_staging(len, &d_in, &d_out, &st); rc != kOk)
            return rc;
        uint32_t* d_flag = nullptr;
        if (int32_t rc = g_flag.get(&d_flag); rc != kOk)
            return rc;
        HIP_TRY_N(hipMemsetAsync(d_flag, 0, sizeof(uint32_t), st), "memset flag");
        HIP_TRY_N(hipMemcpyAsync(d_in, input_ptr, len, hipMemcpyHostToDevice, st), "H2D copy");
        HIP_TRY_N(dxtlt::launch_bc1_any_normalizable(d_in, len / 8, d_flag, st), "kernel launch");
        HIP_TRY_N(hipMemcpyAsync(&any, d_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, st), "D2H flag");
        HIP_TRY_N(hipStreamSynchronize(st), "stream synchronize");
    }
    if (any == 0) {
        // transform.rs:321-331: nothing to normalise -> the regular brute force, mode None
        dxtlt_host::AutoChoice c{};
        const int32_t rc = dxtlt_host::transform_auto(1, input_ptr, output_ptr, len, est, use_all, &c);
        if (out_estimator_error)
            *out_estimator_error = c.estimator_error;
        if (rc == kOk) {
            if (out_color_mode) *out_color_mode = DXTLT_NORMALIZE_NONE;
            if (out_decorrelation_mode) *out_decorrelation_mode = c.mode;
            if (out_split_colour_endpoints) *out_split_colour_endpoints = c.split_colour;
        }
        return rc;
    }

    uint8_t* scratch = nullptr;
    if (max_comp != 0) {
        scratch = static_cast<uint8_t*>(std::aligned_alloc(64, (max_comp + 63) / 64 * 64));
        if (scratch == nullptr)
            return fail(kAllocation, "estimator scratch allocation failed");
    }
    struct Best {
        uint8_t norm, variant;
        bool split;
    } best{DXTLT_NORMALIZE_NONE, 1, true}, last = best;
    bool last_valid = false;
    size_t best_size = SIZE_MAX;
    const Candidate* order = use_all ? kAll : kFast;
    const int count = use_all ? 8 : 4;
    const uint64_t blocks = len / 8;
    int32_t rc = kOk;
    hipError_t herr = hipSuccess;
    for (uint8_t norm = 0; norm <= DXTLT_NORMALIZE_REPLICATE_COLOR && rc == kOk && herr == hipSuccess; ++norm) {
        // the reference estimates on the buffers of normalize_blocks_all_modes, whose `None` buffer still has its
        // fully transparent blocks rewritten (normalize.rs:447-454): internal fused mode 3 reproduces that
        const uint8_t fused = norm == DXTLT_NORMALIZE_NONE ? 3 : norm;
        for (int i = 0; i < count; ++i) {
            rc = dxtlt_host::enqueue(1, false, d_in, d_out, blocks, order[i].variant, false, order[i].split, st, fused);
            if (rc != kOk)
                break;
            herr = hipMemcpyAsync(output_ptr, d_out, len / 2, hipMemcpyDeviceToHost, st);
            if (herr == hipSuccess)
                herr = hipStreamSynchronize(st);
            if (herr != hipSuccess)
                break;
            last = {norm, order[i].variant, order[i].split};
            last_valid = true;
            size_t size = 0;
            if (est->EstimateCompressedSize(est->Context, output_ptr, len / 2, scratch, max_comp, &size) != 0)
                continue;   // transform.rs:403: a failing estimate skips the candidate
            if (size < best_size) {
                best_size = size;
                best = last;
            }
        }
    }
    std::free(scratch);
    if (rc != kOk)
        return rc;
    if (herr != hipSuccess)
        return fail(kDevice, "candidate transform / download", herr);

    // the final transform uses the chosen mode as transform_bc1_with_normalize_blocks defines it (transform.rs:307-313),
    // so a `None` winner is re-run without the transparent rewrite its candidates were estimated with
    const bool resident = last_valid && best.norm != DXTLT_NORMALIZE_NONE && last.norm == best.norm &&
                          last.variant == best.variant && last.split == best.split;
    if (!resident) {
        rc = dxtlt_host::enqueue(1, false, d_in, d_out, blocks, best.variant, false, best.split, st, best.norm);
        if (rc != kOk)
            return rc;
    }
    HIP_TRY_N(hipMemcpyAsync(output_ptr, d_out, len, hipMemcpyDeviceToHost, st), "D2H result");
    HIP_TRY_N(hipStreamSynchronize(st), "stream synchronize");
    if (out_color_mode) *out_color_mode = best.norm;
    if (out_decorrelation_mode) *out_decorrelation_mode = best.variant;
    if (out_split_colour_endpoints) *out_split_colour_endpoints = best.split;
    return kOk;
}

}  // extern "C"
