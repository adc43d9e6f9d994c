// bc23_normalize.hip -- gfx950 kernels for BC2 / BC3 block normalisation (the per-block rules are bc23_normalize.h).
//
// Reference: /root/reference/src/core/dxt-lossless-transform-bc{2,3}/src/experimental/normalize_blocks/normalize.rs --
// normalize_blocks (bc2 :35, bc3 :36), normalize_blocks_all_modes (bc2 :193, bc3 :419),
// normalize_split_blocks_in_place (bc2 :382, bc3 :539).  Element-wise, one 16-byte block per lane, one-shot grid; HBM
// bound (2*len per output buffer).  Pointers that are not 16-byte aligned go through byte accesses.
#include <hip/hip_runtime.h>

#include "bc23_normalize.h"
#include "bcn_launch.h"
#include "launch_grid.h"
#include "streaming_store.h"

namespace dxtlt {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads23 = 256;

__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p, bool a4)
{
    if (a4)
        return *reinterpret_cast<const uint32_t*>(p);
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

__device__ __forceinline__ void st_u32(uint8_t* p, uint32_t v, bool a4)
{
    if (a4) {
        *reinterpret_cast<uint32_t*>(p) = v;
    } else {
        p[0] = (uint8_t)v;
        p[1] = (uint8_t)(v >> 8);
        p[2] = (uint8_t)(v >> 16);
        p[3] = (uint8_t)(v >> 24);
    }
}

__device__ __forceinline__ void load_block(const uint8_t* p, bool vec, uint32_t (&q)[4])
{
    if (vec) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        q[0] = v.x, q[1] = v.y, q[2] = v.z, q[3] = v.w;
    } else {
        const bool a4 = (reinterpret_cast<uintptr_t>(p) & 3) == 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            q[i] = ld_u32(p + 4 * i, a4);
    }
}

__device__ __forceinline__ void store_block16(uint8_t* p, bool vec, const uint32_t (&q)[4])
{
    if (vec) {
        store_streaming16(p, u32x4{q[0], q[1], q[2], q[3]});
    } else {
        const bool a4 = (reinterpret_cast<uintptr_t>(p) & 3) == 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            st_u32(p + 4 * i, q[i], a4);
    }
}

template <int FMT>
__global__ void __launch_bounds__(kThreads23)
normalize23_kernel(const uint8_t* in, uint8_t* out, uint64_t n, int alpha_mode, int color_mode, int vec)
{
    const uint64_t i = workgroup_index() * kThreads23 + threadIdx.x;
    if (i >= n)
        return;
    uint32_t q[4];
    load_block(in + 16 * i, vec != 0, q);
    normalize_block_bc23<FMT>(alpha_mode, color_mode, q);
    store_block16(out + 16 * i, vec != 0, q);
}

struct OutPtrs {
    uint8_t* p[12];
};

// BC2: outputs 0..2 = colour modes; BC3: outputs [alpha_mode * 3 + colour_mode].  The block is classified once.
template <int FMT>
__global__ void __launch_bounds__(kThreads23)
normalize23_all_modes_kernel(const uint8_t* in, OutPtrs outs, uint64_t n, int vec)
{
    const uint64_t i = workgroup_index() * kThreads23 + threadIdx.x;
    if (i >= n)
        return;
    uint32_t q[4];
    load_block(in + 16 * i, vec != 0, q);
    uint32_t solid = 0, alpha = 0;
    const bool is_solid = solid_colour_4c(q[2], q[3], solid);
    const bool is_uniform = FMT == 3 && uniform_alpha_bc3(q[0], q[1], alpha);
    constexpr int kAlphaModes = FMT == 3 ? 4 : 1;
#pragma unroll
    for (int a = 0; a < kAlphaModes; ++a) {
        uint32_t w0 = q[0], w1 = q[1];
        if (is_uniform && a != kAlphaNone) {
            if (alpha == 255 && a == kAlphaOpaqueFillAll) {
                w0 = w1 = 0xFFFFFFFFu;
            } else if (alpha == 255 && a == kAlphaOpaqueZeroAlphaMaxIndices) {
                w0 = 0xFFFF0000u, w1 = 0xFFFFFFFFu;
            } else {
                w0 = alpha, w1 = 0;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uint32_t o[4] = {w0, w1, q[2], q[3]};
            if (is_solid && c != kNormNone) {
                o[2] = c == kNormReplicateColor ? solid | (solid << 16) : solid;
                o[3] = 0;
            }
            store_block16(outs.p[a * 3 + c] + 16 * i, vec != 0, o);
        }
    }
}

// BC2, colours (4 bytes per block) and indices (4 bytes per block) in place; the alpha array is not needed
__global__ void __launch_bounds__(kThreads23)
normalize2_split_kernel(uint8_t* colours, uint8_t* indices, uint64_t n, int color_mode)
{
    const uint64_t i = workgroup_index() * kThreads23 + threadIdx.x;
    if (i >= n)
        return;
    const bool a4 = ((reinterpret_cast<uintptr_t>(colours) | reinterpret_cast<uintptr_t>(indices)) & 3) == 0;
    uint32_t c = ld_u32(colours + 4 * i, a4), x = ld_u32(indices + 4 * i, a4);
    if (normalize_colour_half_4c(color_mode, c, x)) {
        st_u32(colours + 4 * i, c, a4);
        st_u32(indices + 4 * i, x, a4);
    }
}

// BC3, four arrays in place: alpha endpoints (2 bytes per block), alpha indices (6), colour endpoints (4), colour indices (4)
__global__ void __launch_bounds__(kThreads23)
normalize3_split_kernel(uint8_t* aep, uint8_t* aidx, uint8_t* cep, uint8_t* cidx, uint64_t n, int alpha_mode, int color_mode)
{
    const uint64_t i = workgroup_index() * kThreads23 + threadIdx.x;
    if (i >= n)
        return;
    const bool c4 = ((reinterpret_cast<uintptr_t>(cep) | reinterpret_cast<uintptr_t>(cidx)) & 3) == 0;
    uint8_t* pe = aep + 2 * i;
    uint8_t* pi = aidx + 6 * i;
    uint32_t w0 = (uint32_t)pe[0] | ((uint32_t)pe[1] << 8) | ((uint32_t)pi[0] << 16) | ((uint32_t)pi[1] << 24);
    uint32_t w1 = (uint32_t)pi[2] | ((uint32_t)pi[3] << 8) | ((uint32_t)pi[4] << 16) | ((uint32_t)pi[5] << 24);
    if (normalize_alpha_half_bc3(alpha_mode, w0, w1)) {
        pe[0] = (uint8_t)w0;
        pe[1] = (uint8_t)(w0 >> 8);
        pi[0] = (uint8_t)(w0 >> 16);
        pi[1] = (uint8_t)(w0 >> 24);
        pi[2] = (uint8_t)w1;
        pi[3] = (uint8_t)(w1 >> 8);
        pi[4] = (uint8_t)(w1 >> 16);
        pi[5] = (uint8_t)(w1 >> 24);
    }
    uint32_t c = ld_u32(cep + 4 * i, c4), x = ld_u32(cidx + 4 * i, c4);
    if (normalize_colour_half_4c(color_mode, c, x)) {
        st_u32(cep + 4 * i, c, c4);
        st_u32(cidx + 4 * i, x, c4);
    }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline hipError_t grid23(uint64_t n, dim3& g) { return grid_rows(n, kThreads23, g); }

}  // namespace

hipError_t launch_normalize_bc23_blocks(int fmt, const void* in, void* out, uint64_t num_blocks, int alpha_mode, int color_mode,
                                        hipStream_t stream)
{
    if ((fmt != 2 && fmt != 3) || alpha_mode < 0 || alpha_mode > 3 || color_mode < 0 || color_mode > 2 ||
        (fmt == 2 && alpha_mode != 0))
        return hipErrorInvalidValue;
    if (num_blocks == 0)
        return hipSuccess;
    if (alpha_mode == kAlphaNone && color_mode == kNormNone)   // bc2 normalize.rs:50-61, bc3 normalize.rs:52-62
        return in == out ? hipSuccess : hipMemcpyAsync(out, in, num_blocks * 16, hipMemcpyDeviceToDevice, stream);
    dim3 g;
    if (hipError_t e = grid23(num_blocks, g); e != hipSuccess)
        return e;
    const int vec = al16(in) && al16(out);
    auto k = fmt == 2 ? normalize23_kernel<2> : normalize23_kernel<3>;
    hipLaunchKernelGGL(k, g, dim3(kThreads23), 0, stream, static_cast<const uint8_t*>(in), static_cast<uint8_t*>(out),
                       num_blocks, alpha_mode, color_mode, vec);
    return hipGetLastError();
}

hipError_t launch_normalize_bc23_all_modes(int fmt, const void* in, void* const* outs, uint64_t num_blocks, hipStream_t stream)
{
    if (fmt != 2 && fmt != 3)
        return hipErrorInvalidValue;
    if (num_blocks == 0)
        return hipSuccess;
    const int count = fmt == 2 ? 3 : 12;
    OutPtrs o{};
    int vec = al16(in);
    for (int i = 0; i < count; ++i) {
        o.p[i] = static_cast<uint8_t*>(outs[i]);
        vec = vec && al16(outs[i]);
    }
    dim3 g;
    if (hipError_t e = grid23(num_blocks, g); e != hipSuccess)
        return e;
    auto k = fmt == 2 ? normalize23_all_modes_kernel<2> : normalize23_all_modes_kernel<3>;
    // one read, 3 (BC2) or 12 (BC3) copies written per lane: fewer resident workgroups per CU move more (launch_grid.h).
    // BC2, 1 GiB of blocks: 764 us at eight, 658 at six, 681 / 672 at five / four, 908 at three; BC3, 256 MiB: 700 at eight,
    // 664 / 680 / 616 at six / five / four, 584 at three (profiles/r02_o_wgs_per_cu.txt)
    hipLaunchKernelGGL(k, g, dim3(kThreads23), lds_pad_for_wgs_per_cu(wgs_per_cu_or(fmt == 2 ? 6 : 3), kThreads23, 0), stream, static_cast<const uint8_t*>(in), o, num_blocks, vec);
    return hipGetLastError();
}

hipError_t launch_normalize_bc2_split(void* colours, void* indices, uint64_t num_blocks, int color_mode, hipStream_t stream)
{
    if (color_mode < 0 || color_mode > 2)
        return hipErrorInvalidValue;
    if (num_blocks == 0 || color_mode == kNormNone)
        return hipSuccess;
    dim3 g;
    if (hipError_t e = grid23(num_blocks, g); e != hipSuccess)
        return e;
    hipLaunchKernelGGL(normalize2_split_kernel, g, dim3(kThreads23), 0, stream, static_cast<uint8_t*>(colours),
                       static_cast<uint8_t*>(indices), num_blocks, color_mode);
    return hipGetLastError();
}

hipError_t launch_normalize_bc3_split(void* alpha_endpoints, void* alpha_indices, void* color_endpoints, void* color_indices,
                                      uint64_t num_blocks, int alpha_mode, int color_mode, hipStream_t stream)
{
    if (alpha_mode < 0 || alpha_mode > 3 || color_mode < 0 || color_mode > 2)
        return hipErrorInvalidValue;
    if (num_blocks == 0 || (alpha_mode == kAlphaNone && color_mode == kNormNone))
        return hipSuccess;
    dim3 g;
    if (hipError_t e = grid23(num_blocks, g); e != hipSuccess)
        return e;
    hipLaunchKernelGGL(normalize3_split_kernel, g, dim3(kThreads23), 0, stream, static_cast<uint8_t*>(alpha_endpoints),
                       static_cast<uint8_t*>(alpha_indices), static_cast<uint8_t*>(color_endpoints),
                       static_cast<uint8_t*>(color_indices), num_blocks, alpha_mode, color_mode);
    return hipGetLastError();
}

}  // namespace dxtlt
