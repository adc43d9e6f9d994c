// color565_ops.hip -- gfx950 kernels for the array-level colour operations of the reference's common crate (rows a2 /
// a3 of SURVEY.md section 8(a) as stand-alone operations):
//   Color565::decorrelate_ycocg_r_ptr / recorrelate_ycocg_r_ptr   dxt-lossless-transform-common/src/color_565/decorrelate_batch_ptr.rs:336,378
//   Color565::recorrelate_ycocg_r_ptr_split                        .../color_565/decorrelate_batch_split_ptr.rs:324
//   split_color_endpoints                                          .../transforms/split_565_color_endpoints/mod.rs:110
// Element-wise over 16-bit colours, 16 bytes (eight colours) per lane with ycocg_swar.h on colour pairs; pointers that
// are not 16-byte aligned and the last few colours take a one-colour (one-pair) path.  HBM bound, 2 * bytes.
#include <hip/hip_runtime.h>

#include "bcn_launch.h"
#include "launch_grid.h"
#include "streaming_store.h"
#include "ycocg_swar.h"

namespace dxtlt {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kColThreads = 256;

template <int V, bool INVERSE>
__device__ __forceinline__ uint32_t ycocg2(uint32_t v)
{
    return INVERSE ? recorrelate2<V>(v) : decorrelate2<V>(v);
}

template <bool INVERSE>
__device__ __forceinline__ uint32_t ycocg2_rt(int variant, uint32_t v)
{
    switch (variant) {
    case kVar1: return ycocg2<kVar1, INVERSE>(v);
    case kVar2: return ycocg2<kVar2, INVERSE>(v);
    case kVar3: return ycocg2<kVar3, INVERSE>(v);
    default: return v;
    }
}

__device__ __forceinline__ uint32_t ld_u16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ void st_u16(uint8_t* p, uint32_t v)
{
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
}

// lanes [0, vecs) take eight colours as one 16-byte vector, lanes [vecs, vecs + singles) one colour each
template <bool INVERSE>
__global__ void __launch_bounds__(kColThreads)
ycocg_array_kernel(const uint8_t* in, uint8_t* out, uint64_t vecs, uint64_t singles, int variant)
{
    const uint64_t i = workgroup_index() * kColThreads + threadIdx.x;
    if (i < vecs) {
        const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + 16 * i));
        const u32x4 r = {ycocg2_rt<INVERSE>(variant, q.x), ycocg2_rt<INVERSE>(variant, q.y), ycocg2_rt<INVERSE>(variant, q.z),
                         ycocg2_rt<INVERSE>(variant, q.w)};
        store_streaming16(out + 16 * i, r);
    } else if (i < vecs + singles) {
        const uint64_t c = 8 * vecs + (i - vecs);
        st_u16(out + 2 * c, ycocg2_rt<INVERSE>(variant, ld_u16(in + 2 * c)) & 0xFFFFu);
    }
}

__device__ __forceinline__ u32x4 zip16(u32x2 a, u32x2 b)   // a = four c0, b = four c1 -> four (c0, c1) pairs
{
    return u32x4{(a.x & 0xFFFFu) | (b.x << 16), (a.x >> 16) | (b.x & 0xFFFF0000u), (a.y & 0xFFFFu) | (b.y << 16),
                 (a.y >> 16) | (b.y & 0xFFFF0000u)};
}

// interleave src0[k], src1[k] -> dst[2k], dst[2k+1] with recorrelation: lanes [0, vecs) four pairs each (one 16-byte
// store per lane: two per lane at a 32-byte stride leave every 128-byte line to two store instructions, measured
// 0.57 of peak against 0.80)
__global__ void __launch_bounds__(kColThreads)
recorrelate_split_kernel(const uint8_t* src0, const uint8_t* src1, uint8_t* dst, uint64_t vecs, uint64_t singles, int variant)
{
    const uint64_t i = workgroup_index() * kColThreads + threadIdx.x;
    if (i < vecs) {
        const u32x2 a = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(src0 + 8 * i));   // c0 of pairs 4i .. 4i+3
        const u32x2 b = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(src1 + 8 * i));
        u32x4 r = zip16(a, b);
        r = u32x4{ycocg2_rt<true>(variant, r.x), ycocg2_rt<true>(variant, r.y), ycocg2_rt<true>(variant, r.z),
                  ycocg2_rt<true>(variant, r.w)};
        store_streaming16(dst + 16 * i, r);
    } else if (i < vecs + singles) {
        const uint64_t p = 4 * vecs + (i - vecs);
        const uint32_t v = ycocg2_rt<true>(variant, ld_u16(src0 + 2 * p) | (ld_u16(src1 + 2 * p) << 16));
        st_u16(dst + 4 * p, v & 0xFFFFu);
        st_u16(dst + 4 * p + 2, v >> 16);
    }
}

// (c0, c1) pairs -> all c0, then all c1: lanes [0, vecs) eight pairs each
__global__ void __launch_bounds__(kColThreads)
split_endpoints_kernel(const uint8_t* in, uint8_t* out, uint64_t num_pairs, uint64_t vecs, uint64_t singles)
{
    const uint64_t i = workgroup_index() * kColThreads + threadIdx.x;
    uint8_t* out1 = out + 2 * num_pairs;
    if (i < vecs) {
        const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + 32 * i));
        const u32x4 r = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + 32 * i + 16));
        const u32x4 c0 = {(q.x & 0xFFFFu) | (q.y << 16), (q.z & 0xFFFFu) | (q.w << 16), (r.x & 0xFFFFu) | (r.y << 16),
                          (r.z & 0xFFFFu) | (r.w << 16)};
        const u32x4 c1 = {(q.x >> 16) | (q.y & 0xFFFF0000u), (q.z >> 16) | (q.w & 0xFFFF0000u), (r.x >> 16) | (r.y & 0xFFFF0000u),
                          (r.z >> 16) | (r.w & 0xFFFF0000u)};
        // two output streams per lane: the plain nontemporal store measured 0.786 here, the write-through one 0.756
        __builtin_nontemporal_store(c0, reinterpret_cast<u32x4*>(out + 16 * i));
        __builtin_nontemporal_store(c1, reinterpret_cast<u32x4*>(out1 + 16 * i));
    } else if (i < vecs + singles) {
        const uint64_t p = 8 * vecs + (i - vecs);
        st_u16(out + 2 * p, ld_u16(in + 4 * p));
        st_u16(out1 + 2 * p, ld_u16(in + 4 * p + 2));
    }
}

inline bool aligned(const void* p, int a) { return (reinterpret_cast<uintptr_t>(p) & (uintptr_t)(a - 1)) == 0; }

inline hipError_t col_grid(uint64_t lanes, dim3& g) { return grid_rows(lanes, kColThreads, g); }

}  // namespace

hipError_t launch_color565_ycocg(bool inverse, const void* in, void* out, uint64_t num_items, int variant, hipStream_t stream)
{
    if (variant < 0 || variant > 3)
        return hipErrorInvalidValue;
    if (num_items == 0)
        return hipSuccess;
    if (variant == kNone)   // decorrelate_batch_ptr.rs:351-356: a copy, nothing when in place
        return in == out ? hipSuccess : hipMemcpyAsync(out, in, num_items * 2, hipMemcpyDeviceToDevice, stream);
    const bool vec = aligned(in, 16) && aligned(out, 16);
    const uint64_t vecs = vec ? num_items / 8 : 0, singles = num_items - 8 * vecs;
    dim3 g;
    if (hipError_t e = col_grid(vecs + singles, g); e != hipSuccess)
        return e;
    auto k = inverse ? ycocg_array_kernel<true> : ycocg_array_kernel<false>;
    hipLaunchKernelGGL(k, g, dim3(kColThreads), 0, stream, static_cast<const uint8_t*>(in), static_cast<uint8_t*>(out), vecs,
                       singles, variant);
    return hipGetLastError();
}

hipError_t launch_color565_recorrelate_split(const void* src0, const void* src1, void* dst, uint64_t num_items, int variant,
                                             hipStream_t stream)
{
    if (variant < 0 || variant > 3 || (num_items & 1))
        return hipErrorInvalidValue;
    if (num_items == 0)
        return hipSuccess;
    const uint64_t pairs = num_items / 2;
    const bool vec = aligned(src0, 8) && aligned(src1, 8) && aligned(dst, 16);
    const uint64_t vecs = vec ? pairs / 4 : 0, singles = pairs - 4 * vecs;
    dim3 g;
    if (hipError_t e = col_grid(vecs + singles, g); e != hipSuccess)
        return e;
    hipLaunchKernelGGL(recorrelate_split_kernel, g, dim3(kColThreads), 0, stream, static_cast<const uint8_t*>(src0),
                       static_cast<const uint8_t*>(src1), static_cast<uint8_t*>(dst), vecs, singles, variant);
    return hipGetLastError();
}

hipError_t launch_split_565_color_endpoints(const void* in, void* out, uint64_t len_bytes, hipStream_t stream)
{
    if (len_bytes % 4 != 0)
        return hipErrorInvalidValue;
    if (len_bytes == 0)
        return hipSuccess;
    const uint64_t pairs = len_bytes / 4;
    // both halves of the output must be 16-byte aligned for the vector path: out and out + 2 * pairs
    const bool vec = aligned(in, 16) && aligned(out, 16) && (pairs % 8 == 0);
    const uint64_t vecs = vec ? pairs / 8 : 0, singles = pairs - 8 * vecs;
    dim3 g;
    if (hipError_t e = col_grid(vecs + singles, g); e != hipSuccess)
        return e;
    hipLaunchKernelGGL(split_endpoints_kernel, g, dim3(kColThreads), 0, stream, static_cast<const uint8_t*>(in),
                       static_cast<uint8_t*>(out), pairs, vecs, singles);
    return hipGetLastError();
}

}  // namespace dxtlt
