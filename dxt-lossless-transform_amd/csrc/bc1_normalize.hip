// bc1_normalize.hip -- gfx950 kernels for BC1 block normalisation as stand-alone operations (the version fused into
// the forward transform lives in bcn_kernels.hip; the per-block rule is bc1_normalize.h).
//
// Reference (experimental module, /root/reference/src/core/dxt-lossless-transform-bc1/src/experimental/
// normalize_blocks/normalize.rs): normalize_blocks :38-96, normalize_split_blocks_in_place :286-386,
// normalize_blocks_all_modes :417-481.  Element-wise, 8 bytes in / 8 bytes out per block per output: HBM bound,
// 2*len per output buffer.  One 16-byte vector per lane, one-shot grid (the structure that measured best for the
// transform kernels); pointers that are not 16-byte aligned, and the odd blocks at the end, take a lane-per-block
// path with natural-width or byte accesses.
#include <hip/hip_runtime.h>

#include "bc1_normalize.h"
#include "bcn_launch.h"
#include "launch_grid.h"
#include "streaming_store.h"

namespace dxtlt {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kNormThreads = 256;

__device__ __forceinline__ bool is_aligned(const void* p, int a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

__device__ __forceinline__ uint32_t load_u32(const uint8_t* p, bool aligned4)
{
    if (aligned4)
        return *reinterpret_cast<const uint32_t*>(p);
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

__device__ __forceinline__ void store_u32(uint8_t* p, uint32_t v, bool aligned4)
{
    if (aligned4) {
        *reinterpret_cast<uint32_t*>(p) = v;
    } else {
        p[0] = (uint8_t)v;
        p[1] = (uint8_t)(v >> 8);
        p[2] = (uint8_t)(v >> 16);
        p[3] = (uint8_t)(v >> 24);
    }
}

__device__ __forceinline__ void flag_if_any(bool changed, uint32_t* d_any)
{
    // one store per wave at most; every writer stores the same value
    if (d_any != nullptr && __ballot(changed) != 0 && (threadIdx.x & 63) == 0)
        *d_any = 1u;
}

// ---- AoS blocks ------------------------------------------------------------------------------------------
// lanes [0, pairs) take two blocks as one 16-byte vector (pointers 16-byte aligned); lanes [pairs, pairs + singles)
// take one block each with 4-byte or byte accesses
__global__ void __launch_bounds__(kNormThreads)
normalize_blocks_kernel(const uint8_t* in, uint8_t* out, uint64_t pairs, uint64_t singles, int mode)
{
    const uint64_t i = workgroup_index() * kNormThreads + threadIdx.x;
    if (i < pairs) {
        const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + 16 * i));
        uint32_t ca = q.x, xa = q.y, cb = q.z, xb = q.w;
        normalize_bc1_block_rt(mode, ca, xa);
        normalize_bc1_block_rt(mode, cb, xb);
        store_streaming16(out + 16 * i, u32x4{ca, xa, cb, xb});
    } else if (i < pairs + singles) {
        const uint64_t b = 2 * pairs + (i - pairs);
        const bool a4 = is_aligned(in, 4) && is_aligned(out, 4);
        uint32_t c = load_u32(in + 8 * b, a4), x = load_u32(in + 8 * b + 4, a4);
        normalize_bc1_block_rt(mode, c, x);
        store_u32(out + 8 * b, c, a4);
        store_u32(out + 8 * b + 4, x, a4);
    }
}

// in -> three outputs (None, Color0Only, ReplicateColor), classification done once per block
__global__ void __launch_bounds__(kNormThreads)
normalize_all_modes_kernel(const uint8_t* in, uint8_t* out0, uint8_t* out1, uint8_t* out2, uint64_t pairs, uint64_t singles,
                           uint32_t* d_any)
{
    const uint64_t i = workgroup_index() * kNormThreads + threadIdx.x;
    bool changed = false;
    if (i < pairs) {
        const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + 16 * i));
        // the transparent rewrite goes to every output, the `None` one included (normalize.rs:447-454)
        u32x4 z = q, a = q, b = q;
        uint32_t sa = 0, sb = 0;
        const int ca = classify_bc1_block(q.x, q.y, sa), cb = classify_bc1_block(q.z, q.w, sb);
        if (ca == kBlockTransparent) { z.x = z.y = a.x = a.y = b.x = b.y = 0xFFFFFFFFu; }
        if (ca == kBlockSolid) { a.x = sa; a.y = 0; b.x = sa | (sa << 16); b.y = 0; }
        if (cb == kBlockTransparent) { z.z = z.w = a.z = a.w = b.z = b.w = 0xFFFFFFFFu; }
        if (cb == kBlockSolid) { a.z = sb; a.w = 0; b.z = sb | (sb << 16); b.w = 0; }
        changed = ca != kBlockUnchanged || cb != kBlockUnchanged;
        __builtin_nontemporal_store(z, reinterpret_cast<u32x4*>(out0 + 16 * i));
        __builtin_nontemporal_store(a, reinterpret_cast<u32x4*>(out1 + 16 * i));
        __builtin_nontemporal_store(b, reinterpret_cast<u32x4*>(out2 + 16 * i));
    } else if (i < pairs + singles) {
        const uint64_t blk = 2 * pairs + (i - pairs);
        const bool a4 = is_aligned(in, 4) && is_aligned(out0, 4) && is_aligned(out1, 4) && is_aligned(out2, 4);
        const uint32_t c = load_u32(in + 8 * blk, a4), x = load_u32(in + 8 * blk + 4, a4);
        uint32_t s = 0;
        const int cls = classify_bc1_block(c, x, s);
        uint32_t c0 = c, x0 = x, c1 = c, x1 = x, c2 = c, x2 = x;
        if (cls == kBlockTransparent) { c0 = x0 = c1 = x1 = c2 = x2 = 0xFFFFFFFFu; }
        if (cls == kBlockSolid) { c1 = s; c2 = s | (s << 16); x1 = x2 = 0; }
        changed = cls != kBlockUnchanged;
        store_u32(out0 + 8 * blk, c0, a4);
        store_u32(out0 + 8 * blk + 4, x0, a4);
        store_u32(out1 + 8 * blk, c1, a4);
        store_u32(out1 + 8 * blk + 4, x1, a4);
        store_u32(out2 + 8 * blk, c2, a4);
        store_u32(out2 + 8 * blk + 4, x2, a4);
    }
    flag_if_any(changed, d_any);
}

// only the "would anything change" flag (the auto transform uses it to skip the normalisation candidates)
__global__ void __launch_bounds__(kNormThreads)
any_normalizable_kernel(const uint8_t* in, uint64_t pairs, uint64_t singles, uint32_t* d_any)
{
    const uint64_t i = workgroup_index() * kNormThreads + threadIdx.x;
    bool changed = false;
    uint32_t s = 0;
    if (i < pairs) {
        const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + 16 * i));
        changed = classify_bc1_block(q.x, q.y, s) != kBlockUnchanged || classify_bc1_block(q.z, q.w, s) != kBlockUnchanged;
    } else if (i < pairs + singles) {
        const uint64_t blk = 2 * pairs + (i - pairs);
        const bool a4 = is_aligned(in, 4);
        changed = classify_bc1_block(load_u32(in + 8 * blk, a4), load_u32(in + 8 * blk + 4, a4), s) != kBlockUnchanged;
    }
    flag_if_any(changed, d_any);
}

// ---- split blocks: colours[4 * n] and indices[4 * n], in place -----------------------------------------------
// lanes [0, quads) take four blocks (one 16-byte vector of each array); the rest one block each
__global__ void __launch_bounds__(kNormThreads)
normalize_split_kernel(uint8_t* colours, uint8_t* indices, uint64_t quads, uint64_t singles, int mode)
{
    const uint64_t i = workgroup_index() * kNormThreads + threadIdx.x;
    if (i < quads) {
        const u32x4 cv = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(colours + 16 * i));
        const u32x4 xv = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(indices + 16 * i));
        uint32_t c[4] = {cv.x, cv.y, cv.z, cv.w}, x[4] = {xv.x, xv.y, xv.z, xv.w};
        int any = kBlockUnchanged;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            any |= normalize_bc1_block_rt(mode, c[k], x[k]);
        if (any != kBlockUnchanged) {   // in place: untouched vectors need no write
            *reinterpret_cast<u32x4*>(colours + 16 * i) = u32x4{c[0], c[1], c[2], c[3]};
            *reinterpret_cast<u32x4*>(indices + 16 * i) = u32x4{x[0], x[1], x[2], x[3]};
        }
    } else if (i < quads + singles) {
        const uint64_t b = 4 * quads + (i - quads);
        const bool a4 = is_aligned(colours, 4) && is_aligned(indices, 4);
        uint32_t c = load_u32(colours + 4 * b, a4), x = load_u32(indices + 4 * b, a4);
        if (normalize_bc1_block_rt(mode, c, x) != kBlockUnchanged) {
            store_u32(colours + 4 * b, c, a4);
            store_u32(indices + 4 * b, x, a4);
        }
    }
}

inline bool host_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline hipError_t grid_for(uint64_t lanes, dim3& grid) { return grid_rows(lanes, kNormThreads, grid); }

}  // namespace

hipError_t launch_normalize_bc1_blocks(const void* in, void* out, uint64_t num_blocks, int mode, hipStream_t stream)
{
    if (mode < kNormNone || mode > kNormReplicateColor)
        return hipErrorInvalidValue;
    if (num_blocks == 0)
        return hipSuccess;
    if (mode == kNormNone)   // normalize.rs:53-64: a copy, or nothing when in place
        return in == out ? hipSuccess : hipMemcpyAsync(out, in, num_blocks * 8, hipMemcpyDeviceToDevice, stream);
    const bool vec = host_aligned16(in) && host_aligned16(out);
    const uint64_t pairs = vec ? num_blocks / 2 : 0, singles = num_blocks - 2 * pairs;
    dim3 grid;
    if (hipError_t e = grid_for(pairs + singles, grid); e != hipSuccess)
        return e;
    hipLaunchKernelGGL(normalize_blocks_kernel, grid, dim3(kNormThreads), 0, stream, static_cast<const uint8_t*>(in),
                       static_cast<uint8_t*>(out), pairs, singles, mode);
    return hipGetLastError();
}

hipError_t launch_normalize_bc1_split(void* colours, void* indices, uint64_t num_blocks, int mode, hipStream_t stream)
{
    if (mode < kNormNone || mode > kNormReplicateColor)
        return hipErrorInvalidValue;
    if (num_blocks == 0 || mode == kNormNone)   // normalize.rs:293-295
        return hipSuccess;
    const bool vec = host_aligned16(colours) && host_aligned16(indices);
    const uint64_t quads = vec ? num_blocks / 4 : 0, singles = num_blocks - 4 * quads;
    dim3 grid;
    if (hipError_t e = grid_for(quads + singles, grid); e != hipSuccess)
        return e;
    hipLaunchKernelGGL(normalize_split_kernel, grid, dim3(kNormThreads), 0, stream, static_cast<uint8_t*>(colours),
                       static_cast<uint8_t*>(indices), quads, singles, mode);
    return hipGetLastError();
}

hipError_t launch_normalize_bc1_all_modes(const void* in, void* const out[3], uint64_t num_blocks, uint32_t* d_any,
                                          hipStream_t stream)
{
    if (num_blocks == 0)
        return hipSuccess;
    const bool vec = host_aligned16(in) && host_aligned16(out[0]) && host_aligned16(out[1]) && host_aligned16(out[2]);
    const uint64_t pairs = vec ? num_blocks / 2 : 0, singles = num_blocks - 2 * pairs;
    dim3 grid;
    if (hipError_t e = grid_for(pairs + singles, grid); e != hipSuccess)
        return e;
    // 16 bytes read and 48 written per lane: six workgroups per CU instead of eight (1 GiB of blocks: 782 -> 696 us;
    // five 731, four 688, three 1011; profiles/r02_o_wgs_per_cu.txt)
    hipLaunchKernelGGL(normalize_all_modes_kernel, grid, dim3(kNormThreads), lds_pad_for_wgs_per_cu(wgs_per_cu_or(6), kNormThreads, 0), stream,
                       static_cast<const uint8_t*>(in), static_cast<uint8_t*>(out[0]), static_cast<uint8_t*>(out[1]),
                       static_cast<uint8_t*>(out[2]), pairs, singles, d_any);
    return hipGetLastError();
}

hipError_t launch_bc1_any_normalizable(const void* in, uint64_t num_blocks, uint32_t* d_any, hipStream_t stream)
{
    if (num_blocks == 0)
        return hipSuccess;
    const bool vec = host_aligned16(in);
    const uint64_t pairs = vec ? num_blocks / 2 : 0, singles = num_blocks - 2 * pairs;
    dim3 grid;
    if (hipError_t e = grid_for(pairs + singles, grid); e != hipSuccess)
        return e;
    hipLaunchKernelGGL(any_normalizable_kernel, grid, dim3(kNormThreads), 0, stream, static_cast<const uint8_t*>(in),
                       pairs, singles, d_any);
    return hipGetLastError();
}

}  // namespace dxtlt
