// auto_kernels.hip -- transform_bcN_auto, candidate phase: every endpoint section the size estimator will be shown,
// from ONE read of the input.
//
// The reference tries its candidates one full transform at a time and estimates the endpoint section(s) only -- the
// index sections are the same for every candidate and are left out (core/dxt-lossless-transform-bc1/src/transform/
// transform_auto.rs:245-256; BC2 / BC3 twins).  What differs between candidates is
//     the colour section   (4 bytes per block): YCoCg-R variant x {c0/c1 pairs, all c0 then all c1}
//     BC3's alpha endpoints (2 bytes per block): {a0/a1 pairs, all a0 then all a1}
// so 4 (fast search: variants None, Variant1) or 8 colour sections and, for BC3, 2 alpha sections cover all 4 / 8
// (BC1, BC2) or 8 / 16 (BC3) candidates.  This kernel writes them all into an arena:
//     [alpha pairs 2N][alpha split 2N]                         BC3 only
//     for variant in (None, Variant1[, Variant2, Variant3]):  [colour pairs 4N][colour split 4N]
// One 16-byte vector per lane (two BC1 blocks or one BC2 / BC3 block), no LDS: a lane's piece of every section is 2-8
// contiguous bytes and a wave instruction writes 128-512 contiguous bytes of one section.  Traffic: len read once,
// (sections x 4 + 4 [BC3]) bytes per block written -- BC1, fast search: 3 x len against 8 x len for four full
// transforms.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "auto_launch.h"
#include "bcn_launch.h"
#include "launch_grid.h"
#include "ycocg_swar.h"

namespace dxtlt {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

namespace {

template <int FMT, int VARIANT>
__device__ __forceinline__ void colour_sections(uint8_t* __restrict__ pairs, uint64_t n, uint64_t lane_vec, uint32_t ca, uint32_t cb)
{
    // pairs: [c0 c1] dwords at 4 * block; split: c0 at 2 * block, c1 at 2 * n + 2 * block (behind the pairs section)
    uint8_t* split = pairs + 4 * n;
    const uint32_t da = decorrelate2<VARIANT>(ca);
    if constexpr (FMT == kBc1) {
        const uint32_t db = decorrelate2<VARIANT>(cb);
        *reinterpret_cast<u32x2*>(pairs + 8 * lane_vec) = u32x2{da, db};
        *reinterpret_cast<uint32_t*>(split + 4 * lane_vec) = (da & 0xFFFFu) | (db << 16);
        *reinterpret_cast<uint32_t*>(split + 2 * n + 4 * lane_vec) = (da >> 16) | (db & 0xFFFF0000u);
    } else {
        *reinterpret_cast<uint32_t*>(pairs + 4 * lane_vec) = da;
        *reinterpret_cast<uint16_t*>(split + 2 * lane_vec) = (uint16_t)da;
        *reinterpret_cast<uint16_t*>(split + 2 * n + 2 * lane_vec) = (uint16_t)(da >> 16);
    }
}

// n = blocks; the kernel covers the first `vectors` 16-byte vectors (BC1: an odd last block is handled by the caller's
// tail launch with vectors = 0 semantics -- see launch_auto_candidates)
template <int FMT, bool ALL>
__global__ void __launch_bounds__(256)
auto_candidates_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ arena, uint64_t n, uint64_t vectors)
{
    const uint64_t v = workgroup_index() * 256 + threadIdx.x;
    if (v >= vectors)
        return;
    const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + 16 * v));
    uint8_t* colour0 = arena;
    uint32_t ca, cb = 0;
    if constexpr (FMT == kBc1) {
        ca = q.x;
        cb = q.z;
    } else {
        ca = q.z;
        if constexpr (FMT == kBc3) {
            // alpha endpoints: pairs section, then split section
            *reinterpret_cast<uint16_t*>(arena + 2 * v) = (uint16_t)q.x;
            arena[2 * n + v] = (uint8_t)q.x;
            arena[3 * n + v] = (uint8_t)(q.x >> 8);
            colour0 = arena + 4 * n;
        }
    }
    colour_sections<FMT, kNone>(colour0, n, v, ca, cb);
    colour_sections<FMT, kVar1>(colour0 + 8 * n, n, v, ca, cb);
    if constexpr (ALL) {
        colour_sections<FMT, kVar2>(colour0 + 16 * n, n, v, ca, cb);
        colour_sections<FMT, kVar3>(colour0 + 24 * n, n, v, ca, cb);
    }
}

// the odd last block of a BC1 buffer (half a vector): one lane, scalar accesses
template <bool ALL>
__global__ void auto_candidates_bc1_last_block(const uint8_t* __restrict__ in, uint8_t* __restrict__ arena, uint64_t n)
{
    const uint64_t b = n - 1;
    const uint32_t c = *reinterpret_cast<const uint32_t*>(in + 8 * b);
    auto emit = [&](uint8_t* pairs, uint32_t d) {
        *reinterpret_cast<uint32_t*>(pairs + 4 * b) = d;
        *reinterpret_cast<uint16_t*>(pairs + 4 * n + 2 * b) = (uint16_t)d;
        *reinterpret_cast<uint16_t*>(pairs + 6 * n + 2 * b) = (uint16_t)(d >> 16);
    };
    emit(arena, decorrelate2<kNone>(c));
    emit(arena + 8 * n, decorrelate2<kVar1>(c));
    if constexpr (ALL) {
        emit(arena + 16 * n, decorrelate2<kVar2>(c));
        emit(arena + 24 * n, decorrelate2<kVar3>(c));
    }
}

}  // namespace

uint64_t auto_arena_bytes(Format fmt, bool all_variants, uint64_t blocks)
{
    return ((fmt == kBc3 ? 4u : 0u) + (all_variants ? 32u : 16u)) * blocks;
}

uint64_t auto_section_offset(Format fmt, uint64_t blocks, int variant, bool split_colour)
{
    return ((fmt == kBc3 ? 4u : 0u) + 8u * (uint64_t)variant + (split_colour ? 4u : 0u)) * blocks;
}

uint64_t auto_alpha_section_offset(uint64_t blocks, bool split_alpha) { return split_alpha ? 2 * blocks : 0; }

hipError_t launch_auto_candidates(Format fmt, bool all_variants, const void* d_in, void* d_arena, uint64_t blocks,
                                  hipStream_t stream)
{
    if (blocks == 0)
        return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(d_in) & 15) != 0 || (reinterpret_cast<uintptr_t>(d_arena) & 15) != 0)
        return hipErrorInvalidValue;
    const uint8_t* in = static_cast<const uint8_t*>(d_in);
    uint8_t* arena = static_cast<uint8_t*>(d_arena);
    const uint64_t vectors = fmt == kBc1 ? blocks / 2 : blocks;
    if (vectors > 0) {
        dim3 grid;
        if (hipError_t e = grid_rows(vectors, 256, grid); e != hipSuccess)
            return e;
#define DXTLT_AUTO_LAUNCH(F, A) \
        hipLaunchKernelGGL((auto_candidates_kernel<F, A>), grid, dim3(256), 0, stream, in, arena, blocks, vectors)
        if (fmt == kBc1) { if (all_variants) DXTLT_AUTO_LAUNCH(kBc1, true); else DXTLT_AUTO_LAUNCH(kBc1, false); }
        else if (fmt == kBc2) { if (all_variants) DXTLT_AUTO_LAUNCH(kBc2, true); else DXTLT_AUTO_LAUNCH(kBc2, false); }
        else { if (all_variants) DXTLT_AUTO_LAUNCH(kBc3, true); else DXTLT_AUTO_LAUNCH(kBc3, false); }
#undef DXTLT_AUTO_LAUNCH
        if (hipError_t e = hipGetLastError(); e != hipSuccess)
            return e;
    }
    if (fmt == kBc1 && (blocks & 1)) {
        if (all_variants)
            hipLaunchKernelGGL(auto_candidates_bc1_last_block<true>, dim3(1), dim3(1), 0, stream, in, arena, blocks);
        else
            hipLaunchKernelGGL(auto_candidates_bc1_last_block<false>, dim3(1), dim3(1), 0, stream, in, arena, blocks);
        return hipGetLastError();
    }
    return hipSuccess;
}

}  // namespace dxtlt
