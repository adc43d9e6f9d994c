// bc23_normalize.h -- BC2 / BC3 block normalisation in registers (device code; also built for the host by the tests).
//
// Reference (experimental modules, paths under /root/reference/src/core/):
//   dxt-lossless-transform-bc2/src/experimental/normalize_blocks/normalize.rs:118-160, 270-318
//   dxt-lossless-transform-bc3/src/experimental/normalize_blocks/normalize.rs:176-222 (alpha), 241-280 (colour),
//       325-385 (classification)
//   dxt-lossless-transform-bc2/src/util/bc2_decode.rs:14-95, dxt-lossless-transform-bc3/src/util/bc3_decode.rs:12-125
//
// Colour half (BC2 and BC3 alike: always four colours, alpha ignored): if all 16 pixels have one RGB value that
// survives 8888 -> 565 -> 8888, the half becomes (colour, 0, indices 0) or (colour, colour, indices 0).
// BC3 alpha half: if all 16 decoded alpha values are equal, it becomes (alpha, 0, indices 0), or for alpha == 255
// eight 0xFF bytes / (0, 0, indices 0xFF) depending on the mode.  BC2's explicit alpha is never touched.
//
// As in bc1_normalize.h the reference decodes 16 pixels and compares; here the decision comes from the indices and
// endpoints: when c0 != c1 the four colours are pairwise different, and when the alpha endpoints are far enough
// apart the eight alpha values are pairwise different, so "all equal" needs all indices equal -- anything else
// leaves after a few compares.  Only neighbouring alpha endpoints (|a0 - a1| < 7) walk the sixteen indices.
#pragma once
#include <stdint.h>

#include "bc1_normalize.h"

namespace dxtlt {

enum : int { kAlphaNone = 0, kAlphaUniformZeroIndices = 1, kAlphaOpaqueFillAll = 2, kAlphaOpaqueZeroAlphaMaxIndices = 3 };

// four-colour palette, alpha ignored: true when every pixel has the same, round-trippable colour
__host__ __device__ inline bool solid_colour_4c(uint32_t colours, uint32_t indices, uint32_t& solid565)
{
    const uint32_t c0 = colours & 0xFFFFu, c1 = colours >> 16;
    const uint32_t k = indices & 3u;
    if (indices != k * 0x55555555u) {   // several index values: equal pixels need equal palette entries, i.e. c0 == c1
        solid565 = c0;
        return c0 == c1;
    }
    if (k < 2) {
        solid565 = k == 0 ? c0 : c1;
        return true;
    }
    const uint32_t e0 = expand_565(c0), e1 = expand_565(c1);
    const uint32_t r0 = e0 & 255, g0 = (e0 >> 8) & 255, b0 = e0 >> 16;
    const uint32_t r1 = e1 & 255, g1 = (e1 >> 8) & 255, b1 = e1 >> 16;
    uint32_t r, g, b;
    if (k == 2) {
        r = (2 * r0 + r1) / 3, g = (2 * g0 + g1) / 3, b = (2 * b0 + b1) / 3;
    } else {
        r = (r0 + 2 * r1) / 3, g = (g0 + 2 * g1) / 3, b = (b0 + 2 * b1) / 3;
    }
    solid565 = ((r & 0xF8u) << 8) | ((g & 0xFCu) << 3) | (b >> 3);
    return expand_565(solid565) == (r | (g << 8) | (b << 16));
}

__host__ __device__ inline uint32_t bc3_alpha_value(uint32_t a0, uint32_t a1, uint32_t i)
{
    if (i == 0) return a0;
    if (i == 1) return a1;
    if (a0 > a1) return ((8 - i) * a0 + (i - 1) * a1) / 7;
    if (i < 6) return ((6 - i) * a0 + (i - 1) * a1) / 5;
    return i == 6 ? 0u : 255u;
}

// w0 = bytes 0..3 of the block (a0, a1, index bytes 0-1), w1 = bytes 4..7 (index bytes 2-5)
__host__ __device__ inline bool uniform_alpha_bc3(uint32_t w0, uint32_t w1, uint32_t& alpha)
{
    const uint32_t a0 = w0 & 255u, a1 = (w0 >> 8) & 255u;
    const uint64_t bits = (uint64_t)(w0 >> 16) | ((uint64_t)w1 << 16);   // sixteen 3-bit indices
    const uint32_t k = (uint32_t)bits & 7u;
    alpha = bc3_alpha_value(a0, a1, k);
    if (bits == (uint64_t)k * 0x249249249249ull)
        return true;
    // eight pairwise different table entries: interpolation steps of at least 1, constants 0 / 255 not hit
    if (a0 > a1 ? a0 - a1 >= 7 : (a1 - a0 >= 5 && a0 != 0 && a1 != 255))
        return false;
    // neighbouring endpoints: table entries may coincide.  One wave nearly always holds such a lane, so this path is
    // kept short: which table entries equal the first pixel's alpha (8 compares), which index values occur (16
    // shifts), and no occurring value may point at a different entry.
    uint32_t equal = 0;
#pragma unroll
    for (uint32_t v = 0; v < 8; ++v)
        equal |= (bc3_alpha_value(a0, a1, v) == alpha ? 1u : 0u) << v;
    const uint32_t lo = (uint32_t)bits & 0xFFFFFFu, hi = (uint32_t)(bits >> 24);   // eight indices each
    uint32_t used = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        used |= (1u << ((lo >> (3 * i)) & 7u)) | (1u << ((hi >> (3 * i)) & 7u));
    return (used & ~equal) == 0;
}

// colour half in place; returns true when it changed class (solid)
__host__ __device__ inline bool normalize_colour_half_4c(int color_mode, uint32_t& colours, uint32_t& indices)
{
    uint32_t solid = 0;
    if (color_mode == kNormNone || !solid_colour_4c(colours, indices, solid))
        return false;
    colours = color_mode == kNormReplicateColor ? solid | (solid << 16) : solid;
    indices = 0;
    return true;
}

// BC3 alpha half in place (w0 = bytes 0..3, w1 = bytes 4..7)
__host__ __device__ inline bool normalize_alpha_half_bc3(int alpha_mode, uint32_t& w0, uint32_t& w1)
{
    uint32_t alpha = 0;
    if (alpha_mode == kAlphaNone || !uniform_alpha_bc3(w0, w1, alpha))
        return false;
    if (alpha == 255 && alpha_mode == kAlphaOpaqueFillAll) {
        w0 = w1 = 0xFFFFFFFFu;
    } else if (alpha == 255 && alpha_mode == kAlphaOpaqueZeroAlphaMaxIndices) {
        w0 = 0xFFFF0000u;
        w1 = 0xFFFFFFFFu;
    } else {
        w0 = alpha;
        w1 = 0;
    }
    return true;
}

// one whole block, q = {bytes 0-3, 4-7, 8-11, 12-15}; FMT 2 or 3
template <int FMT>
__host__ __device__ inline void normalize_block_bc23(int alpha_mode, int color_mode, uint32_t (&q)[4])
{
    if (FMT == 3)
        normalize_alpha_half_bc3(alpha_mode, q[0], q[1]);
    normalize_colour_half_4c(color_mode, q[2], q[3]);
}

}  // namespace dxtlt
