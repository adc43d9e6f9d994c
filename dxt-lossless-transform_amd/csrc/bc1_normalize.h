// bc1_normalize.h -- BC1 block normalisation in registers (device code, shared by bcn_kernels.hip and
// bc1_normalize.hip).
//
// Reference (experimental module, paths under /root/reference/src/core/):
//   dxt-lossless-transform-bc1/src/experimental/normalize_blocks/normalize.rs:38-96   normalize_blocks
//   ... :118-188  normalize_blocks_impl (block classification)    :214-258  write_normalized_solid_color_block
//   dxt-lossless-transform-bc1/src/util/bc1_decode.rs:42-100       decode_bc1_block
//   dxt-lossless-transform-common/src/color_565/mod.rs:108-116 (from_rgb), :153-186 (red/green/blue expansion)
//   dxt-lossless-transform-common/src/decoded_4x4_block.rs:107     has_identical_pixels
//
// A block is decoded (two RGB565 endpoints expanded by bit replication, two interpolated colours -- thirds when
// c0 > c1, else the midpoint and transparent black), and
//   * if all 16 pixels are identical and transparent            -> the block becomes eight 0xFF bytes;
//   * if all 16 pixels are one opaque colour that survives 8888 -> 565 -> 8888 unchanged
//                                                                -> (colour, 0, indices 0)      [Color0Only]
//                                                                   (colour, colour, indices 0) [ReplicateColor]
//   * anything else is left as it is.
// The reference decodes 16 pixels and compares them; here the indices and endpoints are inspected instead (a pixel is
// palette[index], so "all pixels equal" == "all used palette entries equal" -- see classify_bc1_block), which needs
// no loop over pixels and lets nearly every block of real data leave after two compares.
// tests/test_normalize.py builds this header for the host and compares it with the oracle's pixel-by-pixel statement.
#pragma once
#include <stdint.h>

namespace dxtlt {

// kNormTransparentOnly is internal: normalize_blocks_all_modes rewrites fully transparent blocks in EVERY output,
// the `None` one included (normalize.rs:447-454), and transform_bc1_auto_with_normalization estimates its `None`
// candidates on that buffer -- so the auto path needs "transparent blocks only" as a fused mode of its own.
enum : int { kNormNone = 0, kNormColor0Only = 1, kNormReplicateColor = 2, kNormTransparentOnly = 3 };

enum : int { kBlockUnchanged = 0, kBlockTransparent = 1, kBlockSolid = 2 };

// r | g << 8 | b << 16 of an RGB565 value, channels expanded by bit replication
__host__ __device__ inline uint32_t expand_565(uint32_t v)
{
    const uint32_t r5 = (v >> 11) & 31, g6 = (v >> 5) & 63, b5 = v & 31;
    const uint32_t r = (r5 << 3) | (r5 >> 2), g = (g6 << 2) | (g6 >> 4), b = (b5 << 3) | (b5 >> 2);
    return r | (g << 8) | (b << 16);
}

// Classifies one block (colours = c0 | c1 << 16, little-endian field order) and, for a solid block, returns its
// colour as RGB565.
//
// When c0 != c1 the four palette entries are pairwise different (some channel differs by >= 4 after expansion, so the
// thirds and the midpoint fall strictly between the endpoints and apart from each other; the transparent entry
// differs in alpha) -- tests/test_normalize.py checks this for every pair of channel values.  Hence all pixels are
// equal exactly when
//   (a) all sixteen indices are the same value k, or
//   (b) c0 == c1 (three-colour mode, entries 0 = 1 = 2) and index 3 does not occur,
// and almost every block of real data leaves after two compares.  Only case (a) with k >= 2 has to interpolate.
__host__ __device__ inline int classify_bc1_block(uint32_t colours, uint32_t indices, uint32_t& solid565)
{
    const uint32_t c0 = colours & 0xFFFFu, c1 = colours >> 16;
    const uint32_t k = indices & 3u;
    const bool single = indices == k * 0x55555555u;
    if (!single) {
        if (c0 != c1 || (indices & (indices >> 1) & 0x55555555u) != 0)
            return kBlockUnchanged;
        solid565 = c0;   // (b): every pixel is the expansion of c0, which converts back to c0
        return kBlockSolid;
    }
    if (k < 2) {         // every pixel is an endpoint colour
        solid565 = k == 0 ? c0 : c1;
        return kBlockSolid;
    }
    const bool four = c0 > c1;
    if (k == 3 && !four)
        return kBlockTransparent;
    // every pixel is an interpolated colour: normalisable only if it survives 8888 -> 565 -> 8888
    const uint32_t e0 = expand_565(c0), e1 = expand_565(c1);
    const uint32_t r0 = e0 & 255, g0 = (e0 >> 8) & 255, b0 = e0 >> 16;
    const uint32_t r1 = e1 & 255, g1 = (e1 >> 8) & 255, b1 = e1 >> 16;
    uint32_t r, g, b;
    if (!four) {
        r = (r0 + r1) / 2, g = (g0 + g1) / 2, b = (b0 + b1) / 2;
    } else if (k == 2) {
        r = (2 * r0 + r1) / 3, g = (2 * g0 + g1) / 3, b = (2 * b0 + b1) / 3;
    } else {
        r = (r0 + 2 * r1) / 3, g = (g0 + 2 * g1) / 3, b = (b0 + 2 * b1) / 3;
    }
    solid565 = ((r & 0xF8u) << 8) | ((g & 0xFCu) << 3) | (b >> 3);
    return expand_565(solid565) == (r | (g << 8) | (b << 16)) ? kBlockSolid : kBlockUnchanged;
}

// Normalises one block in place; MODE is kNormColor0Only, kNormReplicateColor or kNormTransparentOnly.  Returns the
// block's class.
template <int MODE>
__host__ __device__ inline int normalize_bc1_block(uint32_t& colours, uint32_t& indices)
{
    if (MODE == kNormNone)
        return kBlockUnchanged;
    uint32_t solid = 0;
    const int cls = classify_bc1_block(colours, indices, solid);
    if (cls == kBlockTransparent) {
        colours = 0xFFFFFFFFu;
        indices = 0xFFFFFFFFu;
    } else if (cls == kBlockSolid && MODE != kNormTransparentOnly) {
        colours = MODE == kNormReplicateColor ? solid | (solid << 16) : solid;
        indices = 0;
    }
    return cls;
}

__host__ __device__ inline int normalize_bc1_block_rt(int mode, uint32_t& colours, uint32_t& indices)
{
    if (mode == kNormColor0Only)
        return normalize_bc1_block<kNormColor0Only>(colours, indices);
    if (mode == kNormReplicateColor)
        return normalize_bc1_block<kNormReplicateColor>(colours, indices);
    if (mode == kNormTransparentOnly)
        return normalize_bc1_block<kNormTransparentOnly>(colours, indices);
    return kBlockUnchanged;
}

}  // namespace dxtlt
