// bcn_decode.h -- BC1 / BC2 / BC3 block -> sixteen RGBA8888 pixels, in registers (device code; also built for the host
// by the tests).
//
// Reference (paths under /root/reference/src/core/):
//   dxt-lossless-transform-bc1/src/util/bc1_decode.rs:42-100   decode_bc1_block ("ideal" DX9 rounding: /3 and /2 on
//                                                                the 8-bit expansions; three-colour mode when c0 <= c1)
//   dxt-lossless-transform-bc2/src/util/bc2_decode.rs:44-124   decode_bc2_block (always four colours; alpha = nibble * 17)
//   dxt-lossless-transform-bc3/src/util/bc3_decode.rs:43-175   decode_bc3_block (always four colours; eight / six value
//                                                                alpha table, 3-bit indices)
//   dxt-lossless-transform-common/src/decoded_4x4_block.rs:56   Decoded4x4Block: pixels[16], row-major, {r, g, b, a} bytes
//
// A pixel is one little-endian dword r | g << 8 | b << 16 | a << 24; a row of the block is four of them.  The palette
// is kept channel-wise (the four reds in one register, ...), so that one byte permute (v_perm_b32) picks a channel
// of four pixels at once from their 2-bit indices, and three more rounds of permutes weave the channels into pixels:
// 15 instructions per row instead of the ~28 a compare-and-select chain costs.
#pragma once
#include <stdint.h>

namespace dxtlt {

// D.byte[i] = selector byte i picks: 0..3 = byte of `lo`, 4..7 = byte of `hi` (v_perm_b32 with S0 = hi, S1 = lo)
__host__ __device__ inline uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const uint64_t src = ((uint64_t)hi << 32) | lo;
    uint32_t out = 0;
    for (int i = 0; i < 4; ++i)
        out |= (uint32_t)((src >> (8 * ((sel >> (8 * i)) & 7u))) & 0xFFu) << (8 * i);
    return out;
#endif
}

// x / 3 for x <= 765, x / 5 for x <= 1020, x / 7 for x <= 1785 (checked exhaustively in tests/test_decode.py)
__host__ __device__ inline uint32_t div3_small(uint32_t x) { return (x * 683u) >> 11; }
__host__ __device__ inline uint32_t div5_small(uint32_t x) { return (x * 13108u) >> 16; }
__host__ __device__ inline uint32_t div7_small(uint32_t x) { return (x * 9363u) >> 16; }

struct Palette4 {
    uint32_t r, g, b, a;   // byte k = channel of palette entry k
};

// colours = c0 | c1 << 16.  BC1_MODES: three-colour + transparent entry when c0 <= c1 (BC1); otherwise always four.
template <bool BC1_MODES>
__host__ __device__ inline Palette4 colour_palette(uint32_t colours)
{
    const uint32_t c0 = colours & 0xFFFFu, c1 = colours >> 16;
    // 565 -> 888 by bit replication (color_565.rs red()/green()/blue())
    const uint32_t r0 = ((c0 >> 8) & 0xF8u) | (c0 >> 13), r1 = ((c1 >> 8) & 0xF8u) | (c1 >> 13);
    const uint32_t g0 = ((c0 >> 3) & 0xFCu) | ((c0 >> 9) & 3u), g1 = ((c1 >> 3) & 0xFCu) | ((c1 >> 9) & 3u);
    const uint32_t b0 = ((c0 << 3) & 0xF8u) | ((c0 >> 2) & 7u), b1 = ((c1 << 3) & 0xF8u) | ((c1 >> 2) & 7u);
    uint32_t r2 = div3_small(2 * r0 + r1), g2 = div3_small(2 * g0 + g1), b2 = div3_small(2 * b0 + b1);
    uint32_t r3 = div3_small(r0 + 2 * r1), g3 = div3_small(g0 + 2 * g1), b3 = div3_small(b0 + 2 * b1);
    uint32_t alpha = 0xFFFFFFFFu;
    if (BC1_MODES && c0 <= c1) {
        r2 = (r0 + r1) >> 1, g2 = (g0 + g1) >> 1, b2 = (b0 + b1) >> 1;
        r3 = g3 = b3 = 0;
        alpha = 0x00FFFFFFu;
    }
    Palette4 p;
    p.r = r0 | (r1 << 8) | (r2 << 16) | (r3 << 24);
    p.g = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
    p.b = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    p.a = alpha;
    return p;
}

// four pixels from four channel words (byte i = that channel of pixel i)
__host__ __device__ inline void weave_row(uint32_t r4, uint32_t g4, uint32_t b4, uint32_t a4, uint32_t px[4])
{
    const uint32_t rg01 = byte_perm(g4, r4, 0x05010400u), rg23 = byte_perm(g4, r4, 0x07030602u);
    const uint32_t ba01 = byte_perm(a4, b4, 0x05010400u), ba23 = byte_perm(a4, b4, 0x07030602u);
    px[0] = byte_perm(ba01, rg01, 0x05040100u);
    px[1] = byte_perm(ba01, rg01, 0x07060302u);
    px[2] = byte_perm(ba23, rg23, 0x05040100u);
    px[3] = byte_perm(ba23, rg23, 0x07060302u);
}

// byte i = 2-bit index of pixel i of a row, from the row's index byte (two shift-or-mask rounds: a single multiply by
// 1 + 2^6 + 2^12 + 2^18 would add overlapping copies and carry into the wanted bits)
__host__ __device__ inline uint32_t spread_2bit(uint32_t index_byte)
{
    const uint32_t t = (index_byte | (index_byte << 12)) & 0x000F000Fu;
    return (t | (t << 6)) & 0x03030303u;
}

// byte i = 3-bit index of pixel i of a row, from the row's twelve index bits
__host__ __device__ inline uint32_t spread_3bit(uint32_t index_bits12)
{
    const uint32_t t = (index_bits12 | (index_bits12 << 10)) & 0x003F003Fu;
    return (t | (t << 5)) & 0x07070707u;
}

__host__ __device__ inline void decode_bc1_block_px(uint32_t colours, uint32_t indices, uint32_t px[16])
{
    const Palette4 p = colour_palette<true>(colours);
    for (int row = 0; row < 4; ++row) {
        const uint32_t sel = spread_2bit((indices >> (8 * row)) & 0xFFu);
        weave_row(byte_perm(p.r, p.r, sel), byte_perm(p.g, p.g, sel), byte_perm(p.b, p.b, sel), byte_perm(p.a, p.a, sel),
                  px + 4 * row);
    }
}

// q = the 16-byte block as four dwords: explicit alpha (8 bytes), colours, indices
__host__ __device__ inline void decode_bc2_block_px(const uint32_t q[4], uint32_t px[16])
{
    const Palette4 p = colour_palette<false>(q[2]);
    for (int row = 0; row < 4; ++row) {
        const uint32_t sel = spread_2bit((q[3] >> (8 * row)) & 0xFFu);
        // four 4-bit alphas of the row, low nibble first (bc2_decode.rs:101-118), scaled by 17
        uint32_t a = (q[row >> 1] >> (16 * (row & 1))) & 0xFFFFu;
        a = (a | (a << 8)) & 0x00FF00FFu;
        a = ((a | (a << 4)) & 0x0F0F0F0Fu) * 17u;
        weave_row(byte_perm(p.r, p.r, sel), byte_perm(p.g, p.g, sel), byte_perm(p.b, p.b, sel), a, px + 4 * row);
    }
}

// the eight alpha values of a BC3 block as bytes of (lo, hi) (bc3_decode.rs:58-100)
__host__ __device__ inline void bc3_alpha_table(uint32_t a0, uint32_t a1, uint32_t& lo, uint32_t& hi)
{
    const bool eight = a0 > a1;
    // entry k (2 <= k): ((W - k) * a0 + (k - 1) * a1) / D with W, D = 8, 7 or 6, 5: a running sum, one step = a1 - a0
    const uint32_t step = a1 - a0;
    uint32_t x = (eight ? 6u : 4u) * a0 + a1;
    uint32_t t[8];
    t[0] = a0, t[1] = a1;
    for (int k = 2; k < 8; ++k) {
        t[k] = eight ? div7_small(x) : div5_small(x);
        x += step;
    }
    if (!eight)   // x ran below zero for k = 7 there; both entries are constants
        t[6] = 0, t[7] = 255;
    lo = t[0] | (t[1] << 8) | (t[2] << 16) | (t[3] << 24);
    hi = t[4] | (t[5] << 8) | (t[6] << 16) | (t[7] << 24);
}

// q = alpha endpoints + 48 bits of 3-bit indices (8 bytes), colours, indices
__host__ __device__ inline void decode_bc3_block_px(const uint32_t q[4], uint32_t px[16])
{
    const Palette4 p = colour_palette<false>(q[2]);
    uint32_t tab_lo, tab_hi;
    bc3_alpha_table(q[0] & 0xFFu, (q[0] >> 8) & 0xFFu, tab_lo, tab_hi);
    const uint64_t abits = (((uint64_t)q[1] << 32) | q[0]) >> 16;   // 48 bits, pixel i at [3i, 3i + 2]
    for (int row = 0; row < 4; ++row) {
        const uint32_t sel = spread_2bit((q[3] >> (8 * row)) & 0xFFu);
        const uint32_t a12 = (uint32_t)(abits >> (12 * row)) & 0xFFFu;
        weave_row(byte_perm(p.r, p.r, sel), byte_perm(p.g, p.g, sel), byte_perm(p.b, p.b, sel),
                  byte_perm(tab_hi, tab_lo, spread_3bit(a12)),
                  px + 4 * row);
    }
}

template <int FMT>
__host__ __device__ inline void decode_block_px(const uint32_t* q, uint32_t px[16])
{
    if (FMT == 1)
        decode_bc1_block_px(q[0], q[1], px);
    else if (FMT == 2)
        decode_bc2_block_px(q, px);
    else
        decode_bc3_block_px(q, px);
}

}  // namespace dxtlt
