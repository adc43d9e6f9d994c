/*
 * dxtlt_oracle_norm.c -- CPU oracle for BC1 block normalisation (the reference's experimental module).
 *
 * TEST INFRASTRUCTURE ONLY (see dxtlt_oracle.h).  Plain-C restatement, block by block and pixel by pixel like the
 * reference, of (paths under /root/reference/src/core/):
 *   dxt-lossless-transform-bc1/src/util/bc1_decode.rs:42-100                                   decode_bc1_block
 *   dxt-lossless-transform-common/src/color_565/mod.rs:108-116, 153-186, 224-226               from_rgb, red/green/blue, to_8888_lossy
 *   dxt-lossless-transform-common/src/decoded_4x4_block.rs:107-120                             has_identical_pixels
 *   dxt-lossless-transform-bc1/src/experimental/normalize_blocks/normalize.rs:38-96            normalize_blocks
 *   ... normalize.rs:118-188 (classification), :214-258 (solid block writer), :286-386 (split, in place),
 *       :417-481 (all modes)
 *   dxt-lossless-transform-bc1/src/experimental/normalize_blocks/transform.rs:65-166           transform_bc1_with_normalize_blocks
 *
 * Pin status: PINNED by the reference's own unit tests, which hold concrete input/expected-output blocks
 * (normalize.rs:505-1076: solid, transparent, mixed, non-round-trippable, multiple blocks, all modes, in place,
 * split in place x2); tests/test_normalize.py replays every one of them against this file.
 */
#include <stdlib.h>
#include <string.h>

#include "dxtlt_oracle.h"

typedef struct {
    uint8_t r, g, b, a;
} px8888;

static uint8_t red_of(uint16_t v) { const unsigned r = (v >> 11) & 31u; return (uint8_t)((r << 3) | (r >> 2)); }
static uint8_t green_of(uint16_t v) { const unsigned g = (v >> 5) & 63u; return (uint8_t)((g << 2) | (g >> 4)); }
static uint8_t blue_of(uint16_t v) { const unsigned b = v & 31u; return (uint8_t)((b << 3) | (b >> 2)); }
static uint16_t from_rgb(uint8_t r, uint8_t g, uint8_t b)
{
    return (uint16_t)((((unsigned)r & 0xF8u) << 8) | (((unsigned)g & 0xFCu) << 3) | ((unsigned)b >> 3));
}

/* bc1_decode.rs:42-100; pixels in row-major order, pixel i uses index bits [2i, 2i+1] */
static void decode_block(const uint8_t *src, px8888 out[16])
{
    const uint16_t c0 = (uint16_t)(src[0] | (src[1] << 8));
    const uint16_t c1 = (uint16_t)(src[2] | (src[3] << 8));
    const uint32_t idx = (uint32_t)src[4] | ((uint32_t)src[5] << 8) | ((uint32_t)src[6] << 16) | ((uint32_t)src[7] << 24);
    const unsigned r0 = red_of(c0), g0 = green_of(c0), b0 = blue_of(c0);
    const unsigned r1 = red_of(c1), g1 = green_of(c1), b1 = blue_of(c1);
    px8888 dict[4];
    dict[0] = (px8888){(uint8_t)r0, (uint8_t)g0, (uint8_t)b0, 255};
    dict[1] = (px8888){(uint8_t)r1, (uint8_t)g1, (uint8_t)b1, 255};
    if (c0 > c1) {
        dict[2] = (px8888){(uint8_t)((2 * r0 + r1) / 3), (uint8_t)((2 * g0 + g1) / 3), (uint8_t)((2 * b0 + b1) / 3), 255};
        dict[3] = (px8888){(uint8_t)((r0 + 2 * r1) / 3), (uint8_t)((g0 + 2 * g1) / 3), (uint8_t)((b0 + 2 * b1) / 3), 255};
    } else {
        dict[2] = (px8888){(uint8_t)((r0 + r1) / 2), (uint8_t)((g0 + g1) / 2), (uint8_t)((b0 + b1) / 2), 255};
        dict[3] = (px8888){0, 0, 0, 0};
    }
    for (int i = 0; i < 16; ++i)
        out[i] = dict[(idx >> (2 * i)) & 3u];
}

void oracle_decode_bc1_block(const uint8_t *src, uint8_t *rgba_out)
{
    px8888 px[16];
    decode_block(src, px);
    for (int i = 0; i < 16; ++i) {
        rgba_out[4 * i + 0] = px[i].r;
        rgba_out[4 * i + 1] = px[i].g;
        rgba_out[4 * i + 2] = px[i].b;
        rgba_out[4 * i + 3] = px[i].a;
    }
}

enum { CASE_TRANSPARENT, CASE_SOLID, CASE_KEEP };

/* normalize.rs:118-188 */
static int classify(const uint8_t *block, uint16_t *color565)
{
    px8888 px[16];
    decode_block(block, px);
    for (int i = 1; i < 16; ++i)
        if (memcmp(&px[i], &px[0], sizeof(px8888)) != 0)
            return CASE_KEEP;
    if (px[0].a == 0)
        return CASE_TRANSPARENT;
    *color565 = from_rgb(px[0].r, px[0].g, px[0].b);
    if (red_of(*color565) == px[0].r && green_of(*color565) == px[0].g && blue_of(*color565) == px[0].b && px[0].a == 255)
        return CASE_SOLID;
    return CASE_KEEP;
}

/* normalize.rs:214-258 */
static void write_solid(uint8_t *dst, const uint8_t *src, uint16_t color565, int mode)
{
    uint8_t tmp[8];
    memcpy(tmp, src, 8);   /* dst may be src */
    dst[0] = (uint8_t)color565;
    dst[1] = (uint8_t)(color565 >> 8);
    if (mode == ORACLE_NORMALIZE_NONE) {
        memcpy(dst, tmp, 8);
    } else if (mode == ORACLE_NORMALIZE_COLOR0_ONLY) {
        memset(dst + 2, 0, 6);
    } else {
        dst[2] = dst[0];
        dst[3] = dst[1];
        memset(dst + 4, 0, 4);
    }
}

void oracle_normalize_bc1_blocks(const uint8_t *in, uint8_t *out, size_t len, int mode)
{
    if (mode == ORACLE_NORMALIZE_NONE) {
        if (in != out)
            memmove(out, in, len);
        return;
    }
    for (size_t o = 0; o + 8 <= len; o += 8) {
        uint16_t c = 0;
        switch (classify(in + o, &c)) {
        case CASE_TRANSPARENT: memset(out + o, 0xFF, 8); break;
        case CASE_SOLID: write_solid(out + o, in + o, c, mode); break;
        default: memmove(out + o, in + o, 8); break;
        }
    }
}

/* normalize.rs:286-386 */
void oracle_normalize_bc1_split_blocks_in_place(uint8_t *colors, uint8_t *indices, size_t num_blocks, int mode)
{
    if (mode == ORACLE_NORMALIZE_NONE)
        return;
    for (size_t b = 0; b < num_blocks; ++b) {
        uint8_t tmp[8];
        memcpy(tmp, colors + 4 * b, 4);
        memcpy(tmp + 4, indices + 4 * b, 4);
        uint16_t c = 0;
        const int cls = classify(tmp, &c);
        if (cls == CASE_TRANSPARENT) {
            memset(colors + 4 * b, 0xFF, 4);
            memset(indices + 4 * b, 0xFF, 4);
        } else if (cls == CASE_SOLID) {
            colors[4 * b + 0] = (uint8_t)c;
            colors[4 * b + 1] = (uint8_t)(c >> 8);
            colors[4 * b + 2] = mode == ORACLE_NORMALIZE_REPLICATE_COLOR ? (uint8_t)c : 0;
            colors[4 * b + 3] = mode == ORACLE_NORMALIZE_REPLICATE_COLOR ? (uint8_t)(c >> 8) : 0;
            memset(indices + 4 * b, 0, 4);
        }
    }
}

/* normalize.rs:417-481; returns 1 when any block was normalised */
int oracle_normalize_bc1_blocks_all_modes(const uint8_t *in, uint8_t *out_none, uint8_t *out_color0,
                                          uint8_t *out_replicate, size_t len)
{
    uint8_t *outs[3] = {out_none, out_color0, out_replicate};
    int any = 0;
    for (size_t o = 0; o + 8 <= len; o += 8) {
        uint16_t c = 0;
        uint8_t src[8];
        memcpy(src, in + o, 8);
        const int cls = classify(src, &c);
        for (int m = 0; m < 3; ++m) {
            if (cls == CASE_TRANSPARENT)
                memset(outs[m] + o, 0xFF, 8);
            else if (cls == CASE_SOLID)
                write_solid(outs[m] + o, src, c, m);
            else
                memcpy(outs[m] + o, src, 8);
        }
        any |= cls != CASE_KEEP;
    }
    return any;
}

/* transform.rs:65-166: every branch equals "normalise, then transform with the plain settings" */
int oracle_transform_bc1_with_normalize_blocks(const uint8_t *in, uint8_t *out, size_t len, int mode, int variant,
                                               int split_colour)
{
    uint8_t *tmp = (uint8_t *)malloc(len ? len : 1);
    if (tmp == NULL)
        return -1;
    oracle_normalize_bc1_blocks(in, tmp, len, mode);
    oracle_transform_bc1(tmp, out, len, variant, split_colour);
    free(tmp);
    return 0;
}

/* =====================================================================================================
 * BC2 / BC3 block normalisation (the reference's experimental modules for those formats):
 *   dxt-lossless-transform-bc2/src/util/bc2_decode.rs:14-95            decode_bc2_block (always four colours, 4-bit alpha x 17)
 *   dxt-lossless-transform-bc2/src/experimental/normalize_blocks/normalize.rs:35-90 (blocks), :118-160 (classification),
 *       :193-258 (all modes), :270-318 (solid block writer), :382-470 (split, in place)
 *   dxt-lossless-transform-bc3/src/util/bc3_decode.rs:12-125           decode_bc3_block (four colours, BC4-style alpha)
 *   dxt-lossless-transform-bc3/src/experimental/normalize_blocks/normalize.rs:36-107 (blocks), :117-156 (modes),
 *       :176-222 (normalize_alpha), :241-280 (normalize_color), :325-385 (classification), :419-500 (all modes),
 *       :539-690 (split, in place)
 * Pin status: PINNED by the unit tests of those two files (tests/test_normalize_bc23.py replays them).
 * ===================================================================================================== */

/* RGB of the sixteen pixels of the colour half (8 bytes: c0, c1, indices), always four-colour mode */
static void decode_colour_4c(const uint8_t *col, px8888 out[16])
{
    const uint16_t c0 = (uint16_t)(col[0] | (col[1] << 8));
    const uint16_t c1 = (uint16_t)(col[2] | (col[3] << 8));
    const uint32_t idx = (uint32_t)col[4] | ((uint32_t)col[5] << 8) | ((uint32_t)col[6] << 16) | ((uint32_t)col[7] << 24);
    const unsigned r0 = red_of(c0), g0 = green_of(c0), b0 = blue_of(c0);
    const unsigned r1 = red_of(c1), g1 = green_of(c1), b1 = blue_of(c1);
    px8888 dict[4];
    dict[0] = (px8888){(uint8_t)r0, (uint8_t)g0, (uint8_t)b0, 255};
    dict[1] = (px8888){(uint8_t)r1, (uint8_t)g1, (uint8_t)b1, 255};
    dict[2] = (px8888){(uint8_t)((2 * r0 + r1) / 3), (uint8_t)((2 * g0 + g1) / 3), (uint8_t)((2 * b0 + b1) / 3), 255};
    dict[3] = (px8888){(uint8_t)((r0 + 2 * r1) / 3), (uint8_t)((g0 + 2 * g1) / 3), (uint8_t)((b0 + 2 * b1) / 3), 255};
    for (int i = 0; i < 16; ++i)
        out[i] = dict[(idx >> (2 * i)) & 3u];
}

static void decode_bc2(const uint8_t *src, px8888 out[16])
{
    decode_colour_4c(src + 8, out);
    for (int i = 0; i < 16; ++i)
        out[i].a = (uint8_t)(((src[i >> 1] >> ((i & 1) * 4)) & 0x0F) * 17);
}

static void decode_bc3(const uint8_t *src, px8888 out[16])
{
    decode_colour_4c(src + 8, out);
    const unsigned a0 = src[0], a1 = src[1];
    uint8_t tab[8];
    tab[0] = (uint8_t)a0;
    tab[1] = (uint8_t)a1;
    if (a0 > a1) {
        for (int k = 2; k < 8; ++k)
            tab[k] = (uint8_t)(((8 - k) * a0 + (k - 1) * a1) / 7);
    } else {
        for (int k = 2; k < 6; ++k)
            tab[k] = (uint8_t)(((6 - k) * a0 + (k - 1) * a1) / 5);
        tab[6] = 0;
        tab[7] = 255;
    }
    uint64_t bits = 0;
    for (int i = 0; i < 6; ++i)
        bits |= (uint64_t)src[2 + i] << (8 * i);
    for (int i = 0; i < 16; ++i)
        out[i].a = tab[(bits >> (3 * i)) & 7u];
}

void oracle_decode_bc2_block(const uint8_t *src, uint8_t *rgba_out)
{
    px8888 px[16];
    decode_bc2(src, px);
    memcpy(rgba_out, px, 64);
}

void oracle_decode_bc3_block(const uint8_t *src, uint8_t *rgba_out)
{
    px8888 px[16];
    decode_bc3(src, px);
    memcpy(rgba_out, px, 64);
}

/* solid colour ignoring alpha + clean 565 round trip (bc2 normalize.rs:131-150, bc3 normalize.rs:352-372) */
static int solid_colour_ignoring_alpha(const px8888 px[16], uint16_t *color565)
{
    *color565 = from_rgb(px[0].r, px[0].g, px[0].b);
    for (int i = 1; i < 16; ++i)
        if (px[i].r != px[0].r || px[i].g != px[0].g || px[i].b != px[0].b)
            return 0;
    return red_of(*color565) == px[0].r && green_of(*color565) == px[0].g && blue_of(*color565) == px[0].b;
}

static void write_colour_half(uint8_t *dst_col, uint16_t c, int color_mode)
{
    dst_col[0] = (uint8_t)c;
    dst_col[1] = (uint8_t)(c >> 8);
    dst_col[2] = color_mode == ORACLE_NORMALIZE_REPLICATE_COLOR ? (uint8_t)c : 0;
    dst_col[3] = color_mode == ORACLE_NORMALIZE_REPLICATE_COLOR ? (uint8_t)(c >> 8) : 0;
    memset(dst_col + 4, 0, 4);
}

void oracle_normalize_bc2_blocks(const uint8_t *in, uint8_t *out, size_t len, int color_mode)
{
    for (size_t o = 0; o + 16 <= len; o += 16) {
        uint8_t src[16];
        memcpy(src, in + o, 16);
        px8888 px[16];
        decode_bc2(src, px);
        uint16_t c = 0;
        memcpy(out + o, src, 16);
        if (color_mode != ORACLE_NORMALIZE_NONE && solid_colour_ignoring_alpha(px, &c))
            write_colour_half(out + o + 8, c, color_mode);
    }
}

void oracle_normalize_bc2_split_blocks_in_place(const uint8_t *alpha, uint8_t *colors, uint8_t *indices, size_t num_blocks,
                                                int color_mode)
{
    if (color_mode == ORACLE_NORMALIZE_NONE)
        return;
    for (size_t b = 0; b < num_blocks; ++b) {
        uint8_t tmp[16];
        memcpy(tmp, alpha + 8 * b, 8);
        memcpy(tmp + 8, colors + 4 * b, 4);
        memcpy(tmp + 12, indices + 4 * b, 4);
        px8888 px[16];
        decode_bc2(tmp, px);
        uint16_t c = 0;
        if (solid_colour_ignoring_alpha(px, &c)) {
            uint8_t half[8];
            write_colour_half(half, c, color_mode);
            memcpy(colors + 4 * b, half, 4);
            memcpy(indices + 4 * b, half + 4, 4);
        }
    }
}

void oracle_normalize_bc2_blocks_all_modes(const uint8_t *in, uint8_t *out_none, uint8_t *out_color0, uint8_t *out_replicate,
                                           size_t len)
{
    uint8_t *outs[3] = {out_none, out_color0, out_replicate};
    for (int m = 0; m < 3; ++m)
        oracle_normalize_bc2_blocks(in, outs[m], len, m);
}

/* bc3 normalize.rs:176-222 */
static void write_alpha_half(uint8_t *dst, const uint8_t *src, uint8_t alpha, int alpha_mode)
{
    if (alpha_mode == ORACLE_ALPHA_OPAQUE_FILL_ALL && alpha == 255) {
        memset(dst, 0xFF, 8);
    } else if (alpha_mode == ORACLE_ALPHA_OPAQUE_ZERO_ALPHA_MAX_INDICES && alpha == 255) {
        dst[0] = dst[1] = 0;
        memset(dst + 2, 0xFF, 6);
    } else if (alpha_mode == ORACLE_ALPHA_NONE) {
        memmove(dst, src, 8);
    } else {
        dst[0] = alpha;
        memset(dst + 1, 0, 7);
    }
}

static int uniform_alpha(const px8888 px[16])
{
    for (int i = 1; i < 16; ++i)
        if (px[i].a != px[0].a)
            return 0;
    return 1;
}

void oracle_normalize_bc3_blocks(const uint8_t *in, uint8_t *out, size_t len, int alpha_mode, int color_mode)
{
    for (size_t o = 0; o + 16 <= len; o += 16) {
        uint8_t src[16];
        memcpy(src, in + o, 16);
        px8888 px[16];
        decode_bc3(src, px);
        uint16_t c = 0;
        memcpy(out + o, src, 16);
        if (alpha_mode != ORACLE_ALPHA_NONE && uniform_alpha(px))
            write_alpha_half(out + o, src, px[0].a, alpha_mode);
        if (color_mode != ORACLE_NORMALIZE_NONE && solid_colour_ignoring_alpha(px, &c))
            write_colour_half(out + o + 8, c, color_mode);
    }
}

void oracle_normalize_bc3_split_blocks_in_place(uint8_t *alpha_endpoints, uint8_t *alpha_indices, uint8_t *color_endpoints,
                                                uint8_t *color_indices, size_t num_blocks, int alpha_mode, int color_mode)
{
    if (alpha_mode == ORACLE_ALPHA_NONE && color_mode == ORACLE_NORMALIZE_NONE)
        return;
    for (size_t b = 0; b < num_blocks; ++b) {
        uint8_t tmp[16], outb[16];
        memcpy(tmp, alpha_endpoints + 2 * b, 2);
        memcpy(tmp + 2, alpha_indices + 6 * b, 6);
        memcpy(tmp + 8, color_endpoints + 4 * b, 4);
        memcpy(tmp + 12, color_indices + 4 * b, 4);
        oracle_normalize_bc3_blocks(tmp, outb, 16, alpha_mode, color_mode);
        memcpy(alpha_endpoints + 2 * b, outb, 2);
        memcpy(alpha_indices + 6 * b, outb + 2, 6);
        memcpy(color_endpoints + 4 * b, outb + 8, 4);
        memcpy(color_indices + 4 * b, outb + 12, 4);
    }
}

/* outs[alpha_mode * 3 + color_mode], bc3 normalize.rs:419-500 */
void oracle_normalize_bc3_blocks_all_modes(const uint8_t *in, uint8_t *const outs[12], size_t len)
{
    for (int a = 0; a < 4; ++a)
        for (int c = 0; c < 3; ++c)
            oracle_normalize_bc3_blocks(in, outs[a * 3 + c], len, a, c);
}

/* =====================================================================================================
 * Array form of the block decoders above (the reference decodes one block per call: bc1_decode.rs:42,
 * bc2_decode.rs:44, bc3_decode.rs:43): `num_blocks` blocks -> num_blocks * 64 bytes, one Decoded4x4Block
 * (decoded_4x4_block.rs:56: sixteen {r, g, b, a}, row-major) per block.  kind = 1, 2, 3.
 * Pin status: PINNED by the reference's decoder unit tests (tests/test_decode.py replays them).
 * ===================================================================================================== */
void oracle_decode_blocks(int kind, const uint8_t *in, uint8_t *rgba_out, size_t num_blocks)
{
    const size_t bs = kind == 1 ? 8 : 16;
    for (size_t i = 0; i < num_blocks; ++i) {
        if (kind == 1)
            oracle_decode_bc1_block(in + bs * i, rgba_out + 64 * i);
        else if (kind == 2)
            oracle_decode_bc2_block(in + bs * i, rgba_out + 64 * i);
        else
            oracle_decode_bc3_block(in + bs * i, rgba_out + 64 * i);
    }
}

/* number of blocks whose sixteen decoded pixels differ between the two arrays (what the reference's normalisation
 * tests assert to be zero, e.g. bc1 normalize.rs tests: decode before == decode after) */
uint64_t oracle_count_pixel_differences(int kind, const uint8_t *a, const uint8_t *b, size_t num_blocks)
{
    const size_t bs = kind == 1 ? 8 : 16;
    uint64_t n = 0;
    for (size_t i = 0; i < num_blocks; ++i) {
        uint8_t pa[64], pb[64];
        oracle_decode_blocks(kind, a + bs * i, pa, 1);
        oracle_decode_blocks(kind, b + bs * i, pb, 1);
        n += memcmp(pa, pb, 64) != 0;
    }
    return n;
}
