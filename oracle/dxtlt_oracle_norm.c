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
