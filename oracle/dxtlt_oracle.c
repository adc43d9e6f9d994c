/*
 * dxtlt_oracle.c -- CPU oracle (test infrastructure, see dxtlt_oracle.h).
 *
 * Every loop below handles ONE block per iteration, like the reference's scalar files, and moves
 * fields with unaligned little-endian loads/stores so that any pointer alignment is accepted
 * (reference: bc1 test_prelude.rs:364-373 offsets pointers by +1 byte).
 *
 * Reference paths are relative to /root/reference/src/core/.
 */
#include "dxtlt_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#if defined(__BYTE_ORDER__) && (__BYTE_ORDER__ != __ORDER_LITTLE_ENDIAN__)
#error "the oracle states the on-wire format in little-endian terms and is only built on LE hosts"
#endif

/* ------------------------------------------------------------------------------------------- */
/* unaligned little-endian field access (reference uses read_unaligned / ptr-utils accessors)  */
/* ------------------------------------------------------------------------------------------- */
static inline uint16_t ld16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline uint32_t ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t ld64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline void st16(uint8_t *p, uint16_t v) { memcpy(p, &v, 2); }
static inline void st32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
static inline void st64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }

/* ------------------------------------------------------------------------------------------- */
/* YCoCg-R on one RGB565 value                                                                  */
/* dxt-lossless-transform-common/src/color_565/decorrelate.rs:101-127 (var1), 187-212 (var2),  */
/* 274-299 (var3); inverse 148-173, 235-260, 321-344; dispatchers 364, 391.                    */
/* ------------------------------------------------------------------------------------------- */
#define ORACLE_INLINE static inline __attribute__((always_inline))

ORACLE_INLINE uint16_t decorrelate_565(uint16_t v, const int variant)
{
    if (variant == ORACLE_YCOCG_NONE)
        return v;

    /* field split shared by the three variants (decorrelate.rs:104-107) */
    int r = (v >> 11) & 0x1F;
    int g = (v >> 6) & 0x1F; /* top five bits of the 6-bit green */
    int g_low = (v >> 5) & 0x1;
    int b = v & 0x1F;

    /* lifting steps, each reduced mod 32 (decorrelate.rs:110-120) */
    int co = (r - b) & 0x1F;
    int t = (b + (co >> 1)) & 0x1F;
    int cg = (g - t) & 0x1F;
    int y = (t + (cg >> 1)) & 0x1F;

    switch (variant) {
    case ORACLE_YCOCG_VAR1: /* decorrelate.rs:126 */
        return (uint16_t)((y << 11) | (co << 6) | (g_low << 5) | cg);
    case ORACLE_YCOCG_VAR2: /* decorrelate.rs:212 */
        return (uint16_t)((g_low << 15) | (y << 10) | (co << 5) | cg);
    default: /* ORACLE_YCOCG_VAR3, decorrelate.rs:299 */
        return (uint16_t)((y << 11) | (co << 6) | (cg << 1) | g_low);
    }
}

uint16_t oracle_decorrelate_565(uint16_t v, int variant) { return decorrelate_565(v, variant); }

ORACLE_INLINE uint16_t recorrelate_565(uint16_t v, const int variant)
{
    int y, co, cg, g_low;
    switch (variant) {
    case ORACLE_YCOCG_NONE:
        return v;
    case ORACLE_YCOCG_VAR1: /* decorrelate.rs:151-154 */
        y = (v >> 11) & 0x1F;
        co = (v >> 6) & 0x1F;
        g_low = (v >> 5) & 0x1;
        cg = v & 0x1F;
        break;
    case ORACLE_YCOCG_VAR2: /* decorrelate.rs:238-241 */
        g_low = v >> 15;
        y = (v >> 10) & 0x1F;
        co = (v >> 5) & 0x1F;
        cg = v & 0x1F;
        break;
    default: /* ORACLE_YCOCG_VAR3, decorrelate.rs:324-327 */
        y = (v >> 11) & 0x1F;
        co = (v >> 6) & 0x1F;
        cg = (v >> 1) & 0x1F;
        g_low = v & 0x1;
        break;
    }
    /* inverse lifting (decorrelate.rs:158-167) */
    int t = (y - (cg >> 1)) & 0x1F;
    int g = (cg + t) & 0x1F;
    int b = (t - (co >> 1)) & 0x1F;
    int r = (b + co) & 0x1F;
    return (uint16_t)((r << 11) | (g << 6) | (g_low << 5) | b);
}

uint16_t oracle_recorrelate_565(uint16_t v, int variant) { return recorrelate_565(v, variant); }

/* The per-format loops below are always_inline bodies taking `variant` / split flags as constants; the
 * DISPATCH_* macros at the end of each section instantiate them once per settings combination so the
 * compiler sees straight-line loops (the reference does the same with const generics, e.g.
 * with_split_colour_and_recorr/transform/generic.rs:39 `transform_split_decorr::<VARIANT>`). */

/* ------------------------------------------------------------------------------------------- */
/* BC1: block = c0:u16 c1:u16 idx:u32                                                           */
/* Stream placement: dxt-lossless-transform-bc1/src/transform/transform_with_settings.rs:39-71  */
/*   split:    c0 @0, c1 @len/4, idx @len/2      (lines 43-58)                                   */
/*   no split: (c0,c1) pairs @0, idx @len/2      (standard/transform/portable32.rs:12-13)        */
/* The `first`/`count` range lets callers (MT baseline, shard tests) run a sub-range of blocks  */
/* of an N-block buffer; first=0,count=N is the reference call.                                 */
/* ------------------------------------------------------------------------------------------- */

#define DISPATCH_VS(BODY, variant, split)                                              \
    switch (((variant) & 3) * 2 + ((split) ? 1 : 0)) {                                 \
    case 0: BODY(in, out, n_total, first, count, 0, 0); break;                         \
    case 1: BODY(in, out, n_total, first, count, 0, 1); break;                         \
    case 2: BODY(in, out, n_total, first, count, 1, 0); break;                         \
    case 3: BODY(in, out, n_total, first, count, 1, 1); break;                         \
    case 4: BODY(in, out, n_total, first, count, 2, 0); break;                         \
    case 5: BODY(in, out, n_total, first, count, 2, 1); break;                         \
    case 6: BODY(in, out, n_total, first, count, 3, 0); break;                         \
    default: BODY(in, out, n_total, first, count, 3, 1); break;                        \
    }

#define DISPATCH_VSS(BODY, variant, sa, sc)                                            \
    switch (((variant) & 3) * 4 + ((sa) ? 2 : 0) + ((sc) ? 1 : 0)) {                   \
    case 0: BODY(in, out, n_total, first, count, 0, 0, 0); break;                      \
    case 1: BODY(in, out, n_total, first, count, 0, 0, 1); break;                      \
    case 2: BODY(in, out, n_total, first, count, 0, 1, 0); break;                      \
    case 3: BODY(in, out, n_total, first, count, 0, 1, 1); break;                      \
    case 4: BODY(in, out, n_total, first, count, 1, 0, 0); break;                      \
    case 5: BODY(in, out, n_total, first, count, 1, 0, 1); break;                      \
    case 6: BODY(in, out, n_total, first, count, 1, 1, 0); break;                      \
    case 7: BODY(in, out, n_total, first, count, 1, 1, 1); break;                      \
    case 8: BODY(in, out, n_total, first, count, 2, 0, 0); break;                      \
    case 9: BODY(in, out, n_total, first, count, 2, 0, 1); break;                      \
    case 10: BODY(in, out, n_total, first, count, 2, 1, 0); break;                     \
    case 11: BODY(in, out, n_total, first, count, 2, 1, 1); break;                     \
    case 12: BODY(in, out, n_total, first, count, 3, 0, 0); break;                     \
    case 13: BODY(in, out, n_total, first, count, 3, 0, 1); break;                     \
    case 14: BODY(in, out, n_total, first, count, 3, 1, 0); break;                     \
    default: BODY(in, out, n_total, first, count, 3, 1, 1); break;                     \
    }

ORACLE_INLINE void bc1_fwd_body(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count,
                            const int variant, const int split)
{
    uint8_t *c0_out = out;                /* or the (c0,c1) pair stream when !split */
    uint8_t *c1_out = out + 2 * n_total;  /* len/4 */
    uint8_t *idx_out = out + 4 * n_total; /* len/2 */
    for (size_t b = first; b < first + count; ++b) {
        const uint8_t *blk = in + 8 * b;
        /* with_split_colour_and_recorr/transform/generic.rs:49-51 reads c0, c1, indices */
        uint16_t c0 = decorrelate_565(ld16(blk), variant);
        uint16_t c1 = decorrelate_565(ld16(blk + 2), variant);
        uint32_t idx = ld32(blk + 4);
        if (split) {
            /* with_split_colour/transform/generic.rs:11-43; ..._and_recorr/generic.rs:74-80 */
            st16(c0_out + 2 * b, c0);
            st16(c1_out + 2 * b, c1);
        } else {
            /* standard/transform/portable32.rs:33-46; with_recorrelate/transform/generic.rs:49-76 */
            st32(c0_out + 4 * b, (uint32_t)c0 | ((uint32_t)c1 << 16));
        }
        st32(idx_out + 4 * b, idx);
    }
}

ORACLE_INLINE void bc1_inv_body(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count,
                            const int variant, const int split)
{
    /* transform_with_settings.rs:100-134 */
    const uint8_t *c0_in = in;
    const uint8_t *c1_in = in + 2 * n_total;
    const uint8_t *idx_in = in + 4 * n_total;
    for (size_t b = first; b < first + count; ++b) {
        uint16_t c0, c1;
        if (split) {
            c0 = ld16(c0_in + 2 * b);
            c1 = ld16(c1_in + 2 * b);
        } else {
            uint32_t pair = ld32(c0_in + 4 * b);
            c0 = (uint16_t)pair;
            c1 = (uint16_t)(pair >> 16);
        }
        uint8_t *blk = out + 8 * b;
        st16(blk, recorrelate_565(c0, variant));
        st16(blk + 2, recorrelate_565(c1, variant));
        st32(blk + 4, ld32(idx_in + 4 * b));
    }
}

static void bc1_fwd_range(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count,
                          int variant, int split)
{
    DISPATCH_VS(bc1_fwd_body, variant, split)
}

static void bc1_inv_range(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count,
                          int variant, int split)
{
    DISPATCH_VS(bc1_inv_body, variant, split)
}

void oracle_transform_bc1(const uint8_t *in, uint8_t *out, size_t len, int variant, int split_colour)
{
    bc1_fwd_range(in, out, len / 8, 0, len / 8, variant, split_colour);
}

void oracle_untransform_bc1(const uint8_t *in, uint8_t *out, size_t len, int variant, int split_colour)
{
    bc1_inv_range(in, out, len / 8, 0, len / 8, variant, split_colour);
}

/* ------------------------------------------------------------------------------------------- */
/* BC2: block = alpha:u64 c0:u16 c1:u16 idx:u32                                                 */
/* dxt-lossless-transform-bc2/src/transform/transform_with_settings.rs:30-73 (fwd), 93-138 (inv)*/
/*   alpha @0 (8N), colours @len/2 (c1 @len/2+len/8 when split), idx @len/2+len/4               */
/* scalar truth: standard/transform/portable32.rs:28-56, with_split_colour/transform/generic.rs */
/* ------------------------------------------------------------------------------------------- */
ORACLE_INLINE void bc2_fwd_body(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count,
                            const int variant, const int split)
{
    uint8_t *alpha_out = out;
    uint8_t *c0_out = out + 8 * n_total;
    uint8_t *c1_out = out + 10 * n_total;
    uint8_t *idx_out = out + 12 * n_total;
    for (size_t b = first; b < first + count; ++b) {
        const uint8_t *blk = in + 16 * b;
        uint16_t c0 = decorrelate_565(ld16(blk + 8), variant);
        uint16_t c1 = decorrelate_565(ld16(blk + 10), variant);
        st64(alpha_out + 8 * b, ld64(blk));
        if (split) {
            st16(c0_out + 2 * b, c0);
            st16(c1_out + 2 * b, c1);
        } else {
            st32(c0_out + 4 * b, (uint32_t)c0 | ((uint32_t)c1 << 16));
        }
        st32(idx_out + 4 * b, ld32(blk + 12));
    }
}

ORACLE_INLINE void bc2_inv_body(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count,
                            const int variant, const int split)
{
    const uint8_t *alpha_in = in;
    const uint8_t *c0_in = in + 8 * n_total;
    const uint8_t *c1_in = in + 10 * n_total;
    const uint8_t *idx_in = in + 12 * n_total;
    for (size_t b = first; b < first + count; ++b) {
        uint16_t c0, c1;
        if (split) {
            c0 = ld16(c0_in + 2 * b);
            c1 = ld16(c1_in + 2 * b);
        } else {
            uint32_t pair = ld32(c0_in + 4 * b);
            c0 = (uint16_t)pair;
            c1 = (uint16_t)(pair >> 16);
        }
        uint8_t *blk = out + 16 * b;
        st64(blk, ld64(alpha_in + 8 * b));
        st16(blk + 8, recorrelate_565(c0, variant));
        st16(blk + 10, recorrelate_565(c1, variant));
        st32(blk + 12, ld32(idx_in + 4 * b));
    }
}

static void bc2_fwd_range(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count,
                          int variant, int split)
{
    DISPATCH_VS(bc2_fwd_body, variant, split)
}

static void bc2_inv_range(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count,
                          int variant, int split)
{
    DISPATCH_VS(bc2_inv_body, variant, split)
}

void oracle_transform_bc2(const uint8_t *in, uint8_t *out, size_t len, int variant, int split_colour)
{
    bc2_fwd_range(in, out, len / 16, 0, len / 16, variant, split_colour);
}

void oracle_untransform_bc2(const uint8_t *in, uint8_t *out, size_t len, int variant, int split_colour)
{
    bc2_inv_range(in, out, len / 16, 0, len / 16, variant, split_colour);
}

/* ------------------------------------------------------------------------------------------- */
/* BC3: block = a0:u8 a1:u8 aidx:6B c0:u16 c1:u16 idx:u32                                       */
/* dxt-lossless-transform-bc3/src/transform/transform_with_settings.rs:32-142 (fwd), 162-272    */
/*   alpha endpoints @0 (pairs, or a0 @0 / a1 @N when split)  lines 54-56                       */
/*   alpha indices @2N (6-byte records, verbatim)                                               */
/*   colours @8N (pairs, or c0 @8N / c1 @10N when split)      lines 76-80                       */
/*   colour indices @12N                                                                        */
/* scalar truth: standard/transform/portable32.rs:38-65,                                        */
/*   with_split_alphas_colour_and_recorr/transform/generic.rs:23-89 and the six siblings.       */
/* ------------------------------------------------------------------------------------------- */
ORACLE_INLINE void bc3_fwd_body(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count,
                            const int variant, const int split_alpha, const int split_colour)
{
    uint8_t *a0_out = out; /* or (a0,a1) pairs */
    uint8_t *a1_out = out + n_total;
    uint8_t *aidx_out = out + 2 * n_total;
    uint8_t *c0_out = out + 8 * n_total; /* or (c0,c1) pairs */
    uint8_t *c1_out = out + 10 * n_total;
    uint8_t *idx_out = out + 12 * n_total;
    for (size_t b = first; b < first + count; ++b) {
        const uint8_t *blk = in + 16 * b;
        uint8_t a0 = blk[0], a1 = blk[1];
        if (split_alpha) {
            a0_out[b] = a0;
            a1_out[b] = a1;
        } else {
            a0_out[2 * b] = a0;
            a0_out[2 * b + 1] = a1;
        }
        /* six index bytes move as u16 + u32 in the reference (generic.rs:44-45, 77-78) */
        st16(aidx_out + 6 * b, ld16(blk + 2));
        st32(aidx_out + 6 * b + 2, ld32(blk + 4));

        uint16_t c0 = decorrelate_565(ld16(blk + 8), variant);
        uint16_t c1 = decorrelate_565(ld16(blk + 10), variant);
        if (split_colour) {
            st16(c0_out + 2 * b, c0);
            st16(c1_out + 2 * b, c1);
        } else {
            st32(c0_out + 4 * b, (uint32_t)c0 | ((uint32_t)c1 << 16));
        }
        st32(idx_out + 4 * b, ld32(blk + 12));
    }
}

ORACLE_INLINE void bc3_inv_body(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count,
                            const int variant, const int split_alpha, const int split_colour)
{
    const uint8_t *a0_in = in;
    const uint8_t *a1_in = in + n_total;
    const uint8_t *aidx_in = in + 2 * n_total;
    const uint8_t *c0_in = in + 8 * n_total;
    const uint8_t *c1_in = in + 10 * n_total;
    const uint8_t *idx_in = in + 12 * n_total;
    for (size_t b = first; b < first + count; ++b) {
        uint8_t *blk = out + 16 * b;
        if (split_alpha) {
            blk[0] = a0_in[b];
            blk[1] = a1_in[b];
        } else {
            blk[0] = a0_in[2 * b];
            blk[1] = a0_in[2 * b + 1];
        }
        st16(blk + 2, ld16(aidx_in + 6 * b));
        st32(blk + 4, ld32(aidx_in + 6 * b + 2));
        uint16_t c0, c1;
        if (split_colour) {
            c0 = ld16(c0_in + 2 * b);
            c1 = ld16(c1_in + 2 * b);
        } else {
            uint32_t pair = ld32(c0_in + 4 * b);
            c0 = (uint16_t)pair;
            c1 = (uint16_t)(pair >> 16);
        }
        st16(blk + 8, recorrelate_565(c0, variant));
        st16(blk + 10, recorrelate_565(c1, variant));
        st32(blk + 12, ld32(idx_in + 4 * b));
    }
}

static void bc3_fwd_range(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count,
                          int variant, int split_alpha, int split_colour)
{
    DISPATCH_VSS(bc3_fwd_body, variant, split_alpha, split_colour)
}

static void bc3_inv_range(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count,
                          int variant, int split_alpha, int split_colour)
{
    DISPATCH_VSS(bc3_inv_body, variant, split_alpha, split_colour)
}

void oracle_transform_bc3(const uint8_t *in, uint8_t *out, size_t len, int variant, int split_alpha,
                          int split_colour)
{
    bc3_fwd_range(in, out, len / 16, 0, len / 16, variant, split_alpha, split_colour);
}

void oracle_untransform_bc3(const uint8_t *in, uint8_t *out, size_t len, int variant, int split_alpha,
                            int split_colour)
{
    bc3_inv_range(in, out, len / 16, 0, len / 16, variant, split_alpha, split_colour);
}

/* ------------------------------------------------------------------------------------------- */
/* safe wrappers: bc1 safe/transform_with_settings.rs:88-118, 192-220 (+ bc2 / bc3 twins)       */
/* order: length check first, then output size.                                                 */
/* ------------------------------------------------------------------------------------------- */
static int validate(size_t in_len, size_t out_len, size_t block)
{
    if (in_len % block != 0)
        return ORACLE_INVALID_LENGTH;
    if (out_len < in_len)
        return ORACLE_OUTPUT_TOO_SMALL;
    return ORACLE_OK;
}

int oracle_transform_bc1_safe(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, int variant,
                              int split_colour)
{
    int rc = validate(in_len, out_len, 8);
    if (rc == ORACLE_OK)
        oracle_transform_bc1(in, out, in_len, variant, split_colour);
    return rc;
}

int oracle_untransform_bc1_safe(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, int variant,
                                int split_colour)
{
    int rc = validate(in_len, out_len, 8);
    if (rc == ORACLE_OK)
        oracle_untransform_bc1(in, out, in_len, variant, split_colour);
    return rc;
}

int oracle_transform_bc2_safe(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, int variant,
                              int split_colour)
{
    int rc = validate(in_len, out_len, 16);
    if (rc == ORACLE_OK)
        oracle_transform_bc2(in, out, in_len, variant, split_colour);
    return rc;
}

int oracle_untransform_bc2_safe(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, int variant,
                                int split_colour)
{
    int rc = validate(in_len, out_len, 16);
    if (rc == ORACLE_OK)
        oracle_untransform_bc2(in, out, in_len, variant, split_colour);
    return rc;
}

int oracle_transform_bc3_safe(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, int variant,
                              int split_alpha, int split_colour)
{
    int rc = validate(in_len, out_len, 16);
    if (rc == ORACLE_OK)
        oracle_transform_bc3(in, out, in_len, variant, split_alpha, split_colour);
    return rc;
}

int oracle_untransform_bc3_safe(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, int variant,
                                int split_alpha, int split_colour)
{
    int rc = validate(in_len, out_len, 16);
    if (rc == ORACLE_OK)
        oracle_untransform_bc3(in, out, in_len, variant, split_alpha, split_colour);
    return rc;
}

/* ------------------------------------------------------------------------------------------- */
/* reference test-data generators                                                               */
/* ------------------------------------------------------------------------------------------- */

/* bc1 test_prelude.rs:81-105: colour bytes count up from 0, index bytes from 128, both mod 256 */
void oracle_generate_bc1_test_data(size_t num_blocks, uint8_t *out)
{
    uint8_t colour = 0, index = 128;
    for (size_t b = 0; b < num_blocks; ++b, out += 8) {
        for (int k = 0; k < 4; ++k) {
            out[k] = (uint8_t)(colour + k);
            out[4 + k] = (uint8_t)(index + k);
        }
        colour = (uint8_t)(colour + 4);
        index = (uint8_t)(index + 4);
    }
}

/* bc2 test_prelude.rs:151-186: alpha from 0x00 (+8), colours from 0x80 (+4), indices from 0xC0 (+4),
 * all plain u8 wrap-around */
void oracle_generate_bc2_test_data(size_t num_blocks, uint8_t *out)
{
    uint8_t alpha = 0x00, colour = 0x80, index = 0xC0;
    for (size_t b = 0; b < num_blocks; ++b, out += 16) {
        for (int k = 0; k < 8; ++k)
            out[k] = (uint8_t)(alpha + k);
        for (int k = 0; k < 4; ++k) {
            out[8 + k] = (uint8_t)(colour + k);
            out[12 + k] = (uint8_t)(index + k);
        }
        alpha = (uint8_t)(alpha + 8);
        colour = (uint8_t)(colour + 4);
        index = (uint8_t)(index + 4);
    }
}

/* bc3 test_prelude.rs:45-101: four bands, each wrapping inside its own band:
 * alpha 0..31 (+2), alpha indices 32..127 (+6), colours 128..191 (+4), indices 192..255 (+4) */
void oracle_generate_bc3_test_data(size_t num_blocks, uint8_t *out)
{
    uint8_t alpha = 0, aidx = 32, colour = 128, index = 192;
    for (size_t b = 0; b < num_blocks; ++b, out += 16) {
        out[0] = alpha;
        out[1] = (uint8_t)(alpha + 1);
        alpha = (uint8_t)(alpha + 2);
        if (alpha >= 32)
            alpha = (uint8_t)(alpha - 32);

        for (int k = 0; k < 6; ++k)
            out[2 + k] = (uint8_t)(aidx + k);
        aidx = (uint8_t)(aidx + 6);
        if (aidx >= 128)
            aidx = (uint8_t)(aidx - 96);

        for (int k = 0; k < 4; ++k)
            out[8 + k] = (uint8_t)(colour + k);
        colour = (uint8_t)(colour + 4);
        if (colour >= 192)
            colour = (uint8_t)(colour - 64);

        for (int k = 0; k < 4; ++k)
            out[12 + k] = (uint8_t)(index + k);
        index = (uint8_t)(index + 4);
        if (index < 192) /* wrapped past 255 */
            index = (uint8_t)(index - 64);
    }
}

/* common/src/transforms/split_565_color_endpoints/mod.rs:110 + tests.rs:140-152 */
void oracle_split_565_color_endpoints(const uint8_t *in, uint8_t *out, size_t len_bytes)
{
    size_t pairs = len_bytes / 4;
    for (size_t i = 0; i < pairs; ++i) {
        st16(out + 2 * i, ld16(in + 4 * i));
        st16(out + 2 * pairs + 2 * i, ld16(in + 4 * i + 2));
    }
}

/* ------------------------------------------------------------------------------------------- */
/* synthetic workload + checksum                                                                */
/* ------------------------------------------------------------------------------------------- */
static inline uint64_t splitmix64_at(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void oracle_fill_splitmix64(uint8_t *out, size_t len_bytes, uint64_t seed, uint64_t first_qword)
{
    size_t q = len_bytes / 8;
    for (size_t i = 0; i < q; ++i)
        st64(out + 8 * i, splitmix64_at(seed, first_qword + i));
    size_t rem = len_bytes - 8 * q;
    if (rem) {
        uint64_t v = splitmix64_at(seed, first_qword + q);
        memcpy(out + 8 * q, &v, rem);
    }
}

uint64_t oracle_sum_u64(const uint8_t *data, size_t len_bytes)
{
    uint64_t s = 0;
    size_t q = len_bytes / 8;
    for (size_t i = 0; i < q; ++i)
        s += ld64(data + 8 * i);
    size_t rem = len_bytes - 8 * q;
    if (rem) {
        uint64_t v = 0;
        memcpy(&v, data + 8 * q, rem);
        s += v;
    }
    return s;
}

/* ------------------------------------------------------------------------------------------- */
/* multi-threaded range split (cpu_baseline only)                                               */
/* ------------------------------------------------------------------------------------------- */
struct mt_job {
    int kind, inverse, variant, split_alpha, split_colour;
    const uint8_t *in;
    uint8_t *out;
    size_t n_total, first, count;
};

static void *mt_worker(void *arg)
{
    struct mt_job *j = (struct mt_job *)arg;
    switch (j->kind * 2 + (j->inverse ? 1 : 0)) {
    case 2: bc1_fwd_range(j->in, j->out, j->n_total, j->first, j->count, j->variant, j->split_colour); break;
    case 3: bc1_inv_range(j->in, j->out, j->n_total, j->first, j->count, j->variant, j->split_colour); break;
    case 4: bc2_fwd_range(j->in, j->out, j->n_total, j->first, j->count, j->variant, j->split_colour); break;
    case 5: bc2_inv_range(j->in, j->out, j->n_total, j->first, j->count, j->variant, j->split_colour); break;
    case 6:
        bc3_fwd_range(j->in, j->out, j->n_total, j->first, j->count, j->variant, j->split_alpha,
                      j->split_colour);
        break;
    case 7:
        bc3_inv_range(j->in, j->out, j->n_total, j->first, j->count, j->variant, j->split_alpha,
                      j->split_colour);
        break;
    default: break;
    }
    return NULL;
}

/* one block range of a buffer of n_total blocks through the scalar loops (tails of the vectorised ports) */
void oracle_transform_range(int kind, int inverse, const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count,
                            int variant, int split_alpha, int split_colour)
{
    struct mt_job j = {kind, inverse, variant, split_alpha, split_colour, in, out, n_total, first, count};
    mt_worker(&j);
}

void oracle_run_mt(int kind, int inverse, const uint8_t *in, uint8_t *out, size_t len, int variant,
                   int split_alpha, int split_colour, int threads)
{
    size_t block = (kind == 1) ? 8 : 16;
    size_t n = len / block;
    if (threads < 1)
        threads = 1;
    if ((size_t)threads > n && n > 0)
        threads = (int)n;
    struct mt_job *jobs = (struct mt_job *)calloc((size_t)threads, sizeof *jobs);
    pthread_t *tids = (pthread_t *)calloc((size_t)threads, sizeof *tids);
    size_t per = threads ? (n + (size_t)threads - 1) / (size_t)threads : 0;
    for (int t = 0; t < threads; ++t) {
        size_t first = per * (size_t)t;
        size_t count = first >= n ? 0 : (first + per > n ? n - first : per);
        struct mt_job j = {kind, inverse, variant, split_alpha, split_colour, in, out, n, first, count};
        jobs[t] = j;
        if (t > 0)
            pthread_create(&tids[t], NULL, mt_worker, &jobs[t]);
    }
    if (threads > 0)
        mt_worker(&jobs[0]);
    for (int t = 1; t < threads; ++t)
        pthread_join(tids[t], NULL);
    free(jobs);
    free(tids);
}
