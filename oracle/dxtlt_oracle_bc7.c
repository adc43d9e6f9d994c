/*
 * dxtlt_oracle_bc7.c -- CPU statement of the BC7 granule-sorted field split, version 2 (docs/BC7_FORMAT.md).
 *
 * TEST INFRASTRUCTURE ONLY (see dxtlt_oracle.h).  PARITY UNPINNED: the reference has no BC7 transform
 * (/root/reference/src/core/dxt-lossless-transform-bc7/src/lib.rs:1-13); the format is defined by this build, and this
 * file is its executable definition, not a restatement of reference behaviour.  What the reference does fix is used:
 * the bit fields of the eight block modes (/root/reference/src/assets/research/dds-bc7-blocks.hexpat:286-654).
 *
 * Written for clarity, one bit field at a time through unsigned __int128; the device code (csrc/bc7_fields.h) does the
 * same moves with compile-time positions on four dwords.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;

#define BC7_GRANULE 1024u /* blocks per sort granule */

/* Bit fields of a block of mode m, LSB first: marker (m zero bits and a one), header (partition / rotation / index
 * selector), n_rgb colour endpoint fields of w_rgb bits (all reds, all greens, all blues), n_a alpha endpoint fields of
 * w_a bits, then the p-bits and the index bits, which fill the block (hexpat:286-654). */
typedef struct {
    int hdr, n_rgb, w_rgb, n_a, w_a;
} Bc7Mode;
static const Bc7Mode kModes[8] = {
    {4, 18, 4, 0, 0}, /* 0: 3 subsets, 4-bit RGB, 6 p-bits, 45 index bits */
    {6, 12, 6, 0, 0}, /* 1: 2 subsets, 6-bit RGB, 2 p-bits, 46 */
    {6, 18, 5, 0, 0}, /* 2: 3 subsets, 5-bit RGB, 29 */
    {6, 12, 7, 0, 0}, /* 3: 2 subsets, 7-bit RGB, 4 p-bits, 30 */
    {3, 6, 5, 2, 6},  /* 4: rotation + index selector, 5-bit RGB, 6-bit A, 31 + 47 */
    {2, 6, 7, 2, 8},  /* 5: rotation, 7-bit RGB, 8-bit A, 31 + 31 */
    {0, 6, 7, 2, 7},  /* 6: 7-bit RGBA, 2 p-bits, 63 */
    {6, 12, 5, 4, 5}, /* 7: 2 subsets, 5-bit RGBA, 4 p-bits, 30 */
};

static inline int bc7_class(uint8_t b0)
{
    if (b0 == 0)
        return 8; /* reserved encoding: its own class, moved unchanged */
    int m = 0;
    while (!(b0 & 1)) {
        b0 >>= 1;
        ++m;
    }
    return m;
}

static inline u128 load128(const uint8_t *p)
{
    u128 v = 0;
    for (int i = 15; i >= 0; --i)
        v = (v << 8) | p[i];
    return v;
}

static inline void store128(uint8_t *p, u128 v)
{
    for (int i = 0; i < 16; ++i) {
        p[i] = (uint8_t)v;
        v >>= 8;
    }
}

static inline u128 bits(u128 v, int pos, int len) { return len == 0 ? 0 : (v >> pos) & ((((u128)1) << len) - 1); }

/* Colour decorrelation of version 2, on the block's own bit layout: every red and every blue endpoint field becomes its
 * difference to the green field of the same endpoint, modulo the field width (sign -1), or gets the green back (sign +1).
 * Endpoint fields lie channel after channel: n reds, n greens, n blues (n = n_rgb / 3), then the alphas (untouched). */
static u128 bc7_green(u128 b, int m, int sign)
{
    const Bc7Mode *d = &kModes[m];
    const int e_start = m + 1 + d->hdr, n = d->n_rgb / 3, w = d->w_rgb;
    const u128 mask = (((u128)1) << w) - 1;
    for (int i = 0; i < n; ++i) {
        const int pr = e_start + i * w, pg = e_start + (n + i) * w, pb = e_start + (2 * n + i) * w;
        const u128 g = bits(b, pg, w);
        const u128 r = sign < 0 ? (bits(b, pr, w) - g) & mask : (bits(b, pr, w) + g) & mask;
        const u128 bl = sign < 0 ? (bits(b, pb, w) - g) & mask : (bits(b, pb, w) + g) & mask;
        b = (b & ~(mask << pr)) | (r << pr);
        b = (b & ~(mask << pb)) | (bl << pb);
    }
    return b;
}

/* Block -> record (both 128 bits).  First red and blue give up their green (bc7_green).  Record, LSB first: marker and
 * header as they are; then the block's last fields (p-bits and index bits); then the low (w - 4) bits of every endpoint
 * field in block order; then the high 4 bits of every endpoint field in block order. */
static u128 bc7_record_of_block(u128 b, int m)
{
    if (m == 8)
        return b;
    b = bc7_green(b, m, -1);
    const Bc7Mode *d = &kModes[m];
    const int e_start = m + 1 + d->hdr;
    const int e_end = e_start + d->n_rgb * d->w_rgb + d->n_a * d->w_a;
    const int tail_len = 128 - e_end;
    u128 r = bits(b, 0, e_start);
    int at = e_start;
    r |= bits(b, e_end, tail_len) << at;
    at += tail_len;
    const int ne = d->n_rgb + d->n_a;
    int pos = e_start;
    for (int e = 0; e < ne; ++e) { /* low parts */
        const int w = e < d->n_rgb ? d->w_rgb : d->w_a;
        r |= bits(b, pos, w - 4) << at;
        at += w - 4;
        pos += w;
    }
    pos = e_start;
    for (int e = 0; e < ne; ++e) { /* high nibbles */
        const int w = e < d->n_rgb ? d->w_rgb : d->w_a;
        r |= bits(b, pos + w - 4, 4) << at;
        at += 4;
        pos += w;
    }
    return r;
}

static u128 bc7_block_of_record(u128 r, int m)
{
    if (m == 8)
        return r;
    const Bc7Mode *d = &kModes[m];
    const int e_start = m + 1 + d->hdr;
    const int e_end = e_start + d->n_rgb * d->w_rgb + d->n_a * d->w_a;
    const int tail_len = 128 - e_end;
    const int ne = d->n_rgb + d->n_a;
    u128 b = bits(r, 0, e_start);
    b |= bits(r, e_start, tail_len) << e_end;
    int lo_at = e_start + tail_len;
    int lo_total = 0;
    for (int e = 0; e < ne; ++e)
        lo_total += (e < d->n_rgb ? d->w_rgb : d->w_a) - 4;
    int hi_at = lo_at + lo_total;
    int pos = e_start;
    for (int e = 0; e < ne; ++e) {
        const int w = e < d->n_rgb ? d->w_rgb : d->w_a;
        b |= (bits(r, lo_at, w - 4) | (bits(r, hi_at, 4) << (w - 4))) << pos;
        lo_at += w - 4;
        hi_at += 4;
        pos += w;
    }
    return bc7_green(b, m, +1);
}

/* One part of the transformed buffer: `n` blocks (a run of whole granules, or the last partial granule) whose streams
 * start at `soa`:  Q8 8n | Q2 2n | B0 n | B1 n | B2 n | B3 n | B4 n | F n.  Record byte 0 -> F (block order); bytes
 * 1..8 -> Q8, 9..10 -> Q2, 11..15 -> B0..B4, all at the block's SORTED position: inside every granule of BC7_GRANULE
 * blocks the blocks are ordered by class (mode 0..7, then the reserved class), blocks of one class keep their order. */
static void bc7_part(const uint8_t *aos, uint8_t *soa, size_t n, int inverse, uint8_t *aos_out)
{
    static const int off[8] = {0, 8, 10, 11, 12, 13, 14, 15};
    for (size_t g0 = 0; g0 < n; g0 += BC7_GRANULE) {
        const size_t gn = n - g0 < BC7_GRANULE ? n - g0 : BC7_GRANULE;
        size_t count[9] = {0}, base[9], next[9];
        for (size_t i = 0; i < gn; ++i) {
            const uint8_t f = inverse ? soa[off[7] * n + g0 + i] : aos[16 * (g0 + i)];
            count[bc7_class(f)]++;
        }
        size_t at = 0;
        for (int c = 0; c < 9; ++c) {
            base[c] = at;
            next[c] = at;
            at += count[c];
        }
        (void)base;
        for (size_t i = 0; i < gn; ++i) {
            uint8_t rec[16];
            if (!inverse) {
                const int c = bc7_class(aos[16 * (g0 + i)]);
                const size_t j = g0 + next[c]++;
                store128(rec, bc7_record_of_block(load128(aos + 16 * (g0 + i)), c));
                soa[off[7] * n + g0 + i] = rec[0];
                memcpy(soa + off[0] * n + 8 * j, rec + 1, 8);
                memcpy(soa + off[1] * n + 2 * j, rec + 9, 2);
                for (int k = 0; k < 5; ++k)
                    soa[off[2 + k] * n + j] = rec[11 + k];
            } else {
                rec[0] = soa[off[7] * n + g0 + i];
                const int c = bc7_class(rec[0]);
                const size_t j = g0 + next[c]++;
                memcpy(rec + 1, soa + off[0] * n + 8 * j, 8);
                memcpy(rec + 9, soa + off[1] * n + 2 * j, 2);
                for (int k = 0; k < 5; ++k)
                    rec[11 + k] = soa[off[2 + k] * n + j];
                store128(aos_out + 16 * (g0 + i), bc7_block_of_record(load128(rec), c));
            }
        }
    }
}

/* Whole buffer = [main part: the first N - N % BC7_GRANULE blocks][tail part: the last N % BC7_GRANULE blocks], each
 * with its own streams (so every stream of the main part starts on a multiple of the granule size). */
void oracle_transform_bc7(const uint8_t *in, uint8_t *out, size_t len)
{
    const size_t n = len / 16, main_n = n - n % BC7_GRANULE;
    bc7_part(in, out, main_n, 0, NULL);
    bc7_part(in + 16 * main_n, out + 16 * main_n, n - main_n, 0, NULL);
}

void oracle_untransform_bc7(const uint8_t *in, uint8_t *out, size_t len)
{
    const size_t n = len / 16, main_n = n - n % BC7_GRANULE;
    bc7_part(NULL, (uint8_t *)in, main_n, 1, out);
    bc7_part(NULL, (uint8_t *)in + 16 * main_n, n - main_n, 1, out + 16 * main_n);
}

/* one block <-> its record, for field-level tests */
void oracle_bc7_record_of_block(const uint8_t *block, uint8_t *record)
{
    store128(record, bc7_record_of_block(load128(block), bc7_class(block[0])));
}

void oracle_bc7_block_of_record(const uint8_t *record, uint8_t *block)
{
    store128(block, bc7_block_of_record(load128(record), bc7_class(record[0])));
}

unsigned oracle_bc7_granule(void) { return BC7_GRANULE; }

/* force a valid mode marker into byte 0 of every block: mode = (byte 15 & 7) */
void oracle_bc7_force_modes(uint8_t *blocks, size_t len)
{
    for (size_t i = 0; i + 16 <= len; i += 16) {
        const int m = blocks[i + 15] & 7;
        const uint8_t low = (uint8_t)((2u << m) - 1u);
        blocks[i] = (uint8_t)((blocks[i] & (uint8_t)~low) | (uint8_t)(1u << m));
    }
}
