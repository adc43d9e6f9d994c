/*
 * dxtlt_oracle_bc7.c -- CPU statement of the BC7 mode-split transform, version 0 (docs/BC7_FORMAT.md).
 *
 * TEST INFRASTRUCTURE ONLY (see dxtlt_oracle.h).  PARITY UNPINNED: the reference has no BC7 transform
 * (/root/reference/src/core/dxt-lossless-transform-bc7/src/lib.rs:1-13); the format is defined by this build, and this
 * file is its executable definition, not a restatement of reference behaviour.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

static const int kHead[9] = {9, 9, 11, 11, 5, 7, 7, 11, 15};

static inline int bc7_mode(uint8_t b0)
{
    if (b0 == 0)
        return 8;
    int m = 0;
    while (!(b0 & 1)) {
        b0 >>= 1;
        ++m;
    }
    return m;
}

/* counts[9] <- histogram of modes over `n` blocks whose byte 0 is found at first[i * stride] */
static void bc7_histogram(const uint8_t *first, size_t stride, size_t n, uint64_t counts[9])
{
    memset(counts, 0, 9 * sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i)
        counts[bc7_mode(first[i * stride])]++;
}

static void bc7_bases(size_t n, const uint64_t counts[9], uint64_t head_base[9], uint64_t tail_base[9])
{
    uint64_t pos = n;
    for (int m = 0; m < 9; ++m) {
        head_base[m] = pos;
        pos += counts[m] * (uint64_t)kHead[m];
        tail_base[m] = pos;
        pos += counts[m] * (uint64_t)(15 - kHead[m]);
    }
}

void oracle_transform_bc7(const uint8_t *in, uint8_t *out, size_t len)
{
    const size_t n = len / 16;
    uint64_t counts[9], hb[9], tb[9], rank[9] = {0};
    bc7_histogram(in, 16, n, counts);
    bc7_bases(n, counts, hb, tb);
    for (size_t i = 0; i < n; ++i) {
        const uint8_t *blk = in + 16 * i;
        const int m = bc7_mode(blk[0]);
        const int h = kHead[m];
        out[i] = blk[0];
        memcpy(out + hb[m] + rank[m] * (uint64_t)h, blk + 1, (size_t)h);
        memcpy(out + tb[m] + rank[m] * (uint64_t)(15 - h), blk + 1 + h, (size_t)(15 - h));
        rank[m]++;
    }
}

void oracle_untransform_bc7(const uint8_t *in, uint8_t *out, size_t len)
{
    const size_t n = len / 16;
    uint64_t counts[9], hb[9], tb[9], rank[9] = {0};
    bc7_histogram(in, 1, n, counts);
    bc7_bases(n, counts, hb, tb);
    for (size_t i = 0; i < n; ++i) {
        uint8_t *blk = out + 16 * i;
        const int m = bc7_mode(in[i]);
        const int h = kHead[m];
        blk[0] = in[i];
        memcpy(blk + 1, in + hb[m] + rank[m] * (uint64_t)h, (size_t)h);
        memcpy(blk + 1 + h, in + tb[m] + rank[m] * (uint64_t)(15 - h), (size_t)(15 - h));
        rank[m]++;
    }
}

/* Synthetic mode-mixed blocks: random bytes, then byte 0's low bits are forced to the
 * marker of mode (r % 8) where r is taken from the block's last byte BEFORE forcing -- deterministic from the seed. */
void oracle_bc7_force_modes(uint8_t *blocks, size_t len)
{
    const size_t n = len / 16;
    for (size_t i = 0; i < n; ++i) {
        uint8_t *b = blocks + 16 * i;
        const int m = b[15] & 7;
        b[0] = (uint8_t)((b[0] & ~((2u << m) - 1u)) | (1u << m));
    }
}
