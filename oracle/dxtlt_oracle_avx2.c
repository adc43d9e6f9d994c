/*
 * dxtlt_oracle_avx2.c -- AVX2 and AVX-512BW ports of the reference's SIMD strategy for BC1 default settings (YCoCg Variant1 + split
 * colour endpoints), forward and inverse.  TEST INFRASTRUCTURE ONLY: it exists so that bench.py's cpu_baseline can also
 * quote a vectorised CPU figure ("the reference's own SIMD path timed on the GPU box's host cores", BASELINE.json);
 * tests/test_oracle.py requires it to equal the scalar oracle byte for byte.
 *
 * Strategy followed (not code): /root/reference/src/core/dxt-lossless-transform-bc1/src/transform/
 * with_split_colour_and_recorr/transform/avx2.rs:13-130 -- 16 blocks (128 B) per iteration, dword de-interleave of
 * colours and indices, YCoCg-R on 16-bit lanes (dxt-lossless-transform-common/src/intrinsics/color_565/decorrelate/
 * avx2.rs:41-74), c0/c1 word split, four stores; the tail falls through to the scalar loop.  Written from that
 * description with this file's own shuffle choices.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "dxtlt_oracle.h"

#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#define HAVE_X86 1
#else
#define HAVE_X86 0
#endif

int oracle_simd_available(void)
{
#if HAVE_X86
    return __builtin_cpu_supports("avx2") ? 1 : 0;
#else
    return 0;
#endif
}

/* 0 = scalar only, 2 = AVX2, 5 = AVX-512BW (what oracle_bc1_default_simd_range will use unless capped) */
static int g_simd_cap = 5;
int oracle_simd_level(void)
{
#if HAVE_X86
    if (g_simd_cap >= 5 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw"))
        return 5;
    if (g_simd_cap >= 2 && __builtin_cpu_supports("avx2"))
        return 2;
#endif
    return 0;
}
/* tests and the bench's ISA comparison: cap the level (0, 2 or 5); returns the level now in effect */
int oracle_simd_set_cap(int cap)
{
    g_simd_cap = cap;
    return oracle_simd_level();
}

#if HAVE_X86

#define TGT __attribute__((target("avx2")))

/* YCoCg-R variant 1 on sixteen RGB565 values in 16-bit lanes */
TGT static inline __m256i decorrelate_var1_epi16(__m256i v)
{
    const __m256i m5 = _mm256_set1_epi16(0x1F);
    const __m256i r = _mm256_srli_epi16(v, 11);
    const __m256i g = _mm256_and_si256(_mm256_srli_epi16(v, 6), m5);
    const __m256i gl = _mm256_and_si256(v, _mm256_set1_epi16(0x20));
    const __m256i b = _mm256_and_si256(v, m5);
    const __m256i co = _mm256_and_si256(_mm256_sub_epi16(r, b), m5);
    const __m256i t = _mm256_and_si256(_mm256_add_epi16(b, _mm256_srli_epi16(co, 1)), m5);
    const __m256i cg = _mm256_and_si256(_mm256_sub_epi16(g, t), m5);
    const __m256i y = _mm256_and_si256(_mm256_add_epi16(t, _mm256_srli_epi16(cg, 1)), m5);
    return _mm256_or_si256(_mm256_or_si256(_mm256_slli_epi16(y, 11), _mm256_slli_epi16(co, 6)), _mm256_or_si256(gl, cg));
}

TGT static inline __m256i recorrelate_var1_epi16(__m256i v)
{
    const __m256i m5 = _mm256_set1_epi16(0x1F);
    const __m256i y = _mm256_srli_epi16(v, 11);
    const __m256i co = _mm256_and_si256(_mm256_srli_epi16(v, 6), m5);
    const __m256i gl = _mm256_and_si256(v, _mm256_set1_epi16(0x20));
    const __m256i cg = _mm256_and_si256(v, m5);
    const __m256i t = _mm256_and_si256(_mm256_sub_epi16(y, _mm256_srli_epi16(cg, 1)), m5);
    const __m256i g = _mm256_and_si256(_mm256_add_epi16(cg, t), m5);
    const __m256i b = _mm256_and_si256(_mm256_sub_epi16(t, _mm256_srli_epi16(co, 1)), m5);
    const __m256i r = _mm256_and_si256(_mm256_add_epi16(b, co), m5);
    return _mm256_or_si256(_mm256_or_si256(_mm256_slli_epi16(r, 11), _mm256_slli_epi16(g, 6)), _mm256_or_si256(gl, b));
}

/* blocks [first, first+count) of an n_total-block buffer; count is a multiple of 16 */
TGT static void bc1_default_fwd_avx2(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count)
{
    uint8_t *c0 = out + 2 * first, *c1 = out + 2 * n_total + 2 * first, *idx = out + 4 * n_total + 4 * first;
    const uint8_t *p = in + 8 * first;
    /* inside each 128-bit lane: low words of the four dwords first, then their high words */
    const __m256i words = _mm256_setr_epi8(0, 1, 4, 5, 8, 9, 12, 13, 2, 3, 6, 7, 10, 11, 14, 15, 0, 1, 4, 5, 8, 9, 12, 13, 2, 3,
                                           6, 7, 10, 11, 14, 15);
    for (size_t i = 0; i < count; i += 16, p += 128, c0 += 32, c1 += 32, idx += 64) {
        const __m256 a = _mm256_loadu_ps((const float *)(p + 0));   /* blocks 0-3 */
        const __m256 b = _mm256_loadu_ps((const float *)(p + 32));  /* blocks 4-7 */
        const __m256 c = _mm256_loadu_ps((const float *)(p + 64));
        const __m256 d = _mm256_loadu_ps((const float *)(p + 96));
        /* even dwords = colours, odd dwords = indices; shuffle_ps works per 128-bit lane, permute4x64 restores order */
        __m256i col_lo = _mm256_permute4x64_epi64(_mm256_castps_si256(_mm256_shuffle_ps(a, b, 0x88)), 0xD8);
        __m256i col_hi = _mm256_permute4x64_epi64(_mm256_castps_si256(_mm256_shuffle_ps(c, d, 0x88)), 0xD8);
        const __m256i idx_lo = _mm256_permute4x64_epi64(_mm256_castps_si256(_mm256_shuffle_ps(a, b, 0xDD)), 0xD8);
        const __m256i idx_hi = _mm256_permute4x64_epi64(_mm256_castps_si256(_mm256_shuffle_ps(c, d, 0xDD)), 0xD8);
        col_lo = decorrelate_var1_epi16(col_lo);
        col_hi = decorrelate_var1_epi16(col_hi);
        /* split (c0,c1) word pairs: [c0 x4 | c1 x4] per lane -> [c0 x8][c1 x8] */
        const __m256i s_lo = _mm256_permute4x64_epi64(_mm256_shuffle_epi8(col_lo, words), 0xD8);
        const __m256i s_hi = _mm256_permute4x64_epi64(_mm256_shuffle_epi8(col_hi, words), 0xD8);
        _mm256_storeu_si256((__m256i *)c0, _mm256_permute2x128_si256(s_lo, s_hi, 0x20));
        _mm256_storeu_si256((__m256i *)c1, _mm256_permute2x128_si256(s_lo, s_hi, 0x31));
        _mm256_storeu_si256((__m256i *)idx, idx_lo);
        _mm256_storeu_si256((__m256i *)(idx + 32), idx_hi);
    }
}

TGT static void bc1_default_inv_avx2(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count)
{
    const uint8_t *c0 = in + 2 * first, *c1 = in + 2 * n_total + 2 * first, *idx = in + 4 * n_total + 4 * first;
    uint8_t *p = out + 8 * first;
    for (size_t i = 0; i < count; i += 16, p += 128, c0 += 32, c1 += 32, idx += 64) {
        const __m256i v0 = _mm256_loadu_si256((const __m256i *)c0);  /* c0 of 16 blocks */
        const __m256i v1 = _mm256_loadu_si256((const __m256i *)c1);
        /* per-lane word interleave gives blocks 0-3,8-11 / 4-7,12-15; fix the order with lane permutes */
        const __m256i lo = _mm256_unpacklo_epi16(v0, v1);
        const __m256i hi = _mm256_unpackhi_epi16(v0, v1);
        __m256i col_a = _mm256_permute2x128_si256(lo, hi, 0x20);  /* blocks 0-7 */
        __m256i col_b = _mm256_permute2x128_si256(lo, hi, 0x31);  /* blocks 8-15 */
        col_a = recorrelate_var1_epi16(col_a);
        col_b = recorrelate_var1_epi16(col_b);
        const __m256i ia = _mm256_loadu_si256((const __m256i *)idx);
        const __m256i ib = _mm256_loadu_si256((const __m256i *)(idx + 32));
        /* dword interleave colours/indices -> blocks */
        const __m256i a_lo = _mm256_unpacklo_epi32(col_a, ia), a_hi = _mm256_unpackhi_epi32(col_a, ia);
        const __m256i b_lo = _mm256_unpacklo_epi32(col_b, ib), b_hi = _mm256_unpackhi_epi32(col_b, ib);
        _mm256_storeu_si256((__m256i *)(p + 0), _mm256_permute2x128_si256(a_lo, a_hi, 0x20));
        _mm256_storeu_si256((__m256i *)(p + 32), _mm256_permute2x128_si256(a_lo, a_hi, 0x31));
        _mm256_storeu_si256((__m256i *)(p + 64), _mm256_permute2x128_si256(b_lo, b_hi, 0x20));
        _mm256_storeu_si256((__m256i *)(p + 96), _mm256_permute2x128_si256(b_lo, b_hi, 0x31));
    }
}

/* ---- AVX-512BW: 32 blocks (256 B) per iteration.  Strategy followed (not code): .../with_split_colour_and_recorr/
 * transform/avx512bw.rs and untransform/avx512bw.rs -- two-source dword permutes separate colours from indices,
 * YCoCg-R on 32 16-bit lanes, one word permute splits c0 / c1.  Index vectors are this file's own. ---- */
#define TGT512 __attribute__((target("avx512f,avx512bw")))

TGT512 static inline __m512i decorrelate_var1_epi16_512(__m512i v)
{
    const __m512i m5 = _mm512_set1_epi16(0x1F);
    const __m512i r = _mm512_srli_epi16(v, 11);
    const __m512i g = _mm512_and_si512(_mm512_srli_epi16(v, 6), m5);
    const __m512i gl = _mm512_and_si512(v, _mm512_set1_epi16(0x20));
    const __m512i b = _mm512_and_si512(v, m5);
    const __m512i co = _mm512_and_si512(_mm512_sub_epi16(r, b), m5);
    const __m512i t = _mm512_and_si512(_mm512_add_epi16(b, _mm512_srli_epi16(co, 1)), m5);
    const __m512i cg = _mm512_and_si512(_mm512_sub_epi16(g, t), m5);
    const __m512i y = _mm512_and_si512(_mm512_add_epi16(t, _mm512_srli_epi16(cg, 1)), m5);
    return _mm512_or_si512(_mm512_or_si512(_mm512_slli_epi16(y, 11), _mm512_slli_epi16(co, 6)), _mm512_or_si512(gl, cg));
}

TGT512 static inline __m512i recorrelate_var1_epi16_512(__m512i v)
{
    const __m512i m5 = _mm512_set1_epi16(0x1F);
    const __m512i y = _mm512_srli_epi16(v, 11);
    const __m512i co = _mm512_and_si512(_mm512_srli_epi16(v, 6), m5);
    const __m512i gl = _mm512_and_si512(v, _mm512_set1_epi16(0x20));
    const __m512i cg = _mm512_and_si512(v, m5);
    const __m512i t = _mm512_and_si512(_mm512_sub_epi16(y, _mm512_srli_epi16(cg, 1)), m5);
    const __m512i g = _mm512_and_si512(_mm512_add_epi16(cg, t), m5);
    const __m512i b = _mm512_and_si512(_mm512_sub_epi16(t, _mm512_srli_epi16(co, 1)), m5);
    const __m512i r = _mm512_and_si512(_mm512_add_epi16(b, co), m5);
    return _mm512_or_si512(_mm512_or_si512(_mm512_slli_epi16(r, 11), _mm512_slli_epi16(g, 6)), _mm512_or_si512(gl, b));
}

/* count is a multiple of 32 */
TGT512 static void bc1_default_fwd_avx512(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count)
{
    uint8_t *c0 = out + 2 * first, *c1 = out + 2 * n_total + 2 * first, *idx = out + 4 * n_total + 4 * first;
    const uint8_t *p = in + 8 * first;
    /* dword i of (a, b): even dwords are colours, odd dwords indices */
    const __m512i even = _mm512_setr_epi32(0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30);
    const __m512i odd = _mm512_setr_epi32(1, 3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31);
    /* word i of (lo, hi): even words are c0, odd words c1 */
    const __m512i w_even = _mm512_set_epi16(62, 60, 58, 56, 54, 52, 50, 48, 46, 44, 42, 40, 38, 36, 34, 32, 30, 28, 26, 24, 22,
                                            20, 18, 16, 14, 12, 10, 8, 6, 4, 2, 0);
    const __m512i w_odd = _mm512_set_epi16(63, 61, 59, 57, 55, 53, 51, 49, 47, 45, 43, 41, 39, 37, 35, 33, 31, 29, 27, 25, 23,
                                           21, 19, 17, 15, 13, 11, 9, 7, 5, 3, 1);
    for (size_t i = 0; i < count; i += 32, p += 256, c0 += 64, c1 += 64, idx += 128) {
        const __m512i a = _mm512_loadu_si512((const void *)(p + 0));     /* blocks 0-7 */
        const __m512i b = _mm512_loadu_si512((const void *)(p + 64));    /* blocks 8-15 */
        const __m512i c = _mm512_loadu_si512((const void *)(p + 128));
        const __m512i d = _mm512_loadu_si512((const void *)(p + 192));
        const __m512i col_lo = decorrelate_var1_epi16_512(_mm512_permutex2var_epi32(a, even, b));   /* blocks 0-15 */
        const __m512i col_hi = decorrelate_var1_epi16_512(_mm512_permutex2var_epi32(c, even, d));   /* blocks 16-31 */
        _mm512_storeu_si512((void *)c0, _mm512_permutex2var_epi16(col_lo, w_even, col_hi));
        _mm512_storeu_si512((void *)c1, _mm512_permutex2var_epi16(col_lo, w_odd, col_hi));
        _mm512_storeu_si512((void *)idx, _mm512_permutex2var_epi32(a, odd, b));
        _mm512_storeu_si512((void *)(idx + 64), _mm512_permutex2var_epi32(c, odd, d));
    }
}

TGT512 static void bc1_default_inv_avx512(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count)
{
    const uint8_t *c0 = in + 2 * first, *c1 = in + 2 * n_total + 2 * first, *idx = in + 4 * n_total + 4 * first;
    uint8_t *p = out + 8 * first;
    /* word interleave of (c0, c1): result word 2k = c0[k], 2k+1 = c1[k]; low half k = 0..15, high half k = 16..31 */
    const __m512i il_lo = _mm512_set_epi16(47, 15, 46, 14, 45, 13, 44, 12, 43, 11, 42, 10, 41, 9, 40, 8, 39, 7, 38, 6, 37, 5,
                                           36, 4, 35, 3, 34, 2, 33, 1, 32, 0);
    const __m512i il_hi = _mm512_set_epi16(63, 31, 62, 30, 61, 29, 60, 28, 59, 27, 58, 26, 57, 25, 56, 24, 55, 23, 54, 22, 53,
                                           21, 52, 20, 51, 19, 50, 18, 49, 17, 48, 16);
    /* dword interleave of (colours, indices): blocks 0-7 then 8-15 of a 16-block group */
    const __m512i dl_lo = _mm512_setr_epi32(0, 16, 1, 17, 2, 18, 3, 19, 4, 20, 5, 21, 6, 22, 7, 23);
    const __m512i dl_hi = _mm512_setr_epi32(8, 24, 9, 25, 10, 26, 11, 27, 12, 28, 13, 29, 14, 30, 15, 31);
    for (size_t i = 0; i < count; i += 32, p += 256, c0 += 64, c1 += 64, idx += 128) {
        const __m512i v0 = _mm512_loadu_si512((const void *)c0);   /* c0 of 32 blocks */
        const __m512i v1 = _mm512_loadu_si512((const void *)c1);
        const __m512i col_a = recorrelate_var1_epi16_512(_mm512_permutex2var_epi16(v0, il_lo, v1));   /* blocks 0-15 */
        const __m512i col_b = recorrelate_var1_epi16_512(_mm512_permutex2var_epi16(v0, il_hi, v1));   /* blocks 16-31 */
        const __m512i ia = _mm512_loadu_si512((const void *)idx);
        const __m512i ib = _mm512_loadu_si512((const void *)(idx + 64));
        _mm512_storeu_si512((void *)(p + 0), _mm512_permutex2var_epi32(col_a, dl_lo, ia));
        _mm512_storeu_si512((void *)(p + 64), _mm512_permutex2var_epi32(col_a, dl_hi, ia));
        _mm512_storeu_si512((void *)(p + 128), _mm512_permutex2var_epi32(col_b, dl_lo, ib));
        _mm512_storeu_si512((void *)(p + 192), _mm512_permutex2var_epi32(col_b, dl_hi, ib));
    }
}

/* ---- BC2, default settings (YCoCg Variant1 + split colour endpoints), AVX2: 8 blocks (128 B) per iteration.
 * Strategy followed (not code): /root/reference/src/core/dxt-lossless-transform-bc2/src/transform/
 * with_split_colour_and_recorr/transform/avx2.rs -- 64-bit unpacks separate the alpha qwords from the (colour, index)
 * qwords, a dword shuffle separates colours from indices, YCoCg-R on 16-bit lanes, a word shuffle splits c0 / c1. ---- */
TGT static void bc2_default_fwd_avx2(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count)
{
    uint8_t *alpha = out + 8 * first, *c0 = out + 8 * n_total + 2 * first, *c1 = out + 10 * n_total + 2 * first,
            *idx = out + 12 * n_total + 4 * first;
    const uint8_t *p = in + 16 * first;
    const __m256i words = _mm256_setr_epi8(0, 1, 4, 5, 8, 9, 12, 13, 2, 3, 6, 7, 10, 11, 14, 15, 0, 1, 4, 5, 8, 9, 12, 13, 2, 3,
                                           6, 7, 10, 11, 14, 15);
    for (size_t i = 0; i < count; i += 8, p += 128, alpha += 64, c0 += 16, c1 += 16, idx += 32) {
        const __m256i v0 = _mm256_loadu_si256((const __m256i *)(p + 0));   /* blocks 0, 1: [alpha, colour|index] x 2 */
        const __m256i v1 = _mm256_loadu_si256((const __m256i *)(p + 32));
        const __m256i v2 = _mm256_loadu_si256((const __m256i *)(p + 64));
        const __m256i v3 = _mm256_loadu_si256((const __m256i *)(p + 96));
        /* qword unpack per 128-bit lane: (b0, b2 | b1, b3) -> permute to block order */
        _mm256_storeu_si256((__m256i *)alpha, _mm256_permute4x64_epi64(_mm256_unpacklo_epi64(v0, v1), 0xD8));
        _mm256_storeu_si256((__m256i *)(alpha + 32), _mm256_permute4x64_epi64(_mm256_unpacklo_epi64(v2, v3), 0xD8));
        const __m256 ci01 = _mm256_castsi256_ps(_mm256_permute4x64_epi64(_mm256_unpackhi_epi64(v0, v1), 0xD8));   /* c i c i | c i c i */
        const __m256 ci23 = _mm256_castsi256_ps(_mm256_permute4x64_epi64(_mm256_unpackhi_epi64(v2, v3), 0xD8));
        __m256i col = _mm256_permute4x64_epi64(_mm256_castps_si256(_mm256_shuffle_ps(ci01, ci23, 0x88)), 0xD8);   /* colours of blocks 0-7 */
        const __m256i ind = _mm256_permute4x64_epi64(_mm256_castps_si256(_mm256_shuffle_ps(ci01, ci23, 0xDD)), 0xD8);
        col = decorrelate_var1_epi16(col);
        const __m256i s = _mm256_permute4x64_epi64(_mm256_shuffle_epi8(col, words), 0xD8);   /* c0 x 8 | c1 x 8 */
        _mm_storeu_si128((__m128i *)c0, _mm256_castsi256_si128(s));
        _mm_storeu_si128((__m128i *)c1, _mm256_extracti128_si256(s, 1));
        _mm256_storeu_si256((__m256i *)idx, ind);
    }
}

TGT static void bc2_default_inv_avx2(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count)
{
    const uint8_t *alpha = in + 8 * first, *c0 = in + 8 * n_total + 2 * first, *c1 = in + 10 * n_total + 2 * first,
                  *idx = in + 12 * n_total + 4 * first;
    uint8_t *p = out + 16 * first;
    for (size_t i = 0; i < count; i += 8, p += 128, alpha += 64, c0 += 16, c1 += 16, idx += 32) {
        const __m128i w0 = _mm_loadu_si128((const __m128i *)c0), w1 = _mm_loadu_si128((const __m128i *)c1);
        /* word interleave: blocks 0-3 | 4-7 */
        __m256i col = _mm256_set_m128i(_mm_unpackhi_epi16(w0, w1), _mm_unpacklo_epi16(w0, w1));
        col = recorrelate_var1_epi16(col);
        const __m256i ind = _mm256_loadu_si256((const __m256i *)idx);
        /* (colour, index) qwords: unpack per lane gives blocks 0,1,4,5 / 2,3,6,7 */
        const __m256i lo = _mm256_unpacklo_epi32(col, ind), hi = _mm256_unpackhi_epi32(col, ind);
        const __m256i ci03 = _mm256_permute2x128_si256(lo, hi, 0x20);   /* blocks 0-3 */
        const __m256i ci47 = _mm256_permute2x128_si256(lo, hi, 0x31);   /* blocks 4-7 */
        const __m256i a03 = _mm256_loadu_si256((const __m256i *)alpha), a47 = _mm256_loadu_si256((const __m256i *)(alpha + 32));
        /* interleave alpha qwords with ci qwords: unpack per lane gives blocks 0,2 / 1,3 -> permute sources first */
        const __m256i a03p = _mm256_permute4x64_epi64(a03, 0xD8), ci03p = _mm256_permute4x64_epi64(ci03, 0xD8);
        const __m256i a47p = _mm256_permute4x64_epi64(a47, 0xD8), ci47p = _mm256_permute4x64_epi64(ci47, 0xD8);
        _mm256_storeu_si256((__m256i *)(p + 0), _mm256_unpacklo_epi64(a03p, ci03p));    /* blocks 0, 1 */
        _mm256_storeu_si256((__m256i *)(p + 32), _mm256_unpackhi_epi64(a03p, ci03p));   /* blocks 2, 3 */
        _mm256_storeu_si256((__m256i *)(p + 64), _mm256_unpacklo_epi64(a47p, ci47p));
        _mm256_storeu_si256((__m256i *)(p + 96), _mm256_unpackhi_epi64(a47p, ci47p));
    }
}

/* ---- BC3, "standard" layout (no decorrelation, no splits), AVX2: 8 blocks per iteration.  Strategy followed (not code):
 * /root/reference/src/core/dxt-lossless-transform-bc3/src/transform/standard/transform/avx2.rs:47-140 -- colours and
 * indices leave as whole vectors, the 2 + 6 alpha bytes of every block through byte shuffles. ---- */
TGT static void bc3_standard_fwd_avx2(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count)
{
    uint8_t *aep = out + 2 * first, *aidx = out + 2 * n_total + 6 * first, *col = out + 8 * n_total + 4 * first,
            *idx = out + 12 * n_total + 4 * first;
    const uint8_t *p = in + 16 * first;
    /* per 128-bit lane holding two alpha qwords: bytes 0-11 = the two 6-byte index records, 12-15 = the two endpoint pairs */
    const __m256i pick = _mm256_setr_epi8(2, 3, 4, 5, 6, 7, 10, 11, 12, 13, 14, 15, 0, 1, 8, 9, 2, 3, 4, 5, 6, 7, 10, 11, 12, 13,
                                          14, 15, 0, 1, 8, 9);
    for (size_t i = 0; i < count; i += 8, p += 128, aep += 16, aidx += 48, col += 32, idx += 32) {
        const __m256i v0 = _mm256_loadu_si256((const __m256i *)(p + 0));
        const __m256i v1 = _mm256_loadu_si256((const __m256i *)(p + 32));
        const __m256i v2 = _mm256_loadu_si256((const __m256i *)(p + 64));
        const __m256i v3 = _mm256_loadu_si256((const __m256i *)(p + 96));
        const __m256i a03 = _mm256_shuffle_epi8(_mm256_permute4x64_epi64(_mm256_unpacklo_epi64(v0, v1), 0xD8), pick);
        const __m256i a47 = _mm256_shuffle_epi8(_mm256_permute4x64_epi64(_mm256_unpacklo_epi64(v2, v3), 0xD8), pick);
        const __m256 ci01 = _mm256_castsi256_ps(_mm256_permute4x64_epi64(_mm256_unpackhi_epi64(v0, v1), 0xD8));
        const __m256 ci23 = _mm256_castsi256_ps(_mm256_permute4x64_epi64(_mm256_unpackhi_epi64(v2, v3), 0xD8));
        _mm256_storeu_si256((__m256i *)col, _mm256_permute4x64_epi64(_mm256_castps_si256(_mm256_shuffle_ps(ci01, ci23, 0x88)), 0xD8));
        _mm256_storeu_si256((__m256i *)idx, _mm256_permute4x64_epi64(_mm256_castps_si256(_mm256_shuffle_ps(ci01, ci23, 0xDD)), 0xD8));
        /* 12 index bytes per lane: a 16-byte store whose last 4 bytes the next store overwrites, then exact 12-byte stores */
        const __m128i l0 = _mm256_castsi256_si128(a03), l1 = _mm256_extracti128_si256(a03, 1);
        const __m128i l2 = _mm256_castsi256_si128(a47), l3 = _mm256_extracti128_si256(a47, 1);
        _mm_storeu_si128((__m128i *)(aidx + 0), l0);
        _mm_storeu_si128((__m128i *)(aidx + 12), l1);
        _mm_storeu_si128((__m128i *)(aidx + 24), l2);
        memcpy(aidx + 36, &l3, 12);
        const uint32_t e[4] = {(uint32_t)_mm_extract_epi32(l0, 3), (uint32_t)_mm_extract_epi32(l1, 3),
                               (uint32_t)_mm_extract_epi32(l2, 3), (uint32_t)_mm_extract_epi32(l3, 3)};
        memcpy(aep, e, 16);
    }
}

TGT static void bc3_standard_inv_avx2(const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count)
{
    const uint8_t *aep = in + 2 * first, *aidx = in + 2 * n_total + 6 * first, *col = in + 8 * n_total + 4 * first,
                  *idx = in + 12 * n_total + 4 * first;
    uint8_t *p = out + 16 * first;
    /* inverse of `pick`: lane bytes = [12 index bytes | 4 endpoint bytes] -> two alpha qwords */
    const __m128i unpick = _mm_setr_epi8(12, 13, 0, 1, 2, 3, 4, 5, 14, 15, 6, 7, 8, 9, 10, 11);
    for (size_t i = 0; i < count; i += 8, p += 128, aep += 16, aidx += 48, col += 32, idx += 32) {
        uint32_t e[4];
        memcpy(e, aep, 16);
        __m128i a[4];
        for (int k = 0; k < 4; ++k) {
            __m128i t;
            memcpy(&t, aidx + 12 * k, 12);   /* exact: the last record must not read past the section */
            a[k] = _mm_shuffle_epi8(_mm_insert_epi32(t, (int)e[k], 3), unpick);   /* alpha qwords of blocks 2k, 2k + 1 */
        }
        const __m256i cv = _mm256_loadu_si256((const __m256i *)col), iv = _mm256_loadu_si256((const __m256i *)idx);
        const __m256i lo = _mm256_unpacklo_epi32(cv, iv), hi = _mm256_unpackhi_epi32(cv, iv);   /* blocks 0,1,4,5 / 2,3,6,7 */
        const __m128i ci01 = _mm256_castsi256_si128(lo), ci23 = _mm256_castsi256_si128(hi);
        const __m128i ci45 = _mm256_extracti128_si256(lo, 1), ci67 = _mm256_extracti128_si256(hi, 1);
        const __m128i ci[4] = {ci01, ci23, ci45, ci67};
        for (int k = 0; k < 4; ++k) {
            _mm_storeu_si128((__m128i *)(p + 32 * k), _mm_unpacklo_epi64(a[k], ci[k]));
            _mm_storeu_si128((__m128i *)(p + 32 * k + 16), _mm_unpackhi_epi64(a[k], ci[k]));
        }
    }
}

#endif /* HAVE_X86 */

/* scalar range kernels live in dxtlt_oracle.c; re-stated minimally here for the tail */
static void bc1_default_scalar_range(int inverse, const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count)
{
    for (size_t b = first; b < first + count; ++b) {
        if (!inverse) {
            uint16_t c0, c1;
            uint32_t ix;
            memcpy(&c0, in + 8 * b, 2);
            memcpy(&c1, in + 8 * b + 2, 2);
            memcpy(&ix, in + 8 * b + 4, 4);
            c0 = oracle_decorrelate_565(c0, ORACLE_YCOCG_VAR1);
            c1 = oracle_decorrelate_565(c1, ORACLE_YCOCG_VAR1);
            memcpy(out + 2 * b, &c0, 2);
            memcpy(out + 2 * n_total + 2 * b, &c1, 2);
            memcpy(out + 4 * n_total + 4 * b, &ix, 4);
        } else {
            uint16_t c0, c1;
            uint32_t ix;
            memcpy(&c0, in + 2 * b, 2);
            memcpy(&c1, in + 2 * n_total + 2 * b, 2);
            memcpy(&ix, in + 4 * n_total + 4 * b, 4);
            c0 = oracle_recorrelate_565(c0, ORACLE_YCOCG_VAR1);
            c1 = oracle_recorrelate_565(c1, ORACLE_YCOCG_VAR1);
            memcpy(out + 8 * b, &c0, 2);
            memcpy(out + 8 * b + 2, &c1, 2);
            memcpy(out + 8 * b + 4, &ix, 4);
        }
    }
}

void oracle_bc1_default_simd_range(int inverse, const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count)
{
    size_t body = 0;
#if HAVE_X86
    const int level = oracle_simd_level();
    if (level == 5) {
        body = count & ~(size_t)31;
        if (inverse)
            bc1_default_inv_avx512(in, out, n_total, first, body);
        else
            bc1_default_fwd_avx512(in, out, n_total, first, body);
    } else if (level == 2) {
        body = count & ~(size_t)15;
        if (inverse)
            bc1_default_inv_avx2(in, out, n_total, first, body);
        else
            bc1_default_fwd_avx2(in, out, n_total, first, body);
    }
#endif
    bc1_default_scalar_range(inverse, in, out, n_total, first + body, count - body);
}

/* contiguous block ranges over `threads` pthreads (cpu_baseline only) */
#include <pthread.h>
#include <stdlib.h>

struct simd_job {
    int inverse;
    const uint8_t *in;
    uint8_t *out;
    size_t n_total, first, count;
};

static void *simd_worker(void *arg)
{
    struct simd_job *j = (struct simd_job *)arg;
    oracle_bc1_default_simd_range(j->inverse, j->in, j->out, j->n_total, j->first, j->count);
    return NULL;
}

void oracle_bc1_default_simd_mt(int inverse, const uint8_t *in, uint8_t *out, size_t len, int threads)
{
    const size_t n = len / 8;
    if (threads < 1)
        threads = 1;
    if ((size_t)threads > n / 16 + 1)
        threads = (int)(n / 16 + 1);
    struct simd_job *jobs = (struct simd_job *)calloc((size_t)threads, sizeof *jobs);
    pthread_t *tids = (pthread_t *)calloc((size_t)threads, sizeof *tids);
    size_t per = (n + (size_t)threads - 1) / (size_t)threads;
    per = (per + 31) & ~(size_t)31; /* keep every range a multiple of the widest vector loop */
    for (int t = 0; t < threads; ++t) {
        size_t first = per * (size_t)t;
        size_t count = first >= n ? 0 : (first + per > n ? n - first : per);
        struct simd_job j = {inverse, in, out, n, first, count};
        jobs[t] = j;
        if (t > 0)
            pthread_create(&tids[t], NULL, simd_worker, &jobs[t]);
    }
    simd_worker(&jobs[0]);
    for (int t = 1; t < threads; ++t)
        pthread_join(tids[t], NULL);
    free(jobs);
    free(tids);
}

/* ---- BC2 default / BC3 standard: AVX2 body + scalar tail (the scalar oracle's own functions on the tail's blocks) ---- */
void oracle_bc23_simd_range(int kind, int inverse, const uint8_t *in, uint8_t *out, size_t n_total, size_t first, size_t count)
{
    /* kind 2 = BC2 {Variant1, split colours}; kind 3 = BC3 standard {None, no splits} */
    size_t body = 0;
#if HAVE_X86
    if (oracle_simd_level() >= 2) {
        body = count & ~(size_t)7;
        if (kind == 2) {
            if (inverse) bc2_default_inv_avx2(in, out, n_total, first, body);
            else bc2_default_fwd_avx2(in, out, n_total, first, body);
        } else {
            if (inverse) bc3_standard_inv_avx2(in, out, n_total, first, body);
            else bc3_standard_fwd_avx2(in, out, n_total, first, body);
        }
    }
#endif
    if (count > body)
        oracle_transform_range(kind, inverse, in, out, n_total, first + body, count - body, kind == 2 ? ORACLE_YCOCG_VAR1 : ORACLE_YCOCG_NONE,
                               0, kind == 2 ? 1 : 0);
}

struct simd23_job {
    int kind, inverse;
    const uint8_t *in;
    uint8_t *out;
    size_t n_total, first, count;
};

static void *simd23_worker(void *arg)
{
    struct simd23_job *j = (struct simd23_job *)arg;
    oracle_bc23_simd_range(j->kind, j->inverse, j->in, j->out, j->n_total, j->first, j->count);
    return NULL;
}

void oracle_bc23_simd_mt(int kind, int inverse, const uint8_t *in, uint8_t *out, size_t len, int threads)
{
    const size_t n = len / 16;
    if (threads < 1)
        threads = 1;
    if ((size_t)threads > n / 8 + 1)
        threads = (int)(n / 8 + 1);
    struct simd23_job *jobs = (struct simd23_job *)calloc((size_t)threads, sizeof *jobs);
    pthread_t *tids = (pthread_t *)calloc((size_t)threads, sizeof *tids);
    size_t per = (n + (size_t)threads - 1) / (size_t)threads;
    per = (per + 7) & ~(size_t)7;
    for (int t = 0; t < threads; ++t) {
        size_t first = per * (size_t)t;
        size_t count = first >= n ? 0 : (first + per > n ? n - first : per);
        struct simd23_job j = {kind, inverse, in, out, n, first, count};
        jobs[t] = j;
        if (t > 0)
            pthread_create(&tids[t], NULL, simd23_worker, &jobs[t]);
    }
    simd23_worker(&jobs[0]);
    for (int t = 1; t < threads; ++t)
        pthread_join(tids[t], NULL);
    free(jobs);
    free(tids);
}
