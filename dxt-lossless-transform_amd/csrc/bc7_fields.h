// bc7_fields.h -- BC7 block <-> record (docs/BC7_FORMAT.md, version 2): the per-mode bit-field permutation, with every
// position a compile-time constant so that it compiles to v_bfe / v_lshl_or / v_alignbit on four dwords.
// Shared by the kernels (bc7_kernels.hip) and, compiled for the host, by tests/cpp/bc7_fields_shim.cpp, which checks it
// against the oracle's one-field-at-a-time statement.
//
// Mode bit fields: /root/reference/src/assets/research/dds-bc7-blocks.hexpat:286-654 (the reference documents the
// modes; it has no BC7 transform).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BC7_HD __host__ __device__ __forceinline__
#else
#define BC7_HD inline
#endif

namespace dxtlt {
namespace bc7 {

constexpr int kGranule = 1024;  // blocks per sort granule

struct B128 {
    uint32_t d[4];
};

// header bits, colour endpoint fields and their width, alpha endpoint fields and their width; the marker is m + 1 bits,
// the p-bits and index bits fill the rest of the block
struct ModeDesc {
    int hdr, n_rgb, w_rgb, n_a, w_a;
};
constexpr ModeDesc kModeDesc[8] = {
    {4, 18, 4, 0, 0}, {6, 12, 6, 0, 0}, {6, 18, 5, 0, 0}, {6, 12, 7, 0, 0},
    {3, 6, 5, 2, 6},  {2, 6, 7, 2, 8},  {0, 6, 7, 2, 7},  {6, 12, 5, 4, 5},
};

constexpr int n_endpoints(int m) { return kModeDesc[m].n_rgb + kModeDesc[m].n_a; }
constexpr int endpoint_width(int m, int e) { return e < kModeDesc[m].n_rgb ? kModeDesc[m].w_rgb : kModeDesc[m].w_a; }
constexpr int endpoints_start(int m) { return m + 1 + kModeDesc[m].hdr; }
constexpr int endpoint_pos(int m, int e)
{
    int p = endpoints_start(m);
    for (int i = 0; i < e; ++i)
        p += endpoint_width(m, i);
    return p;
}
constexpr int endpoints_end(int m) { return endpoint_pos(m, n_endpoints(m)); }
constexpr int tail_len(int m) { return 128 - endpoints_end(m); }          // p-bits + index bits
constexpr int lows_start(int m) { return endpoints_start(m) + tail_len(m); }
constexpr int low_pos(int m, int e)
{
    int p = lows_start(m);
    for (int i = 0; i < e; ++i)
        p += endpoint_width(m, i) - 4;
    return p;
}
constexpr int highs_start(int m) { return low_pos(m, n_endpoints(m)); }
static_assert(highs_start(0) + 4 * n_endpoints(0) == 128 && highs_start(4) + 4 * n_endpoints(4) == 128 &&
                  highs_start(5) + 4 * n_endpoints(5) == 128 && highs_start(7) + 4 * n_endpoints(7) == 128,
              "record layout fills 128 bits");

template <int POS, int LEN>
BC7_HD uint32_t get_bits(const B128& v)
{
    static_assert(LEN >= 0 && LEN <= 32 && POS >= 0 && POS + LEN <= 128, "field inside the block");
    if constexpr (LEN == 0) {
        return 0;
    } else {
        constexpr int w = POS >> 5, s = POS & 31;
        constexpr uint32_t mask = LEN == 32 ? 0xFFFFFFFFu : ((1u << LEN) - 1u);
        if constexpr (s + LEN <= 32) {
            return (v.d[w] >> s) & mask;
        } else {
            return ((v.d[w] >> s) | (v.d[w + 1] << (32 - s))) & mask;
        }
    }
}

// ORs the LEN-bit value x (no bits above LEN set) into v at POS
template <int POS, int LEN>
BC7_HD void put_bits(B128& v, uint32_t x)
{
    static_assert(LEN >= 0 && LEN <= 32 && POS >= 0 && POS + LEN <= 128, "field inside the block");
    if constexpr (LEN > 0) {
        constexpr int w = POS >> 5, s = POS & 31;
        v.d[w] |= x << s;
        if constexpr (s + LEN > 32)
            v.d[w + 1] |= x >> (32 - s);
    }
}

template <int SRC, int DST, int LEN>
BC7_HD void copy_bits(const B128& from, B128& to)
{
    if constexpr (LEN > 0) {
        constexpr int n = LEN < 32 ? LEN : 32;
        put_bits<DST, n>(to, get_bits<SRC, n>(from));
        copy_bits<SRC + n, DST + n, LEN - n>(from, to);
    }
}

// ---- endpoints ---------------------------------------------------------------------------------------------------
// The n = n_rgb / 3 fields of one colour channel lie side by side (n * w <= 30 bits in every mode), so a channel is ONE
// dword of n lanes of w bits, and version 2's colour decorrelation -- red and blue as differences to the green of the same
// endpoint, modulo 2^w -- is one lane-wise (SWAR) subtraction per channel.  H = the top bit of every lane.
constexpr int channel_fields(int m) { return kModeDesc[m].n_rgb / 3; }
constexpr int channel_bits(int m) { return channel_fields(m) * kModeDesc[m].w_rgb; }
constexpr int alpha_bits(int m) { return kModeDesc[m].n_a * kModeDesc[m].w_a; }
constexpr uint32_t lane_tops(int m)
{
    uint32_t h = 0;
    for (int i = 0; i < channel_fields(m); ++i)
        h |= 1u << (i * kModeDesc[m].w_rgb + kModeDesc[m].w_rgb - 1);
    return h;
}
static_assert(channel_bits(2) == 30 && channel_bits(3) == 28 && alpha_bits(7) == 20 && alpha_bits(5) == 16, "one dword per channel");

// lane-wise x - y and x + y modulo 2^w (no carry or borrow crosses a lane: the top bits are set / cleared first and
// put right afterwards)
template <int M>
BC7_HD uint32_t lanes_sub(uint32_t x, uint32_t y)
{
    constexpr uint32_t H = lane_tops(M);
    return ((x | H) - (y & ~H)) ^ ((x ^ ~y) & H);
}
template <int M>
BC7_HD uint32_t lanes_add(uint32_t x, uint32_t y)
{
    constexpr uint32_t H = lane_tops(M);
    return ((x & ~H) + (y & ~H)) ^ ((x ^ y) & H);
}

// field E of the block (block order: reds, greens, blues, alphas) sits in channel dword c[E / n] (alpha: c[3]) at lane
// E % n
template <int M, int E>
BC7_HD void split_endpoints(const uint32_t (&c)[4], B128& r)
{
    if constexpr (E < n_endpoints(M)) {
        constexpr bool rgb = E < kModeDesc[M].n_rgb;
        constexpr int w = endpoint_width(M, E);
        constexpr int ch = rgb ? E / channel_fields(M) : 3;
        constexpr int at = (rgb ? E % channel_fields(M) : E - kModeDesc[M].n_rgb) * w;
        if constexpr (w > 4)
            put_bits<low_pos(M, E), w - 4>(r, (c[ch] >> at) & ((1u << (w - 4)) - 1u));
        put_bits<highs_start(M) + 4 * E, 4>(r, (c[ch] >> (at + w - 4)) & 15u);
        split_endpoints<M, E + 1>(c, r);
    }
}

template <int M, int E>
BC7_HD void join_endpoints(const B128& r, uint32_t (&c)[4])
{
    if constexpr (E < n_endpoints(M)) {
        constexpr bool rgb = E < kModeDesc[M].n_rgb;
        constexpr int w = endpoint_width(M, E);
        constexpr int ch = rgb ? E / channel_fields(M) : 3;
        constexpr int at = (rgb ? E % channel_fields(M) : E - kModeDesc[M].n_rgb) * w;
        if constexpr (w > 4)
            c[ch] |= get_bits<low_pos(M, E), w - 4>(r) << at;
        c[ch] |= get_bits<highs_start(M) + 4 * E, 4>(r) << (at + w - 4);
        join_endpoints<M, E + 1>(r, c);
    }
}

// record, LSB first: marker and header | p-bits and index bits | low (w - 4) bits of every endpoint | high nibbles,
// the endpoints being R - G, G, B - G (lane-wise, modulo 2^w) and A
template <int M>
BC7_HD B128 record_of_block(const B128& b)
{
    B128 r = {{0, 0, 0, 0}};
    copy_bits<0, 0, endpoints_start(M)>(b, r);
    copy_bits<endpoints_end(M), endpoints_start(M), tail_len(M)>(b, r);
    constexpr int cb = channel_bits(M), e0 = endpoints_start(M);
    const uint32_t g = get_bits<e0 + cb, cb>(b);
    const uint32_t c[4] = {lanes_sub<M>(get_bits<e0, cb>(b), g), g, lanes_sub<M>(get_bits<e0 + 2 * cb, cb>(b), g),
                           get_bits<e0 + 3 * cb, alpha_bits(M)>(b)};
    split_endpoints<M, 0>(c, r);
    return r;
}

template <int M>
BC7_HD B128 block_of_record(const B128& r)
{
    B128 b = {{0, 0, 0, 0}};
    copy_bits<0, 0, endpoints_start(M)>(r, b);
    copy_bits<endpoints_start(M), endpoints_end(M), tail_len(M)>(r, b);
    uint32_t c[4] = {0, 0, 0, 0};
    join_endpoints<M, 0>(r, c);
    constexpr int cb = channel_bits(M), e0 = endpoints_start(M);
    put_bits<e0, cb>(b, lanes_add<M>(c[0], c[1]));
    put_bits<e0 + cb, cb>(b, c[1]);
    put_bits<e0 + 2 * cb, cb>(b, lanes_add<M>(c[2], c[1]));
    put_bits<e0 + 3 * cb, alpha_bits(M)>(b, c[3]);
    return b;
}

// class of a block: mode 0..7 = trailing zeros of byte 0; 8 = the reserved byte-0 == 0 encoding (moved unchanged)
BC7_HD int block_class(uint32_t byte0)
{
    const uint32_t v = (byte0 & 0xFFu) | 0x100u;
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_ctz(v);
#else
    int c = 0;
    while (!((v >> c) & 1u))
        ++c;
    return c;
#endif
}

BC7_HD B128 record_of_block_any(const B128& b, int cls)
{
    switch (cls) {
    case 0: return record_of_block<0>(b);
    case 1: return record_of_block<1>(b);
    case 2: return record_of_block<2>(b);
    case 3: return record_of_block<3>(b);
    case 4: return record_of_block<4>(b);
    case 5: return record_of_block<5>(b);
    case 6: return record_of_block<6>(b);
    case 7: return record_of_block<7>(b);
    default: return b;
    }
}

BC7_HD B128 block_of_record_any(const B128& r, int cls)
{
    switch (cls) {
    case 0: return block_of_record<0>(r);
    case 1: return block_of_record<1>(r);
    case 2: return block_of_record<2>(r);
    case 3: return block_of_record<3>(r);
    case 4: return block_of_record<4>(r);
    case 5: return block_of_record<5>(r);
    case 6: return block_of_record<6>(r);
    case 7: return block_of_record<7>(r);
    default: return r;
    }
}

// Byte 0 of the record from the block alone (the forward kernel needs it in block order, before the sort): marker and
// header are in place; only modes 0 and 6 have room left in the byte, for the first 3 / 1 of their p-bits.
BC7_HD uint32_t record_byte0(const B128& b, int cls)
{
    const uint32_t b0 = b.d[0] & 0xFFu;
    const uint32_t m0 = (b0 & 0x1Fu) | (get_bits<endpoints_end(0), 3>(b) << 5);
    const uint32_t m6 = (b0 & 0x7Fu) | (get_bits<endpoints_end(6), 1>(b) << 7);
    return cls == 0 ? m0 : cls == 6 ? m6 : b0;
}
static_assert(endpoints_start(0) == 5 && endpoints_start(6) == 7 && endpoints_start(1) == 8 && endpoints_start(4) == 8 &&
                  endpoints_start(5) == 8 && endpoints_start(2) > 8 && endpoints_start(3) > 8 && endpoints_start(7) > 8,
              "record_byte0: which modes' byte 0 differs from the block's");

}  // namespace bc7
}  // namespace dxtlt
