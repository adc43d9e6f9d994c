// ycocg_swar.h -- YCoCg-R on two RGB565 colours packed in one 32-bit register.
//
// Behaviour follows Color565::{decorrelate,recorrelate}_ycocg_r_var{1,2,3}
// (/root/reference/src/core/dxt-lossless-transform-common/src/color_565/decorrelate.rs:101-344):
//   forward   co=(r-b)&31  t=(b+(co>>1))&31  cg=(g-t)&31  y=(t+(cg>>1))&31
//   inverse   t=(y-(cg>>1))&31  g=(cg+t)&31  b=(t-(co>>1))&31  r=(b+co)&31
//   var1 packs y<<11|co<<6|g_low<<5|cg, var2 g_low<<15|y<<10|co<<5|cg, var3 y<<11|co<<6|cg<<1|g_low.
//
// GPU shape: a BCn block keeps (c0, c1) in one dword, so both colours are processed at once in
// 16-bit lanes of a VGPR.  Subtractions add a +32 bias per lane so no borrow crosses from the
// high colour into the low one; every intermediate is masked back to 5 bits per lane.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DXTLT_HD __host__ __device__ __forceinline__
#else
#define DXTLT_HD inline
#endif

namespace dxtlt {

enum : int { kNone = 0, kVar1 = 1, kVar2 = 2, kVar3 = 3 };  // core YCoCgVariant numbering (decorrelate.rs:72-84)

constexpr uint32_t kM5 = 0x001F001Fu;    // 5-bit field in each 16-bit lane
constexpr uint32_t kM4 = 0x000F000Fu;    // (x >> 1) of a 5-bit field
constexpr uint32_t kBias = 0x00200020u;  // +32 per lane (also the g_low bit position, bit 5)

template <int VARIANT>
DXTLT_HD uint32_t decorrelate2(uint32_t v)
{
    if (VARIANT == kNone)
        return v;
    const uint32_t r = (v >> 11) & kM5;
    const uint32_t g = (v >> 6) & kM5;
    const uint32_t gl = v & kBias;  // g_low kept at bit 5 of each lane
    const uint32_t b = v & kM5;
    const uint32_t co = (r + kBias - b) & kM5;
    const uint32_t t = (b + ((co >> 1) & kM4)) & kM5;
    const uint32_t cg = (g + kBias - t) & kM5;
    const uint32_t y = (t + ((cg >> 1) & kM4)) & kM5;
    if (VARIANT == kVar1)
        return (y << 11) | (co << 6) | gl | cg;
    if (VARIANT == kVar2)
        return (gl << 10) | (y << 10) | (co << 5) | cg;
    return (y << 11) | (co << 6) | (cg << 1) | (gl >> 5);
}

template <int VARIANT>
DXTLT_HD uint32_t recorrelate2(uint32_t v)
{
    if (VARIANT == kNone)
        return v;
    uint32_t y, co, cg, gl;
    if (VARIANT == kVar1) {
        y = (v >> 11) & kM5;
        co = (v >> 6) & kM5;
        gl = v & kBias;
        cg = v & kM5;
    } else if (VARIANT == kVar2) {
        gl = (v >> 10) & kBias;
        y = (v >> 10) & kM5;
        co = (v >> 5) & kM5;
        cg = v & kM5;
    } else {
        y = (v >> 11) & kM5;
        co = (v >> 6) & kM5;
        cg = (v >> 1) & kM5;
        gl = (v << 5) & kBias;
    }
    const uint32_t t = (y + kBias - ((cg >> 1) & kM4)) & kM5;
    const uint32_t g = (cg + t) & kM5;
    const uint32_t b = (t + kBias - ((co >> 1) & kM4)) & kM5;
    const uint32_t r = (b + co) & kM5;
    return (r << 11) | (g << 6) | gl | b;
}

}  // namespace dxtlt
