// Host build (g++) of the device header csrc/bc1_normalize.h, so that the CPU test suite can compare the block
// classification the kernels use with the oracle's pixel-by-pixel statement on millions of blocks.
#include <cstddef>
#include <cstdint>
#include <cstring>
#define __host__
#define __device__
#include "../../dxt-lossless-transform_amd/csrc/bc1_normalize.h"
#include "../../dxt-lossless-transform_amd/csrc/bc23_normalize.h"
#include "../../dxt-lossless-transform_amd/csrc/bcn_decode.h"

extern "C" void shim_normalize_blocks(const uint8_t* in, uint8_t* out, size_t num_blocks, int mode)
{
    for (size_t b = 0; b < num_blocks; ++b) {
        uint32_t c, x;
        std::memcpy(&c, in + 8 * b, 4);
        std::memcpy(&x, in + 8 * b + 4, 4);
        dxtlt::normalize_bc1_block_rt(mode, c, x);
        std::memcpy(out + 8 * b, &c, 4);
        std::memcpy(out + 8 * b + 4, &x, 4);
    }
}

extern "C" void shim_normalize_bc23_blocks(int fmt, const uint8_t* in, uint8_t* out, size_t num_blocks, int alpha_mode,
                                           int color_mode)
{
    for (size_t b = 0; b < num_blocks; ++b) {
        uint32_t q[4];
        std::memcpy(q, in + 16 * b, 16);
        if (fmt == 2)
            dxtlt::normalize_block_bc23<2>(alpha_mode, color_mode, q);
        else
            dxtlt::normalize_block_bc23<3>(alpha_mode, color_mode, q);
        std::memcpy(out + 16 * b, q, 16);
    }
}

// csrc/bcn_decode.h: fmt = 1, 2, 3; 64 bytes of pixels per block
extern "C" void shim_decode_blocks(int fmt, const uint8_t* in, uint8_t* out, size_t num_blocks)
{
    const size_t bs = fmt == 1 ? 8 : 16;
    for (size_t b = 0; b < num_blocks; ++b) {
        uint32_t q[4] = {0, 0, 0, 0}, px[16];
        std::memcpy(q, in + bs * b, bs);
        if (fmt == 1)
            dxtlt::decode_block_px<1>(q, px);
        else if (fmt == 2)
            dxtlt::decode_block_px<2>(q, px);
        else
            dxtlt::decode_block_px<3>(q, px);
        std::memcpy(out + 64 * b, px, 64);
    }
}

extern "C" uint32_t shim_small_division(int d, uint32_t x)
{
    return d == 3 ? dxtlt::div3_small(x) : d == 5 ? dxtlt::div5_small(x) : dxtlt::div7_small(x);
}
