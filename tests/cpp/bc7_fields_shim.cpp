// Host build (g++) of the device header csrc/bc7_fields.h, so that the CPU test suite can compare the compile-time
// bit-field moves the kernels use with the oracle's one-field-at-a-time statement.
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "../../dxt-lossless-transform_amd/csrc/bc7_fields.h"

using dxtlt::bc7::B128;

extern "C" void shim_bc7_records(const uint8_t* in, uint8_t* out, size_t num_blocks, int inverse)
{
    for (size_t i = 0; i < num_blocks; ++i) {
        B128 b;
        std::memcpy(b.d, in + 16 * i, 16);
        const int cls = dxtlt::bc7::block_class(b.d[0]);
        const B128 r = inverse ? dxtlt::bc7::block_of_record_any(b, cls) : dxtlt::bc7::record_of_block_any(b, cls);
        std::memcpy(out + 16 * i, r.d, 16);
    }
}

// byte 0 of the record computed from the block alone (forward kernel, block order)
extern "C" void shim_bc7_record_byte0(const uint8_t* in, uint8_t* out, size_t num_blocks)
{
    for (size_t i = 0; i < num_blocks; ++i) {
        B128 b;
        std::memcpy(b.d, in + 16 * i, 16);
        out[i] = (uint8_t)dxtlt::bc7::record_byte0(b, dxtlt::bc7::block_class(b.d[0]));
    }
}
