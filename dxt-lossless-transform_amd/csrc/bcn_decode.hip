// bcn_decode.hip -- gfx950 kernels for the reference's block decoders as array operations (decode_bcN_block:
// dxt-lossless-transform-bc1/src/util/bc1_decode.rs:42, -bc2/src/util/bc2_decode.rs:44, -bc3/src/util/bc3_decode.rs:43;
// output = one Decoded4x4Block per block, dxt-lossless-transform-common/src/decoded_4x4_block.rs:56), and the
// "decoded pixels are unchanged" check the reference's normalisation tests make, as a device-side count.
//
// decode: one block per lane in registers (bcn_decode.h), then the wave's 4 KiB of pixels go through LDS so that
// every store instruction writes 1 KiB of consecutive bytes (lane l -> 16 bytes at 16 * l): a lane storing its own
// 64 bytes would leave every 128-byte line to four instructions.  LDS rows are padded by 64 bytes, which puts the
// four rows a quad of lanes reads into different banks.  Write-bound: 8 (16) bytes in, 64 out per block.
//
// Measured alternatives (tools/decode_lab.hip, profiles/r01_z/decode_lab.txt; BC1, 8 GiB of pixels, fraction of
// 8 TB/s): stores `sc1 nt` 0.80-0.83 (adopted) / nontemporal builtin 0.75 / plain 0.76; the store side alone with no
// decoding 0.80, i.e. the ~175 VALU instructions per block hide completely; two blocks per lane 0.76, four 0.48; an
// XCD-contiguous tile order 0.74; a fixed grid walking the tiles 0.61-0.68; each lane storing its own 64 bytes 0.17.
// A pure 8 GiB fill runs at 0.86 (torch) / 0.79 (hipMemsetAsync).
#include <hip/hip_runtime.h>

#include "bcn_decode.h"
#include "bcn_launch.h"
#include "launch_grid.h"
#include "streaming_store.h"

namespace dxtlt {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kDecThreads = 256;
constexpr int kRowStride = 64 + 4;                 // u32x4 units: 64 lanes + 64 bytes of padding
constexpr int kWaveStage = 4 * kRowStride;         // four pixel rows per wave
constexpr uint64_t kDifferenceGrid = 256 * 16;    // workgroups of the difference count: 16 per CU

__device__ __forceinline__ void store_streaming(void* p, u32x4 v) { store_streaming16(p, v); }   // streaming_store.h

template <int FMT>
__device__ __forceinline__ void load_block(const uint8_t* in, uint64_t b, uint32_t q[4])
{
    if (FMT == 1) {
        const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(in) + b);
        q[0] = v.x, q[1] = v.y, q[2] = 0, q[3] = 0;
    } else {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in) + b);
        q[0] = v.x, q[1] = v.y, q[2] = v.z, q[3] = v.w;
    }
}

template <int FMT>
__device__ __forceinline__ void load_block_bytes(const uint8_t* in, uint64_t b, uint32_t q[4])
{
    constexpr int BS = FMT == 1 ? 8 : 16;
    q[0] = q[1] = q[2] = q[3] = 0;
    for (int i = 0; i < BS; ++i)
        q[i >> 2] |= (uint32_t)in[BS * b + i] << (8 * (i & 3));
}

template <int FMT>
__global__ void __launch_bounds__(kDecThreads)
decode_blocks_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t num_blocks)
{
    __shared__ u32x4 stage[(kDecThreads / 64) * kWaveStage];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t wave_first = workgroup_index() * kDecThreads + 64 * wave;
    const uint64_t b = wave_first + lane;
    uint32_t q[4] = {0, 0, 0, 0}, px[16];
    if (b < num_blocks)
        load_block<FMT>(in, b, q);
    decode_block_px<FMT>(q, px);
    u32x4* mine = stage + wave * kWaveStage;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        mine[r * kRowStride + lane] = u32x4{px[4 * r], px[4 * r + 1], px[4 * r + 2], px[4 * r + 3]};
    __syncthreads();
    u32x4* dst = reinterpret_cast<u32x4*>(out) + 4 * wave_first;   // 16-byte chunk j of the wave = block j / 4, row j % 4
    const uint64_t chunks = num_blocks > wave_first ? 4 * (num_blocks - wave_first) : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = 64 * k + lane;
        if ((uint64_t)j < chunks)
            store_streaming(dst + j, mine[(j & 3) * kRowStride + (j >> 2)]);
    }
}

// any alignment: byte loads, byte stores; one block per lane
template <int FMT>
__global__ void __launch_bounds__(kDecThreads)
decode_blocks_unaligned_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t num_blocks)
{
    const uint64_t b = workgroup_index() * kDecThreads + threadIdx.x;
    if (b >= num_blocks)
        return;
    uint32_t q[4], px[16];
    load_block_bytes<FMT>(in, b, q);
    decode_block_px<FMT>(q, px);
    uint8_t* o = out + 64 * b;
    for (int i = 0; i < 16; ++i)
        for (int c = 0; c < 4; ++c)
            o[4 * i + c] = (uint8_t)(px[i] >> (8 * c));
}

// number of blocks whose decoded pixels differ; byte-identical blocks need no decoding.  A fixed grid walks the
// blocks and every wave adds its total once: one atomic per 64 blocks on one address costs ~12 ns each (measured:
// 25 ms for 2^27 blocks that all differ), which is slower than the whole read.
template <int FMT, bool ALIGNED, int U>
__global__ void __launch_bounds__(kDecThreads)
pixel_difference_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint64_t num_blocks,
                        unsigned long long* __restrict__ count)
{
    // U blocks per lane and step, the 2 U loads issued before the first compare
    unsigned long long mine = 0;   // wave-uniform
    const uint64_t stride = (uint64_t)gridDim.x * (kDecThreads * U);
    for (uint64_t base = (uint64_t)blockIdx.x * (kDecThreads * U); base < num_blocks; base += stride) {
        // branch-free loads (a load under a per-element condition gets its own branch and `s_waitcnt vmcnt(0)`): lanes
        // past the end read the last block of both arrays, which compares equal to itself only if the arrays agree
        // there -- so their result is masked below
        uint32_t qa[U][4], qb[U][4];
        bool inside[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = base + (uint64_t)(u * kDecThreads) + threadIdx.x;
            inside[u] = i < num_blocks;
            const uint64_t ic = inside[u] ? i : num_blocks - 1;
            if (ALIGNED) {
                load_block<FMT>(a, ic, qa[u]);
                load_block<FMT>(b, ic, qb[u]);
            } else {
                load_block_bytes<FMT>(a, ic, qa[u]);
                load_block_bytes<FMT>(b, ic, qb[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bool differs = false;
            if (inside[u] && (qa[u][0] != qb[u][0] || qa[u][1] != qb[u][1] || qa[u][2] != qb[u][2] || qa[u][3] != qb[u][3])) {
                uint32_t pa[16], pb[16];
                decode_block_px<FMT>(qa[u], pa);
                decode_block_px<FMT>(qb[u], pb);
                uint32_t x = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    x |= pa[k] ^ pb[k];
                differs = x != 0;
            }
            mine += (unsigned long long)__popcll(__ballot(differs));
        }
    }
    if ((threadIdx.x & 63) == 0 && mine != 0)
        atomicAdd(count, mine);
}

inline bool aligned_to(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

template <int FMT>
hipError_t decode_fmt(const void* in, void* out, uint64_t n, hipStream_t stream)
{
    dim3 grid;
    if (hipError_t e = grid_rows(n, kDecThreads, grid); e != hipSuccess)
        return e;
    const bool fast = aligned_to(in, FMT == 1 ? 8 : 16) && aligned_to(out, 16);
    // Workgroups per CU, capped with dynamic LDS nobody touches.  A decoder workgroup writes 16 KiB for 2-4 KiB read, and
    // on this memory system more bytes in flight per CU than it can take LOWER the rate (tools/shape_lab.hip; DESIGN.md
    // section 9): with the eight workgroups per CU the wave slots allow BC2 / BC3 decode at 0.745 of the HBM peak, with six
    // at 0.86-0.87 -- the speed of a plain fill -- (five and four: 0.845; seven: 0.82; three: 0.62).  BC1, which reads half
    // as much per block, is best left at eight (0.80; seven 0.80, six 0.76).  DXTLT_EXPERIMENT_WGS_PER_CU overrides (experiments).
    const unsigned pad_lds = lds_pad_for_wgs_per_cu(wgs_per_cu_or(FMT == 1 ? 8 : 6), kDecThreads, (unsigned)((kDecThreads / 64) * kWaveStage * 16));
    if (fast)
        hipLaunchKernelGGL(decode_blocks_kernel<FMT>, grid, dim3(kDecThreads), pad_lds, stream,
                           static_cast<const uint8_t*>(in), static_cast<uint8_t*>(out), n);
    else
        hipLaunchKernelGGL(decode_blocks_unaligned_kernel<FMT>, grid, dim3(kDecThreads), 0, stream,
                           static_cast<const uint8_t*>(in), static_cast<uint8_t*>(out), n);
    return hipGetLastError();
}

template <int FMT>
hipError_t difference_fmt(const void* a, const void* b, uint64_t n, unsigned long long* count, hipStream_t stream)
{
    // two blocks per lane and step: four loads in flight per lane.  Four blocks (eight loads) were slower on the sparse
    // case -- BC2 0.673 against 0.697 of peak, BC3 0.626 against 0.666 -- and one block no better (0.667 / 0.652): more bytes
    // in flight per CU than the memory system likes lower the rate (launch_grid.h)
    constexpr int kBlocksPerLane = 2;
    uint64_t wgs = (n + kDecThreads * kBlocksPerLane - 1) / (kDecThreads * kBlocksPerLane);
    if (wgs > kDifferenceGrid)
        wgs = kDifferenceGrid;
    const uintptr_t al = FMT == 1 ? 8 : 16;
    const auto* pa = static_cast<const uint8_t*>(a);
    const auto* pb = static_cast<const uint8_t*>(b);
    if (aligned_to(a, al) && aligned_to(b, al))
        hipLaunchKernelGGL((pixel_difference_kernel<FMT, true, kBlocksPerLane>), dim3((unsigned)wgs), dim3(kDecThreads), 0, stream, pa, pb, n, count);
    else
        hipLaunchKernelGGL((pixel_difference_kernel<FMT, false, kBlocksPerLane>), dim3((unsigned)wgs), dim3(kDecThreads), 0, stream, pa, pb, n, count);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_decode_blocks(int fmt, const void* in, void* out, uint64_t num_blocks, hipStream_t stream)
{
    if (num_blocks == 0)
        return hipSuccess;
    switch (fmt) {
    case 1: return decode_fmt<1>(in, out, num_blocks, stream);
    case 2: return decode_fmt<2>(in, out, num_blocks, stream);
    case 3: return decode_fmt<3>(in, out, num_blocks, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_count_pixel_differences(int fmt, const void* a, const void* b, uint64_t num_blocks, uint64_t* d_count,
                                          hipStream_t stream)
{
    if (hipError_t e = hipMemsetAsync(d_count, 0, sizeof(uint64_t), stream); e != hipSuccess)
        return e;
    if (num_blocks == 0)
        return hipSuccess;
    auto* c = reinterpret_cast<unsigned long long*>(d_count);
    switch (fmt) {
    case 1: return difference_fmt<1>(a, b, num_blocks, c, stream);
    case 2: return difference_fmt<2>(a, b, num_blocks, c, stream);
    case 3: return difference_fmt<3>(a, b, num_blocks, c, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace dxtlt
