// launch_grid.h -- grids for element-wise kernels over more than 2^32 lanes.
//
// HIP refuses a launch whose x dimension holds 2^32 or more threads (grid.x * workgroup size: the AQL packet's
// grid_size_x is 32 bits), and one MI355X holds buffers that reach it at 16 bytes per lane (64 GiB).  The element-wise
// kernels therefore take their workgroup number from a two-dimensional grid: up to 2^20 workgroups along x, rows along
// y.  The last row may be partly empty; every kernel already bounds-checks its lane index.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

namespace dxtlt {

constexpr uint64_t kGridRow = 1ull << 20;   // workgroups along x (x 1024 threads still fits 32 bits)

// grid for `lanes` lanes in workgroups of `threads`
inline hipError_t grid_rows(uint64_t lanes, unsigned threads, dim3& grid)
{
    const uint64_t wgs = (lanes + threads - 1) / threads;
    if (wgs <= kGridRow) {
        grid = dim3((unsigned)wgs, 1, 1);
        return hipSuccess;
    }
    const uint64_t rows = (wgs + kGridRow - 1) / kGridRow;
    if (rows > 65535)
        return hipErrorInvalidValue;   // 2^36 workgroups
    grid = dim3((unsigned)kGridRow, (unsigned)rows, 1);
    return hipSuccess;
}

// Dynamic LDS (touched by nobody) that caps a kernel of `threads` lanes and `static_lds` bytes of its own at
// `wgs_per_cu` workgroups per CU (160 KiB of LDS per CU on gfx950); 0 when the cap is not below what the wave slots allow.
// Why a kernel would want fewer resident workgroups: with more than ~32 KiB of loads or stores in flight per CU the memory
// system gets slower, not faster (tools/shape_lab.hip), so a kernel that moves more than 16 bytes per
// lane in one go can gain from it.  Stays below 64 KiB per workgroup (no launch attribute needed) for caps of 3 and more.
inline unsigned lds_pad_for_wgs_per_cu(int wgs_per_cu, unsigned threads, unsigned static_lds)
{
    constexpr unsigned kLdsPerCu = 160u << 10;
    const int by_waves = (int)(2048u / threads);   // 32 wave slots per CU
    if (wgs_per_cu < 3 || wgs_per_cu >= by_waves)
        return 0;
    const unsigned want = kLdsPerCu / (unsigned)wgs_per_cu - 256u;
    return want > static_lds ? want - static_lds : 0u;
}

// the cap a kernel runs with: its measured best; in the experiments side build (-DDXTLT_EXPERIMENTS) DXTLT_EXPERIMENT_WGS_PER_CU=<n>
// (read once per process) overrides it for the kernels that take one (n = 8: no cap)
inline int wgs_per_cu_or(int measured_best)
{
#ifdef DXTLT_EXPERIMENTS
    static const int v = [] { const char* e = std::getenv("DXTLT_EXPERIMENT_WGS_PER_CU"); return e ? std::atoi(e) : 0; }();
    return v > 0 ? v : measured_best;
#else
    return measured_best;
#endif
}

#if defined(__HIPCC__)
__device__ __forceinline__ uint64_t workgroup_index() { return (uint64_t)blockIdx.y * gridDim.x + blockIdx.x; }
#endif

}  // namespace dxtlt
