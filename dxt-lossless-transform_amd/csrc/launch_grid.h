// launch_grid.h -- grids for element-wise kernels over more than 2^32 lanes.
//
// HIP refuses a launch whose x dimension holds 2^32 or more threads (grid.x * workgroup size: the AQL packet's
// grid_size_x is 32 bits), and one MI355X holds buffers that reach it at 16 bytes per lane (64 GiB).  The element-wise
// kernels therefore take their workgroup number from a two-dimensional grid: up to 2^20 workgroups along x, rows along
// y.  The last row may be partly empty; every kernel already bounds-checks its lane index.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

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

#if defined(__HIPCC__)
__device__ __forceinline__ uint64_t workgroup_index() { return (uint64_t)blockIdx.y * gridDim.x + blockIdx.x; }
#endif

}  // namespace dxtlt
