// vmm_helper.cpp -- probe tooling only (tools/placement_vmm_probe.py; never linked into the product library): device buffers
// whose physical backing is chosen by the caller through HIP's virtual-memory API instead of by hipMalloc.
//   hipMemAddressReserve -> one virtual range; hipMemCreate -> physical handles of `chunk` bytes each (chunk == 0: one handle
//   for the whole buffer); hipMemMap -> the handles behind consecutive pieces of the range, in creation order or in a seeded
//   shuffled order; hipMemSetAccess for the device.
// Built by:  hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/vmm_helper.cpp -o tools/libvmm_helper.so
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Mapping {
    size_t va_bytes;
    size_t piece;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};

std::mutex g_mu;
std::map<void*, Mapping> g_maps;
std::string g_err;

int fail(const char* what, hipError_t e)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    g_err = buf;
    return (int)e ? (int)e : -1;
}

uint64_t splitmix(uint64_t& s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

hipMemAllocationProp device_prop(int dev)
{
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    return prop;
}

}  // namespace

extern "C" {

const char* vmm_last_error() { return g_err.c_str(); }

int vmm_granularity(int dev, size_t* min_g, size_t* rec_g)
{
    const hipMemAllocationProp prop = device_prop(dev);
    if (hipError_t e = hipMemGetAllocationGranularity(min_g, &prop, hipMemAllocationGranularityMinimum); e != hipSuccess)
        return fail("hipMemGetAllocationGranularity(min)", e);
    if (hipError_t e = hipMemGetAllocationGranularity(rec_g, &prop, hipMemAllocationGranularityRecommended); e != hipSuccess)
        return fail("hipMemGetAllocationGranularity(recommended)", e);
    return 0;
}

// `bytes` of device memory at a fresh virtual address.  chunk == 0: one physical handle; else ceil(bytes / chunk) handles of
// `chunk` bytes.  shuffle_seed != 0: the handles are mapped in a seeded random order (physical placement decoupled from the
// order the driver handed the pages out in).  `va_align`: alignment of the virtual range, a power of two (0 = 2 MiB).
int vmm_alloc(int dev, size_t bytes, size_t chunk, uint64_t shuffle_seed, size_t va_align, void** out)
{
    std::lock_guard<std::mutex> lock(g_mu);
    *out = nullptr;
    if (bytes == 0)
        return 0;
    if (hipError_t e = hipSetDevice(dev); e != hipSuccess)
        return fail("hipSetDevice", e);
    const hipMemAllocationProp prop = device_prop(dev);
    size_t min_g = 0;
    if (hipError_t e = hipMemGetAllocationGranularity(&min_g, &prop, hipMemAllocationGranularityMinimum); e != hipSuccess)
        return fail("hipMemGetAllocationGranularity", e);
    const size_t piece = chunk ? chunk : (bytes + min_g - 1) / min_g * min_g;
    if (piece % min_g != 0) {
        g_err = "chunk is not a multiple of the minimum granularity";
        return -1;
    }
    const size_t n = (bytes + piece - 1) / piece;
    const size_t va_bytes = n * piece;
    void* va = nullptr;
    if (hipError_t e = hipMemAddressReserve(&va, va_bytes, va_align ? va_align : size_t(2) << 20, nullptr, 0);   // (a power of two)
        e != hipSuccess)
        return fail("hipMemAddressReserve", e);
    Mapping m{va_bytes, piece, {}};
    m.handles.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        hipMemGenericAllocationHandle_t h;
        if (hipError_t e = hipMemCreate(&h, piece, &prop, 0); e != hipSuccess) {
            for (auto hh : m.handles)
                (void)hipMemRelease(hh);
            (void)hipMemAddressFree(va, va_bytes);
            return fail("hipMemCreate", e);
        }
        m.handles.push_back(h);
    }
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; ++i)
        order[i] = (uint32_t)i;
    if (shuffle_seed) {
        uint64_t s = shuffle_seed;
        for (size_t i = n; i > 1; --i)
            std::swap(order[i - 1], order[splitmix(s) % i]);
    }
    for (size_t i = 0; i < n; ++i) {
        if (hipError_t e = hipMemMap(static_cast<uint8_t*>(va) + i * piece, piece, 0, m.handles[order[i]], 0); e != hipSuccess) {
            if (i)
                (void)hipMemUnmap(va, i * piece);
            for (auto hh : m.handles)
                (void)hipMemRelease(hh);
            (void)hipMemAddressFree(va, va_bytes);
            return fail("hipMemMap", e);
        }
    }
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipError_t e = hipMemSetAccess(va, va_bytes, &acc, 1); e != hipSuccess) {
        (void)hipMemUnmap(va, va_bytes);
        for (auto hh : m.handles)
            (void)hipMemRelease(hh);
        (void)hipMemAddressFree(va, va_bytes);
        return fail("hipMemSetAccess", e);
    }
    g_maps.emplace(va, std::move(m));
    *out = va;
    return 0;
}

// Unmaps and releases the physical handles.  The VIRTUAL range stays reserved for the life of the process, on purpose: on this
// stack (ROCm 7.2.0, gfx950) a range that was unmapped, freed and handed out again by hipMemAddressReserve for a new mapping
// showed STALE TRANSLATIONS -- the copy engine and kernels (and kernels on different XCDs) disagreed about the bytes behind one
// address, wrong data without any fault (profiles/r05_placement_vmm.txt, "diagnose").  A range that is never mapped twice
// cannot meet a stale entry.
int vmm_free(void* p)
{
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_maps.find(p);
    if (it == g_maps.end()) {
        g_err = "vmm_free: unknown pointer";
        return -1;
    }
    (void)hipDeviceSynchronize();
    hipError_t first = hipSuccess;
    const Mapping& m = it->second;
    for (size_t i = 0; i < m.handles.size(); ++i) {
        const hipError_t e = hipMemUnmap(static_cast<uint8_t*>(p) + i * m.piece, m.piece);
        if (e != hipSuccess && first == hipSuccess)
            first = e;
    }
    for (auto h : m.handles)
        (void)hipMemRelease(h);
    (void)hipDeviceSynchronize();
    g_maps.erase(it);
    return first == hipSuccess ? 0 : fail("hipMemUnmap", first);
}

int vmm_copy(void* dst, const void* src, size_t n)
{
    hipError_t e = hipMemcpy(dst, src, n, hipMemcpyDeviceToDevice);
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    return e == hipSuccess ? 0 : fail("hipMemcpy", e);
}

int vmm_zero(void* dst, size_t n)
{
    hipError_t e = hipMemset(dst, 0, n);
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    return e == hipSuccess ? 0 : fail("hipMemset", e);
}

// Copies and comparisons done by KERNELS (16 bytes per lane, grid-stride): the runtime's own hipMemcpy / hipMemset look the
// pointer up in their allocation map and may take another path for a range that several physical handles back.
__global__ void __launch_bounds__(256) copy16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t vecs)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < vecs; i += (size_t)gridDim.x * 256)
        dst[i] = src[i];
}
__global__ void __launch_bounds__(256) fill16_kernel(uint4* __restrict__ dst, size_t vecs, uint32_t value)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < vecs; i += (size_t)gridDim.x * 256)
        dst[i] = make_uint4(value, value, value, value);
}
__global__ void __launch_bounds__(256) differ16_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, size_t vecs,
                                                        unsigned long long* __restrict__ count, unsigned long long* __restrict__ first)
{
    unsigned long long mine = 0, at = ~0ull;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < vecs; i += (size_t)gridDim.x * 256) {
        const uint4 x = a[i], y = b[i];
        if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) {
            ++mine;
            at = at < i ? at : i;
        }
    }
    if (mine) {
        atomicAdd(count, mine);
        atomicMin(first, at);
    }
}

int vmm_copy_by_kernel(void* dst, const void* src, size_t n)
{
    if (n % 16 != 0) {
        g_err = "vmm_copy_by_kernel: length must be a multiple of 16";
        return -1;
    }
    hipLaunchKernelGGL(copy16_kernel, dim3(256 * 32), dim3(256), 0, nullptr, static_cast<const uint4*>(src), static_cast<uint4*>(dst), n / 16);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    return e == hipSuccess ? 0 : fail("copy16_kernel", e);
}

int vmm_fill_by_kernel(void* dst, size_t n, uint32_t value)
{
    if (n % 16 != 0) {
        g_err = "vmm_fill_by_kernel: length must be a multiple of 16";
        return -1;
    }
    hipLaunchKernelGGL(fill16_kernel, dim3(256 * 32), dim3(256), 0, nullptr, static_cast<uint4*>(dst), n / 16, value);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    return e == hipSuccess ? 0 : fail("fill16_kernel", e);
}

// number of 16-byte vectors in which a and b differ (-1 on error); *first_vec = index of the first one
long long vmm_differ_by_kernel(const void* a, const void* b, size_t n, unsigned long long* first_vec)
{
    unsigned long long* d = nullptr;
    if (hipError_t e = hipMalloc(&d, 16); e != hipSuccess)
        return fail("hipMalloc", e), -1;
    const unsigned long long init[2] = {0, ~0ull};
    (void)hipMemcpy(d, init, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(differ16_kernel, dim3(256 * 32), dim3(256), 0, nullptr, static_cast<const uint4*>(a), static_cast<const uint4*>(b), n / 16, d, d + 1);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    unsigned long long out[2] = {0, 0};
    if (e == hipSuccess)
        e = hipMemcpy(out, d, 16, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess)
        return fail("differ16_kernel", e), -1;
    if (first_vec)
        *first_vec = out[1];
    return (long long)out[0];
}

// plain hipMalloc / hipFree, so that the "allocator's own" arm runs through the same raw-pointer plumbing
int plain_alloc(int dev, size_t bytes, void** out)
{
    if (hipError_t e = hipSetDevice(dev); e != hipSuccess)
        return fail("hipSetDevice", e);
    hipError_t e = hipMalloc(out, bytes);
    return e == hipSuccess ? 0 : fail("hipMalloc", e);
}

int plain_free(void* p)
{
    (void)hipDeviceSynchronize();
    hipError_t e = hipFree(p);
    return e == hipSuccess ? 0 : fail("hipFree", e);
}

}  // extern "C"
