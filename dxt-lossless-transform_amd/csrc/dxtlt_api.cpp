// dxtlt_api.cpp -- C ABI of libdxtlt_gfx950.so (include/dxtlt_gfx950.h): argument validation, the
// host-pointer staging path, the device-pointer path and the single-process multi-GPU shard path.
// All arithmetic lives in bcn_kernels.hip; nothing here touches block bytes on the CPU.
#include "../../include/dxtlt_gfx950.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <thread>
#include <vector>

#include "bc7_launch.h"
#include "bcn_launch.h"
#include "host_common.h"

namespace {

using dxtlt::Format;
using dxtlt::Range;
using dxtlt::Settings;

thread_local std::string g_last_error;
thread_local std::vector<DxtltShardStat> g_shard_stats;   // of this thread's last dxtlt_transform_sharded call
std::atomic<int> g_tile_threads{0};
std::atomic<int> g_force_generic{0};
std::atomic<int> g_xcd_remap{-1};

}  // namespace

int32_t dxtlt_host::fail(int32_t code, const char* what, hipError_t e)
{
    char buf[256];
    if (e != hipSuccess)
        std::snprintf(buf, sizeof buf, "dxtlt: %s: %s (%d)", what, hipGetErrorString(e), (int)e);
    else
        std::snprintf(buf, sizeof buf, "dxtlt: %s", what);
    g_last_error = buf;
    return code;
}

namespace {
using dxtlt_host::fail;

#define HIP_TRY(expr, what)                                \
    do {                                                   \
        hipError_t e_ = (expr);                            \
        if (e_ != hipSuccess)                              \
            return fail(DXTLT_E_DEVICE, what, e_);         \
    } while (0)

dxtlt::LaunchTuning current_tuning()
{
    dxtlt::LaunchTuning t;
    t.tile_threads = g_tile_threads.load(std::memory_order_relaxed);
    t.force_generic = g_force_generic.load(std::memory_order_relaxed);
    t.xcd_remap = g_xcd_remap.load(std::memory_order_relaxed);
    return t;
}

int32_t check_common(int32_t format, size_t len, uint8_t mode, const void* a, const void* b)
{
    if (format < 1 || format > 3)
        return fail(DXTLT_E_INVALID_ARGUMENT, "format must be 1 (BC1), 2 (BC2) or 3 (BC3)");
    if (len % (size_t)dxtlt::block_bytes((Format)format) != 0)
        return fail(DXTLT_E_INVALID_LENGTH, "len is not a multiple of the block size");
    if (mode > 3)
        return fail(DXTLT_E_INVALID_ARGUMENT, "decorrelation_mode must be 0..3");
    if (len > 0 && (a == nullptr || b == nullptr))
        return fail(DXTLT_E_INVALID_ARGUMENT, "NULL buffer with len > 0");
    return DXTLT_OK;
}

int32_t device_range(int32_t format, bool inverse, const void* d_src, void* d_dst, uint64_t total, uint64_t first,
                     uint64_t num, uint8_t mode, bool sa, bool sc, hipStream_t stream, uint8_t normalize = 0)
{
    if (format < 1 || format > 3)
        return fail(DXTLT_E_INVALID_ARGUMENT, "format must be 1 (BC1), 2 (BC2) or 3 (BC3)");
    if (mode > 3)
        return fail(DXTLT_E_INVALID_ARGUMENT, "decorrelation_mode must be 0..3");
    if (first > total || num > total - first)
        return fail(DXTLT_E_INVALID_ARGUMENT, "block range exceeds total_blocks");
    if (num == 0)
        return DXTLT_OK;
    if (d_src == nullptr || d_dst == nullptr)
        return fail(DXTLT_E_INVALID_ARGUMENT, "NULL device buffer with a non-empty range");
    // 3 = transparent blocks only: internal, used by the auto transform (bc1_normalize.h); the public entry points
    // check color_mode <= 2 before they get here
    if (normalize != 0 && (format != 1 || inverse || normalize > 3))
        return fail(DXTLT_E_INVALID_ARGUMENT, "block normalisation: BC1 forward only, color_mode 0..2");
    Settings s{(int)mode, sa, sc, (int)normalize};
    Range r{total, first, num};
    dxtlt::LaunchTuning t = current_tuning();
    HIP_TRY(dxtlt::launch_transform((Format)format, inverse, s, d_src, d_dst, r, stream, &t), "kernel launch");
    return DXTLT_OK;
}

// ---------------------------------------------------------------------------------------------------
// Host-pointer path.  Per thread and device: one stream and a grow-only pair of device buffers, so that
// repeated calls (the reference's callers transform file after file) pay allocation once.
// ---------------------------------------------------------------------------------------------------
struct HostCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    void* d_in = nullptr;
    void* d_out = nullptr;
    size_t cap = 0;

    ~HostCtx() { release(); }

    void release()
    {
        if (device >= 0) {
            // best effort; the runtime may already be shutting down at thread exit
            if (d_in) (void)hipFree(d_in);
            if (d_out) (void)hipFree(d_out);
            if (stream) (void)hipStreamDestroy(stream);
        }
        d_in = d_out = nullptr;
        stream = nullptr;
        cap = 0;
        device = -1;
    }

    int32_t prepare(size_t bytes)
    {
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count <= 0)
            return fail(DXTLT_E_NO_DEVICE, "no HIP device available (this library has no CPU fallback)", e);
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev), "hipGetDevice");
        if (dev != device) {
            release();
            device = dev;
            HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
        }
        if (bytes > cap) {
            if (d_in) (void)hipFree(d_in);
            if (d_out) (void)hipFree(d_out);
            d_in = d_out = nullptr;
            cap = 0;
            size_t want = bytes + bytes / 8;  // a little headroom for the next, slightly larger file
            if (hipMalloc(&d_in, want) != hipSuccess || hipMalloc(&d_out, want) != hipSuccess) {
                (void)hipGetLastError();
                if (d_in) (void)hipFree(d_in);
                d_in = d_out = nullptr;
                want = bytes;
                HIP_TRY(hipMalloc(&d_in, want), "hipMalloc(input staging)");
                HIP_TRY(hipMalloc(&d_out, want), "hipMalloc(output staging)");
            }
            cap = want;
        }
        return DXTLT_OK;
    }
};

thread_local HostCtx g_host_ctx;

// Small host buffers: a pair of MAPPED pinned staging buffers per thread.  The caller's bytes are copied in by the CPU,
// the kernel reads them over PCIe and writes its result straight into the second buffer, the CPU copies that out: one
// launch and one wait, no copy-engine transfers (each of which is a queue hand-over of its own; a 64 KiB call through
// two hipMemcpyAsync is 37-39 us).  Used up to kMappedMaxBytes (DXTLT_MAPPED_MAX_BYTES; 0 turns it off).
struct MappedPair {
    int device = -1;
    void* h_in = nullptr;
    void* h_out = nullptr;
    void* d_in = nullptr;   // device-side addresses of the two host buffers
    void* d_out = nullptr;
    size_t cap = 0;

    ~MappedPair() { release(); }
    void release()
    {
        if (h_in) (void)hipHostFree(h_in);
        if (h_out) (void)hipHostFree(h_out);
        h_in = h_out = d_in = d_out = nullptr;
        cap = 0;
        device = -1;
    }
    hipError_t reserve(int dev, size_t bytes)
    {
        if (dev == device && bytes <= cap)
            return hipSuccess;
        release();
        const size_t want = std::max<size_t>(bytes, 64u << 10);
        hipError_t e = hipHostMalloc(&h_in, want, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostMalloc(&h_out, want, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer(&d_in, h_in, 0);
        if (e == hipSuccess) e = hipHostGetDevicePointer(&d_out, h_out, 0);
        if (e != hipSuccess) {
            release();
            return e;
        }
        device = dev;
        cap = want;
        return hipSuccess;
    }
};
thread_local MappedPair g_mapped;

// ---------------------------------------------------------------------------------------------------
// Chunked host path: H2D of chunk k+1, the kernel of chunk k and D2H of chunk k-1 overlap.
// Copies from/to pageable host memory block the calling thread while the runtime stages them, so the two
// directions are driven by two host threads: the caller uploads and launches (stream `up`), a helper thread
// downloads (stream `down`) as soon as the chunk's event has fired.  PCIe is full duplex; the kernel time is
// negligible next to either copy.  A chunk is a block range of the whole array (dxtlt_transform_range_device
// semantics), so on the SoA side every chunk moves one slice per stream.
// ---------------------------------------------------------------------------------------------------
// Pipeline thresholds; overridable once per process through the environment for experiments
// (DXTLT_PIPELINE_MIN_BYTES, DXTLT_PIPELINE_CHUNK_BYTES).
size_t env_bytes(const char* name, size_t fallback)
{
    const char* v = std::getenv(name);
    if (v == nullptr || *v == 0)
        return fallback;
    const unsigned long long x = std::strtoull(v, nullptr, 10);
    return x ? (size_t)x : fallback;
}
// Measured (profiles/r01_j_*): one-shot H2D + kernel + D2H runs at ~25.5 GiB/s at every size; the pipeline costs
// ~150 us per chunk and only wins from ~100 MiB up (16 MiB chunks: 32 / 36 / 38 GiB/s at 128 / 256 / 512 MiB; 32 MiB
// chunks: 40-42 GiB/s from 512 MiB up).
const size_t kPipelineMinBytes = env_bytes("DXTLT_PIPELINE_MIN_BYTES", 96u << 20);
// Up to 1 MiB the mapped staging pair wins (tools/host_path_latency.py: 4 KiB 30 -> 17 us per call, 64 KiB 37 -> 20,
// 256 KiB 52 -> 34, 1 MiB 120 -> 101; at 4 MiB it loses, 347 against 194: lanes reading host memory reach ~12 GiB/s
// where the copy engines reach 25).
const size_t kMappedMaxBytes = env_bytes("DXTLT_MAPPED_MAX_BYTES", 1u << 20);
const uint64_t kPipelineChunkOverride = env_bytes("DXTLT_PIPELINE_CHUNK_BYTES", 0) & ~(uint64_t)0xFFFF;
// Chunk size.  BC3's six streams include two of a sixteenth of the data each: with 16 MiB chunks their downloads are
// 1 MiB copies and the pipeline falls to 22-30 GiB/s between 256 MiB and 1 GiB; 32 MiB chunks give 35-40 there
// (tools/host_chunk_sweep.py, round 2).  BC1 / BC2 keep 16 MiB chunks below 256 MiB (one more GiB/s at 128 MiB).
inline uint64_t pipeline_chunk_bytes(uint64_t len, int32_t format)
{
    if (kPipelineChunkOverride)
        return kPipelineChunkOverride;
    return (format == 3 || len >= (256ull << 20)) ? (32ull << 20) : (16ull << 20);
}
std::atomic<int> g_host_pipeline{1};

struct PipeShared {
    std::mutex m;
    std::condition_variable cv;
    int launched = 0;     // chunks whose kernel (and event) have been enqueued
    bool failed = false;  // uploader gave up
};

// One pipelined job: blocks [first, first + count) of a host-resident array of `total` blocks, on device `dev`.
// The device holds the range as a stand-alone array of `count` blocks (d_in / d_out of count * B bytes: AoS slice
// and compact SoA, which IS the range's slice of every stream, packed); host offsets are those of the whole array.
// With first = 0 and count = total this is the whole-buffer pipeline of the host-pointer entry points; with a proper
// sub-range it is one shard of dxtlt_transform_sharded.
struct PipeJob {
    int dev;
    hipStream_t up;
    void* d_in;
    void* d_out;
    int32_t format;
    bool inverse;
    const uint8_t* in;
    uint8_t* out;
    uint64_t total, first, count;
    uint8_t mode;
    bool sa, sc;
    uint8_t normalize;
    uint64_t chunk_bytes;
};

int32_t pipelined_range(const PipeJob& j)
{
    // format 7 = the main part of a BC7 buffer (include/dxtlt_bc7.h): eight streams over whole 1024-block granules, chunks
    // are granule multiples, the kernels are the BC7 range launches; everything else is the same pipeline
    const bool bc7 = j.format == 7;
    const uint64_t B = bc7 ? 16 : (uint64_t)dxtlt::block_bytes((Format)j.format);
    struct {
        int n;
        int off[8], width[8];
    } S{};
    if (bc7) {
        const int off[8] = {0, 8, 10, 11, 12, 13, 14, 15}, width[8] = {8, 2, 1, 1, 1, 1, 1, 1};
        S.n = 8;
        for (int s = 0; s < 8; ++s) {
            S.off[s] = off[s];
            S.width[s] = width[s];
        }
    } else {
        const dxtlt::Streams bs = dxtlt::make_streams(j.format, j.format == 3 && j.sa, j.sc);
        S.n = bs.n;
        for (int s = 0; s < bs.n; ++s) {
            S.off[s] = bs.off[s];
            S.width[s] = bs.width[s];
        }
    }
    auto launch = [&](bool inv, const void* src, void* dst, uint64_t range_total, uint64_t range_first, uint64_t range_count) -> int32_t {
        if (!bc7)
            return device_range(j.format, inv, src, dst, range_total, range_first, range_count, j.mode, j.sa, j.sc, j.up, inv ? 0 : j.normalize);
        const hipError_t e = dxtlt::bc7::launch_range(inv, src, dst, range_total, range_first, range_count, j.up);
        return e == hipSuccess ? DXTLT_OK : fail(DXTLT_E_DEVICE, "BC7 kernel launch", e);
    };
    const uint64_t chunk_blocks = j.chunk_bytes / B;  // a multiple of every tile size (and of the BC7 granule)
    const int nchunks = (int)((j.count + chunk_blocks - 1) / chunk_blocks);
    const int dev = j.dev;
    const bool inverse = j.inverse;
    const uint64_t total = j.total, base = j.first, blocks = j.count;
    const uint8_t* in = j.in;
    uint8_t* out = j.out;

    hipStream_t down = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&down, hipStreamNonBlocking), "hipStreamCreate(download)");
    std::vector<hipEvent_t> ev((size_t)nchunks, nullptr);
    for (auto& e : ev) {
        hipError_t err = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (err != hipSuccess) {
            for (auto& e2 : ev) if (e2) (void)hipEventDestroy(e2);
            (void)hipStreamDestroy(down);
            return fail(DXTLT_E_DEVICE, "hipEventCreate", err);
        }
    }

    PipeShared sh;
    hipError_t down_err = hipSuccess;
    std::thread downloader([&] {
        hipError_t e = hipSetDevice(dev);
        for (int k = 0; k < nchunks && e == hipSuccess; ++k) {
            {
                std::unique_lock<std::mutex> lk(sh.m);
                sh.cv.wait(lk, [&] { return sh.launched > k || sh.failed; });
                if (sh.launched <= k)
                    break;  // uploader failed before this chunk
            }
            const uint64_t first = (uint64_t)k * chunk_blocks;
            const uint64_t count = std::min<uint64_t>(chunk_blocks, blocks - first);
            e = hipStreamWaitEvent(down, ev[(size_t)k], 0);
            if (!inverse) {
                for (int s = 0; s < S.n && e == hipSuccess; ++s) {
                    const uint64_t w = (uint64_t)S.width[s], off = (uint64_t)S.off[s];
                    e = hipMemcpyAsync(out + off * total + w * (base + first),
                                       (const uint8_t*)j.d_out + off * blocks + w * first, (size_t)(w * count),
                                       hipMemcpyDeviceToHost, down);
                }
            } else if (e == hipSuccess) {
                e = hipMemcpyAsync(out + (base + first) * B, (const uint8_t*)j.d_out + first * B, (size_t)(count * B),
                                   hipMemcpyDeviceToHost, down);
            }
        }
        // drain whatever was enqueued, also after a failure: the events and the stream die with this call
        hipError_t e2 = hipStreamSynchronize(down);
        down_err = e != hipSuccess ? e : e2;
    });

    hipError_t up_err = hipSuccess;
    int32_t rc = DXTLT_OK;
    for (int k = 0; k < nchunks; ++k) {
        const uint64_t first = (uint64_t)k * chunk_blocks;
        const uint64_t count = std::min<uint64_t>(chunk_blocks, blocks - first);
        if (!inverse) {
            up_err = hipMemcpyAsync((uint8_t*)j.d_in + first * B, in + (base + first) * B, (size_t)(count * B),
                                    hipMemcpyHostToDevice, j.up);
            if (up_err == hipSuccess)
                rc = launch(false, (const uint8_t*)j.d_in + first * B, j.d_out, blocks, first, count);
        } else {
            for (int s = 0; s < S.n && up_err == hipSuccess; ++s) {
                const uint64_t w = (uint64_t)S.width[s], off = (uint64_t)S.off[s];
                up_err = hipMemcpyAsync((uint8_t*)j.d_in + off * blocks + w * first, in + off * total + w * (base + first),
                                        (size_t)(w * count), hipMemcpyHostToDevice, j.up);
            }
            if (up_err == hipSuccess)
                rc = launch(true, j.d_in, (uint8_t*)j.d_out + first * B, blocks, first, count);
        }
        if (up_err == hipSuccess && rc == DXTLT_OK)
            up_err = hipEventRecord(ev[(size_t)k], j.up);
        {
            std::lock_guard<std::mutex> lk(sh.m);
            if (up_err == hipSuccess && rc == DXTLT_OK)
                sh.launched = k + 1;
            else
                sh.failed = true;
        }
        sh.cv.notify_all();
        if (up_err != hipSuccess || rc != DXTLT_OK)
            break;
    }
    downloader.join();
    // every exit drains the upload stream before the events go away and the staging buffers can be reused
    // (a failed copy or launch leaves earlier chunks queued)
    const hipError_t drain = hipStreamSynchronize(j.up);
    if (up_err == hipSuccess && rc == DXTLT_OK)
        up_err = drain;
    for (auto& e : ev) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(down);
    if (rc != DXTLT_OK)
        return rc;
    if (up_err != hipSuccess)
        return fail(DXTLT_E_DEVICE, "pipelined upload/launch", up_err);
    if (down_err != hipSuccess)
        return fail(DXTLT_E_DEVICE, "pipelined download", down_err);
    return DXTLT_OK;
}

int32_t pipelined_transform(HostCtx& c, int32_t format, bool inverse, const uint8_t* in, uint8_t* out, uint64_t blocks,
                            uint8_t mode, bool sa, bool sc, uint8_t normalize)
{
    const uint64_t B = (uint64_t)dxtlt::block_bytes((Format)format);
    PipeJob j{c.device, c.stream, c.d_in, c.d_out, format, inverse, in, out, blocks, 0, blocks, mode, sa, sc, normalize,
              pipeline_chunk_bytes(blocks * B, format)};
    return pipelined_range(j);
}

}  // namespace

// The main part (whole granules) of a large BC7 host buffer through the chunked pipeline; false = too small / switched off
bool dxtlt_host::pipelined_bc7_main(bool inverse, const uint8_t* in, uint8_t* out, uint64_t main_blocks, int32_t* rc)
{
    const uint64_t bytes = main_blocks * 16;
    if (bytes < kPipelineMinBytes || g_host_pipeline.load(std::memory_order_relaxed) == 0)
        return false;
    HostCtx& c = g_host_ctx;
    *rc = c.prepare((size_t)bytes);
    if (*rc != DXTLT_OK)
        return true;
    // eight downloads per chunk, five of them a sixteenth of it: larger chunks than BC1-3 (16 MiB chunks lose to the
    // one-shot path below 1 GiB; 32 MiB: 30 / 34 / 37 GiB/s at 128 / 256 / 512 MiB; 64 MiB: 40-41 from 1 GiB up;
    // tools/bc7_host_bench.py)
    const uint64_t chunk = kPipelineChunkOverride ? kPipelineChunkOverride : bytes >= (1ull << 30) ? (64ull << 20) : (32ull << 20);
    PipeJob j{c.device, c.stream, c.d_in, c.d_out, 7, inverse, in, out, main_blocks, 0, main_blocks, 0, false, false, 0, chunk};
    *rc = pipelined_range(j);
    return true;
}

int32_t dxtlt_host::acquire_staging(size_t bytes, void** d_in, void** d_out, hipStream_t* stream)
{
    HostCtx& c = g_host_ctx;
    int32_t rc = c.prepare(bytes);
    if (rc != DXTLT_OK)
        return rc;
    *d_in = c.d_in;
    *d_out = c.d_out;
    *stream = c.stream;
    return DXTLT_OK;
}

int32_t dxtlt_host::acquire_mapped_staging(size_t bytes, MappedStaging* out)
{
    out->usable = false;
    if (bytes > kMappedMaxBytes)
        return DXTLT_OK;
    HostCtx& c = g_host_ctx;
    int32_t rc = c.prepare(0);   // device and stream only
    if (rc != DXTLT_OK)
        return rc;
    MappedPair& m = g_mapped;
    HIP_TRY(m.reserve(c.device, bytes), "hipHostMalloc(mapped staging)");
    *out = MappedStaging{true, m.h_in, m.h_out, m.d_in, m.d_out, c.stream};
    return DXTLT_OK;
}

int32_t dxtlt_host::enqueue(int32_t format, bool inverse, const void* d_src, void* d_dst, uint64_t blocks, uint8_t mode,
                            bool sa, bool sc, hipStream_t stream, uint8_t normalize)
{
    return device_range(format, inverse, d_src, d_dst, blocks, 0, blocks, mode, sa, sc, stream, normalize);
}

int32_t dxtlt_host::transform(int32_t format, bool inverse, const uint8_t* in, uint8_t* out, size_t len, uint8_t mode,
                              bool sa, bool sc, uint8_t normalize)
{
    int32_t rc = check_common(format, len, mode, in, out);
    if (rc != DXTLT_OK)
        return rc;
    if (len == 0)
        return DXTLT_OK;  // zero blocks: nothing to do, no device needed
    HostCtx& c = g_host_ctx;
    const uint64_t blocks = len / (size_t)dxtlt::block_bytes((Format)format);
    MappedStaging m;
    rc = acquire_mapped_staging(len, &m);
    if (rc != DXTLT_OK)
        return rc;
    if (m.usable) {
        std::memcpy(m.h_in, in, len);
        rc = device_range(format, inverse, m.d_in, m.d_out, blocks, 0, blocks, mode, sa, sc, m.stream, normalize);
        const hipError_t drained = hipStreamSynchronize(m.stream);
        if (rc != DXTLT_OK)
            return rc;
        HIP_TRY(drained, "stream synchronize");
        std::memcpy(out, m.h_out, len);
        return DXTLT_OK;
    }
    rc = c.prepare(len);
    if (rc != DXTLT_OK)
        return rc;
    if (len >= kPipelineMinBytes && g_host_pipeline.load(std::memory_order_relaxed) != 0)
        return pipelined_transform(c, format, inverse, in, out, blocks, mode, sa, sc, normalize);
    // Every failure exit drains the stream first: the staging buffers belong to this thread's next call, which may
    // free or regrow them while an earlier copy or kernel of this one is still queued.
    hipError_t e = hipMemcpyAsync(c.d_in, in, len, hipMemcpyHostToDevice, c.stream);
    const char* what = "H2D copy";
    if (e == hipSuccess) {
        rc = device_range(format, inverse, c.d_in, c.d_out, blocks, 0, blocks, mode, sa, sc, c.stream, normalize);
        if (rc != DXTLT_OK) {
            (void)hipStreamSynchronize(c.stream);
            return rc;
        }
        e = hipMemcpyAsync(out, c.d_out, len, hipMemcpyDeviceToHost, c.stream);
        what = "D2H copy";
    }
    const hipError_t drained = hipStreamSynchronize(c.stream);
    if (e != hipSuccess)
        return fail(DXTLT_E_DEVICE, what, e);
    HIP_TRY(drained, "stream synchronize");
    return DXTLT_OK;
}

namespace {
using dxtlt_host::transform;
inline int32_t host_call(int32_t format, bool inverse, const uint8_t* in, uint8_t* out, size_t len, uint8_t mode,
                         bool sa, bool sc)
{
    return transform(format, inverse, in, out, len, mode, sa, sc);
}

// ---------------------------------------------------------------------------------------------------
// Single-process multi-GPU shard path: contiguous block ranges, one host thread per
// device, no collective.  A shard is transformed as a stand-alone buffer on its device (blocks are
// independent, so its compact SoA result holds exactly this shard's slice of every stream); the
// "host concat" is one D2H copy per stream straight to the slice's final place.
// ---------------------------------------------------------------------------------------------------
struct ShardPlan {
    uint64_t first;
    uint64_t count;
};

std::vector<ShardPlan> plan_shards(uint64_t total_blocks, int shards, uint64_t align_blocks)
{
    // equal shares rounded down to a multiple of `align_blocks` (keeps every per-stream slice 16-byte
    // aligned and tile-sized); the last shard takes the remainder
    std::vector<ShardPlan> p((size_t)shards);
    uint64_t share = total_blocks / (uint64_t)shards;
    share -= share % align_blocks;
    uint64_t at = 0;
    for (int i = 0; i < shards; ++i) {
        uint64_t n = (i == shards - 1) ? total_blocks - at : share;
        p[(size_t)i] = {at, n};
        at += n;
    }
    return p;
}

// Per-device shard contexts (a stream and a grow-only pair of device buffers), kept across dxtlt_transform_sharded calls:
// the shard threads are new on every call, so thread-local staging as in the host-pointer path would be allocated and
// freed each time -- two hipMalloc / hipFree of the shard's size per call cost a 4 GiB BC3 array 30 ms of its 130 (pinned
// host memory) and far more with pageable memory (13 against 41 GiB/s through the single-buffer entry point;
// tools/pinned_host_probe.py).  dxtlt_release_thread_resources() frees the idle ones.
struct ShardCtx {
    int dev = -1;
    hipStream_t st = nullptr;
    void* a = nullptr;
    void* b = nullptr;
    size_t cap = 0;
    bool busy = false;
};
std::mutex g_shard_pool_mutex;
std::vector<ShardCtx*> g_shard_pool;

ShardCtx* shard_ctx_acquire(int dev, size_t bytes, hipError_t* err)
{
    ShardCtx* c = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_shard_pool_mutex);
        for (ShardCtx* x : g_shard_pool)
            if (!x->busy && x->dev == dev && (c == nullptr || x->cap > c->cap))
                c = x;
        if (c == nullptr) {
            c = new ShardCtx();
            c->dev = dev;
            g_shard_pool.push_back(c);
        }
        c->busy = true;
    }
    *err = hipSuccess;
    if (c->st == nullptr)
        *err = hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking);
    if (*err == hipSuccess && c->cap < bytes) {
        if (c->a) (void)hipFree(c->a);
        if (c->b) (void)hipFree(c->b);
        c->a = c->b = nullptr;
        c->cap = 0;
        *err = hipMalloc(&c->a, bytes);
        if (*err == hipSuccess)
            *err = hipMalloc(&c->b, bytes);
        if (*err == hipSuccess) {
            c->cap = bytes;
        } else {
            if (c->a) (void)hipFree(c->a);
            c->a = c->b = nullptr;
        }
    }
    if (*err != hipSuccess) {
        (void)hipGetLastError();
        std::lock_guard<std::mutex> lk(g_shard_pool_mutex);
        c->busy = false;
        return nullptr;
    }
    return c;
}

// At most kIdleShardCtxPerDevice idle contexts stay per device (the largest ones): a call with 64 round-robin shards on
// one device would otherwise leave 64 streams and 2 x the array size of HBM behind until someone calls
// dxtlt_release_thread_resources().  Retained memory per device is thus bounded by 2 buffers x the largest shard x 2.
constexpr int kIdleShardCtxPerDevice = 2;

// Handing a context back only marks it idle: hipFree synchronises the whole device, so nothing is freed on a shard's
// completion path while other shards of the call are still moving data.  The surplus is trimmed by the call itself, after
// its workers have joined (shard_pool_trim).
void shard_ctx_release(ShardCtx* c)
{
    std::lock_guard<std::mutex> lk(g_shard_pool_mutex);
    c->busy = false;
}

void shard_pool_trim()
{
    std::vector<ShardCtx*> drop;
    {
        std::lock_guard<std::mutex> lk(g_shard_pool_mutex);
        std::vector<ShardCtx*> idle;
        for (ShardCtx* x : g_shard_pool)
            if (!x->busy)
                idle.push_back(x);
        // per device: keep the kIdleShardCtxPerDevice largest idle contexts
        std::sort(idle.begin(), idle.end(), [](const ShardCtx* l, const ShardCtx* r) { return l->dev != r->dev ? l->dev < r->dev : l->cap > r->cap; });
        int run = 0;
        for (size_t i = 0; i < idle.size(); ++i) {
            run = (i > 0 && idle[i]->dev == idle[i - 1]->dev) ? run + 1 : 0;
            if (run >= kIdleShardCtxPerDevice)
                drop.push_back(idle[i]);
        }
        for (ShardCtx* x : drop)
            g_shard_pool.erase(std::find(g_shard_pool.begin(), g_shard_pool.end(), x));
    }
    if (drop.empty())
        return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (ShardCtx* x : drop) {
        if (hipSetDevice(x->dev) == hipSuccess) {
            if (x->a) (void)hipFree(x->a);
            if (x->b) (void)hipFree(x->b);
            if (x->st) (void)hipStreamDestroy(x->st);
        }
        delete x;
    }
    (void)hipSetDevice(prev);
}

void shard_pool_clear()
{
    std::vector<ShardCtx*> idle;
    {
        std::lock_guard<std::mutex> lk(g_shard_pool_mutex);
        std::vector<ShardCtx*> keep;
        for (ShardCtx* x : g_shard_pool)
            (x->busy ? keep : idle).push_back(x);
        g_shard_pool.swap(keep);
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (ShardCtx* x : idle) {
        if (hipSetDevice(x->dev) == hipSuccess) {
            if (x->a) (void)hipFree(x->a);
            if (x->b) (void)hipFree(x->b);
            if (x->st) (void)hipStreamDestroy(x->st);
        }
        delete x;
    }
    (void)hipSetDevice(prev);
}

int32_t shard_worker(int dev, int32_t format, bool inverse, const uint8_t* in, uint8_t* out, uint64_t total,
                     ShardPlan sp, uint8_t mode, bool sa, bool sc)
{
    if (sp.count == 0)
        return DXTLT_OK;
    const uint64_t B = (uint64_t)dxtlt::block_bytes((Format)format);
    const size_t bytes = (size_t)(sp.count * B);
    const dxtlt::Streams S = dxtlt::make_streams(format, format == 3 && sa, sc);
    HIP_TRY(hipSetDevice(dev), "hipSetDevice");
    hipError_t acquire_err = hipSuccess;
    ShardCtx* ctx = shard_ctx_acquire(dev, bytes, &acquire_err);
    if (ctx == nullptr)
        return fail(DXTLT_E_DEVICE, "shard stream / buffers", acquire_err);
    hipStream_t st = ctx->st;
    void *d_a = ctx->a, *d_b = ctx->b;
    int32_t rc = DXTLT_OK;
    auto done = [&](int32_t code) {   // every path below has drained `st` before it gets here
        shard_ctx_release(ctx);
        return code;
    };

    hipError_t e = hipSuccess;
    if (bytes >= kPipelineMinBytes && g_host_pipeline.load(std::memory_order_relaxed) != 0) {
        // large shard: upload, kernel and the per-stream downloads of consecutive chunks overlap
        PipeJob j{dev, st, d_a, d_b, format, inverse, in, out, total, sp.first, sp.count, mode, sa, sc, 0,
                  pipeline_chunk_bytes(bytes, format)};
        return done(pipelined_range(j));
    }
    if (!inverse) {
        // AoS slice in, compact SoA out, then scatter the stream slices to their final host offsets
        e = hipMemcpyAsync(d_a, in + sp.first * B, bytes, hipMemcpyHostToDevice, st);
        if (e == hipSuccess)
            rc = device_range(format, false, d_a, d_b, sp.count, 0, sp.count, mode, sa, sc, st);
        for (int s = 0; s < S.n && e == hipSuccess && rc == DXTLT_OK; ++s) {
            const uint64_t w = (uint64_t)S.width[s], off = (uint64_t)S.off[s];
            e = hipMemcpyAsync(out + off * total + w * sp.first, (const uint8_t*)d_b + off * sp.count,
                               (size_t)(w * sp.count), hipMemcpyDeviceToHost, st);
        }
    } else {
        // gather this shard's slice of every stream into a compact SoA buffer, untransform, copy AoS back
        for (int s = 0; s < S.n && e == hipSuccess; ++s) {
            const uint64_t w = (uint64_t)S.width[s], off = (uint64_t)S.off[s];
            e = hipMemcpyAsync((uint8_t*)d_a + off * sp.count, in + off * total + w * sp.first,
                               (size_t)(w * sp.count), hipMemcpyHostToDevice, st);
        }
        if (e == hipSuccess)
            rc = device_range(format, true, d_a, d_b, sp.count, 0, sp.count, mode, sa, sc, st);
        if (e == hipSuccess && rc == DXTLT_OK)
            e = hipMemcpyAsync(out + sp.first * B, d_b, bytes, hipMemcpyDeviceToHost, st);
    }
    // drained on every exit: the buffers and the stream are freed below
    const hipError_t drained = hipStreamSynchronize(st);
    if (e == hipSuccess && rc == DXTLT_OK)
        e = drained;
    if (rc != DXTLT_OK)
        return done(rc);
    if (e != hipSuccess)
        return done(fail(DXTLT_E_DEVICE, "shard copy/launch", e));
    return done(DXTLT_OK);
}

}  // namespace

int32_t dxtlt_host::acquire_shard_buffers(int dev, size_t bytes, ShardBuffers* out)
{
    hipError_t err = hipSuccess;
    ShardCtx* c = shard_ctx_acquire(dev, bytes, &err);
    if (c == nullptr)
        return fail(DXTLT_E_DEVICE, "shard stream / buffers", err);
    *out = ShardBuffers{c->st, c->a, c->b, c};
    return DXTLT_OK;
}

void dxtlt_host::release_shard_buffers(const ShardBuffers& sb) { shard_ctx_release(static_cast<ShardCtx*>(sb.handle)); }
void dxtlt_host::trim_idle_shard_buffers() { shard_pool_trim(); }
void dxtlt_host::init_runtime_for_devices(int devices)
{
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (int d = 0; d < devices; ++d)
        if (hipSetDevice(d) == hipSuccess)
            (void)hipFree(nullptr);
    (void)hipSetDevice(prev);
}

bool dxtlt_host::pipelined_bc7_shard(const ShardBuffers& sb, int dev, bool inverse, const uint8_t* in, uint8_t* out,
                                     uint64_t total_main, uint64_t first, uint64_t count, int32_t* rc)
{
    const uint64_t bytes = count * 16;
    if (bytes < kPipelineMinBytes || g_host_pipeline.load(std::memory_order_relaxed) == 0)
        return false;
    const uint64_t chunk = kPipelineChunkOverride ? kPipelineChunkOverride : bytes >= (1ull << 30) ? (64ull << 20) : (32ull << 20);
    PipeJob j{dev, sb.stream, sb.a, sb.b, 7, inverse, in, out, total_main, first, count, 0, false, false, 0, chunk};
    *rc = pipelined_range(j);
    return true;
}


extern "C" {

// ---- host pointers ------------------------------------------------------------------------------
int32_t dxtlt_transform_bc1_with_settings(const uint8_t* i, uint8_t* o, size_t len, uint8_t mode, bool sc)
{
    return host_call(1, false, i, o, len, mode, false, sc);
}
int32_t dxtlt_untransform_bc1_with_settings(const uint8_t* i, uint8_t* o, size_t len, uint8_t mode, bool sc)
{
    return host_call(1, true, i, o, len, mode, false, sc);
}
int32_t dxtlt_transform_bc2_with_settings(const uint8_t* i, uint8_t* o, size_t len, uint8_t mode, bool sc)
{
    return host_call(2, false, i, o, len, mode, false, sc);
}
int32_t dxtlt_untransform_bc2_with_settings(const uint8_t* i, uint8_t* o, size_t len, uint8_t mode, bool sc)
{
    return host_call(2, true, i, o, len, mode, false, sc);
}
int32_t dxtlt_transform_bc3_with_settings(const uint8_t* i, uint8_t* o, size_t len, uint8_t mode, bool sa, bool sc)
{
    return host_call(3, false, i, o, len, mode, sa, sc);
}
int32_t dxtlt_untransform_bc3_with_settings(const uint8_t* i, uint8_t* o, size_t len, uint8_t mode, bool sa, bool sc)
{
    return host_call(3, true, i, o, len, mode, sa, sc);
}

// ---- device pointers, whole buffer -------------------------------------------------------------------
static int32_t device_whole(int32_t format, bool inverse, const void* d_in, void* d_out, size_t len, uint8_t mode,
                            bool sa, bool sc, void* stream)
{
    int32_t rc = check_common(format, len, mode, d_in, d_out);
    if (rc != DXTLT_OK)
        return rc;
    const uint64_t blocks = len / (size_t)dxtlt::block_bytes((Format)format);
    return device_range(format, inverse, d_in, d_out, blocks, 0, blocks, mode, sa, sc, (hipStream_t)stream);
}

int32_t dxtlt_transform_bc1_with_settings_device(const void* i, void* o, size_t len, uint8_t mode, bool sc, void* st)
{
    return device_whole(1, false, i, o, len, mode, false, sc, st);
}
int32_t dxtlt_untransform_bc1_with_settings_device(const void* i, void* o, size_t len, uint8_t mode, bool sc, void* st)
{
    return device_whole(1, true, i, o, len, mode, false, sc, st);
}
int32_t dxtlt_transform_bc2_with_settings_device(const void* i, void* o, size_t len, uint8_t mode, bool sc, void* st)
{
    return device_whole(2, false, i, o, len, mode, false, sc, st);
}
int32_t dxtlt_untransform_bc2_with_settings_device(const void* i, void* o, size_t len, uint8_t mode, bool sc, void* st)
{
    return device_whole(2, true, i, o, len, mode, false, sc, st);
}
int32_t dxtlt_transform_bc3_with_settings_device(const void* i, void* o, size_t len, uint8_t mode, bool sa, bool sc,
                                                 void* st)
{
    return device_whole(3, false, i, o, len, mode, sa, sc, st);
}
int32_t dxtlt_untransform_bc3_with_settings_device(const void* i, void* o, size_t len, uint8_t mode, bool sa, bool sc,
                                                   void* st)
{
    return device_whole(3, true, i, o, len, mode, sa, sc, st);
}

int32_t dxtlt_transform_range_device(int32_t format, bool inverse, const void* d_src, void* d_dst,
                                     uint64_t total_blocks, uint64_t first_block, uint64_t num_blocks, uint8_t mode,
                                     bool sa, bool sc, void* stream)
{
    return device_range(format, inverse, d_src, d_dst, total_blocks, first_block, num_blocks, mode, sa, sc,
                        (hipStream_t)stream);
}

// ---- single-process multi-GPU ---------------------------------------------------------------------------
int32_t dxtlt_transform_sharded(int32_t format, bool inverse, const uint8_t* in, uint8_t* out, size_t len,
                                uint8_t mode, bool sa, bool sc, int32_t num_devices)
{
    int32_t rc = check_common(format, len, mode, in, out);
    if (rc != DXTLT_OK)
        return rc;
    if (len == 0)
        return DXTLT_OK;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(DXTLT_E_NO_DEVICE, "no HIP device available (this library has no CPU fallback)", e);
    // more shards than devices are dealt round robin (a 1-GPU box runs the multi-shard placement that way)
    int shards = num_devices <= 0 ? count : std::min(num_devices, 64);
    const uint64_t total = len / (size_t)dxtlt::block_bytes((Format)format);
    if ((uint64_t)shards > total)
        shards = (int)total;
    int prev = 0;
    (void)hipGetDevice(&prev);

    // 2048 blocks = one BC1 tile = two BC2/BC3 tiles; also a multiple of 16 so all slices stay 16-B aligned
    std::vector<ShardPlan> plan = plan_shards(total, shards, 2048);
    std::vector<int32_t> codes((size_t)shards, DXTLT_OK);
    std::vector<std::string> msgs((size_t)shards);
    std::vector<DxtltShardStat> stats((size_t)shards);
    std::vector<std::thread> threads;
    // The workers narrow their own affinity before their first HIP call, and threads the HIP / ROCr runtime starts lazily
    // from a worker would inherit that mask for the life of the process.  So the runtime is brought up for every device
    // this call uses HERE, on the caller's unbound thread, before any worker exists (DXTLT_NUMA_BIND in the header).
    dxtlt_host::init_runtime_for_devices(std::min(shards, count));
    // thread creation can fail (EAGAIN under a process limit): whatever was started is joined before the error leaves
    int32_t spawn_rc = DXTLT_OK;
    for (int d = 0; d < shards && spawn_rc == DXTLT_OK; ++d) {
        try {
            threads.emplace_back([&, d] {
                // this thread is the library's own: put it next to its device before it submits anything (the pipeline's
                // downloader thread is created from it and inherits the mask)
                const int bound = dxtlt_host::bind_this_thread_near_device(d % count);
                const auto t0 = std::chrono::steady_clock::now();
                codes[(size_t)d] = shard_worker(d % count, format, inverse, in, out, total, plan[(size_t)d], mode, sa, sc);
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                stats[(size_t)d] = DxtltShardStat{d % count, bound, plan[(size_t)d].first, plan[(size_t)d].count, dt};
                if (codes[(size_t)d] != DXTLT_OK)
                    msgs[(size_t)d] = g_last_error;
            });
        } catch (const std::exception&) {
            spawn_rc = fail(DXTLT_E_ALLOCATION, "could not start a shard worker thread");   // a host resource ran out (batch_host, auto pool: the same code)
        }
    }
    for (auto& t : threads)
        t.join();
    shard_pool_trim();   // idle contexts beyond the cap, now that no shard of this call is moving data
    g_shard_stats = stats;
    if (spawn_rc != DXTLT_OK) {
        (void)hipSetDevice(prev);
        return spawn_rc;
    }
    (void)hipSetDevice(prev);
    for (int d = 0; d < shards; ++d) {
        if (codes[(size_t)d] != DXTLT_OK) {
            g_last_error = msgs[(size_t)d];
            return codes[(size_t)d];
        }
    }
    return DXTLT_OK;
}

int32_t dxtlt_sharded_last_stats(DxtltShardStat* out, int32_t cap)
{
    const int32_t n = (int32_t)g_shard_stats.size();
    for (int32_t i = 0; out != nullptr && i < n && i < cap; ++i)
        out[i] = g_shard_stats[(size_t)i];
    return n;
}

// ---- plumbing -------------------------------------------------------------------------------------------
int32_t dxtlt_fill_splitmix64_device(void* d_dst, size_t len_bytes, uint64_t seed, uint64_t first_qword, void* stream)
{
    if (len_bytes > 0 && d_dst == nullptr)
        return fail(DXTLT_E_INVALID_ARGUMENT, "NULL device buffer");
    HIP_TRY(dxtlt::launch_fill_splitmix64(d_dst, len_bytes, seed, first_qword, (hipStream_t)stream), "fill launch");
    return DXTLT_OK;
}

const char* dxtlt_last_error(void) { return g_last_error.c_str(); }

int32_t dxtlt_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess)
        return 0;
    return count;
}

// The host-pointer crossover a size-routing caller should use (include/dxtlt_gfx950.h).  32 MiB: the size at which the
// chunked host pipeline (upload | kernel | download) overtakes one core of the reference's SIMD path on this class of host
// (profiles/r01_w_host_path_latency_vs_cpu.json, r01_j_host_pointer_pipeline_threshold.txt).
static std::atomic<size_t> g_host_route_threshold{size_t(32) << 20};

size_t dxtlt_host_route_threshold_bytes(void)
{
    if (const char* v = std::getenv("DXTLT_HOST_ROUTE_THRESHOLD_BYTES")) {
        char* end = nullptr;
        const unsigned long long x = std::strtoull(v, &end, 10);
        if (end != v)
            return (size_t)x;
    }
    return g_host_route_threshold.load();
}

void dxtlt_set_host_route_threshold_bytes(size_t bytes) { g_host_route_threshold.store(bytes); }

int32_t dxtlt_tuning_mask(void) { return dxtlt::launch_force_mask(); }

int32_t dxtlt_debug_plan_transform(int32_t format, int32_t inverse, int32_t variant, int32_t split_alpha, int32_t split_colour,
                                   uint64_t src_address, uint64_t dst_address, uint64_t total_blocks, uint64_t first_block,
                                   uint64_t num_blocks, DxtltDebugPlannedLaunch* out, int32_t cap)
{
    if (format < 1 || format > 3 || cap < 0 || (cap > 0 && out == nullptr) || first_block > total_blocks || num_blocks > total_blocks - first_block)
        return -1;
    std::vector<dxtlt::DebugPlannedLaunch> tmp((size_t)cap);
    const Settings s{variant, split_alpha != 0, split_colour != 0, 0};
    const dxtlt::LaunchTuning t = current_tuning();
    const int n = dxtlt::debug_plan_transform((Format)format, inverse != 0, s, src_address, dst_address, Range{total_blocks, first_block, num_blocks},
                                              &t, tmp.data(), cap);
    for (int i = 0; i < n && i < cap; ++i) {
        const dxtlt::DebugPlannedLaunch& l = tmp[(size_t)i];
        DxtltDebugPlannedLaunch& o = out[i];
        o.kind = l.kind;
        o.threads = l.threads;
        o.workgroups = l.workgroups;
        o.full_tiles = l.full_tiles;
        o.range_blocks = l.range_blocks;
        o.aos_offset = l.aos_offset;
        std::memcpy(o.shift, l.shift, sizeof o.shift);
        o.halo_vecs = l.halo_vecs;
        o.natural = l.natural;
        std::memcpy(o.gbase, l.gbase, sizeof o.gbase);
    }
    return n;
}

void dxtlt_set_tuning(int32_t tile_threads, int32_t force_path)
{
    g_tile_threads.store(tile_threads);
    // Only the bits this build knows (bcn_kernels.hip, kForceMask): the shipped library honours 2 and 0x20 -- test levers that
    // select paths some address pattern selects by itself, results exact -- and nothing else; in particular it contains no switch
    // that changes results.  The experiments side build (-DDXTLT_EXPERIMENTS) adds the rest; its 0x10, a timing experiment with
    // WRONG output, additionally needs DXTLT_TIMING_EXPERIMENTS in the environment.
    int32_t allowed = dxtlt::launch_force_mask();
#ifdef DXTLT_EXPERIMENTS
    static const bool timing_experiments = std::getenv("DXTLT_TIMING_EXPERIMENTS") != nullptr;
    if (!timing_experiments)
        allowed &= ~0x10;
#endif
    const int32_t bits = force_path & allowed;
    g_force_generic.store(bits);
    g_xcd_remap.store((bits & 0x100) ? 0 : (bits & 0x200) ? 1 : -1);   // (experiments build: XCD-contiguous tile order off / on)
}

const char* dxtlt_version(void) { return "dxtlt-gfx950 0.2.0"; }

void dxtlt_release_thread_resources(void)
{
    g_host_ctx.release();
    g_mapped.release();
    shard_pool_clear();   // process-wide: the idle per-device contexts of dxtlt_transform_sharded
    dxtlt_host::release_bc7_thread_scratch();
    dxtlt_host::release_normalize_thread_flag();
    dxtlt_host::release_batch_thread_tables();
    dxtlt_host::release_auto_thread_arena();
}

}  // extern "C"
