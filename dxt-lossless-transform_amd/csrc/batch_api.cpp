// batch_api.cpp -- dxtlt_transform_batch_device: many device-resident buffers in one call.
//
// The reference transforms file after file (its CLI fans files out over rayon workers); a texture is typically
// 0.1-20 MiB.  On MI355X one such buffer cannot fill the chip (a 1 MiB BC1 texture is 256 workgroups for 256 CUs) and
// a launch costs the host ~5 us, longer than the kernel runs: measured 186 GiB/s for 1024 x 1 MiB through one call per
// buffer, even when spread over eight streams (profiles/r01_x).  So a batch becomes ONE launch per (format, direction)
// present in it: the host lays the buffers' workgroups end to end in a table (48 bytes per buffer plus a coarse
// workgroup -> buffer index), copies the table to the device on the caller's stream, and launches batch_kernel
// (bcn_kernels.hip), in which every workgroup looks its buffer up and runs one shifted tile or 256 blocks of the
// element path with that buffer's settings.  Asynchronous and ordered like a single call on the caller's stream.
#include <hip/hip_runtime_api.h>

#include <cstring>
#include <vector>

#include "../../include/dxtlt_gfx950.h"
#include "bcn_launch.h"
#include "host_common.h"

namespace {

using namespace dxtlt_host;
using dxtlt::BatchEntry;

// Table staging: a ring of pinned host buffers with device twins.  A slot is reused only after the copy that last read
// it has finished (its event), so the call never blocks unless more than kSlots batches are in flight.
constexpr int kSlots = 4;

struct TableSlot {
    void* host = nullptr;
    void* dev = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;
    bool pending = false;
};

struct TableRing {
    int device = -1;
    TableSlot slots[kSlots];
    int next = 0;

    ~TableRing() { release(); }
    void release()
    {
        if (device < 0)
            return;
        for (auto& s : slots) {
            if (s.host) (void)hipHostFree(s.host);
            if (s.dev) (void)hipFree(s.dev);
            if (s.done) (void)hipEventDestroy(s.done);
            s = TableSlot{};
        }
        device = -1;
        next = 0;
    }
    hipError_t acquire(size_t bytes, TableSlot** out)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess)
            return e;
        if (dev != device) {
            release();
            device = dev;
        }
        TableSlot& s = slots[next];
        next = (next + 1) % kSlots;
        if (s.pending) {
            e = hipEventSynchronize(s.done);
            if (e != hipSuccess)
                return e;
            s.pending = false;
        }
        if (s.done == nullptr) {
            e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
            if (e != hipSuccess)
                return e;
        }
        if (s.cap < bytes) {
            if (s.host) (void)hipHostFree(s.host);
            if (s.dev) (void)hipFree(s.dev);
            s.host = s.dev = nullptr;
            s.cap = 0;
            const size_t want = bytes + bytes / 2 + 4096;
            e = hipHostMalloc(&s.host, want, hipHostMallocDefault);
            if (e == hipSuccess)
                e = hipMalloc(&s.dev, want);
            if (e != hipSuccess)
                return e;
            s.cap = want;
        }
        *out = &s;
        return hipSuccess;
    }
};

thread_local TableRing g_ring;

}  // namespace

void dxtlt_host::release_batch_thread_tables() { g_ring.release(); }

extern "C" int32_t dxtlt_transform_batch_device(const DxtltBatchItem* items, size_t count, void* hip_stream)
{
    if (count == 0)
        return kOk;
    if (items == nullptr)
        return fail(kInvalidArgument, "NULL item array with count > 0");
    // validate everything first: a batch is enqueued whole or not at all
    for (size_t i = 0; i < count; ++i) {
        const DxtltBatchItem& it = items[i];
        if (it.format < 1 || it.format > 3)
            return fail(kInvalidArgument, "batch item: format must be 1 (BC1), 2 (BC2) or 3 (BC3)");
        if (it.len % (it.format == 1 ? 8u : 16u) != 0)
            return fail(kInvalidLength, "batch item: len is not a multiple of the block size");
        if (it.decorrelation_mode > 3)
            return fail(kInvalidArgument, "batch item: decorrelation_mode must be 0..3");
        if (it.len > 0 && (it.d_input == nullptr || it.d_output == nullptr))
            return fail(kInvalidArgument, "batch item: NULL device buffer with len > 0");
    }
    hipStream_t user = static_cast<hipStream_t>(hip_stream);

    // one table per (format, direction) group; groups are launched one after the other on the caller's stream
    struct Group {
        std::vector<BatchEntry> entries;
        uint32_t wgs = 0;
    };
    Group groups[6];
    for (size_t i = 0; i < count; ++i) {
        const DxtltBatchItem& it = items[i];
        if (it.len == 0)
            continue;
        if (it.len >= (size_t(64) << 30))
            return fail(kInvalidArgument, "batch item of 64 GiB or more: use the single-buffer entry point");
        Group& g = groups[(it.format - 1) * 2 + (it.inverse ? 1 : 0)];
        BatchEntry e{};
        e.src = static_cast<const uint8_t*>(it.d_input);
        e.dst = static_cast<uint8_t*>(it.d_output);
        e.blocks = it.len / (it.format == 1 ? 8u : 16u);
        e.variant = it.decorrelation_mode;
        e.split_alpha = it.format == 3 && it.split_alpha_endpoints ? 1 : 0;
        e.split_colour = it.split_colour_endpoints ? 1 : 0;
        g.wgs = (g.wgs + 7u) & ~7u;   // first workgroup on XCD 0: the kernel orders each buffer's tiles per XCD
        e.first_wg = g.wgs;
        const uint32_t wgs = dxtlt::plan_batch_entry((dxtlt::Format)it.format, it.inverse != 0, e);
        // one launch holds fewer than 2^32 threads = 2^24 workgroups of 256 (64 GiB of blocks per format and direction)
        if ((uint64_t)g.wgs + wgs > 0xFFFFFFull)
            return fail(kInvalidArgument, "batch too large for one launch (64 GiB or more of one format and direction)");
        g.wgs += wgs;
        g.entries.push_back(e);
    }

    for (int gi = 0; gi < 6; ++gi) {
        Group& g = groups[gi];
        if (g.entries.empty())
            continue;
        const size_t n = g.entries.size();
        const size_t coarse_n = ((size_t)g.wgs + 63) / 64;
        const size_t entry_bytes = (n * sizeof(BatchEntry) + 15) & ~(size_t)15;
        const size_t bytes = entry_bytes + coarse_n * sizeof(uint32_t);
        TableSlot* slot = nullptr;
        hipError_t e = g_ring.acquire(bytes, &slot);
        if (e != hipSuccess)
            return fail(kDevice, "batch table staging", e);
        std::memcpy(slot->host, g.entries.data(), n * sizeof(BatchEntry));
        uint32_t* coarse = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(slot->host) + entry_bytes);
        size_t cur = 0;
        for (size_t k = 0; k < coarse_n; ++k) {
            const uint32_t wg = (uint32_t)(k * 64);
            while (cur + 1 < n && g.entries[cur + 1].first_wg <= wg)
                ++cur;
            coarse[k] = (uint32_t)cur;
        }
        e = hipMemcpyAsync(slot->dev, slot->host, bytes, hipMemcpyHostToDevice, user);
        if (e == hipSuccess)
            e = dxtlt::launch_batch((dxtlt::Format)(gi / 2 + 1), (gi & 1) != 0, static_cast<const BatchEntry*>(slot->dev),
                                    reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(slot->dev) + entry_bytes),
                                    (uint32_t)n, g.wgs, user);
        // the event marks both the copy and the kernel that reads the device table
        hipError_t ev = hipEventRecord(slot->done, user);
        slot->pending = ev == hipSuccess;
        if (e != hipSuccess)
            return fail(kDevice, "batch table copy / launch", e);
        if (ev != hipSuccess)
            return fail(kDevice, "batch event", ev);
    }
    return kOk;
}
