// batch_api.cpp -- dxtlt_transform_batch_device: many device-resident buffers in one call.
//
// The reference transforms file after file (its CLI fans files out over rayon workers); a texture is typically
// 0.1-20 MiB.  On MI355X one such buffer cannot fill the chip (a 1 MiB BC1 texture is 256 workgroups for 256 CUs) and
// a launch costs the host ~5 us, longer than the kernel runs: measured 186 GiB/s for 1024 x 1 MiB through one call per
// buffer, even when spread over eight streams (profiles/r01_x).  So a batch becomes ONE launch per (format, direction
// and settings combination) present in it: the host lays the buffers' workgroups end to end in a table (96 bytes per buffer
// plus a two-level workgroup -> buffer index: bcn_launch.h), sends the tables of all its launches to the device in ONE upload on
// the caller's stream (a small kernel reads the mapped pinned slot: no copy-engine hand-over in front of the batch kernel) and
// launches batch_kernel (batch_kernels.hip), in which every workgroup looks its buffer up and runs one tile -- aligned, halo
// or shifted, as the single-buffer call would choose for that buffer -- or the buffer's edge tile.  Asynchronous and ordered
// like a single call on the caller's stream.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/dxtlt_bc7.h"
#include "../../include/dxtlt_gfx950.h"
#include "bc7_launch.h"
#include "bcn_launch.h"
#include "host_common.h"

namespace {

using namespace dxtlt_host;
using dxtlt::BatchEntry;

// Table staging: a ring of pinned host buffers with device twins.  A slot is reused only after the copy and the kernels that last
// read it have finished (its event).  A call takes exactly ONE slot -- the tables of its two BC7 launches and of all its BC1-3
// groups share one staged buffer and one upload -- so it never waits for its own work, and its acquire blocks the host only when
// kSlots earlier calls of this thread are all still in flight.  (Until round 6 a call with BC7 forward, BC7 inverse and BC1-3
// items took three slots: the next such call's second acquire landed on a slot the previous call had left pending and waited in
// hipEventSynchronize for that call's kernels -- an "asynchronous" call that host-blocked with a single earlier call in flight,
// which dxtlt_transform_batch_host hit on every chunk.  Sixteen mixed calls back to back: enqueued in 0.66 ms instead of 1.0 ms, finished
// 20 % sooner: profiles/r06_batch_one_slot.txt.)
constexpr int kSlots = 4;

struct TableSlot {
    void* host = nullptr;
    void* host_mapped = nullptr;   // the device-side address of `host`
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
            e = hipHostMalloc(&s.host, want, hipHostMallocMapped);
            if (e == hipSuccess)
                e = hipHostGetDevicePointer(&s.host_mapped, s.host, 0);
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

// DXTLT_BATCH_TABLE_COPY=1: the table travels by hipMemcpyAsync (the first version; kept for the comparison in
// profiles/r02_m_batch_kernel.txt)
const bool kTableByCopyEngine = dxtlt::experiment_env("DXTLT_BATCH_TABLE_COPY") != nullptr && dxtlt::experiment_env("DXTLT_BATCH_TABLE_COPY")[0] == '1';

hipError_t upload_table(TableSlot* slot, size_t bytes, hipStream_t stream)
{
    if (kTableByCopyEngine)
        return hipMemcpyAsync(slot->dev, slot->host, bytes, hipMemcpyHostToDevice, stream);
    return dxtlt::launch_table_upload(slot->host_mapped, slot->dev, bytes, stream);
}

}  // namespace

extern "C" int32_t dxtlt_transform_batch_device(const DxtltBatchItem* items, size_t count, void* hip_stream)
{
    if (count == 0)
        return kOk;
    if (items == nullptr)
        return fail(kInvalidArgument, "NULL item array with count > 0");
    // validate everything first: a batch is enqueued whole or not at all
    for (size_t i = 0; i < count; ++i) {
        const DxtltBatchItem& it = items[i];
        if ((it.format < 1 || it.format > 3) && it.format != 7)
            return fail(kInvalidArgument, "batch item: format must be 1 (BC1), 2 (BC2), 3 (BC3) or 7 (BC7, this build's own format)");
        if (it.len % (it.format == 1 ? 8u : 16u) != 0)
            return fail(kInvalidLength, "batch item: len is not a multiple of the block size");
        if (it.decorrelation_mode > 3 && it.format != 7)
            return fail(kInvalidArgument, "batch item: decorrelation_mode must be 0..3");
        if (it.len > 0 && (it.d_input == nullptr || it.d_output == nullptr))
            return fail(kInvalidArgument, "batch item: NULL device buffer with len > 0");
    }
    // one launch holds fewer than 2^32 threads = 2^24 workgroups of 256 lanes: for BC7 that is 256 GiB of granules per
    // direction.  Checked here, before anything is enqueued (a batch goes out whole or not at all).
    {
        uint64_t granules[2] = {0, 0};
        for (size_t i = 0; i < count; ++i)
            if (items[i].format == 7)
                granules[items[i].inverse ? 1 : 0] += items[i].len / 16 / 1024;
        if (granules[0] > 0xFFFFFFull || granules[1] > 0xFFFFFFull)
            return fail(kInvalidArgument, "batch too large for one launch (256 GiB or more of BC7 in one direction)");
    }
    hipStream_t user = static_cast<hipStream_t>(hip_stream);

    // one table per (format, direction, settings) group, planned (and checked against the launch limit) before anything is
    // enqueued; the groups are launched one after the other on the caller's stream, behind the BC7 launches.  The kernel is
    // instantiated per settings combination (no run-time switch in front of every tile): a corpus usually has one or two.
    struct Group {
        std::vector<BatchEntry> entries;
        uint32_t wgs = 0;
        uint32_t uniform_wgs = 0;   // workgroups per buffer while all buffers own the same number
        bool uniform = true;
    };
    constexpr int kGroups = 3 * 2 * 16;   // format, direction, variant x split_alpha x split_colour
    std::vector<Group> groups(kGroups);
    auto group_settings = [](int gi) {
        dxtlt::Settings s{};
        s.variant = (gi >> 2) & 3;
        s.split_alpha = ((gi >> 1) & 1) != 0;
        s.split_colour = (gi & 1) != 0;
        return s;
    };
    std::vector<size_t> singles;   // items the batch kernel does not take (plan_batch_entry): launched alone, behind the batches
    for (size_t i = 0; i < count; ++i) {
        const DxtltBatchItem& it = items[i];
        if (it.len == 0 || it.format == 7)
            continue;
        if (it.len >= (size_t(64) << 30))
            return fail(kInvalidArgument, "batch item of 64 GiB or more: use the single-buffer entry point");
        const int sa = it.format == 3 && it.split_alpha_endpoints ? 1 : 0, sc = it.split_colour_endpoints ? 1 : 0;
        const int gi = (((it.format - 1) * 2 + (it.inverse ? 1 : 0)) << 4) | (it.decorrelation_mode << 2) | (sa << 1) | sc;
        Group& g = groups[gi];
        BatchEntry e{};
        e.src = static_cast<const uint8_t*>(it.d_input);
        e.dst = static_cast<uint8_t*>(it.d_output);
        e.blocks = it.len / (it.format == 1 ? 8u : 16u);
        e.first_wg = g.wgs;
        const uint32_t wgs = dxtlt::plan_batch_entry((dxtlt::Format)it.format, it.inverse != 0, group_settings(gi), e);
        if (wgs == 0xFFFFFFFFu) {
            singles.push_back(i);
            continue;
        }
        // one launch holds fewer than 2^32 threads = 2^24 workgroups of 256 (64 GiB of blocks per format, direction and settings)
        if ((uint64_t)g.wgs + wgs > 0xFFFFFFull)
            return fail(kInvalidArgument, "batch too large for one launch (64 GiB or more of one format, direction and settings)");
        g.wgs += wgs;
        g.entries.push_back(e);
        // buffers of one size: every entry owns the same number of workgroups -- the kernel then divides instead of looking
        // the entry up
        if (g.entries.size() == 1)
            g.uniform_wgs = wgs;
        else if (wgs != g.uniform_wgs)
            g.uniform = false;
    }
    static const bool no_uniform = dxtlt::experiment_env("DXTLT_BATCH_NO_UNIFORM") != nullptr;   // A/B switches of the experiments build (tools/batch_kernel_probe.py)
    static const bool no_strided = dxtlt::experiment_env("DXTLT_BATCH_NO_STRIDED") != nullptr;

    // BC7 items (format 7; no settings): their granules in one launch per direction, their tail parts in a second one.  Planned
    // here, staged and launched below with everything else.
    struct Bc7Plan {
        std::vector<dxtlt::bc7::BatchEntry> entries, tails;
        std::vector<uint32_t> coarse;
        uint64_t wgs = 0;
        size_t at = 0, entry_bytes = 0, tail_bytes = 0, bytes = 0;
    };
    Bc7Plan bc7_plans[2];
    size_t table_bytes = 0;
    for (int inverse = 0; inverse < 2; ++inverse) {
        Bc7Plan& p = bc7_plans[inverse];
        for (size_t i = 0; i < count; ++i) {
            const DxtltBatchItem& it = items[i];
            if (it.format != 7 || it.len == 0 || (it.inverse != 0) != (inverse != 0))
                continue;
            const uint64_t blocks = it.len / 16, tail = blocks % 1024, main = blocks - tail;
            const uint8_t* src = static_cast<const uint8_t*>(it.d_input);
            uint8_t* dst = static_cast<uint8_t*>(it.d_output);
            if (main != 0) {
                p.entries.push_back({src, dst, main, (uint32_t)p.wgs, 0});
                p.wgs += main / 1024;
            }
            if (tail != 0)
                p.tails.push_back({src + main * 16, dst + main * 16, 0, 0, (uint32_t)tail});
        }
        if (p.entries.empty() && p.tails.empty())
            continue;
        p.coarse.resize(((size_t)p.wgs + 63) / 64);
        size_t cur = 0;
        for (size_t k = 0; k < p.coarse.size(); ++k) {
            while (cur + 1 < p.entries.size() && p.entries[cur + 1].first_wg <= (uint32_t)(k * 64))
                ++cur;
            p.coarse[k] = (uint32_t)cur;
        }
        p.entry_bytes = p.entries.size() * sizeof(dxtlt::bc7::BatchEntry);
        p.tail_bytes = p.tails.size() * sizeof(dxtlt::bc7::BatchEntry);
        p.bytes = (p.entry_bytes + p.tail_bytes + p.coarse.size() * sizeof(uint32_t) + 15) & ~(size_t)15;
        p.at = table_bytes;
        table_bytes += p.bytes;
    }

    // Every table of the call -- the two BC7 tables above, then every group's (entries, then the workgroup index) -- goes into ONE
    // staged buffer and ONE upload, at 16-byte aligned offsets: one ring slot per call.  (One slot per group made a call with many
    // settings combinations -- rare in a corpus, routine in the fuzz: up to 96 groups -- wait in hipEventSynchronize for kernels this
    // same call had enqueued: the "asynchronous" call then drained its own work, and would deadlock under a caller whose stream is
    // gated on an event recorded after the call returns.  Three slots per call -- BC7 forward, BC7 inverse, the groups -- still let
    // the NEXT mixed call wait for this one's kernels; see the ring's comment.)
    struct Placed {
        int gi;
        size_t at, entry_bytes;
    };
    std::vector<Placed> placed;
    for (int gi = 0; gi < kGroups; ++gi) {
        const Group& g = groups[gi];
        if (g.entries.empty())
            continue;
        const size_t entry_bytes = (g.entries.size() * sizeof(BatchEntry) + 15) & ~(size_t)15;
        placed.push_back({gi, table_bytes, entry_bytes});
        table_bytes += entry_bytes + dxtlt::batch_index_bytes(g.wgs);
    }
    if (table_bytes != 0) {
        TableSlot* slot = nullptr;
        hipError_t e = g_ring.acquire(table_bytes, &slot);
        if (e != hipSuccess)
            return fail(kDevice, "batch table staging", e);
        for (const Bc7Plan& p : bc7_plans) {
            if (p.bytes == 0)
                continue;
            uint8_t* h = static_cast<uint8_t*>(slot->host) + p.at;
            if (p.entry_bytes) std::memcpy(h, p.entries.data(), p.entry_bytes);
            if (p.tail_bytes) std::memcpy(h + p.entry_bytes, p.tails.data(), p.tail_bytes);
            if (!p.coarse.empty()) std::memcpy(h + p.entry_bytes + p.tail_bytes, p.coarse.data(), p.coarse.size() * sizeof(uint32_t));
        }
        std::vector<bool> wide(placed.size());
        for (size_t k = 0; k < placed.size(); ++k) {
            const Group& g = groups[placed[k].gi];
            uint8_t* h = static_cast<uint8_t*>(slot->host) + placed[k].at;
            std::memcpy(h, g.entries.data(), g.entries.size() * sizeof(BatchEntry));
            wide[k] = dxtlt::build_batch_index(g.entries.data(), g.entries.size(), g.wgs, h + placed[k].entry_bytes);
        }
        e = upload_table(slot, table_bytes, user);
        const char* what = "batch table copy / launch";
        for (int inverse = 0; inverse < 2 && e == hipSuccess; ++inverse) {
            const Bc7Plan& p = bc7_plans[inverse];
            if (p.bytes == 0)
                continue;
            const uint8_t* d = static_cast<const uint8_t*>(slot->dev) + p.at;
            e = dxtlt::bc7::launch_batch(inverse != 0, reinterpret_cast<const dxtlt::bc7::BatchEntry*>(d),
                                         reinterpret_cast<const uint32_t*>(d + p.entry_bytes + p.tail_bytes), (uint32_t)p.entries.size(),
                                         (uint32_t)p.wgs, reinterpret_cast<const dxtlt::bc7::BatchEntry*>(d + p.entry_bytes),
                                         (uint32_t)p.tails.size(), user);
            if (e != hipSuccess)
                what = "BC7 batch table copy / launch";
        }
        for (size_t k = 0; k < placed.size() && e == hipSuccess; ++k) {
            const int gi = placed[k].gi;
            Group& g = groups[gi];
            const size_t n = g.entries.size();
            const bool uniform = g.uniform && !no_uniform;
            // a regular array of buffers: one size and tile form, pointers a constant stride apart
            bool strided = uniform && !no_strided;   // (one buffer is a regular array too)
            const int64_t src_stride = n >= 2 ? (int64_t)(g.entries[1].src - g.entries[0].src) : 0;
            const int64_t dst_stride = n >= 2 ? (int64_t)(g.entries[1].dst - g.entries[0].dst) : 0;
            for (size_t i = 1; i < n && strided; ++i) {
                const BatchEntry &a = g.entries[0], &b = g.entries[i];
                strided = b.blocks == a.blocks && b.full_tiles == a.full_tiles && b.form == a.form && b.halo_vecs == a.halo_vecs &&
                          std::memcmp(b.shift, a.shift, sizeof a.shift) == 0 && std::memcmp(b.gbase, a.gbase, sizeof a.gbase) == 0 &&
                          b.src == a.src + (int64_t)i * src_stride && b.dst == a.dst + (int64_t)i * dst_stride;
            }
            const uint8_t* d = static_cast<const uint8_t*>(slot->dev) + placed[k].at;
            e = dxtlt::launch_batch((dxtlt::Format)((gi >> 5) + 1), ((gi >> 4) & 1) != 0, group_settings(gi),
                                    reinterpret_cast<const BatchEntry*>(d), d + placed[k].entry_bytes, (uint32_t)n, g.wgs,
                                    uniform ? g.uniform_wgs : 0, wide[k], user, strided ? &g.entries[0] : nullptr, src_stride, dst_stride);
        }
        // the event marks both the copy and the kernels that read the device table
        hipError_t ev = hipEventRecord(slot->done, user);
        slot->pending = ev == hipSuccess;
        if (e != hipSuccess)
            return fail(kDevice, what, e);
        if (ev != hipSuccess)
            return fail(kDevice, "batch event", ev);
    }
    // buffers whose transformed-side pointer is not 8-byte aligned: the single-buffer call, one launch each
    for (size_t i : singles) {
        const DxtltBatchItem& it = items[i];
        dxtlt::Settings s{};
        s.variant = it.decorrelation_mode;
        s.split_alpha = it.format == 3 && it.split_alpha_endpoints;
        s.split_colour = it.split_colour_endpoints != 0;
        const uint64_t blocks = it.len / (it.format == 1 ? 8u : 16u);
        const hipError_t e = dxtlt::launch_transform((dxtlt::Format)it.format, it.inverse != 0, s, it.d_input, it.d_output,
                                                     dxtlt::Range{blocks, 0, blocks}, user);
        if (e != hipSuccess)
            return fail(kDevice, "batch item launch", e);
    }
    return kOk;
}

// Test hook (no device needed, nothing is dereferenced): plans buffers of one format, direction and settings exactly as
// dxtlt_transform_batch_device does for one launch -- plan_batch_entry per buffer, then build_batch_index -- and hands back the
// planned entries and the workgroup -> entry index, so that the host logic can be checked on a machine without a GPU
// (tests/test_batch_plan.py).  Returns the launch's workgroups; 0xFFFFFFFF when `index_capacity` is too small or a buffer is
// one the batch kernel does not take (dxtlt_transform_batch_device launches those alone).
extern "C" uint32_t dxtlt_debug_plan_batch(int32_t format, int32_t inverse, int32_t variant, int32_t split_alpha, int32_t split_colour,
                                           const uint64_t* src_addresses, const uint64_t* dst_addresses, const uint64_t* blocks, size_t count,
                                           DxtltDebugPlannedEntry* entries_out, uint8_t* index_out, size_t index_capacity,
                                           uint32_t* index_is_wide_out)
{
    if (index_is_wide_out)
        *index_is_wide_out = 0;
    if (format < 1 || format > 3 || (count != 0 && (!src_addresses || !dst_addresses || !blocks || !entries_out)))
        return 0xFFFFFFFFu;
    dxtlt::Settings s{};
    s.variant = variant;
    s.split_alpha = format == 3 && split_alpha != 0;
    s.split_colour = split_colour != 0;
    std::vector<BatchEntry> entries;
    uint32_t total = 0;
    for (size_t i = 0; i < count; ++i) {
        BatchEntry e{};
        e.src = reinterpret_cast<const uint8_t*>(static_cast<uintptr_t>(src_addresses[i]));
        e.dst = reinterpret_cast<uint8_t*>(static_cast<uintptr_t>(dst_addresses[i]));
        e.blocks = blocks[i];
        e.first_wg = total;
        const uint32_t wgs = dxtlt::plan_batch_entry((dxtlt::Format)format, inverse != 0, s, e);
        if (wgs == 0xFFFFFFFFu || (uint64_t)total + wgs > 0xFFFFFFull)
            return 0xFFFFFFFFu;
        if (wgs == 0)
            e.end_wg = e.first_wg;
        total += wgs;
        DxtltDebugPlannedEntry& o = entries_out[i];
        o.first_wg = e.first_wg;
        o.end_wg = e.end_wg;
        o.full_tiles = e.full_tiles;
        o.form = e.form;
        o.halo_vecs = e.halo_vecs;
        std::memcpy(o.shift, e.shift, sizeof o.shift);
        std::memcpy(o.gbase, e.gbase, sizeof o.gbase);
        if (wgs != 0)
            entries.push_back(e);
    }
    if (total != 0) {
        if (index_out == nullptr || index_capacity < dxtlt::batch_index_bytes(total))
            return 0xFFFFFFFFu;
        const bool wide = dxtlt::build_batch_index(entries.data(), entries.size(), total, index_out);
        if (index_is_wide_out)
            *index_is_wide_out = wide ? 1 : 0;
    }
    return total;
}

// ---------------------------------------------------------------------------------------------------------------
// dxtlt_transform_batch_host: the same, for HOST buffers -- the reference's actual call pattern: one call per file,
// host pointers, textures of 0.1-20 MiB (tools/dxt-lossless-transform-cli/src/commands/transform/mod.rs:154-199).
// Through the single-buffer entry points every texture pays a PCIe round trip of its own (1 MiB: 149 us = 6.6 GiB/s,
// DESIGN.md section 5).  Here the batch is cut into chunks of about 64 MiB and every chunk takes five steps:
//     pack      a few host threads copy the chunk's buffers side by side into a PINNED arena
//     upload    one asynchronous copy of the whole chunk
//     kernels   one batch launch per (format, direction) present in the chunk
//     download  one asynchronous copy into a second pinned arena
//     unpack    host threads copy every result to its caller's buffer
// with two arenas per direction, so that packing chunk k + 1, moving chunk k and unpacking chunk k - 1 overlap and PCIe
// runs in both directions at once.  (A first version let several threads issue one small pageable copy per buffer
// straight from / to the callers' memory: 21 GiB/s at 1 MiB per buffer, 16 at 4 MiB -- the runtime pins pageable memory
// on the fly above 1 MiB; profiles/r02_d_batch_host.txt.)  Synchronous; every failure exit joins the threads and drains
// the streams before the arenas are released for reuse.
// ---------------------------------------------------------------------------------------------------------------
namespace {

// chunk size and copy threads per direction; DXTLT_BATCH_CHUNK_MIB / DXTLT_BATCH_THREADS override them (experiments)
size_t env_number(const char* name, size_t fallback)
{
    const char* v = std::getenv(name);
    const unsigned long long x = v ? std::strtoull(v, nullptr, 10) : 0;
    return x ? (size_t)x : fallback;
}
const size_t kHostBatchChunkBytes = env_number("DXTLT_BATCH_CHUNK_MIB", 64) << 20;
const int kCopyThreads = (int)std::min<size_t>(64, env_number("DXTLT_BATCH_THREADS", 6));

// pinned arenas of the calling thread: two per direction, grow-only
struct PinnedArenas {
    void* in[2] = {nullptr, nullptr};
    void* out[2] = {nullptr, nullptr};
    size_t cap = 0;
    ~PinnedArenas() { release(); }
    void release()
    {
        for (int i = 0; i < 2; ++i) {
            if (in[i]) (void)hipHostFree(in[i]);
            if (out[i]) (void)hipHostFree(out[i]);
            in[i] = out[i] = nullptr;
        }
        cap = 0;
    }
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap)
            return hipSuccess;
        release();
        for (int i = 0; i < 2; ++i) {
            hipError_t e = hipHostMalloc(&in[i], bytes, hipHostMallocDefault);
            if (e == hipSuccess)
                e = hipHostMalloc(&out[i], bytes, hipHostMallocDefault);
            if (e != hipSuccess) {
                release();
                return e;
            }
        }
        cap = bytes;
        return hipSuccess;
    }
};
thread_local PinnedArenas g_pinned;
#define g_pinned_in(c) (pinned_in[(c) & 1])
#define g_pinned_out(c) (pinned_out[(c) & 1])

struct HostBatchShared {
    std::mutex m;
    std::condition_variable cv;
    std::vector<int> packed;     // per chunk: packer threads that have finished it
    std::vector<int> unpacked;   // per chunk: unpacker threads that have finished it
    int uploaded_recorded = 0;   // chunks whose upload event has been recorded
    int kernels_issued = 0;      // chunks whose upload and kernels have been enqueued (ev_kernels recorded)
    int enqueued = 0;            // chunks whose download has been enqueued too (ev_downloaded recorded)
    bool failed = false;
    hipError_t error = hipSuccess;
};

}  // namespace

void dxtlt_host::release_batch_thread_tables()
{
    g_ring.release();
    g_pinned.release();
}

extern "C" int32_t dxtlt_transform_batch_host(const DxtltBatchItem* items, size_t count)
{
    if (count == 0)
        return kOk;
    if (items == nullptr)
        return fail(kInvalidArgument, "NULL item array with count > 0");
    uint64_t total = 0;
    for (size_t i = 0; i < count; ++i) {
        const DxtltBatchItem& it = items[i];
        if ((it.format < 1 || it.format > 3) && it.format != 7)
            return fail(kInvalidArgument, "batch item: format must be 1 (BC1), 2 (BC2), 3 (BC3) or 7 (BC7, this build's own format)");
        if (it.len % (it.format == 1 ? 8u : 16u) != 0)
            return fail(kInvalidLength, "batch item: len is not a multiple of the block size");
        if (it.decorrelation_mode > 3 && it.format != 7)
            return fail(kInvalidArgument, "batch item: decorrelation_mode must be 0..3");
        if (it.len > 0 && (it.d_input == nullptr || it.d_output == nullptr))
            return fail(kInvalidArgument, "batch item: NULL buffer with len > 0");
        if (it.len >= (size_t(4) << 30))
            return fail(kInvalidArgument, "batch item of 4 GiB or more: use the single-buffer entry point");
        total += (it.len + 255) & ~uint64_t(255);
    }
    if (total == 0)
        return kOk;

    // Items of two chunks or more do not belong in a chunk: the arenas (four pinned ones and two device slots per
    // direction, all of the largest chunk's size) would grow to the item's size and the item would move as one unpipelined
    // upload, kernel, download.  They go through the single-buffer host entry points instead, which run their own chunked
    // pipeline from 96 MiB up; what is left keeps every arena below 3 x the chunk size.
    const uint64_t big_item = 2 * (uint64_t)kHostBatchChunkBytes;
    std::vector<DxtltBatchItem> small;
    small.reserve(count);
    for (size_t i = 0; i < count; ++i) {
        const DxtltBatchItem& it = items[i];
        if (it.len < big_item) {
            small.push_back(it);
            continue;
        }
        int32_t rc;
        if (it.format == 7)
            rc = it.inverse ? dxtlt_untransform_bc7(static_cast<const uint8_t*>(it.d_input), static_cast<uint8_t*>(it.d_output), it.len)
                            : dxtlt_transform_bc7(static_cast<const uint8_t*>(it.d_input), static_cast<uint8_t*>(it.d_output), it.len);
        else
            rc = dxtlt_host::transform(it.format, it.inverse != 0, static_cast<const uint8_t*>(it.d_input),
                                       static_cast<uint8_t*>(it.d_output), it.len, it.decorrelation_mode,
                                       it.format == 3 && it.split_alpha_endpoints, it.split_colour_endpoints != 0);
        if (rc != kOk)
            return rc;
    }
    if (small.empty())
        return kOk;
    items = small.data();
    count = small.size();

    // chunks, and every item's 256-byte aligned slot inside its chunk (the same offset in all four arenas and on the device)
    std::vector<uint64_t> slot(count);
    std::vector<size_t> chunk_first;   // first item of every chunk, plus the end
    std::vector<uint64_t> chunk_bytes;
    {
        uint64_t in_chunk = 0;
        chunk_first.push_back(0);
        for (size_t i = 0; i < count; ++i) {
            const uint64_t padded = (items[i].len + 255) & ~uint64_t(255);
            // a chunk is closed BEFORE the item that would push it past the cap (so a chunk holds at most the cap, or one
            // item of less than two chunks): arena_bytes stays bounded whatever the item sizes
            if (in_chunk > 0 && in_chunk + padded > kHostBatchChunkBytes) {
                chunk_first.push_back(i);
                chunk_bytes.push_back(in_chunk);
                in_chunk = 0;
            }
            slot[i] = in_chunk;
            in_chunk += padded;
        }
        chunk_first.push_back(count);
        chunk_bytes.push_back(in_chunk);
    }
    const int nchunks = (int)chunk_bytes.size();
    const uint64_t arena_bytes = *std::max_element(chunk_bytes.begin(), chunk_bytes.end());

    // device: two chunk-sized slots per direction inside this thread's staging buffers
    void *d_in = nullptr, *d_out = nullptr;
    hipStream_t up = nullptr;
    if (int32_t rc = acquire_staging((size_t)(2 * arena_bytes), &d_in, &d_out, &up); rc != kOk)
        return rc;
    if (hipError_t e = g_pinned.reserve((size_t)arena_bytes); e != hipSuccess)
        return fail(kDevice, "hipHostMalloc(batch arenas)", e);
    // worker threads do not see this thread's thread_local arenas: hand them the pointers
    void* const pinned_in[2] = {g_pinned.in[0], g_pinned.in[1]};
    void* const pinned_out[2] = {g_pinned.out[0], g_pinned.out[1]};
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipStream_t down = nullptr;
    if (hipError_t e = hipStreamCreateWithFlags(&down, hipStreamNonBlocking); e != hipSuccess)
        return fail(kDevice, "hipStreamCreate(download)", e);

    std::vector<hipEvent_t> ev((size_t)nchunks * 3, nullptr);   // per chunk: uploaded, kernels done, downloaded
    for (auto& e : ev)
        if (hipError_t err = hipEventCreateWithFlags(&e, hipEventDisableTiming); err != hipSuccess) {
            for (auto& e2 : ev) if (e2) (void)hipEventDestroy(e2);
            (void)hipStreamDestroy(down);
            return fail(kDevice, "hipEventCreate", err);
        }
    auto ev_uploaded = [&](int c) { return ev[(size_t)c * 3]; };
    auto ev_kernels = [&](int c) { return ev[(size_t)c * 3 + 1]; };
    auto ev_downloaded = [&](int c) { return ev[(size_t)c * 3 + 2]; };

    HostBatchShared sh;
    sh.packed.assign((size_t)nchunks, 0);
    sh.unpacked.assign((size_t)nchunks, 0);
    auto set_failed = [&](hipError_t e) {
        std::lock_guard<std::mutex> lk(sh.m);
        if (!sh.failed) {
            sh.failed = true;
            sh.error = e;
        }
        sh.cv.notify_all();
    };

    // item i of a chunk belongs to copy thread (i - chunk_first) % kCopyThreads
    auto packer = [&](int tid) {
        hipError_t e = hipSetDevice(dev);
        for (int c = 0; c < nchunks && e == hipSuccess; ++c) {
            if (c < 2) {   // (chunks 0 and 1 wait for nothing -- but not for a call that is already being abandoned)
                std::lock_guard<std::mutex> lk(sh.m);
                if (sh.failed)
                    return;
            }
            if (c >= 2) {
                // arena c % 2 is free once chunk c - 2 has left it
                {
                    std::unique_lock<std::mutex> lk(sh.m);
                    sh.cv.wait(lk, [&] { return sh.uploaded_recorded > c - 2 || sh.failed; });
                    if (sh.failed)
                        return;
                }
                e = hipEventSynchronize(ev_uploaded(c - 2));
                if (e != hipSuccess)
                    break;
            }
            uint8_t* arena = static_cast<uint8_t*>(g_pinned_in(c));
            for (size_t i = chunk_first[(size_t)c] + (size_t)tid; i < chunk_first[(size_t)c + 1]; i += kCopyThreads)
                if (items[i].len)
                    std::memcpy(arena + slot[i], items[i].d_input, items[i].len);
            {
                std::lock_guard<std::mutex> lk(sh.m);
                sh.packed[(size_t)c]++;
            }
            sh.cv.notify_all();
        }
        if (e != hipSuccess)
            set_failed(e);
    };
    auto unpacker = [&](int tid) {
        hipError_t e = hipSetDevice(dev);
        for (int c = 0; c < nchunks && e == hipSuccess; ++c) {
            {
                std::unique_lock<std::mutex> lk(sh.m);
                sh.cv.wait(lk, [&] { return sh.enqueued > c || sh.failed; });
                if (sh.enqueued <= c)
                    return;
            }
            e = hipEventSynchronize(ev_downloaded(c));
            if (e != hipSuccess)
                break;
            const uint8_t* arena = static_cast<const uint8_t*>(g_pinned_out(c));
            for (size_t i = chunk_first[(size_t)c] + (size_t)tid; i < chunk_first[(size_t)c + 1]; i += kCopyThreads)
                if (items[i].len)
                    std::memcpy(items[i].d_output, arena + slot[i], items[i].len);
            {
                std::lock_guard<std::mutex> lk(sh.m);
                sh.unpacked[(size_t)c]++;
            }
            sh.cv.notify_all();
        }
        if (e != hipSuccess)
            set_failed(e);
    };

    // Thread creation can fail (EAGAIN under a process limit).  Every stage below waits on counts that only the full set
    // of threads reaches, so a partial set is told to leave (failed), joined, the streams drained, and the call reports it.
    std::vector<std::thread> threads;
    bool spawn_failed = false;
    auto spawn = [&](auto&& fn, auto... args) {
        if (spawn_failed)
            return;
        try {
            threads.emplace_back(fn, args...);
        } catch (const std::exception&) {
            spawn_failed = true;
        }
    };
    for (int t = 0; t < kCopyThreads; ++t) spawn(packer, t);
    for (int t = 0; t < kCopyThreads; ++t) spawn(unpacker, t);

    // The downloads are issued by a thread of their own.  A chunk's download may only be enqueued once the unpackers have
    // emptied its pinned output arena (chunk c - 2) -- a HOST-side condition -- and when this thread waited for that before
    // it issued the chunk's upload, only two chunks were ever between upload and unpack: the steady state was
    // (upload + kernels + download + unpack) / 2 per chunk, 33 GiB/s, not the slowest stage.
    auto downloader = [&] {
        hipError_t e = hipSetDevice(dev);
        for (int c = 0; c < nchunks && e == hipSuccess; ++c) {
            {
                std::unique_lock<std::mutex> lk(sh.m);
                sh.cv.wait(lk, [&] { return (sh.kernels_issued > c && (c < 2 || sh.unpacked[(size_t)c - 2] == kCopyThreads)) || sh.failed; });
                if (sh.failed)
                    return;
            }
            const uint8_t* dout = static_cast<const uint8_t*>(d_out) + (size_t)(c & 1) * arena_bytes;
            e = hipStreamWaitEvent(down, ev_kernels(c), 0);
            if (e == hipSuccess)
                e = hipMemcpyAsync(g_pinned_out(c), dout, (size_t)chunk_bytes[(size_t)c], hipMemcpyDeviceToHost, down);
            if (e == hipSuccess)
                e = hipEventRecord(ev_downloaded(c), down);
            if (e != hipSuccess)
                break;
            {
                std::lock_guard<std::mutex> lk(sh.m);
                sh.enqueued = c + 1;
            }
            sh.cv.notify_all();
        }
        if (e != hipSuccess)
            set_failed(e);
    };
    spawn(downloader);
    if (spawn_failed) {
        set_failed(hipErrorOutOfMemory);
        for (auto& t : threads)
            t.join();
        for (auto& e : ev) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(down);
        return fail(kAllocation, "could not start the batch copy threads");
    }

    // this thread: upload and kernels of every chunk, as soon as it is packed and its device slots are free
    int32_t rc = kOk;
    hipError_t err = hipSuccess;
    std::vector<DxtltBatchItem> dev_items;
    for (int c = 0; c < nchunks && rc == kOk && err == hipSuccess; ++c) {
        {
            std::unique_lock<std::mutex> lk(sh.m);
            // packed; and the download of chunk c - 2 has been ENQUEUED, so that its event exists to be waited for below
            sh.cv.wait(lk, [&] { return (sh.packed[(size_t)c] == kCopyThreads && (c < 2 || sh.enqueued > c - 2)) || sh.failed; });
            if (sh.failed)
                break;
        }
        uint8_t* di = static_cast<uint8_t*>(d_in) + (size_t)(c & 1) * arena_bytes;
        uint8_t* dout = static_cast<uint8_t*>(d_out) + (size_t)(c & 1) * arena_bytes;
        // device input slot c % 2 is rewritten here: wait for chunk c - 2's kernels (same stream: already ordered; kept
        // for clarity)
        if (c >= 2)
            err = hipStreamWaitEvent(up, ev_kernels(c - 2), 0);
        if (err == hipSuccess)
            err = hipMemcpyAsync(di, g_pinned_in(c), (size_t)chunk_bytes[(size_t)c], hipMemcpyHostToDevice, up);
        if (err == hipSuccess)
            err = hipEventRecord(ev_uploaded(c), up);
        {
            std::lock_guard<std::mutex> lk(sh.m);
            if (err == hipSuccess)
                sh.uploaded_recorded = c + 1;
        }
        sh.cv.notify_all();
        if (err != hipSuccess)
            break;
        // the kernels write device output slot c % 2: chunk c - 2's download must have read it
        if (c >= 2)
            err = hipStreamWaitEvent(up, ev_downloaded(c - 2), 0);
        if (err != hipSuccess)
            break;
        dev_items.clear();
        for (size_t i = chunk_first[(size_t)c]; i < chunk_first[(size_t)c + 1]; ++i) {
            DxtltBatchItem it = items[i];
            it.d_input = di + slot[i];
            it.d_output = dout + slot[i];
            dev_items.push_back(it);
        }
        rc = dxtlt_transform_batch_device(dev_items.data(), dev_items.size(), up);
        if (rc != kOk)
            break;
        err = hipEventRecord(ev_kernels(c), up);
        if (err != hipSuccess)
            break;
        {
            std::lock_guard<std::mutex> lk(sh.m);
            sh.kernels_issued = c + 1;
        }
        sh.cv.notify_all();
    }
    if (rc != kOk || err != hipSuccess) {
        std::lock_guard<std::mutex> lk(sh.m);
        sh.failed = true;
        if (err != hipSuccess && sh.error == hipSuccess)
            sh.error = err;
        sh.cv.notify_all();
    }
    for (auto& t : threads)
        t.join();
    (void)hipStreamSynchronize(up);
    (void)hipStreamSynchronize(down);
    for (auto& e : ev) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(down);
    if (rc != kOk)
        return rc;
    if (sh.failed)
        return fail(kDevice, "host batch copy", sh.error);
    return kOk;
}
