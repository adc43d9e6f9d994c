// bc7_sharded.cpp -- single-process multi-GPU sharding of the BC7 granule-sorted field split:
// contiguous block ranges that start on sort granules, one host thread per shard, no collective and -- unlike version
// 0 of the format -- no exchange of counters either: a granule's output depends on that granule alone, so a shard
// transformed as a stand-alone buffer holds exactly its slice of each of the eight main streams, packed, and (the last
// shard only) the tail part behind them.  The "host concat" is eight copies per shard (nine for the last one).
// dxtlt_bc7_shard_pieces is that placement table on its own (pure host code; the CPU tests drive it with the oracle).
#include <hip/hip_runtime_api.h>

#include <string>
#include <algorithm>
#include <exception>
#include <thread>
#include <vector>

#include "../../include/dxtlt_bc7.h"
#include "../../include/dxtlt_gfx950.h"
#include "bc7_fields.h"
#include "bc7_launch.h"
#include "host_common.h"

namespace {

using namespace dxtlt_host;

constexpr uint64_t kT = dxtlt::bc7::kGranule;
constexpr uint64_t kOff[8] = {0, 8, 10, 11, 12, 13, 14, 15};
constexpr uint64_t kWidth[8] = {8, 2, 1, 1, 1, 1, 1, 1};

struct Piece {
    uint64_t global_off, local_off, bytes;
};

// pieces 0..7: the shard's slice of every main stream; piece 8: the tail part (last shard only, else empty)
int32_t pieces_for(uint64_t total_blocks, uint64_t first_block, uint64_t num_blocks, Piece (&p)[9])
{
    const uint64_t main_total = total_blocks - total_blocks % kT;
    if (first_block % kT != 0 || first_block > total_blocks || num_blocks > total_blocks - first_block ||
        ((first_block + num_blocks) % kT != 0 && first_block + num_blocks != total_blocks))
        return fail(kInvalidArgument, "BC7 shard: a shard starts on a sort granule (1024 blocks) and ends on one or at the end");
    const uint64_t end = first_block + num_blocks;
    const uint64_t main_count = first_block >= main_total ? 0 : (end > main_total ? main_total : end) - first_block;
    for (int s = 0; s < 8; ++s)
        p[s] = {kOff[s] * main_total + kWidth[s] * first_block, kOff[s] * main_count, kWidth[s] * main_count};
    const uint64_t tail = num_blocks - main_count;   // non-zero only when the shard reaches the end
    p[8] = {16 * main_total, 16 * main_count, 16 * tail};
    return kOk;
}

struct Shard {
    uint64_t first, count;
};

std::vector<Shard> plan(uint64_t total_blocks, int shards)
{
    // equal shares of whole granules; the last shard takes the remainder and the tail part
    std::vector<Shard> p((size_t)shards);
    uint64_t share = total_blocks / (uint64_t)shards;
    share -= share % kT;
    uint64_t at = 0;
    for (int i = 0; i < shards; ++i) {
        const uint64_t n = i == shards - 1 ? total_blocks - at : share;
        p[(size_t)i] = {at, n};
        at += n;
    }
    return p;
}

int32_t shard_worker(int dev, bool inverse, const uint8_t* in, uint8_t* out, uint64_t total_blocks, Shard sh)
{
    if (sh.count == 0)
        return kOk;
    Piece pc[9];
    if (int32_t rc = pieces_for(total_blocks, sh.first, sh.count, pc); rc != kOk)
        return rc;
    const size_t bytes = (size_t)sh.count * 16;
    hipError_t e = hipSetDevice(dev);
    if (e != hipSuccess)
        return fail(kDevice, "hipSetDevice", e);
    // stream and buffers of this device are kept across calls (host_common.h): a fresh stream and two hipMallocs of the
    // shard's size per call cost more than the transfers they serve
    ShardBuffers sb{};
    if (int32_t rc = acquire_shard_buffers(dev, bytes, &sb); rc != kOk)
        return rc;
    hipStream_t st = sb.stream;
    void *d_a = sb.a, *d_b = sb.b;
    auto done = [&](int32_t code) {
        (void)hipStreamSynchronize(st);   // nothing may still target the buffers the next call will reuse
        release_shard_buffers(sb);
        return code;
    };
    // a large shard: its whole granules through the chunked pipeline (upload | kernel | eight per-stream downloads of
    // consecutive chunks overlap), its share of the array's tail part -- the last shard only -- afterwards, one shot
    const uint64_t main_total = total_blocks - total_blocks % 1024;
    const uint64_t in_main = sh.first >= main_total ? 0 : std::min<uint64_t>(sh.first + sh.count, main_total) - sh.first;
    int32_t prc = kOk;
    if (in_main != 0 && pipelined_bc7_shard(sb, dev, inverse, in, out, main_total, sh.first, in_main, &prc)) {
        if (prc != kOk)
            return done(prc);
        const uint64_t tail = sh.count - in_main;   // blocks of the array's tail part
        if (tail != 0) {
            const uint8_t* src = in + main_total * 16;
            uint8_t* dst = out + main_total * 16;
            e = hipMemcpyAsync(d_a, src, (size_t)tail * 16, hipMemcpyHostToDevice, st);
            if (e == hipSuccess)
                e = dxtlt::bc7::launch(inverse, d_a, d_b, tail, st);
            if (e == hipSuccess)
                e = hipMemcpyAsync(dst, d_b, (size_t)tail * 16, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess)
                e = hipStreamSynchronize(st);
            if (e != hipSuccess)
                return done(fail(kDevice, "BC7 shard tail part", e));
        }
        return done(kOk);
    }
    if (!inverse) {
        e = hipMemcpyAsync(d_a, in + sh.first * 16, bytes, hipMemcpyHostToDevice, st);
        if (e == hipSuccess)
            e = dxtlt::bc7::launch(false, d_a, d_b, sh.count, st);   // stand-alone: its streams are the shard's slices, packed
        for (int p = 0; p < 9 && e == hipSuccess; ++p)
            if (pc[p].bytes)
                e = hipMemcpyAsync(out + pc[p].global_off, (const uint8_t*)d_b + pc[p].local_off, (size_t)pc[p].bytes,
                                   hipMemcpyDeviceToHost, st);
    } else {
        for (int p = 0; p < 9 && e == hipSuccess; ++p)
            if (pc[p].bytes)
                e = hipMemcpyAsync((uint8_t*)d_a + pc[p].local_off, in + pc[p].global_off, (size_t)pc[p].bytes,
                                   hipMemcpyHostToDevice, st);
        if (e == hipSuccess)
            e = dxtlt::bc7::launch(true, d_a, d_b, sh.count, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(out + sh.first * 16, d_b, bytes, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess)
        e = hipStreamSynchronize(st);
    if (e != hipSuccess)
        return done(fail(kDevice, "BC7 shard copy/launch", e));
    return done(kOk);
}

int32_t sharded(bool inverse, const uint8_t* in, uint8_t* out, size_t len, int32_t num_shards)
{
    if (len % 16 != 0)
        return fail(kInvalidLength, "len is not a multiple of 16 (BC7 block size)");
    if (len == 0)
        return kOk;
    if (in == nullptr || out == nullptr)
        return fail(kInvalidArgument, "NULL buffer with len > 0");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(kNoDevice, "no HIP device available (this library has no CPU fallback)", e);
    const uint64_t total = len / 16;
    int shards = num_shards <= 0 ? count : (num_shards > 64 ? 64 : num_shards);
    const uint64_t granules = (total + kT - 1) / kT;
    if ((uint64_t)shards > granules)
        shards = (int)granules;
    int prev = 0;
    (void)hipGetDevice(&prev);
    const std::vector<Shard> pl = plan(total, shards);
    std::vector<int32_t> codes((size_t)shards, kOk);
    std::vector<std::string> msgs((size_t)shards);
    std::vector<std::thread> threads;
    bool spawn_failed = false;
    dxtlt_host::init_runtime_for_devices(shards < count ? shards : count);   // on the caller's thread, before any worker binds itself
    for (int s = 0; s < shards && !spawn_failed; ++s) {
        try {
            threads.emplace_back([&, s] {
                (void)dxtlt_host::bind_this_thread_near_device(s % count);   // the library's own thread: next to its device
                codes[(size_t)s] = shard_worker(s % count, inverse, in, out, total, pl[(size_t)s]);
                if (codes[(size_t)s] != kOk)
                    msgs[(size_t)s] = dxtlt_last_error();
            });
        } catch (const std::exception&) {
            spawn_failed = true;   // EAGAIN under a process limit: join what was started, then report
        }
    }
    for (auto& t : threads)
        t.join();
    dxtlt_host::trim_idle_shard_buffers();
    (void)hipSetDevice(prev);
    if (spawn_failed)
        return fail(kAllocation, "could not start a shard worker thread");   // a host resource ran out, as in every other spawn path
    for (int s = 0; s < shards; ++s)
        if (codes[(size_t)s] != kOk)
            return fail(codes[(size_t)s], msgs[(size_t)s].c_str());
    return kOk;
}

}  // namespace

extern "C" {

int32_t dxtlt_transform_bc7_sharded(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len, int32_t num_shards)
{
    return sharded(false, input_ptr, output_ptr, len, num_shards);
}

int32_t dxtlt_untransform_bc7_sharded(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len, int32_t num_shards)
{
    return sharded(true, input_ptr, output_ptr, len, num_shards);
}

int32_t dxtlt_bc7_shard_pieces(uint64_t total_blocks, uint64_t first_block, uint64_t num_blocks, uint64_t* global_off,
                               uint64_t* local_off, uint64_t* bytes)
{
    if (global_off == nullptr || local_off == nullptr || bytes == nullptr)
        return fail(kInvalidArgument, "dxtlt_bc7_shard_pieces: NULL output array");
    Piece pc[9];
    if (int32_t rc = pieces_for(total_blocks, first_block, num_blocks, pc); rc != kOk)
        return rc;
    for (int p = 0; p < 9; ++p) {
        global_off[p] = pc[p].global_off;
        local_off[p] = pc[p].local_off;
        bytes[p] = pc[p].bytes;
    }
    return kOk;
}

uint32_t dxtlt_bc7_sort_granule(void) { return (uint32_t)kT; }

}  // extern "C"
