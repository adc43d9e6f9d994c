// bc7_sharded.cpp -- single-process multi-GPU sharding of the BC7 mode-split transform (last
// paragraph): contiguous block ranges, one host thread per shard, no collective.  Unlike BC1-3 the placement of a
// shard's output depends on the data of the shards before it, so the work runs in two phases around a host-side
// exchange of nine counters per shard:
//   forward   phase 1: every shard is transformed on its device as a stand-alone buffer (its local layout holds
//                      exactly this shard's piece of each of the 19 streams) and reports its nine mode counts;
//             host:    grand totals -> the 19 global stream bases; exclusive prefix over the shards -> where each
//                      shard's piece of every stream starts;
//             phase 2: 19 device-to-host copies per shard, straight to the pieces' final places.
//   inverse   phase 1: every shard counts the modes of its slice of the `first` stream;
//             host:    the same table;
//             phase 2: 19 host-to-device copies gather the shard's pieces into a local transformed buffer, the
//                      local inverse runs, one contiguous AoS copy comes back.
// dxtlt_bc7_shard_pieces is the placement table on its own (pure host code; the CPU tests drive it with the oracle).
#include <hip/hip_runtime_api.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/dxtlt_bc7.h"
#include "../../include/dxtlt_gfx950.h"
#include "bc7_launch.h"
#include "host_common.h"

namespace {

using namespace dxtlt_host;

constexpr uint64_t kHead[9] = {9, 9, 11, 11, 5, 7, 7, 11, 15};   // docs/BC7_FORMAT.md

// byte offset of piece p (0 = first, 1 + m = head_m, 10 + m = tail_m) in a buffer of n blocks with the given counts
uint64_t layout_offset(const uint64_t counts[9], uint64_t n, int p)
{
    if (p == 0)
        return 0;
    const int m = p <= 9 ? p - 1 : p - 10;
    uint64_t off = n;
    for (int mm = 0; mm < m; ++mm)
        off += counts[mm] * 15;
    if (p > 9)
        off += counts[m] * kHead[m];
    return off;
}

uint64_t piece_width(int p)
{
    if (p == 0)
        return 1;
    const int m = p <= 9 ? p - 1 : p - 10;
    return p <= 9 ? kHead[m] : 15 - kHead[m];
}

int32_t pieces_for(const uint64_t* counts, int32_t num_shards, int32_t shard, const uint64_t* first_block,
                   const uint64_t* num_blocks, uint64_t total_blocks, uint64_t* global_off, uint64_t* local_off,
                   uint64_t* bytes)
{
    if (counts == nullptr || first_block == nullptr || num_blocks == nullptr || global_off == nullptr ||
        local_off == nullptr || bytes == nullptr || num_shards <= 0 || shard < 0 || shard >= num_shards)
        return fail(kInvalidArgument, "dxtlt_bc7_shard_pieces: bad arguments");
    uint64_t totals[9] = {0}, before[9] = {0}, at = 0;
    for (int32_t s = 0; s < num_shards; ++s) {
        uint64_t sum = 0;
        for (int m = 0; m < 9; ++m) {
            const uint64_t c = counts[(size_t)s * 9 + m];
            totals[m] += c;
            if (s < shard)
                before[m] += c;
            sum += c;
        }
        if (sum != num_blocks[s] || first_block[s] != at)
            return fail(kInvalidArgument, "dxtlt_bc7_shard_pieces: counts / ranges do not describe a partition");
        at += num_blocks[s];
    }
    if (at != total_blocks)
        return fail(kInvalidArgument, "dxtlt_bc7_shard_pieces: shards do not cover total_blocks");
    const uint64_t* mine = counts + (size_t)shard * 9;
    for (int p = 0; p < 19; ++p) {
        const uint64_t w = piece_width(p);
        const int m = p == 0 ? 0 : (p <= 9 ? p - 1 : p - 10);
        const uint64_t skipped = p == 0 ? first_block[shard] : before[m];
        const uint64_t have = p == 0 ? num_blocks[shard] : mine[m];
        global_off[p] = layout_offset(totals, total_blocks, p) + skipped * w;
        local_off[p] = layout_offset(mine, num_blocks[shard], p);
        bytes[p] = have * w;
    }
    return kOk;
}

struct Shard {
    int dev = 0;
    uint64_t first = 0, count = 0;
    void *d_in = nullptr, *d_out = nullptr, *d_ws = nullptr;
    size_t ws_bytes = 0;
    hipStream_t st = nullptr;
    uint64_t counts[9] = {0};
    int32_t rc = kOk;
    std::string msg;
};

#define SHARD_TRY(expr, what)                                           \
    do {                                                                \
        hipError_t e_ = (expr);                                         \
        if (e_ != hipSuccess) {                                         \
            sh.rc = fail(kDevice, what, e_);                            \
            sh.msg = dxtlt_last_error();                                \
            return;                                                     \
        }                                                               \
    } while (0)

void shard_setup(Shard& sh)
{
    SHARD_TRY(hipSetDevice(sh.dev), "hipSetDevice");
    SHARD_TRY(hipStreamCreateWithFlags(&sh.st, hipStreamNonBlocking), "hipStreamCreate");
    const size_t bytes = (size_t)(sh.count * 16);
    sh.ws_bytes = dxtlt::bc7::workspace_bytes(sh.count);
    SHARD_TRY(hipMalloc(&sh.d_in, bytes), "hipMalloc(shard input)");
    SHARD_TRY(hipMalloc(&sh.d_out, bytes), "hipMalloc(shard output)");
    SHARD_TRY(hipMalloc(&sh.d_ws, sh.ws_bytes), "hipMalloc(shard workspace)");
}

void shard_release(Shard& sh)
{
    if (hipSetDevice(sh.dev) != hipSuccess)
        return;
    if (sh.d_in) (void)hipFree(sh.d_in);
    if (sh.d_out) (void)hipFree(sh.d_out);
    if (sh.d_ws) (void)hipFree(sh.d_ws);
    if (sh.st) (void)hipStreamDestroy(sh.st);
}

void phase1(Shard& sh, bool inverse, const uint8_t* in)
{
    shard_setup(sh);
    if (sh.rc != kOk)
        return;
    if (!inverse) {
        SHARD_TRY(hipMemcpyAsync(sh.d_in, in + sh.first * 16, (size_t)(sh.count * 16), hipMemcpyHostToDevice, sh.st), "H2D blocks");
        SHARD_TRY(dxtlt::bc7::launch(false, sh.d_in, sh.d_out, sh.count, sh.d_ws, sh.ws_bytes, sh.st), "BC7 transform");
    } else {
        SHARD_TRY(hipMemcpyAsync(sh.d_in, in + sh.first, (size_t)sh.count, hipMemcpyHostToDevice, sh.st), "H2D first stream");
        SHARD_TRY(dxtlt::bc7::launch_counts(sh.d_in, sh.count, sh.d_ws, sh.ws_bytes, sh.st), "BC7 mode counts");
    }
    SHARD_TRY(hipMemcpyAsync(sh.counts, sh.d_ws, sizeof(sh.counts), hipMemcpyDeviceToHost, sh.st), "D2H mode counts");
    SHARD_TRY(hipStreamSynchronize(sh.st), "stream synchronize");
}

void phase2(Shard& sh, bool inverse, const uint8_t* in, uint8_t* out, const uint64_t* g, const uint64_t* l, const uint64_t* n)
{
    SHARD_TRY(hipSetDevice(sh.dev), "hipSetDevice");
    if (!inverse) {
        for (int p = 0; p < 19; ++p)
            if (n[p])
                SHARD_TRY(hipMemcpyAsync(out + g[p], (const uint8_t*)sh.d_out + l[p], (size_t)n[p], hipMemcpyDeviceToHost, sh.st),
                          "D2H stream piece");
    } else {
        for (int p = 1; p < 19; ++p)   // piece 0 (`first`) is already in place from phase 1
            if (n[p])
                SHARD_TRY(hipMemcpyAsync((uint8_t*)sh.d_in + l[p], in + g[p], (size_t)n[p], hipMemcpyHostToDevice, sh.st),
                          "H2D stream piece");
        SHARD_TRY(dxtlt::bc7::launch(true, sh.d_in, sh.d_out, sh.count, sh.d_ws, sh.ws_bytes, sh.st), "BC7 untransform");
        SHARD_TRY(hipMemcpyAsync(out + sh.first * 16, sh.d_out, (size_t)(sh.count * 16), hipMemcpyDeviceToHost, sh.st), "D2H blocks");
    }
    SHARD_TRY(hipStreamSynchronize(sh.st), "stream synchronize");
}

template <class F>
void run_all(std::vector<Shard>& shards, F f)
{
    std::vector<std::thread> threads;
    for (size_t i = 0; i < shards.size(); ++i)
        threads.emplace_back([&, i] {
            if (shards[i].rc == kOk)
                f(shards[i], i);
        });
    for (auto& t : threads)
        t.join();
}

int32_t sharded(bool inverse, const uint8_t* in, uint8_t* out, size_t len, int32_t num_shards)
{
    if (len % 16 != 0)
        return fail(kInvalidLength, "len is not a multiple of 16 (BC7 block size)");
    if (len == 0)
        return kOk;
    if (in == nullptr || out == nullptr)
        return fail(kInvalidArgument, "NULL buffer with len > 0");
    int devices = 0;
    hipError_t e = hipGetDeviceCount(&devices);
    if (e != hipSuccess || devices <= 0)
        return fail(kNoDevice, "no HIP device available (this library has no CPU fallback)", e);
    const uint64_t total = len / 16;
    // num_shards <= 0: one shard per device.  More shards than devices is allowed (round robin), which is how the
    // placement logic is exercised on a single-GPU machine.
    uint64_t S = num_shards <= 0 ? (uint64_t)devices : (uint64_t)num_shards;
    if (S > 64) S = 64;
    if (S > total) S = total;
    int prev = 0;
    (void)hipGetDevice(&prev);

    std::vector<Shard> shards((size_t)S);
    uint64_t share = total / S;
    share -= share % 1024;   // whole tiles, so every shard but the last starts on a tile boundary
    if (share == 0) share = total / S;
    uint64_t at = 0;
    for (uint64_t i = 0; i < S; ++i) {
        shards[i].dev = (int)(i % (uint64_t)devices);
        shards[i].first = at;
        shards[i].count = i == S - 1 ? total - at : share;
        at += shards[i].count;
    }

    run_all(shards, [&](Shard& sh, size_t) { phase1(sh, inverse, in); });

    std::vector<uint64_t> counts(S * 9), firsts(S), nums(S), g(S * 19), l(S * 19), n(S * 19);
    int32_t rc = kOk;
    std::string msg;
    for (uint64_t i = 0; i < S; ++i) {
        if (shards[i].rc != kOk && rc == kOk) {
            rc = shards[i].rc;
            msg = shards[i].msg;
        }
        for (int m = 0; m < 9; ++m)
            counts[i * 9 + m] = shards[i].counts[m];
        firsts[i] = shards[i].first;
        nums[i] = shards[i].count;
    }
    for (uint64_t i = 0; i < S && rc == kOk; ++i) {
        rc = pieces_for(counts.data(), (int32_t)S, (int32_t)i, firsts.data(), nums.data(), total, &g[i * 19], &l[i * 19], &n[i * 19]);
        if (rc != kOk)
            msg = dxtlt_last_error();
    }
    if (rc == kOk) {
        run_all(shards, [&](Shard& sh, size_t i) { phase2(sh, inverse, in, out, &g[i * 19], &l[i * 19], &n[i * 19]); });
        for (uint64_t i = 0; i < S; ++i)
            if (shards[i].rc != kOk && rc == kOk) {
                rc = shards[i].rc;
                msg = shards[i].msg;
            }
    }
    for (auto& sh : shards)
        shard_release(sh);
    (void)hipSetDevice(prev);
    if (rc != kOk)
        return fail(rc, msg.c_str());
    return kOk;
}

}  // namespace

extern "C" {

int32_t dxtlt_bc7_shard_pieces(const uint64_t* counts, int32_t num_shards, int32_t shard, const uint64_t* shard_first_block,
                               const uint64_t* shard_num_blocks, uint64_t total_blocks, uint64_t* global_off,
                               uint64_t* local_off, uint64_t* bytes)
{
    return pieces_for(counts, num_shards, shard, shard_first_block, shard_num_blocks, total_blocks, global_off, local_off,
                      bytes);
}

int32_t dxtlt_transform_bc7_sharded(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len, int32_t num_shards)
{
    return sharded(false, input_ptr, output_ptr, len, num_shards);
}

int32_t dxtlt_untransform_bc7_sharded(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len, int32_t num_shards)
{
    return sharded(true, input_ptr, output_ptr, len, num_shards);
}

}  // extern "C"
