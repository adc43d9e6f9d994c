// auto_transform.cpp -- transform_bcN_auto behind the C ABI: brute-force choice of transform settings with a
// caller-supplied size estimator.
//
// Reference behaviour kept exactly (paths under /root/reference/src/core/):
//   BC1  dxt-lossless-transform-bc1/src/transform/transform_auto.rs:200-270, test orders settings.rs:81-98
//   BC2  dxt-lossless-transform-bc2/src/transform/transform_auto.rs:196-,   test orders settings.rs:81-98
//   BC3  dxt-lossless-transform-bc3/src/transform/transform_auto.rs:196-294, test orders settings.rs:91-121
//   * one max_compressed_size query up front (len/2 for BC1, len/4 for BC2 and BC3), scratch allocated once;
//   * candidates are tried in the reference's order; each is a FULL transform followed by the estimator on the
//     endpoint section(s) only: BC1 [0, len/2); BC2 [len/2, len/2+len/4); BC3 alpha endpoints [0, 2N) plus
//     colour endpoints [len/2, len/2+4N), sizes added;
//   * strict `<` against the running best (first best wins), defaults as the initial best;
//   * if the best candidate was not the last one tried, the data is transformed once more with it.
//
// GPU shape: the input is uploaded once.  ONE fused kernel (auto_kernels.hip) reads it once and writes every endpoint
// section any candidate can show the estimator into a device arena -- 4 / 8 colour sections (YCoCg-R variant x
// split) and, for BC3, 2 alpha-endpoint sections cover all 4 / 8 / 16 candidates; the index sections are the same for
// every candidate and are never produced here (transform_auto.rs:245-256).  Per candidate only its section(s) travel
// back (half or a quarter of the buffer), into the output buffer at the offsets the reference estimates at, and the
// estimator -- on the CPU behind the callback, as in the reference -- is called in the reference's order with the
// reference's bytes.  One transform launch with the winning settings and one download of the whole result finish the
// call.  If the arena (2-4 x len) cannot be allocated the candidates are produced one full transform at a time, as in
// round 1 (same results, len x 2 of traffic per candidate).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <exception>
#include <thread>
#include <vector>

#include "../../include/dxtlt_gfx950.h"
#include "auto_launch.h"
#include "bcn_launch.h"
#include "host_common.h"

namespace {

struct Candidate {
    uint8_t mode;
    bool split_alpha;
    bool split_colour;
};

// bc1/bc2 settings.rs:81-86 and :89-98
const Candidate kFast12[] = {{0, false, false}, {0, false, true}, {1, false, false}, {1, false, true}};
const Candidate kAll12[] = {{2, false, false}, {0, false, false}, {0, false, true}, {3, false, false},
                            {3, false, true},  {2, false, true},  {1, false, false}, {1, false, true}};
// bc3 settings.rs:91-100 and :104-121  (variant, split_alphas, split_colours)
const Candidate kFast3[] = {{1, true, false}, {1, true, true},  {0, true, false},  {0, false, true},
                            {0, true, true},  {1, false, true}, {0, false, false}, {1, false, false}};
const Candidate kAll3[] = {{2, true, false},  {2, true, true},  {3, true, true},   {3, true, false},
                           {1, true, false},  {3, false, true}, {1, true, true},   {2, false, true},
                           {2, false, false}, {3, false, false}, {0, true, false}, {0, false, true},
                           {0, true, true},   {1, false, true}, {0, false, false}, {1, false, false}};

bool same(const Candidate& a, const Candidate& b)
{
    return a.mode == b.mode && a.split_alpha == b.split_alpha && a.split_colour == b.split_colour;
}

// every failure exit drains the stream first: the staging buffers and the arena belong to this thread's next call
#define HIP_TRY_AUTO(expr, what)                                        \
    do {                                                                \
        hipError_t e_ = (expr);                                         \
        if (e_ != hipSuccess) {                                         \
            if (st) (void)hipStreamSynchronize(st);                     \
            std::free(scratch);                                         \
            return dxtlt_host::fail(dxtlt_host::kDevice, what, e_);     \
        }                                                               \
    } while (0)

// per-thread candidate arena (grow-only, like the staging buffers)
struct Arena {
    void* ptr = nullptr;
    size_t cap = 0;
    int device = -1;
    ~Arena() { release(); }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        device = -1;
    }
    void* get(size_t bytes)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess)
            return nullptr;
        if (dev != device || bytes > cap) {
            release();
            if (hipMalloc(&ptr, bytes) != hipSuccess) {
                (void)hipGetLastError();
                ptr = nullptr;
                return nullptr;
            }
            cap = bytes;
            device = dev;
        }
        return ptr;
    }
};
thread_local Arena g_arena;

// ---------------------------------------------------------------------------------------------------------------
// Opt-in: the estimator on several host threads (dxtlt_set_auto_estimator_threads).  The reference evaluates its
// candidates one after the other because each is a transform into the one output buffer; here every section a
// candidate can show the estimator already sits in the arena, so the estimator -- the hot loop of this call
// (transform/mod.rs:32-34: 265 MiB/s with zstd level 1, 1 GiB/s with LTU, one thread) -- can run on all of them at
// once.  And every DISTINCT section is estimated once: BC3's 8 / 16 candidates are 2 alpha-endpoint sections x 4 / 8
// colour sections = 6 / 10 estimator calls instead of 16 / 32.  Same candidates, same sizes, same order of
// comparison and strict `<`: the same choice and the same bytes as the sequential flow -- what changes is the
// sequence of callback invocations (concurrent, once per distinct section), which is why the caller has to ask for it:
// the callbacks must be safe to call from several threads at once with the same Context.
// ---------------------------------------------------------------------------------------------------------------
std::atomic<int> g_estimator_threads{1};
// per-thread cap on top of it (dxtlt_set_auto_estimator_threads_for_this_thread): 0 = none.  A binding whose estimator type
// makes no thread-safety promise (the Rust glue: `T: SizeEstimationOperations` without `Sync`) sets 1 around its call, so that
// its soundness does not rest on nobody in the process having called the process-wide setter.
thread_local int t_estimator_threads_cap = 0;
constexpr size_t kStageCapBytes = size_t(512) << 20;   // pinned staging per wave of sections (one section at least)

struct HostStage {
    void* ptr = nullptr;
    size_t cap = 0;
    ~HostStage() { release(); }
    void release()
    {
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
    void* get(size_t bytes)
    {
        if (bytes > cap) {
            release();
            if (hipHostMalloc(&ptr, bytes, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                ptr = nullptr;
                return nullptr;
            }
            cap = bytes;
        }
        return ptr;
    }
};
thread_local HostStage g_stage;

struct Section {
    const uint8_t* d_src;   // in the arena
    size_t len;
    size_t size = 0;        // the estimator's answer
    uint32_t rc = 0;        // the estimator's status
    size_t slot = 0;        // byte offset in the staging buffer of its wave
};

// Estimates every section; false = a HIP call failed (*hip_error).  Estimator failures are recorded per section.
bool estimate_sections_parallel(std::vector<Section>& sections, const DltSizeEstimator* est, size_t max_comp, int threads,
                                hipStream_t st, hipError_t* hip_error)
{
    size_t largest = 0;
    for (const Section& s : sections)
        largest = std::max(largest, (s.len + 255) & ~size_t(255));
    const size_t cap = std::max(largest, kStageCapBytes);
    size_t want = 0, run = 0;
    for (const Section& s : sections) {   // the largest wave the cap allows, in order
        const size_t b = (s.len + 255) & ~size_t(255);
        if (run + b > cap)
            run = 0;
        run += b;
        want = std::max(want, run);
    }
    uint8_t* stage = static_cast<uint8_t*>(g_stage.get(want));
    if (stage == nullptr) {
        *hip_error = hipErrorOutOfMemory;
        return false;
    }
    threads = std::max(1, std::min<int>(threads, (int)sections.size()));
    std::vector<uint8_t*> scratch((size_t)threads, nullptr);
    bool ok = true;
    if (max_comp != 0)
        for (auto& p : scratch) {
            p = static_cast<uint8_t*>(std::aligned_alloc(64, (max_comp + 63) / 64 * 64));
            ok = ok && p != nullptr;
        }
    *hip_error = ok ? hipSuccess : hipErrorOutOfMemory;
    size_t first = 0;
    while (ok && first < sections.size()) {
        size_t last = first, used = 0;
        while (last < sections.size() && (last == first || used + ((sections[last].len + 255) & ~size_t(255)) <= cap)) {
            sections[last].slot = used;
            used += (sections[last].len + 255) & ~size_t(255);
            ++last;
        }
        for (size_t i = first; i < last && ok; ++i)
            if (sections[i].len) {
                *hip_error = hipMemcpyAsync(stage + sections[i].slot, sections[i].d_src, sections[i].len, hipMemcpyDeviceToHost, st);
                ok = *hip_error == hipSuccess;
            }
        if (ok) {
            *hip_error = hipStreamSynchronize(st);
            ok = *hip_error == hipSuccess;
        }
        if (!ok)
            break;
        std::atomic<size_t> next{first};
        auto worker = [&](int tid) {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= last)
                    return;
                Section& s = sections[i];
                s.rc = est->EstimateCompressedSize(est->Context, stage + s.slot, s.len, scratch[(size_t)tid], max_comp, &s.size);
            }
        };
        // a thread that cannot be started (EAGAIN under a process limit) is simply not part of the pool: the sections are
        // handed out through `next`, so the threads that did start -- this one at least -- take all of them
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; ++t) {
            try {
                pool.emplace_back(worker, t);
            } catch (const std::exception&) {
                break;
            }
        }
        worker(0);
        for (auto& t : pool)
            t.join();
        first = last;
    }
    for (auto p : scratch)
        std::free(p);
    return ok;
}

}  // namespace

extern "C" void dxtlt_set_auto_estimator_threads(int32_t threads)
{
    g_estimator_threads.store(threads < 1 ? 1 : threads > 64 ? 64 : threads, std::memory_order_relaxed);
}

extern "C" int32_t dxtlt_get_auto_estimator_threads(void) { return g_estimator_threads.load(std::memory_order_relaxed); }

extern "C" int32_t dxtlt_set_auto_estimator_threads_for_this_thread(int32_t cap)
{
    const int32_t before = t_estimator_threads_cap;
    t_estimator_threads_cap = cap < 0 ? 0 : cap > 64 ? 64 : cap;
    return before;
}

void dxtlt_host::release_auto_thread_arena()
{
    g_arena.release();
    g_stage.release();
}

int32_t dxtlt_host::transform_auto(int32_t format, const uint8_t* in, uint8_t* out, size_t len,
                                   const DltSizeEstimator* est, bool use_all, AutoChoice* choice)
{
    if (format < 1 || format > 3)
        return fail(kInvalidArgument, "format must be 1 (BC1), 2 (BC2) or 3 (BC3)");
    const size_t block = format == 1 ? 8 : 16;
    if (len % block != 0)
        return fail(kInvalidLength, "len is not a multiple of the block size");
    if (est == nullptr || est->MaxCompressedSize == nullptr || est->EstimateCompressedSize == nullptr || choice == nullptr)
        return fail(kInvalidArgument, "NULL estimator / choice");
    if (len > 0 && (in == nullptr || out == nullptr))
        return fail(kInvalidArgument, "NULL buffer with len > 0");

    const uint64_t blocks = len / block;
    // defaults: Bc1/Bc2 {Variant1, split}, Bc3 {Variant1, split alphas, split colours}
    Candidate best{1, format == 3, true};
    Candidate last = best;
    size_t best_size = SIZE_MAX;
    choice->estimator_error = 0;

    // the section(s) the estimator sees
    const size_t colour_off = format == 1 ? 0 : len / 2;
    const size_t colour_len = format == 1 ? len / 2 : len / 4;
    const size_t alpha_len = format == 3 ? (size_t)blocks * 2 : 0;

    size_t max_comp = 0;
    uint32_t rc_est = est->MaxCompressedSize(est->Context, format == 1 ? len / 2 : len / 4, &max_comp);
    if (rc_est != 0) {
        choice->estimator_error = rc_est;
        return fail(kEstimator, "size estimator: max_compressed_size failed");
    }
    uint8_t* scratch = nullptr;
    if (max_comp != 0) {
        scratch = static_cast<uint8_t*>(std::aligned_alloc(64, (max_comp + 63) / 64 * 64));
        if (scratch == nullptr)
            return fail(kAllocation, "estimator scratch allocation failed");
    }

    void *d_in = nullptr, *d_out = nullptr;
    hipStream_t st = nullptr;
    uint8_t* arena = nullptr;   // device: every candidate section, from one read of the input
    if (len > 0) {
        int32_t rc = acquire_staging(len, &d_in, &d_out, &st);
        if (rc != kOk) {
            std::free(scratch);
            return rc;
        }
        HIP_TRY_AUTO(hipMemcpyAsync(d_in, in, len, hipMemcpyHostToDevice, st), "H2D copy");
        static const bool fused = [] { const char* v = dxtlt::experiment_env("DXTLT_AUTO_FUSED"); return !(v && v[0] == '0'); }();
        if (fused)
            arena = static_cast<uint8_t*>(g_arena.get((size_t)dxtlt::auto_arena_bytes((dxtlt::Format)format, use_all, blocks)));
        if (arena != nullptr)
            HIP_TRY_AUTO(dxtlt::launch_auto_candidates((dxtlt::Format)format, use_all, d_in, arena, blocks, st),
                         "candidate kernel launch");
    }

    const Candidate* order;
    int count;
    if (format == 3) {
        order = use_all ? kAll3 : kFast3;
        count = use_all ? 16 : 8;
    } else {
        order = use_all ? kAll12 : kFast12;
        count = use_all ? 8 : 4;
    }

    int est_threads = g_estimator_threads.load(std::memory_order_relaxed);
    if (t_estimator_threads_cap > 0 && est_threads > t_estimator_threads_cap)
        est_threads = t_estimator_threads_cap;
    const bool parallel = est_threads > 1 && arena != nullptr && len > 0;
    if (parallel) {
        // distinct sections: colour (variant, split) pairs in the arena's order, then BC3's two alpha-endpoint sections
        const int variants = use_all ? 4 : 2;
        std::vector<Section> sections;
        for (int m = 0; m < variants; ++m)
            for (int sp = 0; sp < 2; ++sp)
                sections.push_back(Section{arena + dxtlt::auto_section_offset((dxtlt::Format)format, blocks, m, sp != 0), colour_len});
        const size_t alpha_first = sections.size();
        if (format == 3)
            for (int sp = 0; sp < 2; ++sp)
                sections.push_back(Section{arena + dxtlt::auto_alpha_section_offset(blocks, sp != 0), alpha_len});
        hipError_t herr = hipSuccess;
        if (!estimate_sections_parallel(sections, est, max_comp, est_threads, st, &herr))
            HIP_TRY_AUTO(herr == hipSuccess ? hipErrorUnknown : herr, "parallel estimation (staging / download)");
        for (int i = 0; i < count; ++i) {
            const Candidate c = order[i];
            size_t total = 0;
            uint32_t bad = 0;
            if (format == 3) {   // the reference's order inside a candidate: alpha endpoints, then colour endpoints
                const Section& a = sections[alpha_first + (c.split_alpha ? 1 : 0)];
                bad = a.rc;
                total = a.size;
            }
            const Section& col = sections[(size_t)c.mode * 2 + (c.split_colour ? 1 : 0)];
            if (bad == 0)
                bad = col.rc;
            total += col.size;
            if (bad != 0) {   // the sequential flow stops at the first candidate whose estimate fails
                std::free(scratch);
                choice->estimator_error = bad;
                return fail(kEstimator, "size estimator: estimate_compressed_size failed");
            }
            if (total < best_size) {
                best_size = total;
                best = c;
            }
        }
    }

    // Sequential mode with the arena: the sections of candidate i + 1 travel into the other half of a pinned staging buffer
    // while the estimator works on candidate i -- the reference's sequence of calls and bytes, minus the wait for every
    // download (and minus pageable-memory copies).  The estimator is shown the staged bytes, not the output buffer.
    const size_t alpha_slot = (alpha_len + 255) & ~size_t(255), slot_bytes = alpha_slot + ((colour_len + 255) & ~size_t(255));
    uint8_t* stage = nullptr;
    if (!parallel && arena != nullptr && len > 0 && slot_bytes <= (size_t(512) << 20))
        stage = static_cast<uint8_t*>(g_stage.get(2 * slot_bytes));
    auto issue_sections = [&](int i) -> hipError_t {
        const Candidate c = order[i];
        uint8_t* slot = stage + (size_t)(i & 1) * slot_bytes;
        hipError_t e = hipSuccess;
        if (alpha_len)
            e = hipMemcpyAsync(slot, arena + dxtlt::auto_alpha_section_offset(blocks, c.split_alpha), alpha_len, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(slot + alpha_slot, arena + dxtlt::auto_section_offset((dxtlt::Format)format, blocks, c.mode, c.split_colour),
                               colour_len, hipMemcpyDeviceToHost, st);
        return e;
    };
    if (stage != nullptr)
        HIP_TRY_AUTO(issue_sections(0), "D2H candidate sections");

    for (int i = 0; i < count && !parallel; ++i) {
        const Candidate c = order[i];
        const uint8_t* shown_alpha = out;
        const uint8_t* shown_colour = out + colour_off;
        if (stage != nullptr) {
            HIP_TRY_AUTO(hipStreamSynchronize(st), "stream synchronize");   // candidate i has arrived
            if (i + 1 < count)
                HIP_TRY_AUTO(issue_sections(i + 1), "D2H candidate sections");
            shown_alpha = stage + (size_t)(i & 1) * slot_bytes;
            shown_colour = shown_alpha + alpha_slot;
        } else if (len > 0) {
            const uint8_t* alpha_src = (const uint8_t*)d_out;
            const uint8_t* colour_src = (const uint8_t*)d_out + colour_off;
            if (arena != nullptr) {
                alpha_src = arena + dxtlt::auto_alpha_section_offset(blocks, c.split_alpha);
                colour_src = arena + dxtlt::auto_section_offset((dxtlt::Format)format, blocks, c.mode, c.split_colour);
            } else {
                int32_t rc = enqueue(format, false, d_in, d_out, blocks, c.mode, c.split_alpha, c.split_colour, st);
                if (rc != kOk) {
                    (void)hipStreamSynchronize(st);
                    std::free(scratch);
                    return rc;
                }
                last = c;
            }
            if (alpha_len)
                HIP_TRY_AUTO(hipMemcpyAsync(out, alpha_src, alpha_len, hipMemcpyDeviceToHost, st), "D2H alpha endpoints");
            HIP_TRY_AUTO(hipMemcpyAsync(out + colour_off, colour_src, colour_len, hipMemcpyDeviceToHost, st),
                         "D2H colour endpoints");
            HIP_TRY_AUTO(hipStreamSynchronize(st), "stream synchronize");
        } else {
            last = c;
        }

        size_t total = 0, part = 0;
        if (format == 3) {
            rc_est = est->EstimateCompressedSize(est->Context, shown_alpha, alpha_len, scratch, max_comp, &part);
            if (rc_est == 0) {
                total = part;
                part = 0;
                rc_est = est->EstimateCompressedSize(est->Context, shown_colour, colour_len, scratch, max_comp, &part);
                total += part;
            }
        } else {
            rc_est = est->EstimateCompressedSize(est->Context, shown_colour, colour_len, scratch, max_comp, &total);
        }
        if (rc_est != 0) {
            if (st) (void)hipStreamSynchronize(st);   // a download of the next candidate may be in flight
            std::free(scratch);
            choice->estimator_error = rc_est;
            return fail(kEstimator, "size estimator: estimate_compressed_size failed");
        }
        if (total < best_size) {
            best_size = total;
            best = c;
        }
    }

    if (len > 0) {
        // with the arena no full transform has run yet; without it the last candidate's is in d_out
        if (arena != nullptr || !same(best, last)) {
            int32_t rc = enqueue(format, false, d_in, d_out, blocks, best.mode, best.split_alpha, best.split_colour, st);
            if (rc != kOk) {
                (void)hipStreamSynchronize(st);
                std::free(scratch);
                return rc;
            }
        }
        HIP_TRY_AUTO(hipMemcpyAsync(out, d_out, len, hipMemcpyDeviceToHost, st), "D2H result");
        HIP_TRY_AUTO(hipStreamSynchronize(st), "stream synchronize");
    }
    std::free(scratch);
    choice->mode = best.mode;
    choice->split_alpha = best.split_alpha;
    choice->split_colour = best.split_colour;
    return kOk;
}

extern "C" {

int32_t dxtlt_transform_bc1_auto(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len,
                                 const DltSizeEstimator* estimator, bool use_all_decorrelation_modes,
                                 uint8_t* out_decorrelation_mode, bool* out_split_colour_endpoints,
                                 uint32_t* out_estimator_error)
{
    dxtlt_host::AutoChoice c{};
    int32_t rc = dxtlt_host::transform_auto(1, input_ptr, output_ptr, len, estimator, use_all_decorrelation_modes, &c);
    if (out_estimator_error) *out_estimator_error = c.estimator_error;
    if (rc == DXTLT_OK) {
        if (out_decorrelation_mode) *out_decorrelation_mode = c.mode;
        if (out_split_colour_endpoints) *out_split_colour_endpoints = c.split_colour;
    }
    return rc;
}

int32_t dxtlt_transform_bc2_auto(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len,
                                 const DltSizeEstimator* estimator, bool use_all_decorrelation_modes,
                                 uint8_t* out_decorrelation_mode, bool* out_split_colour_endpoints,
                                 uint32_t* out_estimator_error)
{
    dxtlt_host::AutoChoice c{};
    int32_t rc = dxtlt_host::transform_auto(2, input_ptr, output_ptr, len, estimator, use_all_decorrelation_modes, &c);
    if (out_estimator_error) *out_estimator_error = c.estimator_error;
    if (rc == DXTLT_OK) {
        if (out_decorrelation_mode) *out_decorrelation_mode = c.mode;
        if (out_split_colour_endpoints) *out_split_colour_endpoints = c.split_colour;
    }
    return rc;
}

int32_t dxtlt_transform_bc3_auto(const uint8_t* input_ptr, uint8_t* output_ptr, size_t len,
                                 const DltSizeEstimator* estimator, bool use_all_decorrelation_modes,
                                 uint8_t* out_decorrelation_mode, bool* out_split_alpha_endpoints,
                                 bool* out_split_colour_endpoints, uint32_t* out_estimator_error)
{
    dxtlt_host::AutoChoice c{};
    int32_t rc = dxtlt_host::transform_auto(3, input_ptr, output_ptr, len, estimator, use_all_decorrelation_modes, &c);
    if (out_estimator_error) *out_estimator_error = c.estimator_error;
    if (rc == DXTLT_OK) {
        if (out_decorrelation_mode) *out_decorrelation_mode = c.mode;
        if (out_split_alpha_endpoints) *out_split_alpha_endpoints = c.split_alpha;
        if (out_split_colour_endpoints) *out_split_colour_endpoints = c.split_colour;
    }
    return rc;
}

}  // extern "C"
