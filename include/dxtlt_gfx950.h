/*
 * dxtlt_gfx950.h -- C ABI of libdxtlt_gfx950.so: the MI355X (gfx950) implementation of the BCn block
 * transform hot path of Sewer56/dxt-lossless-transform.
 *
 * These are the entry points the reference's Rust crates bind through `extern "C"` so that the bodies of
 * their `transform_bcN_with_settings` / `untransform_bcN_with_settings` become one FFI call (the shim is
 * shown in INTEGRATION.md).  Reference signatures replaced (paths under /root/reference/src/core/):
 *
 *   transform_bc1_with_settings     dxt-lossless-transform-bc1/src/transform/transform_with_settings.rs:31
 *   untransform_bc1_with_settings   dxt-lossless-transform-bc1/src/transform/transform_with_settings.rs:92
 *   transform_bc2_with_settings     dxt-lossless-transform-bc2/src/transform/transform_with_settings.rs:30
 *   untransform_bc2_with_settings   dxt-lossless-transform-bc2/src/transform/transform_with_settings.rs:93
 *   transform_bc3_with_settings     dxt-lossless-transform-bc3/src/transform/transform_with_settings.rs:32
 *   untransform_bc3_with_settings   dxt-lossless-transform-bc3/src/transform/transform_with_settings.rs:162
 *
 * Settings travel as scalars (the Rust settings structs are not repr(C)):
 *   decorrelation_mode  = core YCoCgVariant numbering, None=0 Variant1=1 Variant2=2 Variant3=3
 *                         (dxt-lossless-transform-common/src/color_565/decorrelate.rs:72-84)
 *   split_*_endpoints   = Bc1/Bc2/Bc3TransformSettings fields (bc1 settings.rs:16-27, bc3 settings.rs:16-30)
 *
 * Contract kept from the reference: `len` is in bytes and must be a multiple of the block size (8 for BC1,
 * 16 for BC2/BC3); output length == input length; any pointer alignment; any block count including 0;
 * input and output must not overlap; reentrant, no global mutable state visible to the caller.
 * Difference: the reference's unsafe fns return nothing; these return a status (0 = ok) because a device can
 * fail where a CPU loop cannot.  DXTLT_E_* values below; dxtlt_last_error() gives the text for this thread.
 *
 * Three families:
 *   *_with_settings           host pointers in, host pointers out (H2D + kernel + D2H on the current device)
 *   *_with_settings_device    device pointers, enqueued on a caller-provided HIP stream, no synchronisation
 *   *_with_settings_range     device pointers, one contiguous block range of a larger array
 *                             (multi-GPU shards, chunked host staging)
 */
#ifndef DXTLT_GFX950_H
#define DXTLT_GFX950_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include "dlt_size_estimator.h"

#ifdef __cplusplus
extern "C" {
#endif

/* status codes */
#define DXTLT_OK 0
#define DXTLT_E_INVALID_LENGTH 1   /* len not a multiple of the block size */
#define DXTLT_E_INVALID_ARGUMENT 2 /* bad decorrelation mode, NULL pointer with len > 0, bad range */
#define DXTLT_E_NO_DEVICE 3        /* no usable HIP device */
#define DXTLT_E_DEVICE 4           /* HIP runtime error (allocation, copy, launch) */
#define DXTLT_E_ESTIMATOR 5        /* a size-estimator callback returned non-zero (auto transform) */
#define DXTLT_E_ALLOCATION 6       /* a host resource ran out: the estimator's scratch buffer, or a worker thread the call could not
                                      start (batch_host, the sharded calls: everything already started is joined first) */

/* YCoCgVariant, core numbering */
#define DXTLT_YCOCG_NONE 0
#define DXTLT_YCOCG_VARIANT1 1
#define DXTLT_YCOCG_VARIANT2 2
#define DXTLT_YCOCG_VARIANT3 3

/* ---- host pointers: drop-in bodies of the reference's unsafe fns -------------------------------- */
int32_t dxtlt_transform_bc1_with_settings(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len,
                                          uint8_t decorrelation_mode, bool split_colour_endpoints);
int32_t dxtlt_untransform_bc1_with_settings(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len,
                                            uint8_t decorrelation_mode, bool split_colour_endpoints);
int32_t dxtlt_transform_bc2_with_settings(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len,
                                          uint8_t decorrelation_mode, bool split_colour_endpoints);
int32_t dxtlt_untransform_bc2_with_settings(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len,
                                            uint8_t decorrelation_mode, bool split_colour_endpoints);
int32_t dxtlt_transform_bc3_with_settings(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len,
                                          uint8_t decorrelation_mode, bool split_alpha_endpoints,
                                          bool split_colour_endpoints);
int32_t dxtlt_untransform_bc3_with_settings(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len,
                                            uint8_t decorrelation_mode, bool split_alpha_endpoints,
                                            bool split_colour_endpoints);

/* ---- host pointers: transform_bcN_auto ------------------------------------------------------------
 * Replaces the bodies of (paths under /root/reference/src/core/):
 *   transform_bc1_auto   dxt-lossless-transform-bc1/src/transform/transform_auto.rs:200
 *   transform_bc2_auto   dxt-lossless-transform-bc2/src/transform/transform_auto.rs:196
 *   transform_bc3_auto   dxt-lossless-transform-bc3/src/transform/transform_auto.rs:196
 * `estimator` is the C vtable of the crate's SizeEstimationOperations (a Rust shim wraps any T in one).
 * Candidate order, estimated sections, strict `<` tie-break and the final re-transform follow the reference;
 * on DXTLT_OK the out_* settings (core numbering) describe the data left in output_ptr.  On DXTLT_E_ESTIMATOR
 * *out_estimator_error (may be NULL) receives the callback's error code. */
int32_t dxtlt_transform_bc1_auto(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len,
                                 const DltSizeEstimator *estimator, bool use_all_decorrelation_modes,
                                 uint8_t *out_decorrelation_mode, bool *out_split_colour_endpoints,
                                 uint32_t *out_estimator_error);
int32_t dxtlt_transform_bc2_auto(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len,
                                 const DltSizeEstimator *estimator, bool use_all_decorrelation_modes,
                                 uint8_t *out_decorrelation_mode, bool *out_split_colour_endpoints,
                                 uint32_t *out_estimator_error);
int32_t dxtlt_transform_bc3_auto(const uint8_t *input_ptr, uint8_t *output_ptr, size_t len,
                                 const DltSizeEstimator *estimator, bool use_all_decorrelation_modes,
                                 uint8_t *out_decorrelation_mode, bool *out_split_alpha_endpoints,
                                 bool *out_split_colour_endpoints, uint32_t *out_estimator_error);

/* Additive, opt-in: run the size estimator of dxtlt_transform_bcN_auto (and of everything built on it: the dltbcN auto
 * builders, dxtlt_dds_transform_auto) on `threads` host threads at once (1..64; default 1 = the reference's sequence of
 * calls, core/dxt-lossless-transform-bc1/src/transform/transform_auto.rs:200-270).  The estimator is the hot loop of
 * the auto transform (transform/mod.rs:32-34) and every section a candidate can show it is on the device after ONE
 * kernel, so with threads > 1 every DISTINCT section (4 / 8 colour sections, plus BC3's 2 alpha-endpoint sections: 6 / 10
 * estimator calls for BC3's 8 / 16 candidates) is downloaded to pinned memory and estimated concurrently.  Same
 * candidates, same sizes, same comparison order and strict `<`: the same settings and bytes as with one thread.  What
 * changes is the callback traffic -- concurrent, once per distinct section -- so the estimator's callbacks must be safe
 * to call from several threads with the same Context (each call gets its own scratch buffer).  Process-wide. */
void dxtlt_set_auto_estimator_threads(int32_t threads);
int32_t dxtlt_get_auto_estimator_threads(void);
/* A cap on the above for auto transforms called FROM THE CALLING THREAD (0 = no cap; returns the previous cap).  For bindings
 * whose estimator type makes no thread-safety promise: with a cap of 1 around the call the callbacks run one at a time on the
 * calling thread whatever the process-wide setting says (rust/core-bodies/gfx950_glue.rs: SerialEstimatorCalls). */
int32_t dxtlt_set_auto_estimator_threads_for_this_thread(int32_t cap);

/* ---- device pointers, whole buffer, asynchronous on `hip_stream` (a hipStream_t; NULL = default) -- */
int32_t dxtlt_transform_bc1_with_settings_device(const void *d_input, void *d_output, size_t len,
                                                 uint8_t decorrelation_mode, bool split_colour_endpoints,
                                                 void *hip_stream);
int32_t dxtlt_untransform_bc1_with_settings_device(const void *d_input, void *d_output, size_t len,
                                                   uint8_t decorrelation_mode, bool split_colour_endpoints,
                                                   void *hip_stream);
int32_t dxtlt_transform_bc2_with_settings_device(const void *d_input, void *d_output, size_t len,
                                                 uint8_t decorrelation_mode, bool split_colour_endpoints,
                                                 void *hip_stream);
int32_t dxtlt_untransform_bc2_with_settings_device(const void *d_input, void *d_output, size_t len,
                                                   uint8_t decorrelation_mode, bool split_colour_endpoints,
                                                   void *hip_stream);
int32_t dxtlt_transform_bc3_with_settings_device(const void *d_input, void *d_output, size_t len,
                                                 uint8_t decorrelation_mode, bool split_alpha_endpoints,
                                                 bool split_colour_endpoints, void *hip_stream);
int32_t dxtlt_untransform_bc3_with_settings_device(const void *d_input, void *d_output, size_t len,
                                                   uint8_t decorrelation_mode, bool split_alpha_endpoints,
                                                   bool split_colour_endpoints, void *hip_stream);

/* ---- device pointers, one block range of a larger array -----------------------------------------
 * format: 1, 2, 3 = BC1, BC2, BC3.  inverse: false = transform, true = untransform.
 * The array has `total_blocks` blocks; this call handles blocks [first_block, first_block+num_blocks).
 *   d_aos    points at the AoS bytes of block `first_block` (the range's own slice of the block array)
 *   d_soa    points at byte 0 of the WHOLE transformed buffer (total_blocks * block_size bytes); the call
 *            touches only this range's slice of every stream.
 * Forward reads d_aos and writes d_soa; inverse reads d_soa and writes d_aos.
 * With first_block = 0 and num_blocks = total_blocks this is the whole-buffer call. */
int32_t dxtlt_transform_range_device(int32_t format, bool inverse, const void *d_src, void *d_dst,
                                     uint64_t total_blocks, uint64_t first_block, uint64_t num_blocks,
                                     uint8_t decorrelation_mode, bool split_alpha_endpoints,
                                     bool split_colour_endpoints, void *hip_stream);

/* ---- device pointers, many buffers in one call -----------------------------------------------------
 * Additive (the reference transforms one buffer per call and fans files out over CPU threads): a texture of a few
 * MiB cannot fill 256 CUs and a launch costs the host longer than such a kernel runs, so a batch is ONE kernel launch
 * per (format, direction) present in it -- every workgroup looks up the buffer it belongs to (a small table is copied
 * to the device on `hip_stream` first).  Per-buffer settings, any block counts and alignments.  Asynchronous; ordered
 * like a single call with respect to `hip_stream`; validated as a whole before anything is enqueued.  Items must not
 * overlap one another.  A call stages its tables in ONE of four per-thread slots and returns without waiting for the
 * device; only the fifth call of a thread whose four predecessors are all still in flight waits (on the host) for the oldest
 * of them -- so do not enqueue more than four batch calls of one thread behind an event that is recorded later.
 * Regular batches are recognised and served by shorter paths, with identical results: when all items of one format and
 * direction have ONE size the owning item is found by a division instead of a table walk; when they also share their
 * settings and their inputs and outputs each lie a constant stride apart (an array texture) no table is read at all; and
 * when on top of that every stream of a buffer starts on a 128-byte line and holds a whole number of tiles (any
 * power-of-two texture) the launch is the single-buffer kernel with one more grid dimension.  INTEGRATION.md section 7
 * has the rates and one more caller-side lever (transformed buffers not a power of two apart). */
typedef struct DxtltBatchItem {
    const void *d_input;
    void *d_output;
    uint64_t len;                   /* bytes, a multiple of the block size */
    uint8_t format;                 /* 1, 2, 3 = BC1, BC2, BC3; 7 = BC7 in this build's own format (dxtlt_bc7.h; settings ignored):
                                       its granules go in one launch per direction, its tail parts in a second one */
    uint8_t inverse;                /* 0 = transform, 1 = untransform */
    uint8_t decorrelation_mode;     /* core numbering */
    uint8_t split_alpha_endpoints;  /* BC3 only */
    uint8_t split_colour_endpoints;
    uint8_t reserved[3];
} DxtltBatchItem;
int32_t dxtlt_transform_batch_device(const DxtltBatchItem *items, size_t count, void *hip_stream);
/* Test hook, no device needed: plans `count` buffers of one format, direction and settings the way
 * dxtlt_transform_batch_device plans one launch (addresses are numbers here, nothing is dereferenced) and returns the
 * launch's workgroups; entries_out[i] describes buffer i (an empty buffer owns no workgroup), index_out receives the
 * workgroup -> entry index: uint32 base[ceil(wgs / 4096)], then delta[ceil(wgs / 64)] -- uint8 each, or (*index_is_wide_out
 * == 1: more than 255 buffers begin inside some span of 4096 workgroups) little-endian uint16 each; base[w / 4096] +
 * delta[w / 64] (counting buffers that own workgroups) is the owner of workgroup 64 * (w / 64), and the owner of w is that
 * entry or one of the next w % 64: the first whose end_wg > w.  index_capacity: at least 4 * ceil(wgs / 4096) +
 * 2 * ceil(wgs / 64) + 15.  0xFFFFFFFF: index_capacity too small, or a buffer the batch kernel does not take (stream bases
 * off their element width: launched alone by the batch call). */
typedef struct DxtltDebugPlannedEntry {
    uint32_t first_wg, end_wg;   /* workgroups [first_wg, end_wg): whole tiles first, then the edge tile if there is one */
    uint32_t full_tiles;
    uint8_t form;                /* 1 = aligned tiles, 0 = halo tiles (forward) / shifted tiles (inverse) */
    uint8_t halo_vecs;
    uint8_t shift[6];            /* stream base modulo 64 (forward) or 16 (inverse) */
    uint64_t gbase[6];
} DxtltDebugPlannedEntry;
uint32_t dxtlt_debug_plan_batch(int32_t format, int32_t inverse, int32_t variant, int32_t split_alpha, int32_t split_colour,
                                const uint64_t *src_addresses, const uint64_t *dst_addresses, const uint64_t *blocks, size_t count,
                                DxtltDebugPlannedEntry *entries_out, uint8_t *index_out, size_t index_capacity,
                                uint32_t *index_is_wide_out);

/* Test hook, no device needed: what a single-buffer DEVICE call (dxtlt_{un,}transform_bcN_with_settings_device,
 * dxtlt_transform_range_device) on these addresses would enqueue -- addresses are numbers here, nothing is dereferenced.  One
 * record per kernel launch, in stream order:
 *   kind 0  aligned tiles: every stream base of the range on a 128-byte line; `workgroups` whole tiles of `threads` lanes x 16 bytes
 *   kind 1  forward halo tiles (windows moved back to 64-byte sectors): `full_tiles` whole tiles (tile 0 runs as an edge tile: it
 *           writes the head of every stream) and, when blocks or stream tails are left behind them, one edge-tile workgroup
 *   kind 2  inverse shifted tiles (slices displaced by the base modulo 16): `full_tiles` whole tiles + an edge tile for the rest
 * Behind aligned tiles the rest of a range (fewer blocks than a tile) is one more launch of one edge tile (kind 1 / 2, aos_offset
 * = where it starts).  Ranges of more than 2^31 blocks are planned in pieces of 2^31.  Returns the number of launches (records
 * beyond `cap` are counted, not written); -1 for arguments the call itself would refuse.  Honours dxtlt_set_tuning. */
typedef struct DxtltDebugPlannedLaunch {
    int32_t kind;
    int32_t threads;        /* lanes per workgroup */
    uint32_t workgroups;
    uint32_t full_tiles;    /* whole tiles among them */
    uint64_t range_blocks;  /* blocks the launch covers */
    uint64_t aos_offset;    /* bytes from the call's block-array pointer to the launch's first block */
    uint8_t shift[6];       /* kinds 1, 2: every stream base modulo 64 (forward) / 16 (inverse) */
    uint8_t halo_vecs;      /* kind 1: 16-byte vectors of blocks in front of a tile that have bytes in its windows */
    uint8_t natural;        /* every shift a multiple of its stream's element width */
    uint64_t gbase[6];      /* off_s * total_blocks + w_s * first_block - shift[s] */
} DxtltDebugPlannedLaunch;
int32_t dxtlt_debug_plan_transform(int32_t format, int32_t inverse, int32_t variant, int32_t split_alpha, int32_t split_colour,
                                   uint64_t src_address, uint64_t dst_address, uint64_t total_blocks, uint64_t first_block,
                                   uint64_t num_blocks, DxtltDebugPlannedLaunch *out, int32_t cap);

/* The same for HOST buffers (d_input / d_output of every item are host pointers here) -- the reference's own call
 * pattern: one call per file, host pointers, textures of 0.1-20 MiB (tools/dxt-lossless-transform-cli/src/commands/
 * transform/mod.rs:154-199).  Through the single-buffer host entry points every texture pays a PCIe round trip of its
 * own (1 MiB: 6.6 GiB/s); this call uploads the buffers side by side into one device arena, transforms them with one
 * launch per (format, direction) and chunk of ~64 MiB, and downloads the results, with upload, kernels and download of
 * consecutive chunks overlapped.  Synchronous; items of less than 4 GiB each; validated as a whole first. */
int32_t dxtlt_transform_batch_host(const DxtltBatchItem *items, size_t count);

/* ---- single-process multi-GPU: shard [0, N) by contiguous block range over `num_devices` GPUs -----
 * Host pointers.  Each device receives its slice of the input, runs the range kernel, and its slice of
 * every output stream is copied straight to its final place in `output_ptr` (no collective; see
 * DESIGN.md "Multi-GPU").  num_devices <= 0 means all visible devices; a number above the visible devices (at most
 * 64) is that many shards dealt round robin over them.  Shards of 96 MiB or more run as a chunked pipeline (upload,
 * kernel and the per-stream downloads of consecutive chunks overlap). */
int32_t dxtlt_transform_sharded(int32_t format, bool inverse, const uint8_t *input_ptr, uint8_t *output_ptr,
                                size_t len, uint8_t decorrelation_mode, bool split_alpha_endpoints,
                                bool split_colour_endpoints, int32_t num_devices);

/* What the last dxtlt_transform_sharded call on this thread did, shard by shard (a reporting aid: bench.py prints the
 * per-device rates from it).  Returns the number of shards of that call and fills at most `cap` records. */
typedef struct DxtltShardStat {
    int32_t device;        /* HIP device ordinal the shard ran on */
    int32_t cpus_bound;    /* CPUs the shard's worker thread was bound to (0: not bound -- unknown node, or DXTLT_NUMA_BIND=0) */
    uint64_t first_block;  /* the shard's block range */
    uint64_t blocks;
    double seconds;        /* wall time of the shard: upload + kernels + downloads to their final host offsets */
} DxtltShardStat;
int32_t dxtlt_sharded_last_stats(DxtltShardStat *out, int32_t cap);

/* ---- NUMA placement of the library's own host threads -------------------------------------------
 * Every worker thread dxtlt_transform_sharded (and the BC7 sharded calls) starts for a device binds itself to the CPUs
 * the kernel lists as local to that device's PCI function, intersected with the CPUs the process may use; the threads of
 * the caller are never touched.  DXTLT_NUMA_BIND=0 switches it off.  A sharded call brings the HIP runtime up for the
 * devices it uses on the CALLER's thread before it starts a worker: threads the runtime creates lazily would otherwise be
 * born from a worker and keep its narrowed mask for the life of the process.  The three helpers are exported so that a host
 * program can place its own feeder threads -- and its first touch of the arrays -- the same way.
 * dxtlt_pci_local_cpulist: pure host code; reads <sysfs>/bus/pci/devices/<bdf>/local_cpulist (or numa_node -> node
 * cpulist), <sysfs> = $DXTLT_SYSFS_ROOT or /sys; writes e.g. "0-63,128-191" and returns its length, 0 when unknown.
 * dxtlt_device_local_cpulist: the same for a HIP device ordinal (hipDeviceGetPCIBusId).
 * dxtlt_bind_thread_to_cpulist: sched_setaffinity of the CALLING thread; returns the number of CPUs bound to, 0 when
 * the list is malformed or none of its CPUs is available to this process (the thread then stays where it is). */
int32_t dxtlt_pci_local_cpulist(const char *pci_bdf, char *out, size_t cap);
int32_t dxtlt_device_local_cpulist(int32_t device, char *out, size_t cap);
int32_t dxtlt_bind_thread_to_cpulist(const char *cpulist);

/* ---- plumbing ---------------------------------------------------------------------------------- */
/* Deterministic synthetic blocks on the device: qword i = splitmix64(seed, first_qword + i). */
int32_t dxtlt_fill_splitmix64_device(void *d_dst, size_t len_bytes, uint64_t seed, uint64_t first_qword,
                                     void *hip_stream);
/* Text of the last failure on the calling thread ("" if none). */
const char *dxtlt_last_error(void);
/* Number of visible HIP devices (0 if the runtime cannot initialise). */
int32_t dxtlt_device_count(void);
/* Size routing for callers that keep a CPU implementation of their own next to this library (the reference's crates can, as an
 * opt-in: rust/core-bodies, feature `cpu-below-threshold`; by default they send every call here).  A host-pointer call is a PCIe round trip: at least ~17 us, at most ~25-43 GiB/s, where one CPU core of
 * the reference moves 20-50 GiB/s out of cache -- below the crossover the caller's own CPU path is faster (64 KiB: 20 us
 * here against 2.9 us there; measured crossover ~32 MiB).  Returns that crossover in bytes: the
 * value of $DXTLT_HOST_ROUTE_THRESHOLD_BYTES when set (0 = route everything to the device), else what
 * the setter below stored, else 32 MiB.  The library itself routes nothing: it has no CPU
 * implementation, and every entry point does what its name says on the device. */
size_t dxtlt_host_route_threshold_bytes(void);
void dxtlt_set_host_route_threshold_bytes(size_t bytes);
/* Knobs for tests and tuning; process-wide.  tile_threads: workgroup size of the aligned tiles, 64/128/256/512 (0 = per-format
 * default).  force_path: 0 = automatic; 2 = the halo (forward) / shifted (inverse) tiles even when every stream base is on a
 * 128-byte line; 0x20 = their generic LDS accesses even for naturally aligned shifts.  Both select kernels that some address
 * pattern selects by itself and leave every result exact.  Bits outside dxtlt_tuning_mask() are ignored: the shipped library
 * has no other switch -- the experiment switches earlier rounds measured with (element-granular kernel, store policies, a
 * wrong-output timing switch) exist only in a side build made with -DDXTLT_EXPERIMENTS (csrc/bcn_device.h). */
void dxtlt_set_tuning(int32_t tile_threads, int32_t force_path);
/* The force_path bits this build honours: 0x22 for the shipped library. */
int32_t dxtlt_tuning_mask(void);
/* "dxtlt-gfx950 <version>" */
const char *dxtlt_version(void);
/* The host-pointer entry points keep, per calling thread and device, one stream and a grow-only pair of staging
 * buffers (so that file-after-file callers pay allocation once), and for buffers of up to 1 MiB a pair of mapped pinned
 * host buffers that the kernel reads and writes directly.  This frees the calling thread's set -- and, process-
 * wide, the idle per-device stream + buffer sets that dxtlt_transform_sharded and the BC7 sharded entry points keep
 * across calls (sets in use by a call in flight are left alone).  Those sets are bounded without this call too: at most
 * two idle ones stay per device, i.e. at most 4 x the largest shard's bytes of HBM per device. */
void dxtlt_release_thread_resources(void);

#ifdef __cplusplus
}
#endif
#endif
