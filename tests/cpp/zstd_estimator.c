/* A DltSizeEstimator in C over the system libzstd (dlopen'ed: the image has libzstd.so.1 but no header) -- what the
 * reference's estimator crate does with zstd-sys (extensions/compressors/dxt-lossless-transform-zstd/src/lib.rs:146-200:
 * ZSTD_compress into the scratch buffer, the compressed size is the estimate).  Thread-safe (ZSTD_compress keeps no
 * state between calls), which the parallel-estimator tests and tools/auto_bench.py rely on; counts its calls.
 * Test / measurement tooling: built by tests (gcc -shared), never part of the product. */
#include <dlfcn.h>
#include <stdatomic.h>
#include <stddef.h>
#include <stdint.h>

typedef size_t (*compress_fn)(void *, size_t, const void *, size_t, int);
typedef size_t (*bound_fn)(size_t);
typedef unsigned (*iserr_fn)(size_t);

static compress_fn p_compress;
static bound_fn p_bound;
static iserr_fn p_iserr;
static atomic_int g_calls, g_active, g_max_active;

int zest_init(void)
{
    void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h)
        return 1;
    p_compress = (compress_fn)dlsym(h, "ZSTD_compress");
    p_bound = (bound_fn)dlsym(h, "ZSTD_compressBound");
    p_iserr = (iserr_fn)dlsym(h, "ZSTD_isError");
    return (p_compress && p_bound && p_iserr) ? 0 : 2;
}

/* Context = (void *)(intptr_t)level */
uint32_t zest_max_compressed_size(void *context, size_t len_bytes, size_t *out_size)
{
    (void)context;
    *out_size = p_bound(len_bytes);
    return 0;
}

uint32_t zest_estimate(void *context, const uint8_t *input, size_t len, uint8_t *scratch, size_t scratch_len, size_t *out_size)
{
    const int now = atomic_fetch_add(&g_active, 1) + 1;
    int seen = atomic_load(&g_max_active);
    while (now > seen && !atomic_compare_exchange_weak(&g_max_active, &seen, now)) {
    }
    atomic_fetch_add(&g_calls, 1);
    const size_t n = len ? p_compress(scratch, scratch_len, input, len, (int)(intptr_t)context) : 0;
    atomic_fetch_sub(&g_active, 1);
    if (len && p_iserr(n))
        return 77;
    *out_size = n;
    return 0;
}

/* constant-time estimator (size = len, like the reference's C dummy estimator): what is left is the library's own work */
uint32_t zest_len_estimate(void *context, const uint8_t *input, size_t len, uint8_t *scratch, size_t scratch_len, size_t *out_size)
{
    (void)context; (void)input; (void)scratch; (void)scratch_len;
    atomic_fetch_add(&g_calls, 1);
    *out_size = len;
    return 0;
}

int zest_calls(void) { return atomic_load(&g_calls); }
int zest_max_concurrency(void) { return atomic_load(&g_max_active); }
void zest_reset(void)
{
    atomic_store(&g_calls, 0);
    atomic_store(&g_max_active, 0);
}
