/*
 * include/apm.h -- C ABI of the MI355X approximate-pattern-matching engine.
 *
 * This is the drop-in boundary for ONE path of
 * linomp/INF560-approximate-pattern-matching: the Levenshtein sliding-window
 * DP + per-pattern match count
 *     levenshtein()                    /root/reference/src/utils.c:76-99
 *     per-pattern scan loop            /root/reference/src/sequential.c:105-144
 * and it replaces the reference's hand-declared extern "C" GPU shim
 *     getDeviceCount / setDevice       src/cuda_utils.cu:10-35
 *     invoke_kernel / write_kernel_result   src/patterns_over_ranks.cu:75-134
 *     initializeGPU / getGPUResult     src/database_over_ranks.cu:137-205
 * (prototypes re-declared at src/main.c:18-19, src/patterns_over_ranks.c:33-36,
 *  src/database_over_ranks.c:18-22).
 *
 * Conventions
 *   - plain C linkage, pointers + sizes only, no C++/torch types;
 *   - every function returns APM_OK (0) or a negative apm_status and NEVER
 *     calls exit(); apm_last_error() gives the message;
 *   - the library owns all device memory it allocates; caller buffers are
 *     borrowed for the duration of the call;
 *   - counts are uint64 (the reference uses int, src/sequential.c:32); values
 *     are identical whenever they fit in int;
 *   - one apm_ctx per host thread (thread-compatible, not thread-safe);
 *   - there is NO CPU fallback: without a usable HIP device every compute
 *     entry point fails with APM_ERR_NO_DEVICE.
 *
 * Semantics (pinned by tests/golden/golden.json, produced by the reference):
 *   counts[i] = #{ j in [0, n-k) : dist(pattern_i[0:size], text[j:j+size]) <= k },
 *   size = min(m_i, n-j), dist = square global unit-cost Levenshtein distance,
 *   bytes compared verbatim (newlines are ordinary bytes).
 */
#ifndef APM_H
#define APM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APM_ABI_VERSION 1

typedef enum apm_status {
    APM_OK = 0,
    APM_ERR_INVALID = -1,     /* bad argument (NULL, negative k, empty pattern ...) */
    APM_ERR_NO_DEVICE = -2,   /* no HIP device / device index out of range */
    APM_ERR_HIP = -3,         /* a HIP runtime call failed */
    APM_ERR_IO = -4,          /* open/read failed (apm_count_file) */
    APM_ERR_NOMEM = -5,
    APM_ERR_UNSUPPORTED = -6, /* e.g. pattern longer than APM_MAX_PATTERN_LEN,
                                 or a kernel variant that cannot run this (m,k) */
    APM_ERR_COMM = -7,        /* RCCL failure in single-process multi-GPU mode */
    APM_ERR_STATE = -8        /* call order (patterns not set, ...) */
} apm_status;

/* Kernel variants.  Every variant returns the SAME exact counts; they differ
 * in how many DP cells they really evaluate.
 *   WAVEFRONT  full m x m DP, lanes own pattern rows, anti-diagonal sweep with
 *              DPP/__shfl column passing (the kernel BASELINE.json names)
 *   BITPAR     full m x m DP, Myers/Hyyro bit-vector columns (32 DP cells per
 *              integer op), exact distance: one window per lane with columns of
 *              up to 32 words (m <= 1024), one window per wavefront beyond
 *              (m <= 4096; the pattern's distinct bytes x 64 or 128 words of Eq
 *              rows must fit 60 KiB of LDS, else GENERIC)
 *   BANDED     exact for the predicate dist<=k: only diagonals |x-y|<=k/2,
 *              early exit, candidates pre-filtered by pigeonhole sub-keys looked up
 *              in LDS tables (needs m<=512, k<=7, m/(k+1)>=4)
 *   NFA        exact for the predicate dist<=k, no filter: the k-error automaton over
 *              the diagonals |x-y|<=k/2, 32 consecutive window starts per lane as
 *              the bits of a word (needs m + k/2 <= 32, k <= 7, <= 16 distinct
 *              pattern bytes); AUTO's choice for short patterns with many errors,
 *              whose pieces are too short for BANDED's filter
 *   GENERIC    literal one-column DP per lane, any m, handles truncated tails
 *   AUTO       fastest applicable exact variant per pattern (default)        */
typedef enum apm_kernel {
    APM_KERNEL_AUTO = 0,
    APM_KERNEL_GENERIC = 1,
    APM_KERNEL_WAVEFRONT = 2,
    APM_KERNEL_BITPAR = 3,
    APM_KERNEL_BANDED = 4,
    APM_KERNEL_NFA = 5
} apm_kernel;

#define APM_MAX_PATTERN_LEN 65535
#define APM_MAX_PATTERNS 65536

typedef struct apm_ctx apm_ctx;

/* Filled by the counting calls (host wall-clock + HIP-event times). */
typedef struct apm_timing {
    double total_ms;        /* host wall clock of the call */
    double h2d_ms;          /* host->device copies (0 for device-resident text) */
    double kernel_ms;       /* HIP events around all kernels of the call, max over devices */
    double reduce_ms;       /* cross-device count reduction */
    double main_kernel_ms;  /* HIP events around the dominant scan kernel(s) only */
    uint64_t text_bytes;    /* bytes of text scanned (sum over devices, halos included) */
    uint64_t windows;       /* (position, pattern) pairs decided */
    double cells_algorithmic; /* sum over windows of size^2 (the BASELINE metric's numerator) */
    double cells_evaluated;   /* DP cells the chosen kernels really computed (estimate for
                                 data-dependent early exit: upper bound) */
    int n_devices;
    int n_launches;         /* scan-kernel launches in the call */
} apm_timing;

/* ---- device probe: replaces getDeviceCount/setDevice (src/cuda_utils.cu:10-35) ---- */
int apm_device_count(void);                  /* >=0, or negative apm_status */
int apm_abi_version(void);

/* ---- context ---- */
/* Single process driving devices 0..n_devices-1 (the CLI's mode): text is
 * sharded over the devices with an (m_max-1)-byte halo and partial counts are
 * summed with one RCCL all-reduce (falls back to a host sum of the P-vectors
 * if librccl cannot be loaded).  n_devices<=0: all visible devices. */
int apm_create(apm_ctx **ctx, int n_devices);
/* One device only (the one-process-per-GPU mode used under torchrun). */
int apm_create_on_device(apm_ctx **ctx, int device_id);
void apm_destroy(apm_ctx *ctx);
const char *apm_last_error(const apm_ctx *ctx); /* ctx may be NULL: last create error */

/* Use an existing HIP stream (hipStream_t passed as void*) for all work of a
 * single-device context, e.g. torch's current stream.  NULL is HIP's null
 * (legacy default) stream; APM_STREAM_OWN restores the context's own stream.
 * The context owns device scratch (candidate list, hit masks, block-distribution counters, staging) that consecutive calls reuse in
 * stream order: when the stream is changed while earlier calls may still run, the caller orders the two streams
 * (event or synchronisation) first. */
#define APM_STREAM_OWN ((void *)(intptr_t)-1)
int apm_set_stream(apm_ctx *ctx, void *hip_stream);

/* ---- patterns: replaces the pattern/size uploads of initializeGPU
 *      (src/database_over_ranks.cu:157-169) and invoke_kernel (:79-94) ---- */
/* pat[i] need not be NUL terminated; len[i] >= 1; k >= 0. */
int apm_set_patterns(apm_ctx *ctx, int n_patterns, const char *const *pat,
                     const int *len, int k);
int apm_set_kernel(apm_ctx *ctx, int kernel /* apm_kernel */);
/* How a multi-device context (apm_create, more than one device) divides the work.
 *   APM_PARTITION_TEXT      (default) the text is cut into owner ranges with a halo, every device scans its range for
 *                           all patterns, the partial counts are summed (RCCL all-reduce): the replacement of the
 *                           reference's DB_OVER_RANKS (src/database_over_ranks.c:141-195), without its seam over-count
 *   APM_PARTITION_PATTERNS  the pattern list is cut into contiguous slices, every device scans the WHOLE text for its
 *                           slice, nothing is reduced: the replacement of PATTERNS_OVER_RANKS
 *                           (src/patterns_over_ranks.c:160-182); pays when the text is short and the patterns are many
 * Counts are the same either way.  No effect on single-device contexts.  The device-pointer entry points
 * (apm_count_shard_device, apm_set_stream) remain single-device calls. */
#define APM_PARTITION_TEXT 0
#define APM_PARTITION_PATTERNS 1
int apm_set_partition(apm_ctx *ctx, int partition);

/* ---- whole-text counting: replaces invoke_kernel+write_kernel_result and
 *      initializeGPU+getGPUResult.  counts[n_patterns], host, overwritten. ---- */
int apm_count_buffer(apm_ctx *ctx, const uint8_t *text, uint64_t n, uint64_t *counts);
/* 64-bit chunked file ingest (replaces read_input_file, src/utils.c:12-68). */
int apm_count_file(apm_ctx *ctx, const char *path, uint64_t *counts);

/* Match positions (no reference equivalent: the reference only counts, src/sequential.c:138-140).
 * Start offsets j of the windows of pattern `pattern_index` (of the current pattern set) with
 * dist <= k, ascending, for the whole text.  *n_found = total number of matches; at most `capacity`
 * positions are written (if *n_found > capacity they are an arbitrary subset: retry with a larger
 * buffer).  Runs the one pattern through the full-DP kernels; the pattern set is left unchanged. */
int apm_find_buffer(apm_ctx *ctx, const uint8_t *text, uint64_t n, int pattern_index, uint64_t *positions,
                    uint64_t capacity, uint64_t *n_found);

/* ---- shard-level API (device-resident text, asynchronous) ----
 * d_text holds the bytes of global positions [text_off, text_off+text_len) on
 * the context's device.  Counts every window whose START j lies in
 * [own_begin, own_end) ∩ [0, n_total-k); windows running past n_total are
 * truncated exactly as the reference does at the END OF THE WHOLE TEXT only
 * (src/sequential.c:131-134) -- never at a shard end (that is the
 * reference's DB_OVER_RANKS over-count, src/database_over_ranks.c:339-343).
 * Requires text_off <= own_begin and
 *          text_off+text_len >= min(n_total, own_end + m_max - 1).
 * d_counts: device array of n_patterns uint64, ADDED into (caller zeroes it).
 * Work is enqueued on the context's stream.  In the steady state the call does not synchronise with the host; the
 * FIRST call with a pattern set, and a call that needs larger scratch than any before it (the sieve's candidate
 * list grows with text_len; the GENERIC kernel's columns), allocate device memory and may synchronise the stream.
 * Readable padding: the scan kernels fetch 16 bytes at a time, so the memory behind d_text must be readable up to
 * the next 16-byte boundary past d_text + text_len, and (for a d_text that is not 16-byte aligned) back to the
 * 16-byte boundary in front of it -- both lie inside any hipMalloc'ed block if the block is 16 bytes larger than
 * the text.  Those bytes are never part of a counted window. */
int apm_count_shard_device(apm_ctx *ctx, const void *d_text, uint64_t text_off,
                           uint64_t text_len, uint64_t n_total,
                           uint64_t own_begin, uint64_t own_end,
                           uint64_t *d_counts);
/* Owner-computes partition helper: start positions [0, max(0,n_total-k)) cut
 * into n_shards contiguous ranges with 16-byte aligned interior boundaries. */
int apm_shard_range(uint64_t n_total, int k, int shard, int n_shards,
                    uint64_t *own_begin, uint64_t *own_end);

/* ---- synthetic text (benchmarks): deterministic counter-based DNA ----
 * byte i = "ACGT"[(splitmix64(seed ^ (i>>5)) >> (2*(i&31))) & 3].
 * Device-side fill of d_dst[0:len) with global positions [global_off, +len). */
int apm_synth_fill_device(apm_ctx *ctx, void *d_dst, uint64_t global_off,
                          uint64_t len, uint64_t seed);
/* Same bytes on the host (used to derive patterns without a device round trip). */
void apm_synth_fill_host(uint8_t *dst, uint64_t global_off, uint64_t len, uint64_t seed);

/* Generate n bytes on the device(s), scan them, return final counts. */
int apm_count_synthetic(apm_ctx *ctx, uint64_t n, uint64_t seed, uint64_t *counts);

/* ---- introspection ---- */
/* apm_count_shard_device brackets its kernels with hipEventRecord on the stream (what
 * apm_get_timing reports).  enabled=0 drops those records from the launch path. */
int apm_set_timing(apm_ctx *ctx, int enabled);
int apm_get_timing(const apm_ctx *ctx, apm_timing *out);
/* Per-launch times of the last counting call on a single-device context (timing enabled): HIP events recorded on
 * the launch stream right behind every scan-kernel launch.  Writes up to `max` durations (ms) and, if labels is
 * not NULL, a static string naming each launch ("sieve", "verify", "tile", "stream", "bitpar", ...); returns the
 * number written (>= 0) or a negative apm_status.  Synchronises with the last launch. */
int apm_get_launch_times(const apm_ctx *ctx, int max, double *ms, const char **labels);
/* Named statistics of the plan / the last counting call on device 0 (introspection for benchmarks and DESIGN.md):
 * "sieve_on", "sieve_stride" (1 or 8), "sieve_rate" (expected hits per lookup), "sieve_fused" (last call used the fused
 * kernel), "sieve_clist" (the last sieve pass handed its survivors over as a candidate list), "sieve_mask_bytes" (bytes of that
 * hand-over: list entries + the mask rows of blocks that overflowed, or all mask rows without a list; synchronises),
 * "sieve_candidates" (the candidates handed over: runs a reduction kernel and synchronises with the stream), "verify_launches", "verify_image_bytes", "verify_blocks_per_cu",
 * "verify_threads".  Unknown names: APM_ERR_INVALID. */
int apm_get_stat(const apm_ctx *ctx, const char *name, double *value);
/* Kernel variant AUTO (or the forced variant) resolves to for pattern i. */
int apm_pattern_kernel(const apm_ctx *ctx, int i);
/* Device memory helpers so that a C host needs no HIP headers. */
int apm_device_alloc(apm_ctx *ctx, void **d_ptr, uint64_t bytes);
int apm_device_free(apm_ctx *ctx, void *d_ptr);
int apm_device_upload(apm_ctx *ctx, void *d_dst, const void *src, uint64_t bytes);
int apm_device_download(apm_ctx *ctx, void *dst, const void *d_src, uint64_t bytes);
int apm_device_memset(apm_ctx *ctx, void *d_dst, int value, uint64_t bytes);
int apm_synchronize(apm_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* APM_H */
