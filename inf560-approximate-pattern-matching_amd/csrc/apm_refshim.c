/*
 * apm_refshim.c -- the reference's own six extern "C" GPU entry points, implemented over the C ABI
 * (include/apm.h), so that the reference's host files link against libapm_hip.so UNMODIFIED:
 *
 *   getDeviceCount / setDevice            declared at /root/reference/src/main.c:18-19
 *                                         (CUDA form: src/cuda_utils.cu:10-35)
 *   invoke_kernel / write_kernel_result   declared at src/patterns_over_ranks.c:33-36
 *                                         (CUDA form: src/patterns_over_ranks.cu:75-134)
 *   initializeGPU / getGPUResult          declared at src/database_over_ranks.c:18-22
 *                                         (CUDA form: src/database_over_ranks.cu:137-205)
 *
 * Plain C, no HIP headers.  Semantics = what the reference's kernels compute, minus their races:
 *   invoke_kernel(buf, n, pat, m, k, &c)  ->  *c + #{ j in [0, n-k) : dist(pat[0:size], buf[j:j+size]) <= k },
 *       size = min(m, n-j)  (ComputeMatches, patterns_over_ranks.cu:33-69, without the unsynchronised ++);
 *   initializeGPU(...)  ->  for every pattern i < lastPatternAnalyzedByGPU the scan of searchPattern
 *       (database_over_ranks.cu:81-127): r in [indexStartMyPiece, end_i - k), end_i = indexFinishMyPieceWithoutExtra
 *       (+ m_i - 1 unless myRank is the last), windows truncated at end_i.  That per-rank truncation is the
 *       reference's DB_OVER_RANKS arithmetic (it over-counts at shard seams); the shim reproduces
 *       the reference's numbers, the engine's own sharded entry points (apm_count_*) do not have the problem.
 * Work is enqueued asynchronously where the reference's was (the host's OpenMP region overlaps it) and
 * collected by write_kernel_result / getGPUResult.
 */
#include "../../include/apm.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static apm_ctx *g_ctx;

static apm_ctx *shim_ctx(void) {
    if (!g_ctx && apm_create_on_device(&g_ctx, 0) != APM_OK) {
        fprintf(stderr, "apm: %s\n", apm_last_error(NULL));
        g_ctx = NULL;
    }
    return g_ctx;
}

static void shim_fail(const char *what) {
    fprintf(stderr, "apm: %s failed: %s\n", what, g_ctx ? apm_last_error(g_ctx) : apm_last_error(NULL));
}

/* ---- src/cuda_utils.cu:10-35 ---- */
void getDeviceCount(int *deviceCountPtr) {
    const int n = apm_device_count();
    if (n < 0) { /* the reference prints and exits on a runtime error (cuda_utils.cu:13-18) */
        printf("hipGetDeviceCount returned %d\n-> %s\n", n, apm_last_error(NULL));
        printf("Result = FAIL\n");
        exit(EXIT_FAILURE);
    }
    *deviceCountPtr = n;
}

void setDevice(int rank, int deviceCount) {
    (void)rank;
    if (deviceCount == 0) printf("There are no available device(s) that support HIP\n");
    else (void)shim_ctx(); /* device 0, as the reference (cuda_utils.cu:33) */
}

/* ---- src/patterns_over_ranks.cu:75-134 ---- */
typedef struct shim_job { /* what the "device pointer" handed back to the caller really points at */
    int base;             /* *local_matches at the time of the call */
    void *d_text;
    void *d_count;
    int failed;
} shim_job;

int *invoke_kernel(char *buf, int n_bytes, char *my_pattern, int pattern_length, int approx_factor, int *local_matches) {
    shim_job *job = (shim_job *)calloc(1, sizeof *job);
    if (!job) return NULL;
    job->base = local_matches ? *local_matches : 0;
    apm_ctx *ctx = shim_ctx();
    const char *pats[1] = {my_pattern};
    const int lens[1] = {pattern_length};
    const uint64_t n = n_bytes > 0 ? (uint64_t)n_bytes : 0;
    if (!ctx || apm_set_patterns(ctx, 1, pats, lens, approx_factor) != APM_OK ||
        apm_device_alloc(ctx, &job->d_text, n + 16) != APM_OK || apm_device_alloc(ctx, &job->d_count, 8) != APM_OK ||
        apm_device_upload(ctx, job->d_text, buf, n) != APM_OK || apm_device_memset(ctx, job->d_count, 0, 8) != APM_OK ||
        /* the whole of buf[0:n) is "the text": windows are cut at n exactly as the kernel cuts them at n_bytes */
        apm_count_shard_device(ctx, job->d_text, 0, n, n, 0, n, (uint64_t *)job->d_count) != APM_OK) {
        shim_fail("invoke_kernel");
        job->failed = 1;
    }
    return (int *)job; /* opaque to the caller, like the reference's device pointer */
}

void write_kernel_result(int *local_matches, int *d_local_matches) {
    shim_job *job = (shim_job *)d_local_matches;
    if (!job) return;
    uint64_t c = 0;
    if (!job->failed && g_ctx && apm_device_download(g_ctx, &c, job->d_count, 8) != APM_OK) shim_fail("write_kernel_result");
    if (local_matches) *local_matches = job->base + (int)c;
    if (g_ctx) {
        if (job->d_text) apm_device_free(g_ctx, job->d_text);
        if (job->d_count) apm_device_free(g_ctx, job->d_count);
    }
    free(job);
}

/* ---- src/database_over_ranks.cu:137-205 (result in a file-scope global there too, :18) ---- */
static void *g_db_text;
static int *g_db_init;    /* numberOfMatchesInitialized */
static int g_db_patterns, g_db_last;
/* one scan per group of equal-length patterns, enqueued by initializeGPU and collected by getGPUResult: the group's
   count vector on the device and the pattern slots its entries belong to */
typedef struct shim_group { void *d_counts; int *slot; int n; } shim_group;
static shim_group *g_db_groups;
static int g_db_n_groups;

static void shim_free_groups(apm_ctx *ctx) {
    for (int i = 0; i < g_db_n_groups; ++i) {
        if (ctx && g_db_groups[i].d_counts) apm_device_free(ctx, g_db_groups[i].d_counts);
        free(g_db_groups[i].slot);
    }
    free(g_db_groups);
    g_db_groups = NULL;
    g_db_n_groups = 0;
}

int initializeGPU(char *buf, int n_bytes, char **pattern, int nb_patterns, int lastPatternAnalyzedByGPU, int *sizePatterns,
                  int indexFinishMyPieceWithoutExtra, int myRank, int numberProcesses, int indexStartMyPiece,
                  int approx_factor, int *numberOfMatchesInitialized) {
    apm_ctx *ctx = shim_ctx();
    g_db_patterns = nb_patterns;
    g_db_last = lastPatternAnalyzedByGPU < nb_patterns ? lastPatternAnalyzedByGPU : nb_patterns;
    free(g_db_init);
    g_db_init = (int *)malloc(sizeof(int) * (size_t)(nb_patterns > 0 ? nb_patterns : 1));
    for (int i = 0; i < nb_patterns; ++i) g_db_init[i] = numberOfMatchesInitialized ? numberOfMatchesInitialized[i] : 0;
    if (!ctx || g_db_last <= 0) return 1;
    const uint64_t n = n_bytes > 0 ? (uint64_t)n_bytes : 0;
    if (g_db_text) apm_device_free(ctx, g_db_text), g_db_text = NULL;
    shim_free_groups(ctx);
    if (apm_device_alloc(ctx, &g_db_text, n + 16) != APM_OK || apm_device_upload(ctx, g_db_text, buf, n) != APM_OK) {
        shim_fail("initializeGPU");
        return 1;
    }
    /* searchPattern treats buf[0:end_i) as the whole text of pattern i (database_over_ranks.cu:81-97); end_i depends
       on m_i only through "+ m_i - 1", so the patterns are scanned in groups of equal length.  Nothing is downloaded
       here: like the reference's launch (database_over_ranks.cu:180-189) the call returns with the scans enqueued and
       the host's own share of the patterns (the OpenMP region of database_over_ranks.c) runs beside them; getGPUResult
       waits.  (With several length groups the plan of group g is replaced while building group g + 1, which waits for
       group g's kernels: only the last group overlaps the host then.) */
    char *done = (char *)calloc((size_t)g_db_last, 1);
    g_db_groups = (shim_group *)calloc((size_t)g_db_last, sizeof *g_db_groups);
    for (int i = 0; i < g_db_last && done && g_db_groups; ++i) {
        if (done[i]) continue;
        const int m = sizePatterns[i];
        int n_grp = 0;
        for (int j = i; j < g_db_last; ++j) n_grp += (!done[j] && sizePatterns[j] == m);
        const char **pats = (const char **)malloc(sizeof(char *) * (size_t)n_grp);
        int *lens = (int *)malloc(sizeof(int) * (size_t)n_grp), *slot = (int *)malloc(sizeof(int) * (size_t)n_grp);
        void *d_grp = NULL;
        int g = 0;
        for (int j = i; j < g_db_last; ++j)
            if (!done[j] && sizePatterns[j] == m) { pats[g] = pattern[j]; lens[g] = m; slot[g++] = j; done[j] = 1; }
        long end = indexFinishMyPieceWithoutExtra;
        if (myRank != numberProcesses - 1) end += m - 1;
        if (end > (long)n) end = (long)n;
        const long start = indexStartMyPiece > 0 ? indexStartMyPiece : 0;
        int ok = end > start && approx_factor >= 0;
        if (ok) {
            ok = apm_set_patterns(ctx, n_grp, pats, lens, approx_factor) == APM_OK &&
                 apm_device_alloc(ctx, &d_grp, 8 * (uint64_t)n_grp) == APM_OK &&
                 apm_device_memset(ctx, d_grp, 0, 8 * (uint64_t)n_grp) == APM_OK &&
                 apm_count_shard_device(ctx, g_db_text, 0, (uint64_t)end, (uint64_t)end, (uint64_t)start, (uint64_t)end,
                                        (uint64_t *)d_grp) == APM_OK;
            if (!ok) shim_fail("initializeGPU");
        }
        if (ok) {
            g_db_groups[g_db_n_groups].d_counts = d_grp;
            g_db_groups[g_db_n_groups].slot = slot;
            g_db_groups[g_db_n_groups].n = n_grp;
            ++g_db_n_groups;
        } else {
            if (d_grp) apm_device_free(ctx, d_grp);
            free(slot);
        }
        free(pats); free(lens);
    }
    free(done);
    return 1; /* (the reference returns 1 always, database_over_ranks.cu:190) */
}

int *getGPUResult(int nb_patterns) {
    int *out = (int *)malloc(sizeof(int) * (size_t)(nb_patterns > 0 ? nb_patterns : 1)); /* caller-owned, as the reference's */
    if (!out) return NULL;
    for (int i = 0; i < nb_patterns; ++i) out[i] = (g_db_init && i < g_db_patterns) ? g_db_init[i] : 0;
    for (int gi = 0; g_ctx && gi < g_db_n_groups; ++gi) { /* the downloads wait for the scans initializeGPU enqueued */
        const shim_group *grp = &g_db_groups[gi];
        uint64_t *c = (uint64_t *)calloc((size_t)grp->n, 8);
        if (c && apm_device_download(g_ctx, c, grp->d_counts, 8 * (uint64_t)grp->n) == APM_OK) {
            for (int g = 0; g < grp->n; ++g)
                if (grp->slot[g] < nb_patterns) out[grp->slot[g]] += (int)c[g];
        } else {
            shim_fail("getGPUResult");
        }
        free(c);
    }
    return out;
}
