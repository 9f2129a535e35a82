/*
 * apm_runtime.hip -- implementation of the C ABI in include/apm.h.
 *
 * Host-side replacement for the reference's dispatch layer around its GPU shim:
 *   MPI master/worker + shard bounds   /root/reference/src/database_over_ranks.c:141-195
 *   GPU shims                          /root/reference/src/*.cu (see include/apm.h)
 * Design: owner-computes text sharding with an (m_max-1)-byte halo, truncation
 * only at the end of the WHOLE text, counts summed with one RCCL
 * all-reduce (single-process mode) or by the caller's collective
 * (one-process-per-GPU mode: apm_count_shard_device + torch.distributed).
 *
 * There is no CPU fallback in this file by design.
 */
#include "../../include/apm.h"
#include "apm_internal.h"
#include "apm_sieve.h"
#include "apm_core.h"

#include <algorithm>
#include <atomic>
#include <functional>
#include <mutex>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

// --------------------------------------------------------------------------
// internal kernels defined here (tiny)
// --------------------------------------------------------------------------
// statistics: number of set bits in the sieve's hit masks
__global__ void apm_popcount_kernel(const uint32_t *w, unsigned long long n, unsigned long long *sum) {
    unsigned long long acc = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        acc += (unsigned long long)__popc(w[i]);
    for (int d = 32; d; d >>= 1) acc += __shfl_down(acc, d, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(sum, acc);
}

// the same over the rows of the listed blocks, plus the entries of the candidate list's regions (the statistics of a pass
// that ran with the list: the other rows were never written)
__global__ void apm_popcount_listed_kernel(const uint32_t *w, const uint32_t *blist, const uint32_t *n_listed, const uint32_t *clist_cnt, int regions, unsigned long long *sum) {
    unsigned long long acc = 0;
    const unsigned long long n = (unsigned long long)*n_listed * 64ull, t0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long i = t0; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        acc += (unsigned long long)__popc(w[(unsigned long long)blist[i >> 6] * 64ull + (i & 63ull)]);
    for (unsigned long long i = t0; i < (unsigned long long)regions; i += (unsigned long long)gridDim.x * blockDim.x) acc += clist_cnt[i];
    for (int d = 32; d; d >>= 1) acc += __shfl_down(acc, d, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(sum, acc);
}

// k >= m: every window start matches (the DP never exceeds m); one launch adds the window count to all of them
__global__ void apm_add_const_kernel(unsigned long long *counts, const int *idx, int n, unsigned long long v) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) atomicAdd(&counts[idx[i]], v);
}

namespace {

thread_local std::string g_create_error;

using clk = std::chrono::steady_clock;
double ms_since(clk::time_point t0) {
    return std::chrono::duration<double, std::milli>(clk::now() - t0).count();
}

struct PatternInfo {
    std::string bytes;
    int m = 0;
    int kernel = APM_KERNEL_BITPAR; // resolved variant, or -1 for "k >= m: every window matches"
};
constexpr int KERNEL_TRIVIAL = -1;

struct TiledLaunch {       // host description of one tiled scan launch
    int kind = 0;          // APM_KERNEL_BITPAR | APM_KERNEL_WAVEFRONT
    std::vector<ApmPatDesc> descs;
    std::vector<uint8_t> bytes;
    std::vector<uint32_t> tables;
    uint8_t lut[256];
    std::vector<ApmKey> keys;         // BANDED: sub-keys
    std::vector<uint16_t> piece_off;  // BANDED: piece offsets, per pattern contiguous
    std::vector<uint16_t> table;      // BANDED: nb x 8 16-bit tags
    std::vector<uint16_t> table_kid;  // BANDED: nb x 8 key ids
    std::vector<uint32_t> ovf;        // BANDED: {tag, kid16} pairs
    std::vector<uint32_t> kinfo;      // BANDED: per key pat | off<<12 | piece<<21
    std::vector<uint32_t> pinfo;      // BANDED: per pattern {byte_off | m<<16, aux_off}
    std::vector<uint8_t> image;       // BANDED: LDS image (bytes | table | kids | ovf | kinfo | pinfo)
    int o_tab = 0, o_kid = 0, o_ovf = 0, o_kinfo = 0, o_pinfo = 0, o_next = 0, o_poff = 0;
    int o_bmp = 0, code_shift = 1; // per-position classes: key bitmap over 2-bit byte codes (leads the image)
    int o_pat = 0;                 // pattern bytes inside the image
    int o_kext = 0;                // per-position classes: packed pre-check record per key
    int key_len = 0, stride = 0;      // BANDED: (16,16), (8,8) or (8,1)
    bool sieved = false;              // BANDED per-position launch fed by the shared sieve pass (ctx->sieve)
    int nb = 0, lg_nb = 0, qcap = 0;
    int a_max = 0;                    // BANDED: largest key offset
    int blocks_per_cu[3] = {0, 0, 0}; // BANDED: resident workgroups per CU (occupancy query, cached) [tile, tile+dma, stream]
    int m_max = 0, m_min = 0, tile = 0;
};

/* Largest LDS image of a verify launch (bytes): one 512-thread workgroup with its wave buffers still fits a CU.  There is
 * no density limit on the key set any more: measured on 64 MiB of DNA (tools/density_probe.py) the pipeline beats the tile
 * kernels by 3.6x at 19 % of all code words set (800 patterns of 30, k = 3), by 90x at 48 % (200 x 16, k = 3). */
#define APM_VERIFY_IMAGE_MAX (112 * 1024)

struct VerifyLaunch {      // one apm_verify_kernel launch: a group of patterns and its LDS image (apm_sieve.hip)
    std::vector<ApmPatDesc> descs;    // m, index, byte_off (into bytes), aux_off = first key, w = number of keys
    std::vector<uint8_t> bytes;       // raw pattern bytes
    std::vector<uint32_t> kinfo;      // per key = nomination unit: pat | off << 12 | unit index << 21 (a pattern's units are consecutive keys)
    std::vector<uint32_t> kpart;      // per key: partner offset inside the pattern | partner length << 16
    std::vector<uint32_t> pinfo;      // per pattern: {byte_off | m << 16, id of its first key}
    std::vector<uint8_t> image;       // bitmap16 | prefix | r2s | slots | kext | pattern bytes
    int o_prefix = 0, o_r2s = 0, o_slots = 0, o_kext = 0, o_pat = 0, o_masks = 0, o_kinfo = 0, o_pinfo = 0, o_rc = 0;
    int m_max = 0, m_min = 0;
    int blocks_per_cu = 0, threads = 256; // launch geometry (occupancy query, cached)
    int fused_blocks_per_cu = 0, fused_threads = 0; // the same for the fused form (threads < 0: it does not fit a CU)
    // stride 1 with the code filter: the launch has a SIEVE PASS OF ITS OWN -- the 18-bit bitmap of its keys alone and the
    // code-filter image over its key numbering (ApmSieve2Args::cf_image: tbl | rrec | lrec); empty: the set's shared sieve
    std::vector<uint32_t> bitmap18;
    std::vector<uint8_t> cf_image;
    int cf_o_rrec = 0, cf_o_lrec = 0;
    int cf_threads = 0, cf_blocks_per_cu = 0; // launch geometry (occupancy query, cached; threads < 0: does not fit a CU)
};

struct SievePlan {         // ONE text pass (apm_sieve2_kernel) for every per-position key of the pattern set
    bool on = false;
    int stride = 1;                   // 1: every position (per-position keys present); 8: sampled (all pieces >= 15 bytes)
    int code_shift = 1;
    int m_max = 0;
    double rate = 0;                  // expected hits per lookup on uniform codes (bitmap density)
    std::vector<uint32_t> bitmap;     // 32 KiB over the 18-bit code words of 9-byte windows: dword x & 8191, bit x >> 13
    std::vector<VerifyLaunch> launches;
    double weak_frac = 0;             // share of the key words that belong to units the code filter cannot add to
    bool per_launch_sieve = false;    // stride 1 with the code filter: every verify launch is preceded by its own sieve pass (VerifyLaunch::bitmap18)
};

struct DevVerify {
    uint32_t *d_bmp18 = nullptr;   // VerifyLaunch::bitmap18
    uint8_t *d_cf = nullptr;       // VerifyLaunch::cf_image
    ApmPatDesc *d_descs = nullptr;
    uint8_t *d_image = nullptr;
    uint32_t *d_kinfo = nullptr;
    uint32_t *d_pinfo = nullptr;
    uint32_t *d_kpart = nullptr;
};

struct GenericGroup {      // patterns scanned by the generic kernel, one launch (grid.y = pattern)
    std::vector<ApmPatDesc> descs; // byte_off into the all-pattern pool
    int m_max = 0;
};

struct DevTiled {
    ApmPatDesc *d_descs = nullptr;
    uint8_t *d_bytes = nullptr;
    uint32_t *d_tables = nullptr;
    uint8_t *d_lut = nullptr;
    uint8_t *d_image = nullptr;
};

struct DeviceState {
    int dev = -1;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    uint8_t *d_allpat = nullptr;              // every pattern's raw bytes, concatenated
    ApmPatDesc *d_tail_descs = nullptr;       // tails of tiled-kernel patterns with m > 128 (generic kernel)
    ApmPatDesc *d_stail_descs = nullptr;      // tails of tiled-kernel patterns with m <= 128 (tail kernel)
    ApmPatDesc *d_wtail_descs = nullptr;      // ... with 128 < m <= 512 (wide tail kernel)
    ApmPatDesc *d_xtail_descs = nullptr;      // ... with 512 < m <= 1024 (32-word tail kernel)
    ApmPatDesc *d_long_descs = nullptr;       // generic full-scan patterns
    int *d_trivial = nullptr;                 // indices of the patterns with k >= m
    std::vector<DevTiled> tiled;
    unsigned long long *d_counts = nullptr;   // P
    uint16_t *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    unsigned long long *d_pos_out = nullptr;   // apm_find_buffer: match positions (cap entries) + 1 counter
    unsigned long long *d_pos_count = nullptr;
    unsigned long long pos_cap = 0;
    uint8_t *d_text = nullptr;
    size_t text_cap = 0;
    hipEvent_t ev_stage[32] = {};             // apm_count_file: staging buffer b copied out (this device's stream)
    uint32_t *d_sieve_bmp = nullptr;           // sieve bitmap of the whole set (32 KiB)
    std::vector<DevVerify> verify;
    uint32_t *d_masks = nullptr;               // the sieve's hit masks: one dword per lane and 4 KiB block (n / 16 bytes)
    size_t masks_cap = 0;                      // dwords
    int64_t last_mask_blocks = 0;              // blocks the last call's sieve wrote (statistics)
    uint32_t *d_work = nullptr;                // block-distribution counters of the verify / fused launches (ApmVerifyArgs::work)
    int work_epoch = 0;
    uint32_t *d_blist = nullptr;               // the sieve's list of non-empty blocks (one dword per 4 KiB block at most)
    size_t blist_cap = 0;
    int sieve_epoch = 0;                       // which of the two list counters the next sieve launch counts in
    uint32_t *d_clist = nullptr;               // the sieve's candidate list (ApmSieve2Args::clist) and its per-region counts
    uint32_t *d_clist_cnt = nullptr;
    size_t clist_cap = 0;                      // entries allocated
    int last_clist_regions = 0;                // the last sieve pass ran with the list: its regions and block-list counter (statistics)
    const uint32_t *last_blist_ctr = nullptr;
    unsigned long long *d_stats = nullptr;     // 8 counters (statistics kernel; measurement build: verify counters)
    bool last_fused = false;                   // the last call used the fused form of the pipeline
    hipEvent_t ev_start = nullptr, ev_kstart = nullptr, ev_mstart = nullptr, ev_mstop = nullptr, ev_stop = nullptr;
    bool events_recorded = false;
    // per-launch event stamps (apm_get_launch_times): stamp i is recorded right behind scan launch i, so the time
    // between two stamps is one launch as the stream saw it (the first one is measured from ev_mstart)
    static constexpr int MAX_STAMPS = 32;
    hipEvent_t ev_launch[MAX_STAMPS] = {};
    const char *launch_label[MAX_STAMPS] = {};
    int n_stamps = 0;
    // per-call accounting
    uint64_t text_bytes = 0;
    int launches = 0;
};

struct RcclApi {
    void *handle = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    std::vector<void *> comms;
    bool ready = false;
};

} // namespace

struct apm_ctx {
    std::vector<DeviceState> devs;
    std::vector<PatternInfo> pats;
    std::vector<TiledLaunch> tiled;
    SievePlan sieve;
    GenericGroup tails;   // tiled-kernel patterns with m > 128: tails by the generic kernel
    GenericGroup stails;  // tiled-kernel patterns with m <= 128: tails by the bit-vector tail kernel
    GenericGroup wtails;  // ... with 128 < m <= 512: by its 16-word form
    GenericGroup xtails;  // ... with 512 < m <= 1024: by its 32-word form (apm_bitlong.hip)
    GenericGroup longs;   // patterns scanned fully by the generic kernel
    std::vector<int> trivial; // indices with k >= m
    std::vector<uint8_t> allpat;
    int k = 0;
    int kernel = APM_KERNEL_AUTO;
    int m_max = 0; // over non-trivial patterns
    bool patterns_set = false;
    bool timing_on = true;   // hipEvent bracketing of every call (apm_set_timing)
    bool find_active = false; // apm_find_buffer in progress: kernels also push match positions
    std::string err;
    apm_timing timing{};
    RcclApi rccl;
    bool multi = false; // created by apm_create (single process, >=1 devices)
    // PATTERN-SHARDED partition (apm_set_partition): one single-device child context per device, child g holds the patterns
    // [pat_first[g], pat_first[g + 1]) and scans the WHOLE text; the count vectors are disjoint, nothing is reduced
    int partition = APM_PARTITION_TEXT;
    std::vector<apm_ctx *> children;
    std::vector<int> pat_first; // children.size() + 1 entries
    // apm_count_file: pinned staging ring (kept for the life of the context) and its "copied out" events
    static constexpr int N_STAGE = 32;                 // two per reader thread, allocated on first use
    static constexpr size_t STAGE_BYTES = (size_t)8 << 20;
    uint8_t *stage[N_STAGE] = {};
};

namespace {

int fail(apm_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    static std::mutex mu; // (the per-device staging threads of count_sharded may fail side by side)
    std::lock_guard<std::mutex> lock(mu);
    if (ctx) ctx->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return fail(ctx, APM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                        __FILE__, __LINE__);                                                    \
    } while (0)

// bookkeeping behind every scan-kernel launch: count it and, with timing on, stamp the stream
int note_launch(apm_ctx *ctx, DeviceState &ds, const char *label) {
    ds.launches++;
    if (!ctx->timing_on || ds.n_stamps >= DeviceState::MAX_STAMPS) return APM_OK;
    hipEvent_t &e = ds.ev_launch[ds.n_stamps];
    if (!e) HIP_TRY(ctx, hipEventCreate(&e));
    HIP_TRY(ctx, hipEventRecord(e, ds.stream));
    ds.launch_label[ds.n_stamps++] = label;
    return APM_OK;
}

// ---------------------------------------------------------------------------
// plan: which kernel scans which pattern, in which launch
// ---------------------------------------------------------------------------
int wavefront_rows_per_lane(int m) {
    // minimise VALU work per window: steps (m + Lm - 1) x (overhead + 4R ops) / S windows per sweep
    int best_r = 0;
    double best = 1e30;
    for (int r : {1, 2, 4}) {
        const int lm = (m + r - 1) / r;
        if (lm > 64) continue;
        const int s = 64 / lm;
        const double cost = double(m + lm - 1) * (11.0 + 4.0 * r) / s;
        if (cost < best) { best = cost; best_r = r; }
    }
    return best_r; // 0: does not fit (m > 256)
}

int resolve_kernel(int forced, int m, int k, std::string *why) {
    if (forced == APM_KERNEL_AUTO) {
        if (k >= m) return KERNEL_TRIVIAL;
        if (m <= APM_BANDED_MAX_M && k <= APM_BANDED_MAX_K && m / (k + 1) >= APM_BANDED_MIN_PIECE) return APM_KERNEL_BANDED;
        if (k <= APM_NFA_MAX_K && m + k / 2 <= 32) return APM_KERNEL_NFA; // short and loose: the automaton over 32 window starts per lane (<= 16 distinct bytes: build_plan)
        if (m <= APM_BITPAR_MAX_M) return APM_KERNEL_BITPAR; // short or loose (BANDED's pigeonhole pieces too short), or long: bit-vector columns
        return APM_KERNEL_GENERIC;                           // m > 4096 only (and long patterns over big alphabets: build_plan)
    }
    switch (forced) {
    case APM_KERNEL_GENERIC: return APM_KERNEL_GENERIC;
    case APM_KERNEL_WAVEFRONT:
        if (m > APM_WAVEFRONT_MAX_M) { *why = "WAVEFRONT kernel supports pattern length <= 256"; return -100; }
        return APM_KERNEL_WAVEFRONT;
    case APM_KERNEL_BITPAR:
        if (m > APM_BITPAR_MAX_M) { *why = "BITPAR kernel supports pattern length <= 4096"; return -100; }
        return APM_KERNEL_BITPAR;
    case APM_KERNEL_NFA:
        if (k > APM_NFA_MAX_K || m + k / 2 > 32) { *why = "NFA kernel needs m + k/2 <= 32 and k <= 7"; return -100; }
        return APM_KERNEL_NFA;
    case APM_KERNEL_BANDED:
        if (m > APM_BANDED_MAX_M || k > APM_BANDED_MAX_K || m / (k + 1) < APM_BANDED_MIN_PIECE) {
            *why = "BANDED kernel needs m <= 512, k <= 7 and m/(k+1) >= 4 (pigeonhole keys of >= 4 bytes)";
            return -100;
        }
        return APM_KERNEL_BANDED;
    default: *why = "unknown kernel variant"; return -100;
    }
}

void free_device_plan(DeviceState &ds) {
    hipSetDevice(ds.dev);
    for (auto &t : ds.tiled) {
        if (t.d_descs) hipFree(t.d_descs);
        if (t.d_bytes) hipFree(t.d_bytes);
        if (t.d_tables) hipFree(t.d_tables);
        if (t.d_lut) hipFree(t.d_lut);
        if (t.d_image) hipFree(t.d_image);
    }
    ds.tiled.clear();
    if (ds.d_allpat) hipFree(ds.d_allpat), ds.d_allpat = nullptr;
    if (ds.d_tail_descs) hipFree(ds.d_tail_descs), ds.d_tail_descs = nullptr;
    if (ds.d_stail_descs) hipFree(ds.d_stail_descs), ds.d_stail_descs = nullptr;
    if (ds.d_wtail_descs) hipFree(ds.d_wtail_descs), ds.d_wtail_descs = nullptr;
    if (ds.d_xtail_descs) hipFree(ds.d_xtail_descs), ds.d_xtail_descs = nullptr;
    if (ds.d_long_descs) hipFree(ds.d_long_descs), ds.d_long_descs = nullptr;
    if (ds.d_trivial) hipFree(ds.d_trivial), ds.d_trivial = nullptr;
    if (ds.d_counts) hipFree(ds.d_counts), ds.d_counts = nullptr;
    if (ds.d_sieve_bmp) hipFree(ds.d_sieve_bmp), ds.d_sieve_bmp = nullptr;
    for (auto &v : ds.verify) {
        if (v.d_bmp18) hipFree(v.d_bmp18);
        if (v.d_cf) hipFree(v.d_cf);
        if (v.d_descs) hipFree(v.d_descs);
        if (v.d_image) hipFree(v.d_image);
        if (v.d_kinfo) hipFree(v.d_kinfo);
        if (v.d_pinfo) hipFree(v.d_pinfo);
        if (v.d_kpart) hipFree(v.d_kpart);
    }
    ds.verify.clear();
}

template <typename T>
int upload_vec(apm_ctx *ctx, T **dptr, const std::vector<T> &v) {
    *dptr = nullptr;
    const size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    HIP_TRY(ctx, hipMalloc((void **)dptr, bytes));
    if (!v.empty()) HIP_TRY(ctx, hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return APM_OK;
}

// Enter one per-position key into an 8 KiB presence bitmap over 8-byte code words (2-bit codes
// (b >> shift) & 3, byte z of the window in bits 2z..): the piece itself (its first min(len, 8) bytes) must be
// intact; what the window shows behind a piece shorter than 8 bytes is the text that follows it.  If the
// piece's partner of the pair pre-check lies there (forward partner), only continuations that can still pass
// the one-edit extension (apm_ext1_core16 semantics, bytes beyond the window = wildcards) are entered -- a
// superset of what the pre-check accepts, several times smaller than "every continuation", which is what a
// piece with its partner in front of it (or none) gets.
// Without the pair pre-check (band 0: k <= 1) a nomination is just "the key bytes match", and the dedup of the
// kernels relies on exactly that predicate -- so there only the key itself is entered, with every continuation.
// fn(xx) for every 16-bit code word xx the 8-byte window at the start of piece q may show (see above).
// pat = the pattern's bytes, poffs = its `pieces` piece offsets, m its length; plain_len = the key length used
// without the pair pre-check.
template <typename F>
void enum_key_windows(const uint8_t *pat, int m, const uint16_t *poffs, int pieces, int q, int plain_len, int shift, bool pairs, F fn) {
    auto piece_begin = [&](int qq) { return qq >= pieces ? m : (int)poffs[qq]; };
    auto code = [&](int y) { return (uint32_t)((pat[y] >> shift) & 3); };
    const int at = piece_begin(q);
    const int len = pairs ? piece_begin(q + 1) - at : plain_len; // (stride 1: the key starts the piece)
    const int vis = std::min(len, 8), ext = 8 - vis;
    uint32_t x = 0;
    for (int z = 0; z < vis; ++z) x |= code(at + z) << (2 * z);
    const int pq = q ^ 1;
    const bool forward = pairs && pq < pieces && pq > q;
    const int n = forward ? piece_begin(pq + 1) - piece_begin(pq) : 0;
    const int pa = at + len; // partner start (forward case)
    for (uint32_t p = 0; p < (1u << (2 * ext)); ++p) {
        bool ok = true;
        if (forward && ext > 0) {
            auto t = [&](int j) { return (p >> (2 * j)) & 3u; }; // visible text code j behind the piece
            int i = 0;
            while (i < n && i < ext && t(i) == code(pa + i)) ++i;
            if (!(i >= ext || i >= n - 1)) {
                ok = true; // substitution at i
                for (int j = i + 1; j < n && j < ext && ok; ++j) ok = t(j) == code(pa + j);
                if (!ok) {
                    ok = true; // pattern byte i has no text counterpart
                    for (int j = i + 1; j < n && j - 1 < ext && ok; ++j) ok = t(j - 1) == code(pa + j);
                }
                if (!ok) {
                    ok = true; // one extra text byte before pattern byte i
                    for (int j = i; j < n && j + 1 < ext && ok; ++j) ok = t(j + 1) == code(pa + j);
                }
            }
        }
        if (ok) fn(x | (p << (2 * vis)));
    }
}

void mark_key_windows(std::vector<uint8_t> &bmp, const TiledLaunch &L, const ApmKey &kk, int pieces, int shift, bool pairs) {
    const ApmPatDesc &dd = L.descs[kk.pat];
    enum_key_windows(L.bytes.data() + dd.byte_off, (int)dd.m, L.piece_off.data() + dd.aux_off, pieces, (int)kk.piece, L.key_len, shift, pairs,
                     [&](uint32_t xx) { bmp[xx & 8191u] |= (uint8_t)(1u << (xx >> 13)); });
}

// Plan of the sieve + verify pipeline for all BANDED patterns of the set (see build_plan): one sieve bitmap for the set,
// the patterns split into verify launches by LDS image size.  Leaves ctx->sieve.on false only when a launch does not fit
// the index formats (the splitting keeps clear of that).
int build_sieve_plan(apm_ctx *ctx, int stride) {
    SievePlan &S = ctx->sieve;
    S.stride = stride;
    const int P = (int)ctx->pats.size();
    const int pieces = ctx->k + 1;
    const bool pairs = ctx->k / 2 >= 1;
    std::vector<int> idx;
    for (int i = 0; i < P; ++i)
        if (ctx->pats[i].kernel == APM_KERNEL_BANDED) idx.push_back(i);
    // one code shift for the whole set: spread the pattern bytes over the four 2-bit codes as evenly as possible
    // (s = 1 separates A,C,G,T and a,c,g,t exactly)
    long best = -1;
    for (int sft = 0; sft < 7; ++sft) {
        long hist[4] = {0, 0, 0, 0};
        for (int i : idx)
            for (unsigned char c : ctx->pats[i].bytes) ++hist[(c >> sft) & 3];
        const long score = std::min(std::min(hist[0], hist[1]), std::min(hist[2], hist[3])) * 4 +
                           (hist[0] > 0) + (hist[1] > 0) + (hist[2] > 0) + (hist[3] > 0) + (sft == 1);
        if (score > best) { best = score; S.code_shift = sft; }
    }
    S.bitmap.assign(8192, 0u);
    std::vector<uint8_t> seen16(8192, 0); // union of the launches' 16-bit code words (byte x & 8191, bit x >> 13)
    std::vector<uint32_t> even18(8192, 0); // union of the units' 18-bit words over nine bytes (dword x & 8191, bit x >> 13)
    // nomination units of a pattern (apm_core.h, ApmUnit): per pair of pigeonhole pieces (A, B) either the two
    // piece units "A intact + B within one edit behind it" and "B intact + A within one edit in front of it", or --
    // when both pieces are short -- ONE pair unit "A+B within one edit"; whichever shows fewer 8-byte code words to
    // the sieve.  The unpaired last piece (even k) is a unit without partner; without the pair pre-check (k <= 1)
    // every piece is.  Every window with <= k edits has a unit whose predicate holds at the right text position.
    auto count_words = [&](const uint8_t *pat, const std::vector<ApmUnit> &us) {
        std::vector<uint32_t> w;
        for (const ApmUnit &u : us) apm_enum_unit_windows(pat, u, S.code_shift, [&](uint32_t x) { w.push_back(x); });
        std::sort(w.begin(), w.end());
        return (size_t)(std::unique(w.begin(), w.end()) - w.begin());
    };
    auto units_of = [&](const uint8_t *pat, int m) {
        std::vector<ApmUnit> us;
        auto a = [&](int q) { return q >= pieces ? m : (int)((int64_t)q * m / pieces); };
        for (int q = 0; q < pieces; q += pairs ? 2 : 1) {
            const int lenA = a(q + 1) - a(q);
            if (!pairs || q + 1 >= pieces) {
                us.push_back(ApmUnit{a(q), lenA, 0, 0, 0});
                continue;
            }
            const int lenB = a(q + 2) - a(q + 1);
            const std::vector<ApmUnit> by_piece = {ApmUnit{a(q), lenA, a(q + 1), lenB, 1}, ApmUnit{a(q + 1), lenB, a(q), lenA, 2}};
            const std::vector<ApmUnit> by_pair = {ApmUnit{a(q), 0, a(q), lenA + lenB, 1}};
            if (lenA + lenB <= 16 && lenA < 8 && lenB < 8 && count_words(pat, by_pair) < count_words(pat, by_piece)) us.push_back(by_pair[0]);
            else us.insert(us.end(), by_piece.begin(), by_piece.end());
        }
        return us;
    };
    static const int cf_env = getenv("APM_SIEVE_CF") ? atoi(getenv("APM_SIEVE_CF")) : 1;
    const bool cf_on = cf_env && stride == 1;
    double words_weak = 0, words_strong = 0; // key words of units the code filter can / cannot add to (see `weak` below)
    for (size_t pos = 0; pos < idx.size();) {
        VerifyLaunch V;
        std::vector<uint8_t> v_seen16(8192, 0);  // this launch's 16-bit code words / 18-bit words (as seen16 / even18 of the set)
        std::vector<uint32_t> v_even18(8192, 0);
        std::vector<ApmUnit> units; // per key, offsets relative to the pattern
        size_t n_words = 0;         // code words of the launch's units, counted per pattern (>= the distinct ones)
        for (; pos < idx.size(); ++pos) {
            const PatternInfo &pi = ctx->pats[idx[pos]];
            const std::vector<ApmUnit> us = units_of(reinterpret_cast<const uint8_t *>(pi.bytes.data()), pi.m);
            const size_t pw = stride == 8 ? us.size() * 8 : count_words(reinterpret_cast<const uint8_t *>(pi.bytes.data()), us);
            // the image must fit a CU's LDS beside the wave buffers of one workgroup, and the slot indices 15 bits:
            // bitmap + prefix (12 KiB), rank -> key and key lists (<= 2 + 2 bytes per word), key records, pattern bytes
            const size_t est = 12288 + 4 * (n_words + pw) + 8 * (V.kinfo.size() + us.size()) + 8 * (V.descs.size() + 1) + V.bytes.size() + (size_t)pi.m + 512;
            // (with the code filter the launch's sieve pass keeps 8 bytes per key word in LDS beside its 32 KiB bitmap: <= 80 KiB)
            if (!V.descs.empty() && (V.bytes.size() + (size_t)pi.m > 24576 || V.kinfo.size() + us.size() > (stride == 8 ? 2048u : 8192u) || V.descs.size() >= 4096 ||
                                     est > APM_VERIFY_IMAGE_MAX || n_words + pw >= 0x7000 || (cf_on && 8 * (n_words + pw) > 80 * 1024)))
                break;
            n_words += pw;
            ApmPatDesc d{};
            d.m = (uint32_t)pi.m;
            d.index = (uint32_t)idx[pos];
            d.byte_off = (uint32_t)V.bytes.size();
            d.aux_off = (uint32_t)V.kinfo.size(); // first key
            V.bytes.insert(V.bytes.end(), pi.bytes.begin(), pi.bytes.end());
            d.w = (uint32_t)us.size();
            for (size_t ui = 0; ui < us.size(); ++ui) {
                V.kinfo.push_back((uint32_t)V.descs.size() | ((uint32_t)us[ui].off << 12) | ((uint32_t)ui << 21));
                V.kpart.push_back((uint32_t)us[ui].poff | ((uint32_t)us[ui].plen << 16));
                units.push_back(us[ui]);
            }
            V.pinfo.push_back(d.byte_off | (d.m << 16));
            V.pinfo.push_back(d.aux_off);
            V.descs.push_back(d);
            V.m_max = std::max(V.m_max, pi.m);
            V.m_min = V.m_min ? std::min(V.m_min, pi.m) : pi.m;
        }
        while (V.bytes.size() % 16) V.bytes.push_back(0);
        // (code word, key) pairs in rank order: the verify kernel keeps the words as dword x & 2047, bit x >> 11
        auto rank_key = [](uint32_t x) { return ((x & 2047u) << 5) | (x >> 11); };
        std::vector<uint64_t> wk;
        std::vector<uint32_t> kext, krec;
        for (size_t kid = 0; kid < units.size(); ++kid) {
            const ApmUnit &u = units[kid];
            const ApmPatDesc &dd = V.descs[V.kinfo[kid] & 0xfffu];
            if (stride == 8) {
                // sampled: whatever the piece's position, one of its blocks [r, r+8), r = 0..7, starts at a multiple of 8 in
                // the text; the key-list payload carries r above the key id (11 bits)
                for (uint32_t r = 0; r < 8; ++r) {
                    uint32_t xx = 0;
                    for (int z = 0; z < 8; ++z) xx |= (uint32_t)((V.bytes[dd.byte_off + (uint32_t)u.off + r + (uint32_t)z] >> S.code_shift) & 3) << (2 * z);
                    wk.push_back(((uint64_t)rank_key(xx) << 32) | ((uint64_t)xx << 16) | (uint64_t)(kid | (r << 11)));
                }
            } else {
                const size_t wk0 = wk.size();
                apm_enum_unit_windows(V.bytes.data() + dd.byte_off, u, S.code_shift,
                                      [&](uint32_t xx) { wk.push_back(((uint64_t)rank_key(xx) << 32) | ((uint64_t)xx << 16) | (uint64_t)kid); });
                // a unit the code filter cannot judge any better than the 8-byte bitmap has: everything it would test lies inside the window
                const bool weak = (u.side == 0 && u.len <= 9) || (u.side == 1 && u.len == 0 && u.plen <= 9);
                (weak ? words_weak : words_strong) += (double)(wk.size() - wk0);
                // the sieve looks at NINE bytes where the key window starts at an even position: the unit's 18-bit words
                // (a ninth exact byte, or what one edit leaves of the partner there)
                apm_enum_unit_windows(V.bytes.data() + dd.byte_off, u, S.code_shift, [&](uint32_t x18) { even18[x18 & 8191u] |= 1u << (x18 >> 13); v_even18[x18 & 8191u] |= 1u << (x18 >> 13); }, 9);
            }
            // packed pre-check record: byte offset of the exact part in the pattern pool | its length << 16 |
            // partner length << 24 (31 = beyond 16) | side << 29
            kext.push_back((uint32_t)(dd.byte_off + (uint32_t)u.off) | (std::min<uint32_t>((uint32_t)u.len, 255u) << 16) |
                           ((u.plen > 16 ? 31u : (uint32_t)u.plen) << 24) | ((uint32_t)u.side << 29));
            uint32_t rx, ry;
            apm_cf_record(V.bytes.data() + dd.byte_off, u, S.code_shift, &rx, &ry);
            krec.push_back(rx);
            krec.push_back(ry);
        }
        std::sort(wk.begin(), wk.end());
        wk.erase(std::unique(wk.begin(), wk.end()), wk.end());
        std::vector<uint32_t> bmp16(2048, 0u);
        std::vector<uint16_t> prefix(2048, 0), r2s, slots;
        for (size_t i = 0; i < wk.size();) {
            size_t j = i;
            while (j < wk.size() && (wk[j] >> 32) == (wk[i] >> 32)) ++j;
            const uint32_t xx = (uint32_t)(wk[i] >> 16) & 0xffffu;
            bmp16[xx & 2047u] |= 1u << (xx >> 11);
            seen16[xx & 8191u] |= (uint8_t)(1u << (xx >> 13));
            v_seen16[xx & 8191u] |= (uint8_t)(1u << (xx >> 13));
            if (j - i == 1) {
                r2s.push_back((uint16_t)(0x8000u | (wk[i] & 0x7fffu)));
            } else {
                r2s.push_back((uint16_t)slots.size());
                for (size_t z = i; z < j; ++z) slots.push_back((uint16_t)((wk[z] & 0x7fffu) | (z + 1 == j ? 0x8000u : 0u)));
            }
            i = j;
        }
        if (slots.size() >= 0x8000 || V.kinfo.size() > 0x7fffu) return APM_OK; // (15-bit slot / key indices; the splitting above keeps clear of it)
        uint32_t run = 0;
        for (int w = 0; w < 2048; ++w) {
            prefix[w] = (uint16_t)run;
            run += (uint32_t)__builtin_popcount(bmp16[w]);
        }
        auto append = [&](const void *src, size_t bytes) {
            const size_t at = V.image.size();
            V.image.resize(at + ((bytes + 15) & ~(size_t)15), 0);
            if (bytes) memcpy(V.image.data() + at, src, bytes);
            return (int)at;
        };
        append(bmp16.data(), bmp16.size() * 4); // = 0
        V.o_prefix = append(prefix.data(), prefix.size() * 2);
        V.o_r2s = append(r2s.data(), r2s.size() * 2);
        V.o_slots = append(slots.data(), slots.size() * 2);
        V.o_kext = append(kext.data(), kext.size() * 4);
        {
            std::vector<uint8_t> masks(17 * 16, 0);
            for (int n = 0; n <= 16; ++n)
                for (int b = 0; b < n; ++b) masks[(size_t)n * 16 + (size_t)b] = 0xff;
            V.o_masks = append(masks.data(), masks.size());
        }
        V.o_pat = append(V.bytes.data(), V.bytes.size());
        V.o_kinfo = append(V.kinfo.data(), V.kinfo.size() * 4);
        V.o_pinfo = append(V.pinfo.data(), V.pinfo.size() * 4);
        // sampled sets of up to 128 units: the operands of the fused form's REGISTER COMPARE, ready made -- per (unit, offset r
        // of the sampled block inside its piece, half t of the lane's 16 bytes) the codes of the pattern bytes that face the
        // lane's bytes, packed like the text, and the mask of the code bits the piece covers (apm_verify_body packs them
        // out of the pattern bytes otherwise: five LDS reads and four packs per hit).  16 bytes per (unit, r).
        V.o_rc = 0;
        static const int rc_env = getenv("APM_FUSED_RC") ? atoi(getenv("APM_FUSED_RC")) : 1; // (A/B aid: 0 = pack the operands per hit)
        if (rc_env && stride == 8 && units.size() <= 128) {
            std::vector<uint32_t> rc(units.size() * 8 * 4, 0u);
            for (size_t kid = 0; kid < units.size(); ++kid) {
                const int at = (int)(kext[kid] & 0xffffu), len = (int)((kext[kid] >> 16) & 0xffu);
                for (int r = 0; r < 8; ++r)
                    for (int t = 0; t < 2; ++t) {
                        const int sh8 = 8 * t - r; // lane byte i <-> pattern pool byte at - sh8 + i
                        const int i0 = sh8 > 0 ? sh8 : 0, i1 = len + sh8 < 16 ? len + sh8 : 16;
                        uint32_t pc = 0, mask = 0;
                        for (int i = i0; i < i1; ++i) {
                            pc |= (uint32_t)((V.bytes[(size_t)(at - sh8 + i)] >> S.code_shift) & 3) << (2 * i);
                            mask |= 3u << (2 * i);
                        }
                        rc[((kid * 8 + (size_t)r) * 2 + (size_t)t) * 2] = pc;
                        rc[((kid * 8 + (size_t)r) * 2 + (size_t)t) * 2 + 1] = mask;
                    }
            }
            V.o_rc = append(rc.data(), rc.size() * 4);
        }
        S.m_max = std::max(S.m_max, V.m_max);
        // the launch's own sieve pass (stride 1 with the code filter): the bitmap of ITS keys -- built like the set's below -- and
        // the code-filter tables over its key numbering.  A big set thus scans the text once per launch group, each pass
        // with a sparser bitmap and the filter in front of its verify launch: 2000 patterns of 50 bytes, k = 5, took one
        // sieve + five verify launches of 2.5 - 3 ms per GiB each; a sieve pass is 0.3 and its verify launch then near nothing.
        if (cf_on) {
            V.bitmap18.assign(8192, 0u);
            for (uint32_t x = 0; x < 65536u; ++x) {
                if (!((v_seen16[x & 8191u] >> (x >> 13)) & 1u)) continue;
                for (uint32_t f = 0; f < 4; ++f) {
                    const uint32_t c18 = (x << 2) | f;
                    V.bitmap18[c18 & 8191u] |= 1u << (c18 >> 13);
                }
            }
            for (uint32_t i = 0; i < 8192u; ++i) V.bitmap18[i] |= v_even18[i];
            std::vector<uint32_t> tbl(4096), rrec, lrec;
            for (int w = 0; w < 2048; ++w) { tbl[2 * w] = bmp16[w]; tbl[2 * w + 1] = prefix[w]; }
            for (size_t i = 0; i < wk.size();) { // (rank order, as r2s above)
                size_t j = i;
                while (j < wk.size() && (wk[j] >> 32) == (wk[i] >> 32)) ++j;
                if (j - i == 1) {
                    const uint32_t kid = (uint32_t)(wk[i] & 0x7fffu);
                    rrec.push_back(krec[2 * kid]);
                    rrec.push_back(krec[2 * kid + 1]);
                } else {
                    rrec.push_back(0xC0000000u | (uint32_t)(lrec.size() / 2));
                    rrec.push_back(0u);
                    for (size_t z = i; z < j; ++z) {
                        const uint32_t kid = (uint32_t)(wk[z] & 0x7fffu);
                        lrec.push_back(krec[2 * kid]);
                        lrec.push_back(krec[2 * kid + 1] | (z + 1 == j ? 0x80000000u : 0u));
                    }
                }
                i = j;
            }
            auto cf_append = [&](const std::vector<uint32_t> &v) {
                const size_t at = V.cf_image.size(), bytes = v.size() * 4;
                V.cf_image.resize(at + ((bytes + 15) & ~(size_t)15) + 16, 0); // (+16: a lane without a word reads record 0)
                if (bytes) memcpy(V.cf_image.data() + at, v.data(), bytes);
                return (int)at;
            };
            cf_append(tbl);
            V.cf_o_rrec = cf_append(rrec);
            V.cf_o_lrec = cf_append(lrec);
            if (lrec.size() / 2 > 0xffffu) V.cf_image.clear(); // (list indices are 16 bits)
        }
        S.launches.push_back(std::move(V));
    }
    S.per_launch_sieve = cf_on && !S.launches.empty();
    for (const VerifyLaunch &V : S.launches)
        if (V.cf_image.empty()) S.per_launch_sieve = false;
    S.weak_frac = words_weak + words_strong > 0 ? words_weak / (words_weak + words_strong) : 0.0;
    // A single group whose hits nearly all come from such units gains nothing from the filter and pays its instructions
    // (60 patterns of 16 bytes, k = 3 -- pair units of 8 bytes: 0.395 -> 0.459 ms per 64 MiB); APM_SIEVE_CF=2 keeps it on.
    if (cf_env != 2 && S.launches.size() == 1 && S.weak_frac > 0.8) S.per_launch_sieve = false;
    // the sieve's bitmap.  Stride 1: over 9-byte windows at EVEN positions -- a key window may start at the even position
    // (the unit's own nine-byte words, apm_enum_unit_windows with W = 9) or at the odd one behind it (its 16-bit word x,
    // the first byte free).
    // Stride 8: the 16-bit words themselves (dword x & 2047, bit x >> 11).
    long pop16 = 0, pop18 = 0;
    for (uint32_t x = 0; x < 65536u; ++x) {
        if (!((seen16[x & 8191u] >> (x >> 13)) & 1u)) continue;
        ++pop16;
        if (stride == 8) {
            S.bitmap[x & 2047u] |= 1u << (x >> 11);
            continue;
        }
        for (uint32_t f = 0; f < 4; ++f) { // the key window starts at the odd position behind the lookup: the first byte is free
            const uint32_t c18 = (x << 2) | f;
            S.bitmap[c18 & 8191u] |= 1u << (c18 >> 13);
        }
    }
    if (stride == 1)
        for (uint32_t i = 0; i < 8192u; ++i) { // ... at the lookup's own position: the units' nine-byte words
            S.bitmap[i] |= even18[i];
            pop18 += __builtin_popcount(S.bitmap[i]);
        }
    S.rate = stride == 8 ? (double)pop16 / 65536.0 : (double)pop18 / 262144.0;
    S.on = !S.launches.empty();
    return APM_OK;
}

int build_plan(apm_ctx *ctx) {
    ctx->tiled.clear();
    ctx->tails = GenericGroup();
    ctx->stails = GenericGroup();
    ctx->wtails = GenericGroup();
    ctx->xtails = GenericGroup();
    ctx->longs = GenericGroup();
    ctx->trivial.clear();
    ctx->allpat.clear();
    ctx->m_max = 0;
    const int P = (int)ctx->pats.size();
    std::vector<uint32_t> raw_off(P);
    for (int i = 0; i < P; ++i) {
        std::string why;
        int kv = resolve_kernel(ctx->kernel, ctx->pats[i].m, ctx->k, &why);
        if (kv == -100) return fail(ctx, APM_ERR_UNSUPPORTED, "pattern %d (length %d): %s", i, ctx->pats[i].m, why.c_str());
        if (kv == APM_KERNEL_NFA) { // every distinct pattern byte is a class of the launch: at most 16
            bool seen[256] = {false};
            int nc = 0;
            for (unsigned char c : ctx->pats[i].bytes) if (!seen[c]) { seen[c] = true; ++nc; }
            if (nc > 16) {
                if (ctx->kernel == APM_KERNEL_NFA)
                    return fail(ctx, APM_ERR_UNSUPPORTED, "pattern %d (length %d): NFA kernel takes at most 16 distinct pattern bytes", i, ctx->pats[i].m);
                kv = APM_KERNEL_BITPAR;
            }
        }
        if (kv == APM_KERNEL_BITPAR && ctx->pats[i].m > 1024) {
            // one window per wave: the pattern's Eq rows (64 or 128 words per distinct byte, + the "absent" row) must fit LDS
            bool seen[256] = {false};
            int nc = 1;
            for (unsigned char c : ctx->pats[i].bytes) if (!seen[c]) { seen[c] = true; ++nc; }
            if ((size_t)std::min(nc, 256) * (ctx->pats[i].m <= 2048 ? 64 : 128) * 4 > 60 * 1024) {
                if (ctx->kernel == APM_KERNEL_BITPAR)
                    return fail(ctx, APM_ERR_UNSUPPORTED, "pattern %d (length %d): BITPAR beyond 1024 bytes needs an alphabet whose Eq rows fit 60 KiB of LDS", i, ctx->pats[i].m);
                kv = APM_KERNEL_GENERIC;
            }
        }
        ctx->pats[i].kernel = kv;
        raw_off[i] = (uint32_t)ctx->allpat.size();
        ctx->allpat.insert(ctx->allpat.end(), ctx->pats[i].bytes.begin(), ctx->pats[i].bytes.end());
        if (kv == KERNEL_TRIVIAL) { ctx->trivial.push_back(i); continue; }
        ctx->m_max = std::max(ctx->m_max, ctx->pats[i].m);
        ApmPatDesc d{};
        d.m = (uint32_t)ctx->pats[i].m;
        d.byte_off = raw_off[i];
        d.index = (uint32_t)i;
        if (kv == APM_KERNEL_NFA) { // every distinct pattern byte is a class of the launch: at most 16
            bool seen[256] = {false};
            int nc = 0;
            for (unsigned char c : ctx->pats[i].bytes) if (!seen[c]) { seen[c] = true; ++nc; }
            if (nc > 16) {
                if (ctx->kernel == APM_KERNEL_NFA)
                    return fail(ctx, APM_ERR_UNSUPPORTED, "pattern %d (length %d): NFA kernel takes at most 16 distinct pattern bytes", i, ctx->pats[i].m);
                kv = APM_KERNEL_BITPAR;
            }
        }
        if (kv == APM_KERNEL_BITPAR && ctx->pats[i].m > 1024) {
            // (the one-window-per-wave kernel evaluates its truncated windows itself)
        } else if (kv != APM_KERNEL_GENERIC) { // GENERIC scans truncated windows itself (mode 2)
            GenericGroup &tg = ctx->pats[i].m <= 128 ? ctx->stails : (ctx->pats[i].m <= 512 ? ctx->wtails : ctx->xtails);
            tg.descs.push_back(d);
            tg.m_max = std::max(tg.m_max, ctx->pats[i].m);
        } else {
            ctx->longs.descs.push_back(d);
            ctx->longs.m_max = std::max(ctx->longs.m_max, ctx->pats[i].m);
        }
    }

    // ---- BITPAR launches: group by LDS table budget; one text->code LUT per launch ----
    {
        std::vector<int> idx;
        for (int i = 0; i < P; ++i) if (ctx->pats[i].kernel == APM_KERNEL_BITPAR) idx.push_back(i);
        // width classes, each with launches (and kernels) of its own, picked by the launch's m_max (apm_launch_bitpar):
        // <= 128 bytes (1 - 4 words per column), <= 512 (8 / 16: a register-hungry instantiation), <= 1024 (24 / 32 words,
        // one-pass column step: apm_bitlong.hip), <= 4096 (one window per WAVE, one pattern per launch: apm_bitlong.hip)
        auto width_class = [](int m) { return m <= 128 ? 0 : (m <= 512 ? 1 : (m <= 1024 ? 2 : 3)); };
        std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return width_class(ctx->pats[x].m) < width_class(ctx->pats[y].m); });
        size_t pos = 0;
        while (pos < idx.size()) {
            TiledLaunch L;
            L.kind = APM_KERNEL_BITPAR;
            const int wclass = width_class(ctx->pats[idx[pos]].m);
            L.tile = 1024;
            bool present[256] = {false};
            int n_codes = 1; // code 0 = absent
            size_t words = 0;
            std::vector<int> members;
            while (pos < idx.size() && members.size() < (wclass == 3 ? 1u : 1024u)) {
                const PatternInfo &pi = ctx->pats[idx[pos]];
                if (width_class(pi.m) != wclass) break;
                bool p2[256];
                memcpy(p2, present, sizeof p2);
                int nc = n_codes;
                for (unsigned char c : pi.bytes) if (!p2[c]) { p2[c] = true; ++nc; }
                const int entries = nc > 256 ? 256 : nc;
                // every member's table is re-laid with the launch's final code count: bound with `entries`
                size_t w_total = 0;
                auto stride_of = [](int m) { const int w = (m + 31) / 32; return w <= 2 ? w : (w <= 4 ? 4 : (w <= 8 ? 8 : (w <= 16 ? 16 : (w <= 24 ? 24 : (w <= 32 ? 32 : (w <= 64 ? 64 : 128)))))); };
                for (int mi : members) w_total += (size_t)entries * stride_of(ctx->pats[mi].m);
                w_total += (size_t)entries * stride_of(pi.m);
                if (!members.empty() && w_total * 4 > APM_LDS_TABLE_BUDGET) break;
                memcpy(present, p2, sizeof present);
                n_codes = nc;
                members.push_back(idx[pos]);
                ++pos;
                words = w_total;
            }
            (void)words;
            // LUT: 256 distinct bytes => identity, no "absent" code
            const bool full = n_codes > 256;
            int next = 1;
            for (int c = 0; c < 256; ++c) L.lut[c] = full ? (uint8_t)c : (present[c] ? (uint8_t)next++ : 0);
            const int entries = full ? 256 : n_codes;
            for (int mi : members) {
                const PatternInfo &pi = ctx->pats[mi];
                ApmPatDesc d{};
                d.m = (uint32_t)pi.m;
                const uint32_t w32 = (uint32_t)((pi.m + 31) / 32);      // words of the bit vector: 1, 2, 3, 4, then 8, 16, 24, 32; one window
                d.w = w32 <= 4 ? w32 : (w32 <= 8 ? 8u : (w32 <= 16 ? 16u : (w32 <= 24 ? 24u : (w32 <= 32 ? 32u : (w32 <= 64 ? 64u : 128u))))); // per wave: 64, 128 (the rows past m never reach the distance)
                d.stride = d.w == 3 ? 4 : d.w;
                d.index = (uint32_t)mi;
                d.byte_off = 0;
                while (L.tables.size() % 4) L.tables.push_back(0);
                d.aux_off = (uint32_t)L.tables.size();
                L.tables.resize(L.tables.size() + (size_t)entries * d.stride, 0u);
                for (int y = 0; y < pi.m; ++y) {
                    const uint32_t code = L.lut[(unsigned char)pi.bytes[y]];
                    L.tables[d.aux_off + (size_t)code * d.stride + (y >> 5)] |= 1u << (y & 31);
                }
                L.descs.push_back(d);
                L.m_max = std::max(L.m_max, pi.m);
                L.m_min = L.m_min ? std::min(L.m_min, pi.m) : pi.m;
            }
            ctx->tiled.push_back(std::move(L));
        }
    }
    // ---- NFA launches: <= 16 byte classes and <= 512 patterns per launch ----
    {
        std::vector<int> idx;
        for (int i = 0; i < P; ++i) if (ctx->pats[i].kernel == APM_KERNEL_NFA) idx.push_back(i);
        for (size_t pos = 0; pos < idx.size();) {
            TiledLaunch L;
            L.kind = APM_KERNEL_NFA;
            memset(L.lut, 0, sizeof L.lut);
            int cls_of[256];
            for (int c = 0; c < 256; ++c) cls_of[c] = -1;
            int nc = 0;
            for (; pos < idx.size() && L.descs.size() < 512; ++pos) {
                const PatternInfo &pi = ctx->pats[idx[pos]];
                int add = 0;
                bool seen[256] = {false};
                for (unsigned char c : pi.bytes) if (cls_of[c] < 0 && !seen[c]) { seen[c] = true; ++add; }
                if (!L.descs.empty() && nc + add > 16) break;
                for (unsigned char c : pi.bytes) if (cls_of[c] < 0) { cls_of[c] = nc; L.lut[nc++] = c; }
                ApmPatDesc d{};
                d.m = (uint32_t)pi.m;
                d.index = (uint32_t)idx[pos];
                d.byte_off = (uint32_t)L.bytes.size();
                // 16 bytes per pattern: the class number of pattern byte x in nibble x (the kernel reads them with one
                // scalar 16-byte load and shifts the next one out per column)
                L.bytes.resize(L.bytes.size() + 16, 0);
                for (size_t x = 0; x < pi.bytes.size(); ++x)
                    L.bytes[d.byte_off + x / 2] |= (uint8_t)(cls_of[(unsigned char)pi.bytes[x]] << (4 * (x & 1)));
                L.descs.push_back(d);
                L.m_max = std::max(L.m_max, pi.m);
                L.m_min = L.m_min ? std::min(L.m_min, pi.m) : pi.m;
            }
            L.nb = nc; // classes; their bytes: lut[0 .. nc)
            ctx->tiled.push_back(std::move(L));
        }
    }
    // ---- WAVEFRONT launches: up to 64 patterns, raw bytes in LDS ----
    {
        std::vector<int> idx;
        for (int i = 0; i < P; ++i) if (ctx->pats[i].kernel == APM_KERNEL_WAVEFRONT) idx.push_back(i);
        for (size_t pos = 0; pos < idx.size();) {
            TiledLaunch L;
            L.kind = APM_KERNEL_WAVEFRONT;
            L.tile = 512;
            memset(L.lut, 0, sizeof L.lut);
            for (; pos < idx.size() && L.descs.size() < 64; ++pos) {
                const PatternInfo &pi = ctx->pats[idx[pos]];
                ApmPatDesc d{};
                d.m = (uint32_t)pi.m;
                d.w = (uint32_t)wavefront_rows_per_lane(pi.m);
                d.index = (uint32_t)idx[pos];
                d.byte_off = (uint32_t)L.bytes.size();
                L.bytes.insert(L.bytes.end(), pi.bytes.begin(), pi.bytes.end());
                L.descs.push_back(d);
                L.m_max = std::max(L.m_max, pi.m);
                L.m_min = L.m_min ? std::min(L.m_min, pi.m) : pi.m;
            }
            ctx->tiled.push_back(std::move(L));
        }
    }

    // ---- sieve + verify pipeline (apm_sieve.hip): as soon as one BANDED pattern needs every text position looked at
    // (pieces shorter than 15 bytes), ONE sieve pass serves all BANDED patterns of the set -- those with longer
    // pieces join with one key per piece instead of a sampled family -- and the verify launches work off its
    // candidate list.  The LDS-tile / stream launches of the same patterns are still planned below: they run as
    // for text the sieve cannot take (unaligned, >= 4 GiB).
    // APM_SIEVE=0 switches the pipeline off (A/B aid). ----
    ctx->sieve = SievePlan();
    {
        static const int sieve_env = getenv("APM_SIEVE") ? atoi(getenv("APM_SIEVE")) : 1;
        bool has_s1 = false, has_banded = false;
        size_t stream_keys = 0, stream_bytes = 0, n_banded = 0; // what one stream launch would have to hold (limits of the class loop below)
        for (int i = 0; i < P; ++i) {
            if (ctx->pats[i].kernel != APM_KERNEL_BANDED) continue;
            has_banded = true;
            const int piece = ctx->pats[i].m / (ctx->k + 1);
            if (piece < 15) has_s1 = true;
            stream_keys += (size_t)(ctx->k + 1) * (piece >= 31 ? 16u : 8u);
            stream_bytes += (size_t)ctx->pats[i].m;
            ++n_banded;
        }
        const bool stream_splits = stream_keys > 4096 || stream_bytes > 16384 || n_banded > 1024;
        // sets of long pieces only: the sampled form of the pipeline (one lookup per 8 bytes, sieve and verification fused
        // in one launch) when verification is the heavy part (k >= 2: pair pre-check + banded DP, which stall the stream
        // kernel's loads) or when the set is too big for ONE stream launch (1000 patterns of 32, k = 0: four stream
        // launches 0.60 ms per 256 MiB, two fused ones 0.23); a small set with k <= 1 stays on the stream kernel, which
        // sits on the HBM ceiling there (cfg2: 0.046 ms against 0.060) -- tools/sampled_k_probe.py
        const int stride = has_s1 ? 1 : 8;
#ifdef APM_MEASURE
        static const int sampled_min_k = getenv("APM_SAMPLED_MIN_K") ? atoi(getenv("APM_SAMPLED_MIN_K")) : 2;
#else
        constexpr int sampled_min_k = 2;
#endif
        if (sieve_env && has_banded && (has_s1 || ctx->k >= sampled_min_k || stream_splits)) {
            const int rc = build_sieve_plan(ctx, stride);
            if (rc) return rc;
        }
    }

    // ---- BANDED launches: patterns grouped by (key length, sampling stride); k+1 pigeonhole pieces each ----
    for (int cls = 0; cls < 5; ++cls) {
        static const int kl_of[5] = {16, 8, 8, 6, 4}, st_of[5] = {16, 8, 1, 1, 1};
        const int klen = kl_of[cls];
        const int stride = st_of[cls];
        auto class_of = [&](int m) {
            const int piece = m / (ctx->k + 1);
            if (ctx->sieve.on && ctx->sieve.stride == 1 && piece >= 8) return 2; // (same coverage as the sieve pipeline: these launches are its fallback)
            return piece >= 31 ? 0 : (piece >= 15 ? 1 : (piece >= 8 ? 2 : (piece >= 6 ? 3 : 4)));
        };
        std::vector<int> idx;
        for (int i = 0; i < P; ++i)
            if (ctx->pats[i].kernel == APM_KERNEL_BANDED && class_of(ctx->pats[i].m) == cls) idx.push_back(i);
        auto dword = [](const unsigned char *b) {
            return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
        };
        auto fp8 = [](uint32_t lo, uint32_t hi) { return lo + (hi << 3); };
        auto slot_hash = [](uint32_t f) {
            return (uint32_t)((uint64_t)(f & 0xffffffu) * 0x9E3779u) + (uint32_t)((uint64_t)((f >> 12) & 0xffffffu) * 0x85EBCAu);
        };
        for (size_t pos = 0; pos < idx.size();) {
            TiledLaunch L;
            L.kind = APM_KERNEL_BANDED;
            L.key_len = klen;
            L.stride = stride;
            L.sieved = ctx->sieve.on && (stride == 1 || ctx->sieve.stride == 8);
            L.qcap = stride == 1 ? 1024 : 512;
            memset(L.lut, 0, sizeof L.lut);
            const int pieces = ctx->k + 1;
            for (; pos < idx.size(); ++pos) {
                const PatternInfo &pi = ctx->pats[idx[pos]];
#ifdef APM_MEASURE
                static const size_t max_keys = getenv("APM_MAX_KEYS") ? std::min<size_t>(32767, std::max<long>(1, atol(getenv("APM_MAX_KEYS")))) : 4096;
#else
                constexpr size_t max_keys = 4096; // (15-bit key ids: never above 32767)
#endif
                if (!L.descs.empty() && (L.bytes.size() + (size_t)pi.m > 16384 ||
                                         L.keys.size() + (size_t)pieces * stride > max_keys || L.descs.size() >= 1024 ||
                                         L.piece_off.size() + (size_t)pieces > 60000))
                    break;
                ApmPatDesc d{};
                d.m = (uint32_t)pi.m;
                d.index = (uint32_t)idx[pos];
                d.byte_off = (uint32_t)L.bytes.size();
                d.aux_off = (uint32_t)L.piece_off.size();
                d.w = (uint32_t)pieces;
                L.bytes.insert(L.bytes.end(), pi.bytes.begin(), pi.bytes.end());
                for (int q = 0; q < pieces; ++q) {
                    const int aq = (int)((int64_t)q * pi.m / pieces);
                    L.piece_off.push_back((uint16_t)aq);
                    for (int r = 0; r < stride; ++r) {
                        unsigned char b[16] = {0}; // key bytes, zero padded past the end of the pattern
                        for (int z = 0; z < klen && aq + r + z < pi.m; ++z) b[z] = (unsigned char)pi.bytes[aq + r + z];
                        ApmKey key{};
                        key.pat = (uint16_t)L.descs.size();
                        key.off = (uint16_t)(aq + r);
                        key.piece = (uint16_t)q;
                        key.next = 0;
                        if (klen == 16) key.fp = fp8(dword(b), dword(b + 4)) + (fp8(dword(b + 8), dword(b + 12)) & 0xffffffu) * 0x9E3779u;
                        else if (klen == 4) key.fp = dword(b);
                        else key.fp = fp8(dword(b), dword(b + 4)); // bytes past klen are zero (masked on the device)
                        L.keys.push_back(key);
                        L.a_max = std::max(L.a_max, aq + r);
                    }
                }
                L.descs.push_back(d);
                L.m_max = std::max(L.m_max, pi.m);
                L.m_min = L.m_min ? std::min(L.m_min, pi.m) : pi.m;
            }
            const int band = ctx->k / 2;
            const int front = band > 0 ? 16 : 0;
            L.tile = (APM_FILTER_POS - front - L.m_max - band) & ~31; // every window + its keys inside 4096 staged bytes
            while (L.bytes.size() % 16) L.bytes.push_back(0);
            // compact per-key / per-pattern records the verify stage reads from LDS
            for (const ApmKey &kk : L.keys)
                L.kinfo.push_back((uint32_t)kk.pat | ((uint32_t)kk.off << 12) | ((uint32_t)kk.piece << 21));
            for (const ApmPatDesc &dd : L.descs) {
                L.pinfo.push_back(dd.byte_off | (dd.m << 16));
                L.pinfo.push_back(dd.aux_off);
            }
            // hash table: 8-way buckets of 16-bit tags (low half of the slot hash), bucket = top bits;
            // keys with equal tags in one bucket are chained behind a single entry
            int nb = 16, lg = 4;
            while (nb * 2 < (int)L.keys.size() && nb < 512) { nb *= 2; ++lg; }
            for (;;) {
                L.table.assign((size_t)nb * 8, 0xffffu);
                L.table_kid.assign((size_t)nb * 8, 0xffffu);
                L.ovf.clear();
                std::vector<int> fill((size_t)nb, 0);
                for (auto &kk : L.keys) kk.next = 0;
                for (size_t kid = 0; kid < L.keys.size(); ++kid) {
                    const uint32_t h = klen == 16 ? L.keys[kid].fp : slot_hash(L.keys[kid].fp);
                    const uint32_t slot = h >> (32 - lg);
                    const uint16_t tag = (uint16_t)(h & 0xffffu);
                    int head = -1;
                    uint16_t *head_kid = nullptr;
                    for (int wv = 0; wv < fill[slot]; ++wv)
                        if (L.table[slot * 8 + wv] == tag) {
                            head = L.table_kid[slot * 8 + wv] & 0x7fff;
                            head_kid = &L.table_kid[slot * 8 + wv];
                        }
                    uint32_t *head_ovf = nullptr;
                    if (head < 0)
                        for (size_t o = 0; o + 1 < L.ovf.size(); o += 2)
                            if (L.ovf[o] == tag && (L.ovf[o + 1] >> 16) == slot) {
                                head = (int)(L.ovf[o + 1] & 0x7fff);
                                head_ovf = &L.ovf[o + 1];
                            }
                    if (head >= 0) { // chain behind the existing entry with this tag
                        int tail = head;
                        while (L.keys[tail].next) tail = L.keys[tail].next - 1;
                        L.keys[tail].next = (uint16_t)(kid + 1);
                        if (head_kid) *head_kid |= 0x8000u;
                        if (head_ovf) *head_ovf |= 0x8000u;
                    } else if (fill[slot] < 8) {
                        L.table[slot * 8 + fill[slot]] = tag;
                        L.table_kid[slot * 8 + fill[slot]] = (uint16_t)kid;
                        ++fill[slot];
                    } else {
                        L.ovf.push_back(tag);
                        L.ovf.push_back((uint32_t)kid | (slot << 16));
                    }
                }
                if (L.ovf.size() / 2 <= 4 || nb >= 1024) break;
                nb *= 2;
                ++lg;
            }
            for (size_t o = 1; o < L.ovf.size(); o += 2) L.ovf[o] &= 0xffffu; // drop the slot annotation
            L.nb = nb;
            L.lg_nb = lg;
            // one contiguous image, laid out exactly like its LDS copy
            auto append = [&](const void *src, size_t bytes) {
                const size_t at = L.image.size();
                L.image.resize(at + ((bytes + 15) & ~(size_t)15), 0);
                if (bytes) memcpy(L.image.data() + at, src, bytes);
                return (int)at;
            };
            if (stride == 1) {
                // First-level filter of the per-position classes: a presence bitmap indexed by the 2-bit
                // codes (b >> s) & 3 of the key bytes.  s is picked to spread this launch's pattern bytes
                // over the four codes as evenly as possible (s = 1 separates A,C,G,T and a,c,g,t exactly).
                long best = -1;
                for (int sft = 0; sft < 7; ++sft) {
                    long hist[4] = {0, 0, 0, 0};
                    for (const ApmPatDesc &dd : L.descs)
                        for (uint32_t y = 0; y < dd.m; ++y) ++hist[(L.bytes[dd.byte_off + y] >> sft) & 3];
                    const long score = std::min(std::min(hist[0], hist[1]), std::min(hist[2], hist[3])) * 4 +
                                       (hist[0] > 0) + (hist[1] > 0) + (hist[2] > 0) + (hist[3] > 0) + (sft == 1);
                    if (score > best) { best = score; L.code_shift = sft; }
                }
                std::vector<uint8_t> bmp(8192, 0); // over 8-byte code words whatever the key length
                for (const ApmKey &kk : L.keys) mark_key_windows(bmp, L, kk, pieces, L.code_shift, ctx->k / 2 >= 1);
                L.o_bmp = append(bmp.data(), bmp.size()); // = 0: a compile-time LDS address for the probes
            }
            L.o_pat = append(L.bytes.data(), L.bytes.size());
            L.o_tab = append(L.table.data(), L.table.size() * 2);
            L.o_kid = append(L.table_kid.data(), L.table_kid.size() * 2);
            L.o_ovf = append(L.ovf.data(), L.ovf.size() * 4);
            L.o_kinfo = append(L.kinfo.data(), L.kinfo.size() * 4);
            L.o_pinfo = append(L.pinfo.data(), L.pinfo.size() * 4);
            std::vector<uint16_t> nxt;
            for (const ApmKey &kk : L.keys) nxt.push_back(kk.next);
            L.o_next = append(nxt.data(), nxt.size() * 2);
            L.o_poff = append(L.piece_off.data(), L.piece_off.size() * 2);
            if (stride == 1) { // one packed record per key for the pair pre-check (see ApmFilterArgs::o_kext)
                std::vector<uint32_t> kext;
                for (const ApmKey &kk : L.keys) {
                    const ApmPatDesc &dd = L.descs[kk.pat];
                    auto piece_begin = [&](int q) { return q >= pieces ? (int)dd.m : (int)L.piece_off[dd.aux_off + q]; };
                    const int q = kk.piece, pq = q ^ 1;
                    const uint32_t len = (uint32_t)(piece_begin(q + 1) - piece_begin(q));
                    uint32_t side = 0, plen = 0;
                    if (pq < pieces) {
                        side = pq > q ? 1u : 2u;
                        plen = (uint32_t)(piece_begin(pq + 1) - piece_begin(pq));
                    }
                    kext.push_back((uint32_t)(dd.byte_off + kk.off) | (std::min<uint32_t>(len, 255u) << 16) |
                                   ((plen > 16 ? 31u : plen) << 24) | (side << 29));
                }
                L.o_kext = append(kext.data(), kext.size() * 4);
            }
            if (stride == 1) {
                // the tile kernel's LDS: 4 tile buffers + image + 2 queues + counters + survivor lists (see
                // apm_filter_lds_bytes); a big image (cfg5: 57 KB) leaves room for two workgroups per CU only with
                // the smaller candidate queue -- overflowing it is correct, just slow (dense pass)
                auto lds_with = [&](int qcap) {
                    return (size_t)4 * APM_FILTER_POS + L.image.size() + 2 * (size_t)qcap * 4 + ((L.descs.size() + 3) & ~(size_t)3) * 4 + 32 + 2048 + 16;
                };
                const size_t cu_lds = 160 * 1024;
                if (cu_lds / lds_with(512) > cu_lds / lds_with(1024)) L.qcap = 512;
            }
            ctx->tiled.push_back(std::move(L));
        }
    }

    // ---- upload to every device ----
    for (auto &ds : ctx->devs) {
        free_device_plan(ds);
        HIP_TRY(ctx, hipSetDevice(ds.dev));
        int rc;
        if ((rc = upload_vec(ctx, &ds.d_allpat, ctx->allpat))) return rc;
        if ((rc = upload_vec(ctx, &ds.d_tail_descs, ctx->tails.descs))) return rc;
        if ((rc = upload_vec(ctx, &ds.d_stail_descs, ctx->stails.descs))) return rc;
        if ((rc = upload_vec(ctx, &ds.d_wtail_descs, ctx->wtails.descs))) return rc;
        if ((rc = upload_vec(ctx, &ds.d_xtail_descs, ctx->xtails.descs))) return rc;
        if ((rc = upload_vec(ctx, &ds.d_long_descs, ctx->longs.descs))) return rc;
        if ((rc = upload_vec(ctx, &ds.d_trivial, ctx->trivial))) return rc;
        HIP_TRY(ctx, hipMalloc((void **)&ds.d_counts, std::max<size_t>((size_t)P * 8, 16)));
        if (ctx->sieve.on) {
            if ((rc = upload_vec(ctx, &ds.d_sieve_bmp, ctx->sieve.bitmap))) return rc;
            ds.verify.resize(ctx->sieve.launches.size());
            for (size_t v = 0; v < ctx->sieve.launches.size(); ++v) {
                const VerifyLaunch &V = ctx->sieve.launches[v];
                if (ctx->sieve.per_launch_sieve) {
                    if ((rc = upload_vec(ctx, &ds.verify[v].d_bmp18, V.bitmap18))) return rc;
                    if ((rc = upload_vec(ctx, &ds.verify[v].d_cf, V.cf_image))) return rc;
                }
                if ((rc = upload_vec(ctx, &ds.verify[v].d_descs, V.descs))) return rc;
                if ((rc = upload_vec(ctx, &ds.verify[v].d_image, V.image))) return rc;
                if ((rc = upload_vec(ctx, &ds.verify[v].d_kinfo, V.kinfo))) return rc;
                if ((rc = upload_vec(ctx, &ds.verify[v].d_pinfo, V.pinfo))) return rc;
                if ((rc = upload_vec(ctx, &ds.verify[v].d_kpart, V.kpart))) return rc;
            }
        }
        ds.tiled.resize(ctx->tiled.size());
        for (size_t t = 0; t < ctx->tiled.size(); ++t) {
            const TiledLaunch &L = ctx->tiled[t];
            if ((rc = upload_vec(ctx, &ds.tiled[t].d_descs, L.descs))) return rc;
            if ((rc = upload_vec(ctx, &ds.tiled[t].d_bytes, L.bytes))) return rc;
            if ((rc = upload_vec(ctx, &ds.tiled[t].d_tables, L.tables))) return rc;
            std::vector<uint8_t> lut(L.lut, L.lut + 256);
            if ((rc = upload_vec(ctx, &ds.tiled[t].d_lut, lut))) return rc;
            if ((rc = upload_vec(ctx, &ds.tiled[t].d_image, L.image))) return rc;
        }
    }
    return APM_OK;
}

int ensure_scratch(apm_ctx *ctx, DeviceState &ds, size_t bytes) {
    if (bytes <= ds.scratch_bytes) return APM_OK;
    if (ds.d_scratch) {
        HIP_TRY(ctx, hipStreamSynchronize(ds.stream));
        hipFree(ds.d_scratch);
        ds.d_scratch = nullptr;
        ds.scratch_bytes = 0;
    }
    HIP_TRY(ctx, hipMalloc((void **)&ds.d_scratch, bytes));
    ds.scratch_bytes = bytes;
    return APM_OK;
}

// generic-kernel launch over a pattern group; mode 0 full windows, 1 tails only, 2 everything
int launch_generic_group(apm_ctx *ctx, DeviceState &ds, const GenericGroup &g, const ApmPatDesc *d_descs,
                         int mode, const uint8_t *d_text, int64_t avail, int64_t jb, int64_t je, int64_t nrel,
                         unsigned long long *d_counts, const ApmPosSink &sink) {
    if (g.descs.empty() || je <= jb) return APM_OK;
    int64_t span = je - jb;
    if (mode == 1) span = std::min<int64_t>(span, g.m_max); // at most m-1 tail windows per pattern
    const size_t col = (size_t)g.m_max + 1;
    const size_t budget = (size_t)1 << 30;
    const size_t per_launch = std::min<size_t>(g.descs.size(), 65535); // grid.y limit: more patterns = more launches
    int64_t nbx = (span + APM_BLOCK - 1) / APM_BLOCK;
    const int64_t cap = std::max<int64_t>(1, (int64_t)(budget / (col * 2 * APM_BLOCK * per_launch)));
    nbx = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nbx, cap), 4096));
    const size_t need = col * 2 * APM_BLOCK * (size_t)nbx * per_launch;
    int rc = ensure_scratch(ctx, ds, need);
    if (rc) return rc;
    ApmGenericArgs a{};
    a.text = d_text;
    a.avail = avail;
    a.jb = jb;
    a.je = je;
    a.nrel = nrel;
    a.pats = d_descs;
    a.bytes = ds.d_allpat;
    a.k = ctx->k;
    a.mode = mode;
    a.col_stride = (int)col;
    a.scratch = ds.d_scratch;
    a.counts = d_counts;
    a.pos = sink;
    for (size_t first = 0; first < g.descs.size(); first += per_launch) { // (same scratch: launches of one stream run in order)
        a.pats = d_descs + first;
        HIP_TRY(ctx, apm_launch_generic(a, (int)nbx, (int)std::min(per_launch, g.descs.size() - first), ds.stream));
        { const int nrc = note_launch(ctx, ds, "generic"); if (nrc) return nrc; }
    }
    return APM_OK;
}

int scan_shard_one(apm_ctx *ctx, DeviceState &ds, const uint8_t *d_text, uint64_t text_off, uint64_t text_len,
                   uint64_t n_total, uint64_t own_begin, uint64_t own_end, unsigned long long *d_counts);

// The sieve pipeline addresses its shard with 32 bits.  A bigger shard (a 288 GB device holds a lot of text) is scanned
// in pieces of 3 GiB of window starts, each with its own text window [piece begin rounded down so that the pointer
// keeps its 16-byte alignment, piece end + m_max + 31) -- the same cut a caller sharding the text would make (every
// window lies in exactly one piece; what a piece reads in front of its first window start never decides a match).
int scan_shard(apm_ctx *ctx, DeviceState &ds, const uint8_t *d_text, uint64_t text_off, uint64_t text_len,
               uint64_t n_total, uint64_t own_begin, uint64_t own_end, unsigned long long *d_counts) {
    const uint64_t lim32 = (uint64_t)APM_SIEVE_MAX_BYTES - 4096;
    // an unaligned text pointer into a bigger buffer: start the shard's text at the 16-byte boundary in front of it (those
    // bytes are readable -- apm.h -- and lie in front of every window start of the shard, where nothing decides a match)
    const uint64_t mis = (uint64_t)(reinterpret_cast<uintptr_t>(d_text) & 15u);
    if (ctx->sieve.on && mis != 0 && text_off >= mis && text_len > 0) {
        d_text -= mis;
        text_off -= mis;
        text_len += mis;
    }
    if (!ctx->sieve.on || text_len < lim32 || (reinterpret_cast<uintptr_t>(d_text) & 15u) != 0 || own_begin < text_off)
        return scan_shard_one(ctx, ds, d_text, text_off, text_len, n_total, own_begin, own_end, d_counts);
    const uint64_t k = (uint64_t)ctx->k;
    const uint64_t oe = std::min(own_end, n_total > k ? n_total - k : 0);
    const uint64_t m_max = (uint64_t)std::max(ctx->m_max, 1), step = (uint64_t)3 << 30;
    for (uint64_t b = own_begin; b < oe;) {
        const uint64_t e = std::min(oe, b + step);
        const uint64_t sb = text_off + ((b - text_off) & ~(uint64_t)15);
        const uint64_t se = std::min(text_off + text_len, e + m_max + 31);
        const int rc = scan_shard_one(ctx, ds, d_text + (sb - text_off), sb, se - sb, n_total, b, e, d_counts);
        if (rc) return rc;
        b = e;
    }
    return APM_OK;
}

// the shard scan proper, all on ds.stream, no host sync
int scan_shard_one(apm_ctx *ctx, DeviceState &ds, const uint8_t *d_text, uint64_t text_off, uint64_t text_len,
                   uint64_t n_total, uint64_t own_begin, uint64_t own_end, unsigned long long *d_counts) {
    const uint64_t k = (uint64_t)ctx->k;
    const uint64_t limit = n_total > k ? n_total - k : 0;
    const uint64_t ob = own_begin, oe = std::min(own_end, limit);
    if (oe <= ob) return APM_OK;
    if (text_off > ob) return fail(ctx, APM_ERR_INVALID, "shard text starts after own_begin");
    const uint64_t m_max = (uint64_t)std::max(ctx->m_max, 1);
    const uint64_t need_end = std::min<uint64_t>(n_total, oe + m_max - 1);
    if (text_off + text_len < need_end)
        return fail(ctx, APM_ERR_INVALID, "shard text too short: halo of m_max-1 = %llu bytes required",
                    (unsigned long long)(m_max - 1));
    HIP_TRY(ctx, hipSetDevice(ds.dev));
    const int64_t jb = (int64_t)(ob - text_off), je = (int64_t)(oe - text_off);
    const int64_t nrel = (int64_t)(n_total - text_off), avail = (int64_t)text_len;
    ds.text_bytes += need_end - ob;

    // truncated tail windows of the m <= 128 patterns (only the shard owning the end of the text has
    // any): they ride as extra workgroups of the first BANDED launch, else get their own small launch
    ApmTailArgs ta{};
    bool tails_pending = !ctx->stails.descs.empty() && nrel - (int64_t)ctx->stails.m_max + 1 < je;
    ApmPosSink sink{};
    if (ctx->find_active) {
        sink.out = ds.d_pos_out;
        sink.count = ds.d_pos_count;
        sink.cap = ds.pos_cap;
        sink.text_off = text_off;
    }
    ta.pos = sink;
    if (tails_pending) {
        ta.text = d_text;
        ta.jb = jb;
        ta.je = je;
        ta.nrel = nrel;
        ta.pats = ds.d_stail_descs;
        ta.bytes = ds.d_allpat;
        ta.counts = d_counts;
        ta.k = ctx->k;
    }

    if (ctx->timing_on) HIP_TRY(ctx, hipEventRecord(ds.ev_mstart, ds.stream));
    // sieve + verify pipeline of the per-position classes (needs 16-byte aligned text and < 4 GiB of it: 32-bit
    // buffer offsets, 32-bit list entries); otherwise the LDS-tile / stream launches below do the whole job
    bool sieve_run = false, fused_run = false;
    if (ctx->sieve.on && (reinterpret_cast<uintptr_t>(d_text) & 15u) == 0) {
        const int band = ctx->k / 2;
        const int64_t avail_pad = avail + (int64_t)((16u - ((reinterpret_cast<uintptr_t>(d_text) + (uintptr_t)avail) & 15u)) & 15u);
        const int64_t p_lo = std::max<int64_t>(0, jb - band) & ~(int64_t)15;
        const int64_t p_hi = std::min<int64_t>(avail, je + ctx->sieve.m_max + band);
        if (p_hi > p_lo && avail_pad >= 16 && avail_pad <= APM_SIEVE_MAX_BYTES) {
            if (!ds.d_work) {
                HIP_TRY(ctx, hipMalloc((void **)&ds.d_work, APM_WORK_BYTES));
                HIP_TRY(ctx, hipMemsetAsync(ds.d_work, 0, APM_WORK_BYTES, ds.stream));
                ds.work_epoch = 0;
                ds.sieve_epoch = 0;
            }
            // FUSED form: one kernel per verify group sieves and verifies; the text leaves HBM once, no masks.  Measured on
            // MI355X (profiles/r02/fused_ab.txt): the sampled pipeline gains 15 % (cfg4 0.268 -> 0.228 ms per GiB: its
            // sieve is a few instructions per KiB, the verification hides behind the stream), the per-position one is
            // latency bound in either form and loses occupancy to the bigger kernel (cfg3 0.50 -> 0.48 at best, cfg5
            // 0.64 -> 1.08).  So: fused when the sieve is sampled; APM_FUSED=1 / 0 forces it on / off (A/B aid, and the
            // tests run both forms).
            static const int fused_env = getenv("APM_FUSED") ? atoi(getenv("APM_FUSED")) : -1;
            bool fused_ok = fused_env < 0 ? ctx->sieve.stride == 8 : fused_env != 0;
            std::vector<ApmFusedArgs> fargs;
            std::vector<size_t> fa_index; // fargs[i] belongs to launches[fa_index[i]]
            for (size_t v = 0; fused_ok && v < ctx->sieve.launches.size(); ++v) {
                VerifyLaunch &V = ctx->sieve.launches[v];
                ApmFusedArgs fa{};
                fa.s.text = d_text;
                fa.s.avail_pad = avail_pad;
                fa.s.tile0 = p_lo;
                fa.s.nchunks = (p_hi - p_lo + 1023) / 1024;
                fa.s.bitmap = reinterpret_cast<const uint4 *>(ds.d_sieve_bmp);
                fa.s.code_shift = ctx->sieve.code_shift;
                fa.s.stride = ctx->sieve.stride;
                ApmVerifyArgs &va = fa.v;
                va.text = d_text;
                va.avail = avail;
                va.avail_pad = avail_pad;
                va.jb = jb;
                va.je = std::min<int64_t>(je, nrel - V.m_min + 1);
                va.nrel = nrel;
                va.image = reinterpret_cast<const uint4 *>(ds.verify[v].d_image);
                va.image_len = (int)V.image.size();
                va.o_prefix = V.o_prefix;
                va.o_r2s = V.o_r2s;
                va.o_slots = V.o_slots;
                va.o_kext = V.o_kext;
                va.o_pat = V.o_pat;
                va.o_masks = V.o_masks;
                va.o_kinfo = V.o_kinfo;
                va.o_pinfo = V.o_pinfo;
                va.o_rc = V.o_rc;
                va.kinfo = ds.verify[v].d_kinfo;
                va.pinfo = reinterpret_cast<const uint2 *>(ds.verify[v].d_pinfo);
                va.kpart = ds.verify[v].d_kpart;
                va.pats = ds.verify[v].d_descs;
                va.counts = d_counts;
                va.n_pats = (int)V.descs.size();
                va.nk = (int)V.kinfo.size();
                va.k = ctx->k;
                va.band = band;
                va.code_shift = ctx->sieve.code_shift;
                va.stride = ctx->sieve.stride;
#ifdef APM_MEASURE
                if (!ds.d_stats) HIP_TRY(ctx, hipMalloc((void **)&ds.d_stats, APM_STATS_BYTES));
                va.stats = ds.d_stats;
#endif
                if (!V.fused_threads) {
                    V.fused_blocks_per_cu = apm_fused_geometry(fa, &V.fused_threads);
                    if (V.fused_blocks_per_cu < 1) V.fused_threads = -1; // does not fit a CU
                }
                if (V.fused_threads < 64) fused_ok = false;
                if (va.je > jb) { fargs.push_back(fa); fa_index.push_back(v); }
            }
            if (fused_ok) {
                for (ApmFusedArgs &fa : fargs) {
                    if (tails_pending) { // the truncated tail windows ride as extra workgroups beside the scan
                        fa.s.n_tail = (int)ctx->stails.descs.size();
                        fa.s.tail = ta;
                        tails_pending = false;
                    }
                    const VerifyLaunch &V = ctx->sieve.launches[fa_index[&fa - fargs.data()]];
#ifdef APM_MEASURE
                    HIP_TRY(ctx, hipMemsetAsync(ds.d_stats, 0, APM_STATS_BYTES, ds.stream));
#endif
                    fa.v.work = ds.d_work;
                    HIP_TRY(ctx, apm_launch_fused(fa, V.fused_threads, ds.n_cu * V.fused_blocks_per_cu, &ds.work_epoch, ds.stream));
                    { const int nrc = note_launch(ctx, ds, "fused"); if (nrc) return nrc; }
                }
                fused_run = true;
            }
            if (!fused_run) {
            // hit masks: one dword per lane and 4 KiB block; every one is written by the sieve, nothing to clear
            const int64_t n_mask_blocks = ((p_hi - p_lo + 1023) / 1024 + 3) / 4;
            const size_t need = (size_t)n_mask_blocks * 64 + 64;
            if (ds.masks_cap < need) {
                if (ds.d_masks) {
                    HIP_TRY(ctx, hipStreamSynchronize(ds.stream)); // (a verify launch of an earlier call may still read it)
                    HIP_TRY(ctx, hipFree(ds.d_masks));
                }
                ds.d_masks = nullptr;
                ds.masks_cap = 0;
                HIP_TRY(ctx, hipMalloc((void **)&ds.d_masks, need * 4));
                ds.masks_cap = need;
            }
            if (ds.blist_cap < (size_t)n_mask_blocks + 64) {
                if (ds.d_blist) {
                    HIP_TRY(ctx, hipStreamSynchronize(ds.stream));
                    HIP_TRY(ctx, hipFree(ds.d_blist));
                }
                ds.d_blist = nullptr;
                ds.blist_cap = 0;
                HIP_TRY(ctx, hipMalloc((void **)&ds.d_blist, ((size_t)n_mask_blocks + 64) * 4));
                ds.blist_cap = (size_t)n_mask_blocks + 64;
            }
            static const int blist_env = getenv("APM_SIEVE_BLIST") ? atoi(getenv("APM_SIEVE_BLIST")) : 1; // (A/B aid: 0 = the verify launches walk every mask row)
            // candidate list of the code-filter form: 32 entries allocated per 4 KiB block (half the bytes of the mask rows).
            // APM_SIEVE_CLIST=0 turns it off (A/B aid); APM_CLIST_REGION_CAP=n (1..64) shrinks every region to n entries (the
            // tests force the overflow path with it)
            static const int clist_env = getenv("APM_SIEVE_CLIST") ? atoi(getenv("APM_SIEVE_CLIST")) : 1;
            static const int clist_cap_env = getenv("APM_CLIST_REGION_CAP") ? std::max(1, std::min(64, atoi(getenv("APM_CLIST_REGION_CAP")))) : 0;
            constexpr int clist_per_block = 32;
            constexpr int kClistMaxRegions = 4096;
            int clist_regions = 0;       // of the pass in hand (0: no list kept)
            uint32_t clist_region_cap = 0;
            ds.last_mask_blocks = n_mask_blocks;
            // one sieve pass: the set's shared bitmap (v < 0), or launch v's own bitmap with its code filter
            const uint32_t *blist_ctr = nullptr; // the list counter of the pass in hand (NULL: no list kept)
            auto sieve_pass = [&](int v) -> int {
                ApmSieve2Args sv{};
                sv.text = d_text;
                sv.avail_pad = avail_pad;
                sv.tile0 = p_lo;
                sv.nchunks = (p_hi - p_lo + 1023) / 1024;
                sv.bitmap = reinterpret_cast<const uint4 *>(v < 0 ? ds.d_sieve_bmp : ds.verify[(size_t)v].d_bmp18);
                sv.code_shift = ctx->sieve.code_shift;
                sv.stride = ctx->sieve.stride;
                sv.masks = ds.d_masks;
                if (v >= 0) { // second stage of the sieve: the code filter
                    VerifyLaunch &V = ctx->sieve.launches[(size_t)v];
                    sv.cf_image = reinterpret_cast<const uint4 *>(ds.verify[(size_t)v].d_cf);
                    sv.cf_len = (int)V.cf_image.size();
                    sv.cf_o_rrec = V.cf_o_rrec;
                    sv.cf_o_lrec = V.cf_o_lrec;
                    sv.cf_threads = V.cf_threads;
                    sv.cf_blocks_per_cu = V.cf_blocks_per_cu;
                }
                // the truncated tail windows ride as extra workgroups beside the scan -- in the code-filter form only a few of
                // them: its workgroups are big (1024 threads, most of a CU's LDS) and 2000 of them, one per pattern, made the
                // pass three times as long (256 cost nothing measurable); beyond 512 they get the small launch of their own at the end of the call
                if (tails_pending && (v < 0 || ctx->stails.descs.size() <= 512)) {
                    sv.n_tail = (int)ctx->stails.descs.size();
                    sv.tail = ta;
                    tails_pending = false;
                }
                const bool use_blist = blist_env && sv.cf_image != nullptr; // (the list is kept by the code-filter form of the sieve only)
                blist_ctr = nullptr;
                if (use_blist) {
                    sv.blist = ds.d_blist;
                    sv.blist_ctr = APM_BLIST_CTR(ds.d_work, ds.sieve_epoch & 1);
                    sv.blist_ctr_next = APM_BLIST_CTR(ds.d_work, (ds.sieve_epoch + 1) & 1);
                    blist_ctr = sv.blist_ctr;
                }
                clist_regions = 0;
                if (use_blist && clist_env) {
                    const size_t want = (size_t)n_mask_blocks * (size_t)clist_per_block + (size_t)kClistMaxRegions * 64;
                    if (ds.clist_cap < want) { // (allocated by the first pass that keeps a list: sets without the code filter never do)
                        if (ds.d_clist) {
                            HIP_TRY(ctx, hipStreamSynchronize(ds.stream));
                            HIP_TRY(ctx, hipFree(ds.d_clist));
                        }
                        ds.d_clist = nullptr;
                        ds.clist_cap = 0;
                        HIP_TRY(ctx, hipMalloc((void **)&ds.d_clist, want * 4));
                        ds.clist_cap = want;
                    }
                    if (!ds.d_clist_cnt) HIP_TRY(ctx, hipMalloc((void **)&ds.d_clist_cnt, (size_t)kClistMaxRegions * 4));
                    const int regions = apm_sieve2cf_blocks(sv, ds.n_cu);
                    if (regions >= 1 && regions <= kClistMaxRegions) {
                        clist_regions = regions;
                        clist_region_cap = clist_cap_env ? (uint32_t)clist_cap_env : (uint32_t)std::max<int64_t>(64, n_mask_blocks * clist_per_block / regions);
                        sv.clist = ds.d_clist;
                        sv.clist_cnt = ds.d_clist_cnt;
                        sv.clist_cap = clist_region_cap;
                    }
                }
                ds.last_clist_regions = clist_regions;
                ds.last_blist_ctr = blist_ctr;
                HIP_TRY(ctx, apm_launch_sieve2(sv, ds.n_cu, ds.stream));
                if (use_blist) ++ds.sieve_epoch; // (a launch that did not run leaves its counter set as it was: still zero)
                return note_launch(ctx, ds, "sieve");
            };
            auto verify_pass = [&](size_t v, int64_t je_v) -> int {
                VerifyLaunch &V = ctx->sieve.launches[v];
                ApmVerifyArgs va{};
                va.text = d_text;
                va.avail = avail;
                va.avail_pad = avail_pad;
                va.jb = jb;
                va.je = je_v;
                va.nrel = nrel;
                va.image = reinterpret_cast<const uint4 *>(ds.verify[v].d_image);
                va.image_len = (int)V.image.size();
                va.o_prefix = V.o_prefix;
                va.o_r2s = V.o_r2s;
                va.o_slots = V.o_slots;
                va.o_kext = V.o_kext;
                va.o_pat = V.o_pat;
                va.o_masks = V.o_masks;
                va.o_kinfo = V.o_kinfo;
                va.o_pinfo = V.o_pinfo;
                va.o_rc = V.o_rc;
                va.kinfo = ds.verify[v].d_kinfo;
                va.pinfo = reinterpret_cast<const uint2 *>(ds.verify[v].d_pinfo);
                va.kpart = ds.verify[v].d_kpart;
                va.pats = ds.verify[v].d_descs;
                va.counts = d_counts;
                va.n_pats = (int)V.descs.size();
                va.nk = (int)V.kinfo.size();
                va.k = ctx->k;
                va.band = band;
                va.code_shift = ctx->sieve.code_shift;
                va.stride = ctx->sieve.stride;
                va.masks = ds.d_masks;
                if (blist_ctr) {
                    va.blist = ds.d_blist;
                    va.blist_ctr = blist_ctr;
                }
                if (clist_regions) {
                    va.clist = ds.d_clist;
                    va.clist_cnt = ds.d_clist_cnt;
                    va.clist_cap = clist_region_cap;
                    va.clist_regions = clist_regions;
                    // (8: with 1 a planted occurrence's nominations scatter over as many waves, with 64 one wave walks two
                    // occurrences of its region one after the other -- 0.06 against 0.037 ms on sparse sets of long patterns,
                    // profiles/r03/clist_ab.txt; APM_CLIST_MIN_BATCH overrides, A/B aid)
                    static const int min_batch_env = getenv("APM_CLIST_MIN_BATCH") ? std::max(1, std::min(64, atoi(getenv("APM_CLIST_MIN_BATCH")))) : 8;
                    va.clist_min_batch = min_batch_env;
                }
                va.tile0 = p_lo;
                va.n_mask_blocks = n_mask_blocks;
#ifdef APM_MEASURE
                if (!ds.d_stats) HIP_TRY(ctx, hipMalloc((void **)&ds.d_stats, APM_STATS_BYTES));
                HIP_TRY(ctx, hipMemsetAsync(ds.d_stats, 0, APM_STATS_BYTES, ds.stream));
                va.stats = ds.d_stats;
#endif
                va.work = ds.d_work;
                if (!V.blocks_per_cu) V.blocks_per_cu = apm_verify_geometry(va, &V.threads);
                HIP_TRY(ctx, apm_launch_verify(va, V.threads, ds.n_cu * V.blocks_per_cu, &ds.work_epoch, ds.stream));
                return note_launch(ctx, ds, "verify");
            };
            // every launch group with a sieve pass of its own (code filter), when all of them fit a CU in that form ...
            bool per_launch = ctx->sieve.per_launch_sieve;
            for (size_t v = 0; per_launch && v < ctx->sieve.launches.size(); ++v) {
                VerifyLaunch &V = ctx->sieve.launches[v];
                if (!V.cf_threads) {
                    V.cf_blocks_per_cu = apm_sieve2cf_geometry((int)V.cf_image.size(), &V.cf_threads);
                    if (V.cf_blocks_per_cu < 1) V.cf_threads = -1; // does not fit a CU
                }
                if (V.cf_threads < 64) per_launch = false;
            }
            bool any_pass = false;
            for (size_t v = 0; v < ctx->sieve.launches.size(); ++v) {
                const int64_t je_v = std::min<int64_t>(je, nrel - ctx->sieve.launches[v].m_min + 1);
                if (je_v <= jb) continue;
                if (per_launch || !any_pass) { // ... else ONE pass over the set's shared bitmap, in front of the first verify launch
                    const int src = sieve_pass(per_launch ? (int)v : -1);
                    if (src) return src;
                    any_pass = true;
                }
                const int vrc = verify_pass(v, je_v);
                if (vrc) return vrc;
            }
            // (no launch had windows to decide: the tails get their own launch at the end of the call)
            sieve_run = true;
            }
        }
    }
    for (size_t t = 0; t < ctx->tiled.size(); ++t) {
        const TiledLaunch &L = ctx->tiled[t];
        if (L.kind == APM_KERNEL_BITPAR && L.m_max > 1024) { // one window per wave, one pattern per launch: full and truncated windows alike
            ApmScanArgs a{};
            a.text = d_text;
            a.avail = avail;
            a.jb = jb;
            a.je = je;
            a.nrel = nrel;
            a.tile0 = jb;
            a.pats = ds.tiled[t].d_descs;
            a.tables = ds.tiled[t].d_tables;
            a.lut = ds.tiled[t].d_lut;
            a.counts = d_counts;
            a.n_pats = 1;
            a.k = ctx->k;
            a.table_words = (int)L.tables.size();
            a.pos = sink;
            HIP_TRY(ctx, apm_launch_bitlong(a, L.m_max, ds.stream));
            { const int nrc = note_launch(ctx, ds, "bitpar"); if (nrc) return nrc; }
            continue;
        }
        const int64_t je_l = std::min<int64_t>(je, nrel - L.m_min + 1);
        if (je_l <= jb) continue;
        if (L.kind == APM_KERNEL_BANDED) {
            if ((fused_run || sieve_run) && L.sieved) continue; // decided by the sieve pipeline above (it cannot overflow: no fallback)
            ApmFilterArgs f{};
            f.text = d_text;
            f.avail = avail;
            f.avail_pad = avail + (int64_t)((16u - ((reinterpret_cast<uintptr_t>(d_text) + (uintptr_t)avail) & 15u)) & 15u);
            f.jb = jb;
            f.je = je_l;
            f.nrel = nrel;
            f.band = ctx->k / 2;
            f.front = f.band > 0 ? 16 : 0;
            f.tile0 = jb - (int64_t)((reinterpret_cast<uintptr_t>(d_text) + (uintptr_t)jb - (uintptr_t)f.front) & 15u);
            f.pats = ds.tiled[t].d_descs;
            f.image = reinterpret_cast<const uint4 *>(ds.tiled[t].d_image);
            f.image_len = (int)L.image.size();
            f.o_tab = L.o_tab;
            f.o_kid = L.o_kid;
            f.o_ovf = L.o_ovf;
            f.o_kinfo = L.o_kinfo;
            f.o_pinfo = L.o_pinfo;
            f.o_next = L.o_next;
            f.o_poff = L.o_poff;
            f.o_bmp = L.o_bmp;
            f.o_pat = L.o_pat;
            f.o_kext = L.o_kext;
            f.code_shift = L.code_shift;
            f.nk = (int)L.keys.size();
            f.nb = L.nb;
            f.lg_nb = L.lg_nb;
            f.n_ovf = (int)(L.ovf.size() / 2);
            f.qcap = L.qcap;
#ifdef APM_MEASURE
            if (L.stride == 1) { // APM_QCAP_S1 overrides the per-tile candidate queue of the per-position classes
                static const int q_env = getenv("APM_QCAP_S1") ? atoi(getenv("APM_QCAP_S1")) : 0;
                if (q_env >= 64 && q_env <= 8192) f.qcap = q_env;
            }
#endif
            f.key_len = L.key_len;
            f.stride = L.stride;
            f.counts = d_counts;
            f.n_cu = ds.n_cu;
            f.n_pats = (int)L.descs.size();
            f.k = ctx->k;
            f.tile_w = L.tile;
            f.tile_len = APM_FILTER_POS;
            f.ntiles = (je_l - f.tile0 + L.tile - 1) / L.tile;
            {
                static const int dma_env = getenv("APM_FILTER_DMA") ? atoi(getenv("APM_FILTER_DMA")) : 1;
                f.use_dma = (dma_env && (reinterpret_cast<uintptr_t>(d_text) & 15u) == 0 && f.avail_pad >= 16) ? 1 : 0;
            }
            // APM_FILTER_STREAM=0 forces the tile kernel (A/B aid); default: stream kernel for the sampled classes
            static const int stream_env = getenv("APM_FILTER_STREAM") ? atoi(getenv("APM_FILTER_STREAM")) : 1;
            // per-position classes stream only when candidates are expected to be rare (verification then
            // reads global text, dense 64-candidate batches); APM_FILTER_STREAM=2 forces, 3 forbids (A/B aid)
            const double hit_rate = (double)L.keys.size() / (double)(1ull << (2 * std::min(L.key_len, 8)));
            const bool stream_ok = L.stride > 1 || (f.band <= 1 && (stream_env == 2 || (stream_env != 3 && hit_rate < 1.0 / 200.0)));
            if (stream_env && stream_ok && (reinterpret_cast<uintptr_t>(d_text) & 15u) == 0 && f.avail_pad >= 16) {
                // wave-autonomous streaming kernel over 1 KiB chunks
                const int64_t p_lo = std::max<int64_t>(0, jb - f.band) & ~(int64_t)15;
                const int64_t p_hi = std::min<int64_t>(avail, je_l + L.m_max + f.band);
                f.tile0 = p_lo;
                f.ntiles = p_hi > p_lo ? (p_hi - p_lo + 1023) / 1024 : 0;
                if (!ctx->tiled[t].blocks_per_cu[2]) ctx->tiled[t].blocks_per_cu[2] = apm_stream_blocks_per_cu(f);
                if (tails_pending) {
                    f.n_tail = (int)ctx->stails.descs.size();
                    f.tail = ta;
                    tails_pending = false;
                }
                HIP_TRY(ctx, apm_launch_stream(f, ds.n_cu * L.blocks_per_cu[2], ds.stream));
                { const int nrc = note_launch(ctx, ds, "stream"); if (nrc) return nrc; }
                continue;
            }
            if (!ctx->tiled[t].blocks_per_cu[f.use_dma])
                ctx->tiled[t].blocks_per_cu[f.use_dma] =
                    apm_filter_blocks_per_cu(f.band, f.key_len, f.stride, f.use_dma, apm_filter_lds_bytes(f));
            if (tails_pending) {
                f.n_tail = (int)ctx->stails.descs.size();
                f.tail = ta;
                tails_pending = false;
            }
            {
                int bpc = L.blocks_per_cu[f.use_dma];
#ifdef APM_MEASURE
                static const int bpc_env = getenv("APM_BPC_CAP") ? atoi(getenv("APM_BPC_CAP")) : 0;
                if (bpc_env > 0) bpc = std::min(bpc_env, bpc);
#endif
                HIP_TRY(ctx, apm_launch_filter(f, ds.n_cu * bpc, ds.stream));
            }
            { const int nrc = note_launch(ctx, ds, "tile"); if (nrc) return nrc; }
            continue;
        }
        if (L.kind == APM_KERNEL_NFA) {
            ApmNfaArgs na{};
            na.text = d_text;
            na.avail = avail;
            na.jb = jb;
            na.je = je_l;
            na.nrel = nrel;
            na.tile0 = jb - (int64_t)((reinterpret_cast<uintptr_t>(d_text) + (uintptr_t)jb) & 15u);
            na.pats = ds.tiled[t].d_descs;
            na.classes = ds.tiled[t].d_bytes;
            na.cls_len = (int)L.bytes.size();
            memcpy(na.class_bytes, L.lut, 16);
            na.n_classes = L.nb;
            na.counts = d_counts;
            na.n_pats = (int)L.descs.size();
            na.k = ctx->k;
            na.pos = sink;
            HIP_TRY(ctx, apm_launch_nfa(na, ds.stream));
            { const int nrc = note_launch(ctx, ds, "nfa"); if (nrc) return nrc; }
            continue;
        }
        ApmScanArgs a{};
        a.text = d_text;
        a.avail = avail;
        a.jb = jb;
        a.je = je_l;
        a.nrel = nrel;
        a.tile0 = jb - (int64_t)((reinterpret_cast<uintptr_t>(d_text) + (uintptr_t)jb) & 15u);
        a.pats = ds.tiled[t].d_descs;
        a.bytes = ds.tiled[t].d_bytes;
        a.tables = ds.tiled[t].d_tables;
        a.lut = ds.tiled[t].d_lut;
        a.counts = d_counts;
        a.n_pats = (int)L.descs.size();
        a.k = ctx->k;
        a.tile = L.tile;
        a.halo = L.m_max - 1;
        a.table_words = (int)L.tables.size();
        a.bytes_len = (int)L.bytes.size();
        a.pos = sink;
        if (L.kind == APM_KERNEL_BITPAR) HIP_TRY(ctx, apm_launch_bitpar(a, ds.stream));
        else HIP_TRY(ctx, apm_launch_wavefront(a, ds.stream));
        { const int nrc = note_launch(ctx, ds, (L.kind == APM_KERNEL_BITPAR ? "bitpar" : "wavefront")); if (nrc) return nrc; }
    }
    ds.last_fused = fused_run;
    if (!sieve_run) { ds.last_mask_blocks = 0; ds.last_clist_regions = 0; }
    int rc = launch_generic_group(ctx, ds, ctx->longs, ds.d_long_descs, 2, d_text, avail, jb, je, nrel, d_counts, sink);
    if (rc) return rc;
    if (ctx->timing_on) HIP_TRY(ctx, hipEventRecord(ds.ev_mstop, ds.stream));
    if (!ctx->tails.descs.empty() && nrel - (int64_t)ctx->tails.m_max + 1 < je) {
        rc = launch_generic_group(ctx, ds, ctx->tails, ds.d_tail_descs, 1, d_text, avail, jb, je, nrel, d_counts, sink);
        if (rc) return rc;
    }
    if (tails_pending) {
        HIP_TRY(ctx, apm_launch_tail(ta, (int)ctx->stails.descs.size(), ds.stream));
        { const int nrc = note_launch(ctx, ds, "tail"); if (nrc) return nrc; }
    }
    if (!ctx->wtails.descs.empty() && nrel - (int64_t)ctx->wtails.m_max + 1 < je) { // truncated windows of the 128 < m <= 512 patterns
        ApmTailArgs tw{};
        tw.text = d_text;
        tw.jb = jb;
        tw.je = je;
        tw.nrel = nrel;
        tw.pats = ds.d_wtail_descs;
        tw.bytes = ds.d_allpat;
        tw.counts = d_counts;
        tw.k = ctx->k;
        tw.pos = sink;
        HIP_TRY(ctx, apm_launch_tail_wide(tw, (int)ctx->wtails.descs.size(), ds.stream));
        { const int nrc = note_launch(ctx, ds, "tail"); if (nrc) return nrc; }
    }
    if (!ctx->xtails.descs.empty() && nrel - (int64_t)ctx->xtails.m_max + 1 < je) { // ... of the 512 < m <= 1024 patterns
        ApmTailArgs tw{};
        tw.text = d_text;
        tw.jb = jb;
        tw.je = je;
        tw.nrel = nrel;
        tw.pats = ds.d_xtail_descs;
        tw.bytes = ds.d_allpat;
        tw.counts = d_counts;
        tw.k = ctx->k;
        tw.pos = sink;
        HIP_TRY(ctx, apm_launch_tail_xwide(tw, (int)ctx->xtails.descs.size(), ds.stream));
        { const int nrc = note_launch(ctx, ds, "tail"); if (nrc) return nrc; }
    }
    if (!ctx->trivial.empty()) {
        const int nt = (int)ctx->trivial.size();
        hipLaunchKernelGGL(apm_add_const_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, ds.stream, d_counts,
                           ds.d_trivial, nt, (unsigned long long)(oe - ob));
    }
    HIP_TRY(ctx, hipGetLastError());
    return APM_OK;
}

void account(apm_ctx *ctx, uint64_t n_total, uint64_t ob, uint64_t oe) {
    // algorithmic / evaluated cells for window starts [ob, oe) (already clipped to n-k)
    const uint64_t k = (uint64_t)ctx->k;
    const uint64_t limit = n_total > k ? n_total - k : 0;
    oe = std::min(oe, limit);
    if (oe <= ob) return;
    for (const auto &p : ctx->pats) {
        const uint64_t m = (uint64_t)p.m;
        const uint64_t full_end = n_total >= m ? std::min<uint64_t>(oe, n_total - m + 1) : 0;
        const uint64_t nfull = full_end > ob ? full_end - ob : 0;
        double cells = double(nfull) * double(m) * double(m);
        for (uint64_t j = std::max(ob, full_end); j < oe; ++j) { // <= m-1 truncated windows
            const double s = double(n_total - j);
            cells += s * s;
        }
        ctx->timing.windows += oe - ob;
        ctx->timing.cells_algorithmic += cells;
        if (p.kernel == APM_KERNEL_BANDED || p.kernel == APM_KERNEL_NFA)
            ctx->timing.cells_evaluated += double(oe - ob) * double(m) * double(2 * (ctx->k / 2) + 1); // upper bound
        else if (p.kernel != KERNEL_TRIVIAL)
            ctx->timing.cells_evaluated += cells;
    }
}

void begin_call(apm_ctx *ctx) {
    ctx->timing = apm_timing{};
    ctx->timing.n_devices = (int)ctx->devs.size();
    for (auto &ds : ctx->devs) {
        ds.text_bytes = 0;
        ds.launches = 0;
        ds.n_stamps = 0;
        ds.events_recorded = false;
    }
}

int load_rccl(apm_ctx *ctx) {
    RcclApi &r = ctx->rccl;
    if (r.ready) return APM_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (!r.handle) return fail(ctx, APM_ERR_COMM, "cannot load librccl: %s", dlerror());
    r.CommInitAll = (int (*)(void **, int, const int *))dlsym(r.handle, "ncclCommInitAll");
    r.CommDestroy = (int (*)(void *))dlsym(r.handle, "ncclCommDestroy");
    r.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(r.handle, "ncclAllReduce");
    r.GroupStart = (int (*)())dlsym(r.handle, "ncclGroupStart");
    r.GroupEnd = (int (*)())dlsym(r.handle, "ncclGroupEnd");
    if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.GroupStart || !r.GroupEnd)
        return fail(ctx, APM_ERR_COMM, "librccl lacks a required symbol");
    std::vector<int> devlist;
    for (auto &ds : ctx->devs) devlist.push_back(ds.dev);
    r.comms.assign(ctx->devs.size(), nullptr);
    const int rc = r.CommInitAll(r.comms.data(), (int)devlist.size(), devlist.data());
    if (rc != 0) return fail(ctx, APM_ERR_COMM, "ncclCommInitAll failed (%d)", rc);
    r.ready = true;
    return APM_OK;
}

// sum the per-device partial count vectors into counts[] (host)
int reduce_counts(apm_ctx *ctx, uint64_t *counts) {
    const int P = (int)ctx->pats.size();
    const auto t0 = clk::now();
    const size_t G = ctx->devs.size();
    bool done = false;
    // APM_FORCE_RCCL=1 runs the collective even on one device (test hook for the RCCL path)
    bool distinct = true; // (APM_DEVICES rehearsal: shards sharing a GPU are summed on the host)
    for (size_t a = 0; a < G; ++a)
        for (size_t b = a + 1; b < G; ++b)
            if (ctx->devs[a].dev == ctx->devs[b].dev) distinct = false;
    if (distinct && (G > 1 || getenv("APM_FORCE_RCCL")) && !getenv("APM_NO_RCCL")) {
        if (load_rccl(ctx) == APM_OK) {
            // one ncclAllReduce(sum, uint64 x P) per device, grouped (RCCL over xGMI);
            // replaces the MPI_Send/Recv + manual sum of database_over_ranks.c:174-195
            RcclApi &r = ctx->rccl;
            int rc = r.GroupStart();
            for (size_t g = 0; g < G && rc == 0; ++g) {
                hipSetDevice(ctx->devs[g].dev);
                rc = r.AllReduce(ctx->devs[g].d_counts, ctx->devs[g].d_counts, (size_t)P, /*ncclUint64*/ 5,
                                 /*ncclSum*/ 0, r.comms[g], ctx->devs[g].stream);
            }
            if (rc == 0) rc = r.GroupEnd();
            if (rc != 0) return fail(ctx, APM_ERR_COMM, "ncclAllReduce failed (%d)", rc);
            HIP_TRY(ctx, hipSetDevice(ctx->devs[0].dev));
            HIP_TRY(ctx, hipMemcpyAsync(counts, ctx->devs[0].d_counts, (size_t)P * 8, hipMemcpyDeviceToHost,
                                        ctx->devs[0].stream));
            for (auto &ds : ctx->devs) {
                HIP_TRY(ctx, hipSetDevice(ds.dev));
                HIP_TRY(ctx, hipStreamSynchronize(ds.stream));
            }
            done = true;
        }
    }
    if (!done) {
        std::vector<uint64_t> tmp((size_t)P);
        for (int i = 0; i < P; ++i) counts[i] = 0;
        for (auto &ds : ctx->devs) {
            HIP_TRY(ctx, hipSetDevice(ds.dev));
            HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), ds.d_counts, (size_t)P * 8, hipMemcpyDeviceToHost, ds.stream));
            HIP_TRY(ctx, hipStreamSynchronize(ds.stream));
            for (int i = 0; i < P; ++i) counts[i] += tmp[i];
        }
    }
    ctx->timing.reduce_ms = ms_since(t0);
    return APM_OK;
}

int collect_event_times(apm_ctx *ctx) {
    double kmax = 0, mmax = 0, hmax = 0;
    uint64_t bytes = 0;
    int launches = 0;
    for (auto &ds : ctx->devs) {
        bytes += ds.text_bytes;
        launches += ds.launches;
        if (!ds.events_recorded) continue;
        HIP_TRY(ctx, hipSetDevice(ds.dev));
        HIP_TRY(ctx, hipEventSynchronize(ds.ev_stop));
        float h = 0, kk = 0, mm = 0;
        hipEventElapsedTime(&h, ds.ev_start, ds.ev_kstart);
        hipEventElapsedTime(&kk, ds.ev_kstart, ds.ev_stop);
        if (hipEventElapsedTime(&mm, ds.ev_mstart, ds.ev_mstop) != hipSuccess) mm = 0;
        hmax = std::max<double>(hmax, h);
        kmax = std::max<double>(kmax, kk);
        mmax = std::max<double>(mmax, mm);
    }
    ctx->timing.h2d_ms = hmax;
    ctx->timing.kernel_ms = kmax;
    ctx->timing.main_kernel_ms = mmax;
    ctx->timing.text_bytes = bytes;
    ctx->timing.n_launches = launches;
    return APM_OK;
}

int ensure_text(apm_ctx *ctx, DeviceState &ds, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes <= ds.text_cap) return APM_OK;
    HIP_TRY(ctx, hipSetDevice(ds.dev));
    if (ds.d_text) {
        HIP_TRY(ctx, hipStreamSynchronize(ds.stream));
        hipFree(ds.d_text);
        ds.d_text = nullptr;
        ds.text_cap = 0;
    }
    HIP_TRY(ctx, hipMalloc((void **)&ds.d_text, bytes));
    ds.text_cap = bytes;
    return APM_OK;
}

int init_device(apm_ctx *ctx, DeviceState &ds, int dev) {
    ds.dev = dev;
    HIP_TRY(ctx, hipSetDevice(dev));
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0) ds.n_cu = ncu;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ds.own_stream, hipStreamNonBlocking));
    ds.stream = ds.own_stream;
    HIP_TRY(ctx, hipEventCreate(&ds.ev_start));
    HIP_TRY(ctx, hipEventCreate(&ds.ev_kstart));
    HIP_TRY(ctx, hipEventCreate(&ds.ev_mstart));
    HIP_TRY(ctx, hipEventCreate(&ds.ev_mstop));
    HIP_TRY(ctx, hipEventCreate(&ds.ev_stop));
    return APM_OK;
}

// the three host-level entry points share this: `stage(g, ds, lo, len)` must enqueue the
// bytes of global positions [lo, lo+len) into ds.d_text on ds.stream.
// Staging runs CONCURRENTLY, one host thread per device (the replacement of the reference's per-rank reads,
// /root/reference/src/database_over_ranks.c:141-166: there every MPI rank read its own piece at the same time; round 2
// staged device g's whole shard before touching device g + 1).  The scan launches follow from the calling thread as
// each device's staging thread returns: they are microseconds of host time, and the plan's cached launch geometry is
// not written from several threads.
template <typename Stage>
int count_sharded(apm_ctx *ctx, uint64_t n, uint64_t *counts, Stage stage) {
    if (!ctx) return APM_ERR_INVALID;
    if (!ctx->patterns_set) return fail(ctx, APM_ERR_STATE, "apm_set_patterns has not been called");
    if (!counts) return fail(ctx, APM_ERR_INVALID, "counts is NULL");
    const auto t0 = clk::now();
    begin_call(ctx);
    const bool timing_saved = ctx->timing_on;
    ctx->timing_on = true; // the host-level calls synchronise anyway; keep their event times
    struct Restore { apm_ctx *c; bool v; ~Restore() { c->timing_on = v; } } restore{ctx, timing_saved};
    const int G = (int)ctx->devs.size();
    const int P = (int)ctx->pats.size();
    const uint64_t halo = (uint64_t)std::max(ctx->m_max, 1) - 1;
    struct Shard { uint64_t ob = 0, oe = 0, lo = 0, len = 0; int rc = APM_OK; };
    std::vector<Shard> sh((size_t)G);
    auto stage_device = [&](int g) {
        DeviceState &ds = ctx->devs[g];
        Shard &S = sh[(size_t)g];
        apm_shard_range(n, ctx->k, g, G, &S.ob, &S.oe);
        S.lo = S.ob;
        const uint64_t hi = std::min<uint64_t>(n, S.oe + halo);
        S.len = hi > S.lo ? hi - S.lo : 0;
        S.rc = [&]() -> int {
            HIP_TRY(ctx, hipSetDevice(ds.dev));
            HIP_TRY(ctx, hipEventRecord(ds.ev_start, ds.stream));
            HIP_TRY(ctx, hipMemsetAsync(ds.d_counts, 0, std::max<size_t>((size_t)P * 8, 16), ds.stream));
            if (S.oe > S.ob) {
                int rc = ensure_text(ctx, ds, (size_t)S.len + 16);
                if (rc) return rc;
                rc = stage(g, ds, S.lo, S.len);
                if (rc) return rc;
            }
            HIP_TRY(ctx, hipEventRecord(ds.ev_kstart, ds.stream));
            return APM_OK;
        }();
    };
    std::vector<std::thread> th;
    for (int g = 1; g < G; ++g) th.emplace_back(stage_device, g);
    stage_device(0);
    int first_rc = APM_OK;
    for (int g = 0; g < G; ++g) {
        if (g > 0) th[(size_t)g - 1].join();
        DeviceState &ds = ctx->devs[g];
        const Shard &S = sh[(size_t)g];
        if (S.rc && !first_rc) first_rc = S.rc;
        if (first_rc) continue; // (keep joining)
        int rc = [&]() -> int {
            HIP_TRY(ctx, hipSetDevice(ds.dev));
            if (S.oe > S.ob) {
                const int r2 = scan_shard(ctx, ds, ds.d_text, S.lo, S.len, n, S.ob, S.oe, ds.d_counts);
                if (r2) return r2;
                account(ctx, n, S.ob, S.oe);
            } else {
                HIP_TRY(ctx, hipEventRecord(ds.ev_mstart, ds.stream));
                HIP_TRY(ctx, hipEventRecord(ds.ev_mstop, ds.stream));
            }
            HIP_TRY(ctx, hipEventRecord(ds.ev_stop, ds.stream));
            ds.events_recorded = true;
            return APM_OK;
        }();
        if (rc && !first_rc) first_rc = rc;
    }
    if (first_rc) return first_rc;
    int rc = reduce_counts(ctx, counts);
    if (rc) return rc;
    rc = collect_event_times(ctx);
    if (rc) return rc;
    ctx->timing.total_ms = ms_since(t0);
    return APM_OK;
}

// Host bytes -> device text through the context's ring of pinned staging buffers (apm_count_buffer, apm_count_file).
// `read(off, dst, len)` must put bytes [off, off+len) of the source into dst (false: I/O error).  Device g of G owns the
// buffers [g * per, (g + 1) * per) of the ring, two per reader thread: a reader takes the next 8 MiB chunk of the
// device's shard, reads it into one of its two buffers, enqueues the copy on the device's stream and reads the next
// chunk into the other while that one travels (a buffer is reused once its copy's event has fired).  A single reader
// runs at a fraction of the PCIe link; several side by side keep it busy (2 -> 35, 4 -> 44, 8 -> 40 GB/s on one
// device), and no pages of the caller's buffer are pinned per call.
int stage_through_ring(apm_ctx *ctx, int g, int G, DeviceState &ds, uint64_t lo, uint64_t len,
                       const std::function<bool(uint64_t, uint8_t *, size_t)> &read) {
    const size_t CH = apm_ctx::STAGE_BYTES;
    static const int n_readers = [] {
        const char *e = getenv("APM_INGEST_THREADS");
        const int hw = (int)std::thread::hardware_concurrency();
        int t = e ? atoi(e) : std::min(4, hw > 1 ? hw / 2 : 1);
        return t < 1 ? 1 : t;
    }();
    const int per = std::max(2, (apm_ctx::N_STAGE / std::max(G, 1)) & ~1); // buffers of this device (G <= N_STAGE / 2)
    if ((g + 1) * per > apm_ctx::N_STAGE) return fail(ctx, APM_ERR_UNSUPPORTED, "more devices than staging buffers");
    const int b0 = g * per;
    const uint64_t n_chunks = (len + CH - 1) / CH;
    const int nt = (int)std::min<uint64_t>((uint64_t)std::min(n_readers, per / 2), n_chunks);
    for (int b = b0; b < b0 + 2 * nt; ++b) { // (current device = ds.dev; these buffers are this device's alone)
        if (!ctx->stage[b] && hipHostMalloc((void **)&ctx->stage[b], CH, hipHostMallocDefault) != hipSuccess)
            return fail(ctx, APM_ERR_NOMEM, "cannot allocate pinned staging buffers");
        if (!ds.ev_stage[b]) HIP_TRY(ctx, hipEventCreateWithFlags(&ds.ev_stage[b], hipEventDisableTiming));
    }
    std::atomic<uint64_t> next{0};
    std::atomic<int> bad{0};
    auto reader = [&](int t) {
        if (hipSetDevice(ds.dev) != hipSuccess) { bad = 2; return; }
        int flip = 0;
        bool used[2] = {false, false};
        for (;;) {
            const uint64_t c = next.fetch_add(1);
            if (c >= n_chunks || bad.load()) break;
            const int b = b0 + 2 * t + flip;
            if (used[flip] && hipEventSynchronize(ds.ev_stage[b]) != hipSuccess) { bad = 2; break; }
            const uint64_t off = c * CH;
            const size_t want = (size_t)std::min<uint64_t>(CH, len - off);
            if (!read(lo + off, ctx->stage[b], want)) { bad = 1; break; }
            if (hipMemcpyAsync(ds.d_text + off, ctx->stage[b], want, hipMemcpyHostToDevice, ds.stream) != hipSuccess ||
                hipEventRecord(ds.ev_stage[b], ds.stream) != hipSuccess) { bad = 2; break; }
            used[flip] = true;
            flip ^= 1;
        }
        for (int f = 0; f < 2; ++f) // the buffers are free again when this call returns (the next call may be another device's)
            if (used[f] && hipEventSynchronize(ds.ev_stage[b0 + 2 * t + f]) != hipSuccess) bad = 2;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(reader, t);
    if (nt > 0) reader(0);
    for (auto &t : th) t.join();
    if (bad.load() == 1) return fail(ctx, APM_ERR_IO, "Unable to copy %llu byte(s) from text file", (unsigned long long)len);
    if (bad.load()) return fail(ctx, APM_ERR_HIP, "staging copy failed: %s", hipGetErrorString(hipGetLastError()));
    return APM_OK;
}

} // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

int apm_abi_version(void) { return APM_ABI_VERSION; }

int apm_device_count(void) {
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice) return 0;
    if (e != hipSuccess) {
        g_create_error = std::string("hipGetDeviceCount failed: ") + hipGetErrorString(e);
        return APM_ERR_HIP;
    }
    return n;
}

const char *apm_last_error(const apm_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

static int create_common(apm_ctx **out, const std::vector<int> &devs, bool multi) {
    apm_ctx *ctx = new (std::nothrow) apm_ctx();
    if (!ctx) return fail(nullptr, APM_ERR_NOMEM, "out of memory");
    ctx->multi = multi;
    ctx->devs.resize(devs.size());
    for (size_t i = 0; i < devs.size(); ++i) {
        const int rc = init_device(ctx, ctx->devs[i], devs[i]);
        if (rc) {
            g_create_error = ctx->err;
            apm_destroy(ctx);
            return rc;
        }
    }
    *out = ctx;
    return APM_OK;
}

int apm_create(apm_ctx **ctx, int n_devices) {
    if (!ctx) return fail(nullptr, APM_ERR_INVALID, "ctx is NULL");
    *ctx = nullptr;
    const int have = apm_device_count();
    if (have < 0) return have;
    if (have == 0) return fail(nullptr, APM_ERR_NO_DEVICE, "no HIP device visible (this engine has no CPU fallback)");
    std::vector<int> devs;
    // APM_DEVICES=a,b,...: the device ids of the context, in shard order, instead of 0 .. n_devices-1.  An id may repeat
    // ("0,0"): several shards then share one GPU, each with its own stream, text buffer and count vector -- the rehearsal of
    // the multi-device path on a one-GPU box (tests); the partial counts are then summed on the host (RCCL wants one rank
    // per device).
    if (const char *e = getenv("APM_DEVICES")) {
        for (const char *p = e; *p;) {
            char *end = nullptr;
            const long id = strtol(p, &end, 10);
            if (end == p || id < 0 || id >= have) return fail(nullptr, APM_ERR_NO_DEVICE, "APM_DEVICES: bad device list <%s> (%d visible)", e, have);
            devs.push_back((int)id);
            p = *end == ',' ? end + 1 : end;
            if (*end && *end != ',') return fail(nullptr, APM_ERR_NO_DEVICE, "APM_DEVICES: bad device list <%s>", e);
        }
        if (devs.empty() || (int)devs.size() > apm_ctx::N_STAGE / 2) return fail(nullptr, APM_ERR_NO_DEVICE, "APM_DEVICES: bad device list <%s>", e);
        if (n_devices > 0 && n_devices != (int)devs.size())
            return fail(nullptr, APM_ERR_NO_DEVICE, "%d devices requested, APM_DEVICES lists %d", n_devices, (int)devs.size());
        return create_common(ctx, devs, true);
    }
    if (n_devices <= 0) n_devices = have;
    if (n_devices > have) return fail(nullptr, APM_ERR_NO_DEVICE, "%d devices requested, %d visible", n_devices, have);
    if (n_devices > apm_ctx::N_STAGE / 2) return fail(nullptr, APM_ERR_NO_DEVICE, "at most %d devices per context", apm_ctx::N_STAGE / 2);
    for (int i = 0; i < n_devices; ++i) devs.push_back(i);
    return create_common(ctx, devs, true);
}

int apm_create_on_device(apm_ctx **ctx, int device_id) {
    if (!ctx) return fail(nullptr, APM_ERR_INVALID, "ctx is NULL");
    *ctx = nullptr;
    const int have = apm_device_count();
    if (have < 0) return have;
    if (have == 0) return fail(nullptr, APM_ERR_NO_DEVICE, "no HIP device visible (this engine has no CPU fallback)");
    if (device_id < 0 || device_id >= have)
        return fail(nullptr, APM_ERR_NO_DEVICE, "device %d out of range (%d visible)", device_id, have);
    return create_common(ctx, std::vector<int>{device_id}, false);
}

void apm_destroy(apm_ctx *ctx) {
    if (!ctx) return;
    for (apm_ctx *ch : ctx->children) apm_destroy(ch);
    ctx->children.clear();
    if (ctx->rccl.ready)
        for (void *c : ctx->rccl.comms) if (c) ctx->rccl.CommDestroy(c);
    for (auto &ds : ctx->devs) {
        if (ds.dev < 0) continue;
        hipSetDevice(ds.dev);
        if (ds.own_stream) hipStreamSynchronize(ds.own_stream);
        free_device_plan(ds);
        if (ds.d_scratch) hipFree(ds.d_scratch);
        if (ds.d_text) hipFree(ds.d_text);
        if (ds.d_masks) hipFree(ds.d_masks);
        if (ds.d_stats) hipFree(ds.d_stats);
        if (ds.d_work) hipFree(ds.d_work);
        if (ds.d_blist) hipFree(ds.d_blist);
        if (ds.d_clist) hipFree(ds.d_clist);
        if (ds.d_clist_cnt) hipFree(ds.d_clist_cnt);
        for (hipEvent_t e : ds.ev_stage) if (e) hipEventDestroy(e);
        for (hipEvent_t e : ds.ev_launch) if (e) hipEventDestroy(e);
        for (hipEvent_t e : {ds.ev_start, ds.ev_kstart, ds.ev_mstart, ds.ev_mstop, ds.ev_stop}) if (e) hipEventDestroy(e);
        if (ds.own_stream) hipStreamDestroy(ds.own_stream);
    }
    for (int b = 0; b < apm_ctx::N_STAGE; ++b)
        if (ctx->stage[b]) hipHostFree(ctx->stage[b]);
    delete ctx;
}

int apm_set_stream(apm_ctx *ctx, void *hip_stream) {
    if (!ctx) return APM_ERR_INVALID;
    if (ctx->devs.size() != 1) return fail(ctx, APM_ERR_STATE, "apm_set_stream needs a single-device context");
    ctx->devs[0].stream = hip_stream == APM_STREAM_OWN ? ctx->devs[0].own_stream : (hipStream_t)hip_stream;
    return APM_OK;
}

static bool pattern_sharded(const apm_ctx *ctx) { return ctx->partition == APM_PARTITION_PATTERNS && ctx->devs.size() > 1; }

// (re)build the children of a pattern-sharded context from ctx->pats / ctx->k / ctx->kernel: contiguous slices of the
// pattern list, as even as they come (the replacement of /root/reference/src/patterns_over_ranks.c:160-182, where the
// master dealt patterns to the ranks by a cost model and every rank read the whole file)
static int build_children(apm_ctx *ctx) {
    const int G = (int)ctx->devs.size(), P = (int)ctx->pats.size();
    if (ctx->children.empty()) {
        ctx->children.assign((size_t)G, nullptr);
        for (int g = 0; g < G; ++g) {
            const int rc = create_common(&ctx->children[(size_t)g], std::vector<int>{ctx->devs[(size_t)g].dev}, false);
            if (rc) return fail(ctx, rc, "pattern-sharded context: device %d: %s", ctx->devs[(size_t)g].dev, g_create_error.c_str());
        }
    }
    ctx->pat_first.assign((size_t)G + 1, 0);
    for (int g = 0; g <= G; ++g) ctx->pat_first[(size_t)g] = (int)((long long)P * g / G);
    for (int g = 0; g < G; ++g) {
        apm_ctx *ch = ctx->children[(size_t)g];
        const int a = ctx->pat_first[(size_t)g], b = ctx->pat_first[(size_t)g + 1];
        ch->timing_on = ctx->timing_on;
        ch->patterns_set = false;
        if (b <= a) continue; // (fewer patterns than devices)
        std::vector<const char *> pp;
        std::vector<int> ll;
        for (int i = a; i < b; ++i) { pp.push_back(ctx->pats[(size_t)i].bytes.data()); ll.push_back(ctx->pats[(size_t)i].m); }
        ch->kernel = ctx->kernel;
        const int rc = apm_set_patterns(ch, b - a, pp.data(), ll.data(), ctx->k);
        if (rc) return fail(ctx, rc, "%s", ch->err.c_str());
        for (int i = a; i < b; ++i) ctx->pats[(size_t)i].kernel = ch->pats[(size_t)(i - a)].kernel;
    }
    return APM_OK;
}

// run fn(child, counts of its slice) on every child at once, one host thread per device
static int for_children(apm_ctx *ctx, uint64_t *counts, const std::function<int(apm_ctx *, uint64_t *)> &fn) {
    if (!ctx->patterns_set) return fail(ctx, APM_ERR_STATE, "apm_set_patterns has not been called");
    if (!counts) return fail(ctx, APM_ERR_INVALID, "counts is NULL");
    const auto t0 = clk::now();
    const int G = (int)ctx->children.size();
    std::vector<int> rcs((size_t)G, APM_OK);
    auto work = [&](int g) {
        const int a = ctx->pat_first[(size_t)g], b = ctx->pat_first[(size_t)g + 1];
        if (b > a) rcs[(size_t)g] = fn(ctx->children[(size_t)g], counts + a);
    };
    std::vector<std::thread> th;
    for (int g = 1; g < G; ++g) th.emplace_back(work, g);
    work(0);
    for (auto &t : th) t.join();
    ctx->timing = apm_timing{};
    ctx->timing.n_devices = G;
    for (int g = 0; g < G; ++g) {
        if (rcs[(size_t)g]) return fail(ctx, rcs[(size_t)g], "%s", ctx->children[(size_t)g]->err.c_str());
        if (ctx->pat_first[(size_t)g + 1] <= ctx->pat_first[(size_t)g]) continue;
        const apm_timing &t = ctx->children[(size_t)g]->timing;
        ctx->timing.h2d_ms = std::max(ctx->timing.h2d_ms, t.h2d_ms);
        ctx->timing.kernel_ms = std::max(ctx->timing.kernel_ms, t.kernel_ms);
        ctx->timing.main_kernel_ms = std::max(ctx->timing.main_kernel_ms, t.main_kernel_ms);
        ctx->timing.text_bytes += t.text_bytes;
        ctx->timing.windows += t.windows;
        ctx->timing.cells_algorithmic += t.cells_algorithmic;
        ctx->timing.cells_evaluated += t.cells_evaluated;
        ctx->timing.n_launches += t.n_launches;
    }
    ctx->timing.total_ms = ms_since(t0);
    return APM_OK;
}

int apm_set_partition(apm_ctx *ctx, int partition) {
    if (!ctx) return APM_ERR_INVALID;
    if (partition != APM_PARTITION_TEXT && partition != APM_PARTITION_PATTERNS) return fail(ctx, APM_ERR_INVALID, "unknown partition %d", partition);
    if (partition == ctx->partition) return APM_OK;
    ctx->partition = partition;
    if (ctx->pats.empty()) return APM_OK;
    ctx->patterns_set = false;
    const int rc = pattern_sharded(ctx) ? build_children(ctx) : build_plan(ctx);
    if (rc) return rc;
    ctx->patterns_set = true;
    return APM_OK;
}

int apm_set_patterns(apm_ctx *ctx, int n_patterns, const char *const *pat, const int *len, int k) {
    if (!ctx) return APM_ERR_INVALID;
    if (n_patterns <= 0 || n_patterns > APM_MAX_PATTERNS || !pat || !len)
        return fail(ctx, APM_ERR_INVALID, "need 1..%d patterns", APM_MAX_PATTERNS);
    if (k < 0) return fail(ctx, APM_ERR_INVALID, "distance must be >= 0 (the reference reads out of bounds for k<0)");
    std::vector<PatternInfo> v((size_t)n_patterns);
    for (int i = 0; i < n_patterns; ++i) {
        if (!pat[i] || len[i] <= 0) return fail(ctx, APM_ERR_INVALID, "pattern %d is empty", i);
        if (len[i] > APM_MAX_PATTERN_LEN)
            return fail(ctx, APM_ERR_UNSUPPORTED, "pattern %d longer than %d bytes", i, APM_MAX_PATTERN_LEN);
        v[i].bytes.assign(pat[i], pat[i] + len[i]);
        v[i].m = len[i];
    }
    ctx->pats.swap(v);
    ctx->k = k;
    ctx->patterns_set = false;
    const int rc = pattern_sharded(ctx) ? build_children(ctx) : build_plan(ctx);
    if (rc) return rc;
    ctx->patterns_set = true;
    return APM_OK;
}

int apm_set_timing(apm_ctx *ctx, int enabled) {
    if (!ctx) return APM_ERR_INVALID;
    ctx->timing_on = enabled != 0;
    for (apm_ctx *ch : ctx->children) if (ch) ch->timing_on = ctx->timing_on;
    return APM_OK;
}

int apm_set_kernel(apm_ctx *ctx, int kernel) {
    if (!ctx) return APM_ERR_INVALID;
    if (kernel < APM_KERNEL_AUTO || kernel > APM_KERNEL_NFA) return fail(ctx, APM_ERR_INVALID, "unknown kernel variant %d", kernel);
    const int old = ctx->kernel;
    ctx->kernel = kernel;
    if (ctx->patterns_set || !ctx->pats.empty()) {
        ctx->patterns_set = false;
        auto rebuild = [&]() { return pattern_sharded(ctx) ? build_children(ctx) : build_plan(ctx); };
        const int rc = rebuild();
        if (rc) {
            const std::string msg = ctx->err;
            ctx->kernel = old;
            if (rebuild() == APM_OK) ctx->patterns_set = true;
            ctx->err = msg;
            return rc;
        }
        ctx->patterns_set = true;
    }
    return APM_OK;
}

int apm_pattern_kernel(const apm_ctx *ctx, int i) {
    if (!ctx || i < 0 || i >= (int)ctx->pats.size()) return APM_ERR_INVALID;
    return ctx->pats[i].kernel == KERNEL_TRIVIAL ? APM_KERNEL_AUTO : ctx->pats[i].kernel;
}

int apm_shard_range(uint64_t n_total, int k, int shard, int n_shards, uint64_t *own_begin, uint64_t *own_end) {
    if (!own_begin || !own_end || n_shards <= 0 || shard < 0 || shard >= n_shards || k < 0) return APM_ERR_INVALID;
    const uint64_t limit = n_total > (uint64_t)k ? n_total - (uint64_t)k : 0;
    auto cut = [&](int s) -> uint64_t {
        if (s <= 0) return 0;
        if (s >= n_shards) return limit;
        const uint64_t c = (uint64_t)((unsigned __int128)limit * (unsigned)s / (unsigned)n_shards);
        return std::min<uint64_t>(limit, c & ~(uint64_t)15);
    };
    *own_begin = cut(shard);
    *own_end = cut(shard + 1);
    return APM_OK;
}

int apm_count_shard_device(apm_ctx *ctx, const void *d_text, uint64_t text_off, uint64_t text_len,
                           uint64_t n_total, uint64_t own_begin, uint64_t own_end, uint64_t *d_counts) {
    if (!ctx) return APM_ERR_INVALID;
    if (!ctx->patterns_set) return fail(ctx, APM_ERR_STATE, "apm_set_patterns has not been called");
    if (ctx->devs.size() != 1) return fail(ctx, APM_ERR_STATE, "apm_count_shard_device needs a single-device context");
    if (!d_counts || (!d_text && text_len)) return fail(ctx, APM_ERR_INVALID, "NULL device pointer");
    if (text_off + text_len > n_total || own_begin > own_end)
        return fail(ctx, APM_ERR_INVALID, "inconsistent shard description");
    begin_call(ctx);
    DeviceState &ds = ctx->devs[0];
    HIP_TRY(ctx, hipSetDevice(ds.dev));
    if (ctx->timing_on) {
        HIP_TRY(ctx, hipEventRecord(ds.ev_start, ds.stream));
        HIP_TRY(ctx, hipEventRecord(ds.ev_kstart, ds.stream));
    }
    const int rc = scan_shard(ctx, ds, (const uint8_t *)d_text, text_off, text_len, n_total, own_begin, own_end,
                              (unsigned long long *)d_counts);
    if (rc) return rc;
    if (ctx->timing_on) {
        HIP_TRY(ctx, hipEventRecord(ds.ev_stop, ds.stream));
        ds.events_recorded = true;
    }
    account(ctx, n_total, own_begin, own_end);
    return APM_OK;
}

int apm_count_buffer(apm_ctx *ctx, const uint8_t *text, uint64_t n, uint64_t *counts) {
    if (!ctx) return APM_ERR_INVALID;
    if (!text && n) return fail(ctx, APM_ERR_INVALID, "text is NULL");
    if (pattern_sharded(ctx)) return for_children(ctx, counts, [&](apm_ctx *ch, uint64_t *c) { return apm_count_buffer(ch, text, n, c); });
    const int G = (int)ctx->devs.size();
    return count_sharded(ctx, n, counts, [&](int g, DeviceState &ds, uint64_t lo, uint64_t len) -> int {
        if (len < (1u << 20)) { // small: one pageable copy (the runtime stages it itself)
            HIP_TRY(ctx, hipMemcpyAsync(ds.d_text, text + lo, (size_t)len, hipMemcpyHostToDevice, ds.stream));
            return APM_OK;
        }
        // through the pinned ring: round 2 pinned the caller's whole buffer with hipHostRegister on every call
        return stage_through_ring(ctx, g, G, ds, lo, len, [&](uint64_t off, uint8_t *dst, size_t want) {
            memcpy(dst, text + off, want);
            return true;
        });
    });
}

int apm_count_file(apm_ctx *ctx, const char *path, uint64_t *counts) {
    if (!ctx) return APM_ERR_INVALID;
    if (!path) return fail(ctx, APM_ERR_INVALID, "path is NULL");
    if (pattern_sharded(ctx)) { // every device reads the whole file (as every rank of the reference's PATTERNS_OVER_RANKS did)
        const int fdt = open(path, O_RDONLY);
        if (fdt < 0) return fail(ctx, APM_ERR_IO, "Unable to open the text file <%s>", path);
        close(fdt);
        return for_children(ctx, counts, [&](apm_ctx *ch, uint64_t *c) { return apm_count_file(ch, path, c); });
    }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(ctx, APM_ERR_IO, "Unable to open the text file <%s>", path);
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        close(fd);
        return fail(ctx, APM_ERR_IO, "Unable to stat the text file <%s>", path);
    }
    const uint64_t n = (uint64_t)st.st_size;
    const int G = (int)ctx->devs.size();
    // chunked ingest, all devices at once (stage_through_ring): pread out of the page cache into pinned buffers
    const int rc = count_sharded(ctx, n, counts, [&](int g, DeviceState &ds, uint64_t lo, uint64_t len) -> int {
        return stage_through_ring(ctx, g, G, ds, lo, len, [&](uint64_t off, uint8_t *dst, size_t want) {
            size_t got = 0;
            while (got < want) {
                const ssize_t r = pread(fd, dst + got, want - got, (off_t)(off + got));
                if (r <= 0) return false;
                got += (size_t)r;
            }
            return true;
        });
    });
    close(fd);
    return rc;
}

int apm_find_buffer(apm_ctx *ctx, const uint8_t *text, uint64_t n, int pattern_index, uint64_t *positions,
                    uint64_t capacity, uint64_t *n_found) {
    if (!ctx) return APM_ERR_INVALID;
    if (!ctx->patterns_set) return fail(ctx, APM_ERR_STATE, "apm_set_patterns has not been called");
    if (pattern_index < 0 || pattern_index >= (int)ctx->pats.size() || !n_found || (!positions && capacity) || (!text && n))
        return fail(ctx, APM_ERR_INVALID, "bad argument to apm_find_buffer");
    if (pattern_sharded(ctx)) { // the device that holds the pattern
        size_t g = 0;
        while (g + 1 < ctx->children.size() && pattern_index >= ctx->pat_first[g + 1]) ++g;
        const int rc = apm_find_buffer(ctx->children[g], text, n, pattern_index - ctx->pat_first[g], positions, capacity, n_found);
        if (rc) return fail(ctx, rc, "%s", ctx->children[g]->err.c_str());
        return APM_OK;
    }
    // run the one pattern through the full-DP kernels with a position sink, then restore the plan
    const std::vector<PatternInfo> saved = ctx->pats;
    const int saved_kernel = ctx->kernel;
    const PatternInfo one = saved[(size_t)pattern_index];
    auto restore = [&]() {
        ctx->pats = saved;
        ctx->kernel = saved_kernel;
        ctx->find_active = false;
        for (auto &ds : ctx->devs) {
            hipSetDevice(ds.dev);
            if (ds.d_pos_out) hipFree(ds.d_pos_out), ds.d_pos_out = nullptr;
            if (ds.d_pos_count) hipFree(ds.d_pos_count), ds.d_pos_count = nullptr;
        }
        const std::string keep = ctx->err;
        ctx->patterns_set = (build_plan(ctx) == APM_OK);
        if (!keep.empty()) ctx->err = keep;
    };
    ctx->pats.assign(1, one);
    ctx->kernel = one.m <= APM_BITPAR_MAX_M ? APM_KERNEL_BITPAR : APM_KERNEL_GENERIC;
    int rc = build_plan(ctx);
    if (rc) { restore(); return rc; }
    for (auto &ds : ctx->devs) {
        if (hipSetDevice(ds.dev) != hipSuccess ||
            hipMalloc((void **)&ds.d_pos_out, (size_t)std::max<uint64_t>(capacity, 1) * 8) != hipSuccess ||
            hipMalloc((void **)&ds.d_pos_count, 16) != hipSuccess || hipMemset(ds.d_pos_count, 0, 16) != hipSuccess) {
            restore();
            return fail(ctx, APM_ERR_NOMEM, "cannot allocate the position buffer (%llu entries)", (unsigned long long)capacity);
        }
        ds.pos_cap = capacity;
    }
    ctx->find_active = true;
    ctx->err.clear();
    uint64_t cnt1 = 0;
    rc = apm_count_buffer(ctx, text, n, &cnt1);
    std::vector<uint64_t> all;
    uint64_t total = 0;
    if (rc == APM_OK) {
        for (auto &ds : ctx->devs) {
            unsigned long long c = 0;
            if (hipSetDevice(ds.dev) != hipSuccess || hipMemcpy(&c, ds.d_pos_count, 8, hipMemcpyDeviceToHost) != hipSuccess) {
                rc = fail(ctx, APM_ERR_HIP, "cannot read back the match positions");
                break;
            }
            total += c;
            const size_t take = (size_t)std::min<uint64_t>(c, capacity);
            const size_t at = all.size();
            all.resize(at + take);
            if (take && hipMemcpy(all.data() + at, ds.d_pos_out, take * 8, hipMemcpyDeviceToHost) != hipSuccess) {
                rc = fail(ctx, APM_ERR_HIP, "cannot read back the match positions");
                break;
            }
        }
    }
    if (rc == APM_OK) {
        std::sort(all.begin(), all.end());
        for (size_t i = 0; i < all.size() && i < capacity; ++i) positions[i] = all[i];
        *n_found = total;
        if (total != cnt1) rc = fail(ctx, APM_ERR_STATE, "position sink count %llu != match count %llu", (unsigned long long)total, (unsigned long long)cnt1);
    }
    restore();
    return rc;
}

int apm_synth_fill_device(apm_ctx *ctx, void *d_dst, uint64_t global_off, uint64_t len, uint64_t seed) {
    if (!ctx) return APM_ERR_INVALID;
    if (ctx->devs.size() != 1) return fail(ctx, APM_ERR_STATE, "apm_synth_fill_device needs a single-device context");
    if (!d_dst && len) return fail(ctx, APM_ERR_INVALID, "NULL device pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->devs[0].dev));
    HIP_TRY(ctx, apm_launch_synth((uint8_t *)d_dst, global_off, len, seed, ctx->devs[0].stream));
    return APM_OK;
}

void apm_synth_fill_host(uint8_t *dst, uint64_t global_off, uint64_t len, uint64_t seed) {
    for (uint64_t i = 0; i < len; ++i) dst[i] = apm_synth_byte(global_off + i, seed);
}

int apm_count_synthetic(apm_ctx *ctx, uint64_t n, uint64_t seed, uint64_t *counts) {
    if (!ctx) return APM_ERR_INVALID;
    if (pattern_sharded(ctx)) return for_children(ctx, counts, [&](apm_ctx *ch, uint64_t *c) { return apm_count_synthetic(ch, n, seed, c); });
    return count_sharded(ctx, n, counts, [&](int, DeviceState &ds, uint64_t lo, uint64_t len) -> int {
        HIP_TRY(ctx, apm_launch_synth(ds.d_text, lo, len, seed, ds.stream));
        return APM_OK;
    });
}

int apm_get_timing(const apm_ctx *cctx, apm_timing *out) {
    apm_ctx *ctx = const_cast<apm_ctx *>(cctx);
    if (!ctx || !out) return APM_ERR_INVALID;
    if (pattern_sharded(ctx)) { *out = ctx->timing; return APM_OK; } // (aggregated by the call itself)
    const int rc = collect_event_times(ctx);
    if (rc) return rc;
    *out = ctx->timing;
    return APM_OK;
}

int apm_get_launch_times(const apm_ctx *cctx, int max, double *ms, const char **labels) {
    apm_ctx *ctx = const_cast<apm_ctx *>(cctx);
    if (!ctx || max < 0 || (max > 0 && !ms)) return APM_ERR_INVALID;
    if (ctx->devs.empty()) return 0;
    DeviceState &ds = ctx->devs[0];
    if (!ds.events_recorded || ds.n_stamps == 0) return 0;
    HIP_TRY(ctx, hipSetDevice(ds.dev));
    HIP_TRY(ctx, hipEventSynchronize(ds.ev_launch[ds.n_stamps - 1]));
    const int n = std::min(max, ds.n_stamps);
    for (int i = 0; i < n; ++i) {
        float t = 0;
        if (hipEventElapsedTime(&t, i ? ds.ev_launch[i - 1] : ds.ev_mstart, ds.ev_launch[i]) != hipSuccess) t = 0;
        ms[i] = t;
        if (labels) labels[i] = ds.launch_label[i];
    }
    return n;
}

int apm_get_stat(const apm_ctx *cctx, const char *name, double *value) {
    apm_ctx *ctx = const_cast<apm_ctx *>(cctx);
    if (!ctx || !name || !value || ctx->devs.empty()) return APM_ERR_INVALID;
    DeviceState &ds = ctx->devs[0];
    const std::string n = name;
    if (n == "sieve_on") { *value = ctx->sieve.on ? 1 : 0; return APM_OK; }
    if (n == "sieve_rate") { *value = ctx->sieve.rate; return APM_OK; }
    if (n == "sieve_fused") { *value = ds.last_fused ? 1 : 0; return APM_OK; }
    if (n == "sieve_cf") { *value = (ctx->sieve.on && ctx->sieve.per_launch_sieve && ctx->sieve.launches[0].cf_threads >= 64) ? (double)ctx->sieve.launches[0].cf_threads : 0.0; return APM_OK; } // (after a call: workgroup size of the code-filter form, 0 = plain sieve)
    if (n == "sieve_weak_frac") { *value = ctx->sieve.weak_frac; return APM_OK; }
    if (n == "sieve_cf_bytes") { *value = ctx->sieve.per_launch_sieve ? (double)ctx->sieve.launches[0].cf_image.size() : 0.0; return APM_OK; }
    if (n == "sieve_stride") { *value = ctx->sieve.on ? (double)ctx->sieve.stride : 0.0; return APM_OK; }
    if (n == "sieve_clist") { *value = ds.last_clist_regions ? 1.0 : 0.0; return APM_OK; }
    if (n == "sieve_mask_bytes") { // what the last sieve pass handed over: mask rows, or list entries + the rows of the overflow blocks
        *value = (double)ds.last_mask_blocks * 256.0;
        if (ds.last_clist_regions && ds.last_mask_blocks > 0) {
            HIP_TRY(ctx, hipSetDevice(ds.dev));
            std::vector<uint32_t> cnt((size_t)ds.last_clist_regions);
            uint32_t listed = 0;
            HIP_TRY(ctx, hipMemcpyAsync(cnt.data(), ds.d_clist_cnt, cnt.size() * 4, hipMemcpyDeviceToHost, ds.stream));
            HIP_TRY(ctx, hipMemcpyAsync(&listed, ds.last_blist_ctr, 4, hipMemcpyDeviceToHost, ds.stream));
            HIP_TRY(ctx, hipStreamSynchronize(ds.stream));
            double entries = 0;
            for (uint32_t c : cnt) entries += (double)c;
            *value = 4.0 * entries + 256.0 * (double)listed;
        }
        return APM_OK;
    }
    if (n == "verify_launches") { *value = (double)ctx->sieve.launches.size(); return APM_OK; }
    if (n == "verify_image_bytes") { *value = ctx->sieve.launches.empty() ? 0.0 : (double)ctx->sieve.launches[0].image.size(); return APM_OK; }
    if (n == "verify_blocks_per_cu") { // (of the form the last call ran)
        *value = ctx->sieve.launches.empty() ? 0.0 : (double)(ds.last_fused ? ctx->sieve.launches[0].fused_blocks_per_cu : ctx->sieve.launches[0].blocks_per_cu);
        return APM_OK;
    }
    if (n == "verify_threads") {
        *value = ctx->sieve.launches.empty() ? 0.0 : (double)(ds.last_fused ? ctx->sieve.launches[0].fused_threads : ctx->sieve.launches[0].threads);
        return APM_OK;
    }
    if (n == "sieve_candidates") { // hits of the last call's sieve: popcount over its masks (synchronises with the stream)
        *value = 0;
        if (!ds.d_masks || ds.last_mask_blocks <= 0) return APM_OK;
        HIP_TRY(ctx, hipSetDevice(ds.dev));
        if (!ds.d_stats) HIP_TRY(ctx, hipMalloc((void **)&ds.d_stats, APM_STATS_BYTES));
        unsigned long long *d_sum = ds.d_stats + 7;
        HIP_TRY(ctx, hipMemsetAsync(d_sum, 0, 8, ds.stream));
        if (ds.last_clist_regions)
            hipLaunchKernelGGL(apm_popcount_listed_kernel, dim3(1024), dim3(256), 0, ds.stream, ds.d_masks, ds.d_blist, ds.last_blist_ctr, ds.d_clist_cnt, ds.last_clist_regions, d_sum);
        else
            hipLaunchKernelGGL(apm_popcount_kernel, dim3(1024), dim3(256), 0, ds.stream, ds.d_masks, (unsigned long long)ds.last_mask_blocks * 64ull, d_sum);
        unsigned long long h = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&h, d_sum, 8, hipMemcpyDeviceToHost, ds.stream));
        HIP_TRY(ctx, hipStreamSynchronize(ds.stream));
        *value = (double)h;
        return APM_OK;
    }
#ifdef APM_MEASURE
    if (n.rfind("verify_", 0) == 0 && ds.d_stats) {
        HIP_TRY(ctx, hipSetDevice(ds.dev));
        HIP_TRY(ctx, hipStreamSynchronize(ds.stream));
        unsigned long long h[8] = {};
        HIP_TRY(ctx, hipMemcpy(h, ds.d_stats, 64, hipMemcpyDeviceToHost));
        if (n == "verify_survivors") { *value = (double)h[1]; return APM_OK; }
        if (n == "verify_dp_items") { *value = (double)h[2]; return APM_OK; }
        if (n == "verify_counted") { *value = (double)h[3]; return APM_OK; }
        if (n.rfind("verify_wave_", 0) == 0) { // per-wave start / end stamps (APM_MEASURE_SKIP bit 9), in us from the first start
            std::vector<unsigned long long> t(2 * APM_STATS_WAVES);
            HIP_TRY(ctx, hipMemcpy(t.data(), ds.d_stats + 8, t.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> st, en;
            unsigned long long t0 = ~0ull;
            for (size_t w = 0; w < APM_STATS_WAVES; ++w) if (t[2 * w + 1]) t0 = std::min(t0, t[2 * w]);
            for (size_t w = 0; w < APM_STATS_WAVES; ++w) if (t[2 * w + 1]) { st.push_back((double)(t[2 * w] - t0) * 0.01); en.push_back((double)(t[2 * w + 1] - t0) * 0.01); }
            if (n.rfind("verify_wave_grp", 0) == 0 || n.rfind("verify_wave_xcd", 0) == 0) { // verify_wave_grpmax<g> / grpmin<g> / xcdavg<x>: end stamps by group / by blockIdx % 8
                const bool by_xcd = n[12] == 'x';
                const int want = atoi(n.c_str() + 18);
                double lo = 1e30, hi = 0, sum = 0; long cnt = 0;
                for (size_t w = 0; w < APM_STATS_WAVES; ++w) {
                    if (!t[2 * w + 1]) continue;
                    const int key = by_xcd ? (int)((w / 4) % 8) : (int)((w ^ (w >> 5)) % APM_WORK_GROUPS);
                    if (key != want) continue;
                    const double e = (double)(t[2 * w + 1] - t0) * 0.01;
                    lo = std::min(lo, e); hi = std::max(hi, e); sum += e; ++cnt;
                }
                *value = n[15] == 'm' && n[16] == 'a' ? hi : (n[15] == 'm' && n[16] == 'i' ? lo : (cnt ? sum / cnt : 0));
                return APM_OK;
            }
            if (en.empty()) { *value = 0; return APM_OK; }
            std::sort(st.begin(), st.end());
            std::sort(en.begin(), en.end());
            if (n == "verify_wave_count") { *value = (double)en.size(); return APM_OK; }
            if (n == "verify_wave_start_max") { *value = st.back(); return APM_OK; }
            if (n == "verify_wave_end_min") { *value = en.front(); return APM_OK; }
            if (n == "verify_wave_end_p10") { *value = en[en.size() / 10]; return APM_OK; }
            if (n == "verify_wave_end_p50") { *value = en[en.size() / 2]; return APM_OK; }
            if (n == "verify_wave_end_p90") { *value = en[en.size() * 9 / 10]; return APM_OK; }
            if (n == "verify_wave_end_max") { *value = en.back(); return APM_OK; }
        }
    }
#endif
    return fail(ctx, APM_ERR_INVALID, "unknown statistic '%s'", name);
}

int apm_device_alloc(apm_ctx *ctx, void **d_ptr, uint64_t bytes) {
    if (!ctx || !d_ptr) return APM_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->devs[0].dev));
    HIP_TRY(ctx, hipMalloc(d_ptr, (size_t)std::max<uint64_t>(bytes, 16)));
    return APM_OK;
}
int apm_device_free(apm_ctx *ctx, void *d_ptr) {
    if (!ctx) return APM_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->devs[0].dev));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->devs[0].stream));
    HIP_TRY(ctx, hipFree(d_ptr));
    return APM_OK;
}
int apm_device_upload(apm_ctx *ctx, void *d_dst, const void *src, uint64_t bytes) {
    if (!ctx) return APM_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->devs[0].dev));
    HIP_TRY(ctx, hipMemcpyAsync(d_dst, src, (size_t)bytes, hipMemcpyHostToDevice, ctx->devs[0].stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->devs[0].stream));
    return APM_OK;
}
int apm_device_download(apm_ctx *ctx, void *dst, const void *d_src, uint64_t bytes) {
    if (!ctx) return APM_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->devs[0].dev));
    HIP_TRY(ctx, hipMemcpyAsync(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->devs[0].stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->devs[0].stream));
    return APM_OK;
}
int apm_device_memset(apm_ctx *ctx, void *d_dst, int value, uint64_t bytes) {
    if (!ctx) return APM_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->devs[0].dev));
    HIP_TRY(ctx, hipMemsetAsync(d_dst, value, (size_t)bytes, ctx->devs[0].stream));
    return APM_OK;
}
int apm_synchronize(apm_ctx *ctx) {
    if (!ctx) return APM_ERR_INVALID;
    for (auto &ds : ctx->devs) {
        HIP_TRY(ctx, hipSetDevice(ds.dev));
        HIP_TRY(ctx, hipStreamSynchronize(ds.stream));
    }
    return APM_OK;
}

} // extern "C"
