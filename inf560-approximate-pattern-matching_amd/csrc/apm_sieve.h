/*
 * apm_sieve.h -- launch arguments of the sieve + verify pipeline (apm_sieve.hip), shared with the runtime only.
 */
#ifndef APM_SIEVE_H
#define APM_SIEVE_H

#include "apm_internal.h"

/* ---- sieve + verify pipeline of the per-position classes (apm_sieve.hip) ----------------------------------
 * SIEVE: ONE pass over the text for all per-position keys of a pattern set.  Every EVEN text position is tested
 * with one LDS lookup: the 2-bit codes (b >> code_shift) & 3 of the 9 bytes at the position form an 18-bit code
 * word x; its bit sits in dword x & 8191, bit x >> 13 of a 32 KiB presence bitmap.  The bitmap holds every code
 * word under which a key can start at the position itself or at the odd position behind it (host: built from the
 * 16-bit code words of the keys' 8-byte windows), so one lookup decides two text positions.  Hits are appended
 * to a global candidate list as (relative position / 2), 32 bits each.
 * VERIFY: list-driven, one candidate per lane: identifies the key(s) through a rank structure over the exact
 * 16-bit presence bitmap (no hashing), runs the piece compare + pair pre-check against global text, collects the
 * survivors per wave and runs the banded DP with stateless dedup on dense lanes. */
struct ApmSieve2Args {
    const uint8_t *text;        /* 16-byte aligned */
    int64_t avail_pad;          /* bytes readable from text (multiple of 16), < 2^32 */
    int64_t tile0;              /* first scanned relative position (multiple of 16) */
    int64_t nchunks;            /* 1 KiB chunks */
    const uint4 *bitmap;        /* 32 KiB (stride 8: the first 8 KiB, a bitmap over the 16-bit code words of 8-byte blocks:
                                   dword x & 2047, bit x >> 11) */
    int code_shift;
    int stride;                 /* 1: every position (two per lookup, see above); 8: sampled -- the keys' pieces are >= 15 bytes
                                   long, so each contains an 8-byte block at a multiple of 8: one lookup per 8 text bytes */
    /* hit masks, the hand-over to the verify launches: one dword per lane and 4 KiB block of text (block b = the four
       chunks 4b .. 4b+3 from tile0 on; dword masks[64 b + lane]): bit 8 j + t = the lane's lookup t in chunk j hit, i.e.
       relative position tile0 + 4096 b + 1024 j + 16 lane + 2 t (stride 8: + 8 t, t < 2).  Every dword of every block is
       written by exactly one wave with one coalesced store: no queue, no atomics, no capacity, nothing to overflow;
       n / 16 bytes. */
    uint32_t *masks;
    /* list of the blocks with at least one hit (stride 1; NULL: none kept), for the verify launches to walk instead of
       every mask row: with the code filter most rows are empty (cfg5: 10 K hits in 262 K rows per GiB) and the walk, a chain
       of dependent loads per wave, was all the verify launch did (0.08 ms).  A wave collects its non-empty block numbers
       in a register and appends them 64 at a time (one atomic on *blist_ctr per flush); at most one entry per block: the
       list cannot overflow.  Order is arbitrary.  blist_ctr points at one of two counters the sieve launches alternate
       between (launch i counts in set i & 1 and zeroes the other for launch i + 1 on the same stream). */
    uint32_t *blist;
    uint32_t *blist_ctr, *blist_ctr_next;
    /* CANDIDATE LIST (code-filter form; NULL: masks and block list only): what survives the filter is a few positions per
       block (cfg3: 6.5 of 2048, cfg5: 0.04), so the wave appends them as 32-bit entries (relative position / 2, what the
       verify launch forms out of a mask bit) to its WORKGROUP's region of the list -- region g = entries
       [g * clist_cap, g * clist_cap + clist_cnt[g]), filled through a counter in LDS, no global atomics -- instead of
       storing a 256-byte mask row per block: 1/10 of the write traffic on cfg3, and the verify launch neither reads the
       rows nor picks the bits out of them.  A block whose entries no longer fit its region keeps the old hand-over for
       what is left of it (mask row + block list): nothing is lost, whatever the density.  The workgroups of a launch walk
       the text interleaved, so the regions fill evenly wherever the candidates cluster. */
    uint32_t *clist;
    uint32_t *clist_cnt;        /* one per scanning workgroup, written when the workgroup ends */
    uint32_t clist_cap;         /* entries per region */
    int n_main_blocks;          /* set by the launcher: scanning workgroups */
    int n_tail;                 /* extra workgroups, one per pattern with truncated tail windows (they run beside the scan) */
    ApmTailArgs tail;
    /* CODE FILTER (stride 1, one verify launch): the sieve's second stage.  A lookup hit says "some key's 8-byte window
       may start here or at the odd position behind"; before its bit goes into the masks the wave identifies the key(s) by
       rank and tests, on the 2-bit codes it holds for the block (a wave-private LDS copy, 16 bytes of halo in front and
       32 behind), what the verify launch would test first on the bytes: bytes 8..15 of the exact part, and the partner
       within one edit (apm_cf_pass in apm_core.h).  A superset of the nomination predicate survives -- on DNA exactly
       the predicate: cfg5 21.6 M hits per GiB -> 10.6 K, and the verify launch no longer re-reads the text to reject
       them.  cf_image = NULL: no second stage.
       cf_image (16-byte aligned parts): tbl: uint2[2048] = {dword x & 2047 of the bitmap over the 16-bit code words of
       8-byte windows (bit x >> 11), set bits in the dwords before it} | rrec at cf_o_rrec: uint2 per set bit in rank
       order = the code-filter record of the word's key (apm_cf_record), or, for a word that several keys share,
       {3 << 30 | index into lrec, 0} | lrec at cf_o_lrec: their records one after the other, bit 31 of .y = last */
    const uint4 *cf_image;
    int cf_len, cf_o_rrec, cf_o_lrec;
    int cf_threads, cf_blocks_per_cu; /* launch geometry of the code-filter form (apm_sieve2cf_geometry) */
#ifdef APM_MEASURE
    int skip_mask;
#endif
};
/* text bytes a sieve launch addresses: 32-bit offsets, and a wave loads up to 4 x (waves of the launch) chunks of 1 KiB
   ahead of the scanned range (<= 4 x 8192 KiB) -- those offsets must not wrap */
#define APM_SIEVE_MAX_BYTES (((int64_t)1 << 32) - ((int64_t)64 << 20))
#define APM_CF_WAVE_BYTES 1552  /* per wave: code strip 260 dwords | survivor masks 64 dwords | hit ring 128 x u16 */
int apm_sieve2cf_geometry(int cf_len, int *threads); /* workgroups per CU; *threads = workgroup size (0: does not fit) */
int apm_sieve2cf_blocks(const ApmSieve2Args &a, int n_cu); /* scanning workgroups the code-filter form will launch = regions of the candidate list */

struct ApmVerifyArgs {
    const uint8_t *text;        /* 16-byte aligned */
    int64_t avail;              /* valid text bytes */
    int64_t avail_pad;          /* readable bytes (multiple of 16), < 2^32 */
    int64_t jb, je, nrel;       /* window starts to decide [jb, je); end of the whole text (relative) */
    /* LDS image (16-byte aligned parts): bitmap over the 16-bit code words of 8-byte windows (dword x & 2047,
       bit x >> 11) | prefix: u16[2048] set bits in the dwords before | r2s: u16 per set bit (rank order):
       0x8000 | kid for a single key, else first index into slots | slots: u16 kid, bit 15 = last of its list |
       kext: u32 per key (see ApmFilterArgs::o_kext) | raw pattern bytes */
    const uint4 *image;
    int image_len, o_prefix, o_r2s, o_slots, o_kext, o_pat;
    int o_masks;                /* 17 x 16 bytes: entry n = n leading 0xff bytes (byte masks of a compare of n <= 16 bytes) */
    int o_rc;                   /* 0: none; else uint2 {codes, mask} per ((unit * 8 + r) * 2 + half): the fused sampled form's register compare, precomputed (<= 128 units) */
    int o_kinfo, o_pinfo;       /* the records of the banded DP and the dedup, in the image too (round 2 kept them in global memory:
                                   with the sieve's code filter in front the DP is most of the launch, and kinfo -> pinfo -> text
                                   was a chain of three memory round trips per DP batch) */
    const uint32_t *kinfo;      /* (global copy, unused by the kernels) per key = nomination unit: pat | off << 12 | unit index inside
                                   the pattern << 21; off = offset of the unit's text position inside the window (window start =
                                   position - off - shift) */
    const uint2 *pinfo;         /* (global copy) per pattern: {byte_off | m << 16, id of its first key}; a pattern's units are consecutive keys */
    const uint32_t *kpart;      /* global, per key: partner offset inside the pattern | partner length << 16 (partners beyond 16 bytes only) */
    const ApmPatDesc *pats;     /* index = counts[] slot */
    unsigned long long *counts;
    int n_pats, nk, k, band, code_shift;
    int stride;                 /* as the sieve's: 1 = hits are even positions and both parities are tried; 8 = hits are multiples
                                   of 8, the key list entries carry the block's offset r inside its piece (bits 11..13 of
                                   the 15-bit payload, key id in bits 0..10) and the piece is tested at position - r */
    const uint32_t *masks;      /* see ApmSieve2Args */
    const uint32_t *blist;      /* see ApmSieve2Args; NULL: every block 0 .. n_mask_blocks - 1 */
    const uint32_t *blist_ctr;  /* entries of blist */
    const uint32_t *clist;      /* see ApmSieve2Args (NULL: none): the candidates come from the list's regions, then from the rows of the listed blocks */
    const uint32_t *clist_cnt;
    uint32_t clist_cap;
    int clist_regions;
    int clist_min_batch;        /* 1..64: a short region is cut into batches of at least this many entries (see apm_verify_body) */
    int64_t tile0;              /* relative position of block 0 */
    int64_t n_mask_blocks;      /* 4 KiB blocks the sieve wrote masks for */
    int n_blocks;               /* set by the launcher */
    /* dynamic block distribution (apm_verify_body): two sets of APM_WORK_GROUPS counters, APM_WORK_STRIDE dwords apart, all
       zero before the first launch; launch e uses set e & 1 and zeroes the other for launch e + 1 (same stream) */
    uint32_t *work;
    int work_epoch;
    int work_groups;            /* set by the launcher: min(APM_WORK_GROUPS, waves of the launch) */
#ifdef APM_MEASURE
    int skip_mask;
    unsigned long long *stats;  /* [0] pre-check evaluations, [1] survivors, [2] DP items run, [3] windows counted */
#endif
};

/* FUSED form of the pipeline: ONE kernel, the text leaves HBM once.  Every wave sieves its own 4 KiB blocks and
 * feeds the hits straight into the verify batches (the same code as the verify launch: apm_verify_body); the windows
 * are gathered from global memory, where the lines the wave has just streamed are still in the caches.  No masks, no
 * second launch.  `s` carries the sieve side (text, tile0, nchunks, bitmap, tail workgroups), `v` the verify image and
 * records (its masks fields are unused). */
struct ApmFusedArgs {
    ApmSieve2Args s;
    ApmVerifyArgs v;
};
#define APM_FUSED_MAX_THREADS 896 /* 14 waves: two workgroups put 7 waves on every SIMD */
hipError_t apm_launch_fused(const ApmFusedArgs &a, int threads, int max_blocks, int *work_epoch, hipStream_t s);
size_t apm_fused_lds_bytes(const ApmFusedArgs &a, int threads);
int apm_fused_geometry(const ApmFusedArgs &a, int *threads); /* workgroups per CU; *threads = workgroup size (0: does not fit) */

#define APM_WORK_GROUPS 32
#define APM_WORK_STRIDE 64
#define APM_WORK_BYTES (2 * APM_WORK_GROUPS * APM_WORK_STRIDE * 4 + 2 * APM_WORK_STRIDE * 4) /* ... + the two block-list counters of the sieve */
#define APM_BLIST_CTR(work, set) ((work) + 2 * APM_WORK_GROUPS * APM_WORK_STRIDE + (set) * APM_WORK_STRIDE)
#define APM_STATS_WAVES 16384                         /* measurement build: per-wave time stamps behind the 8 counters */
#define APM_STATS_BYTES (64 + 16 * APM_STATS_WAVES)
hipError_t apm_launch_sieve2(const ApmSieve2Args &a, int n_cu, hipStream_t s); /* a.blist set: the caller alternates blist_ctr / blist_ctr_next and advances its epoch on success */
hipError_t apm_launch_verify(const ApmVerifyArgs &a, int threads, int max_blocks, int *work_epoch, hipStream_t s);
int apm_verify_geometry(const ApmVerifyArgs &a, int *threads); /* workgroups per CU; *threads = 256 or 512 */

#endif /* APM_SIEVE_H */
