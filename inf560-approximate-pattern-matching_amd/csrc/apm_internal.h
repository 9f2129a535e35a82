/*
 * apm_internal.h -- structures shared by the runtime (apm_runtime.hip) and the
 * kernels (apm_kernels.hip).  Not part of the ABI.
 */
#ifndef APM_INTERNAL_H
#define APM_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define APM_BLOCK 256          /* threads per workgroup: 4 wave64 */
#define APM_TILE_SLACK 320     /* bytes readable past tile+halo in LDS (ramp-down reads, 16B rounding) */
#define APM_BITPAR_MAX_M 4096    /* <= 512: one window per lane, columns of <= 16 words (apm_bitpar.h); <= 1024: 24 / 32 words; <= 4096: one window per wave (apm_bitlong.hip) */
#define APM_WAVEFRONT_MAX_M 256
#define APM_LDS_TABLE_BUDGET (40 * 1024)
#define APM_BANDED_MAX_M 512     /* (unit offsets inside the pattern: 9 bits of the key records) */
#define APM_BANDED_MIN_PIECE 4
#define APM_BANDED_MAX_K 7
#define APM_BANDED_MAX_PATS 64
#define APM_NFA_MAX_K 7          /* apm_nfa.hip: m + k/2 <= 32, <= 16 distinct pattern bytes per launch */

/* Optional sink for match positions (apm_find_buffer): single-pattern launches only. */
struct ApmPosSink {
    unsigned long long *out;    /* device: global start offsets of matching windows (unordered), or NULL */
    unsigned long long *count;  /* device: number of matches pushed (may exceed cap) */
    unsigned long long cap;
    unsigned long long text_off;/* global position of text[0] */
};

/* One pattern as a scan kernel sees it. */
struct ApmPatDesc {
    uint32_t m;         /* length in bytes */
    uint32_t byte_off;  /* offset of its bytes (raw, or remapped codes) in the launch's byte pool */
    uint32_t aux_off;   /* BITPAR: offset (in uint32) of its Eq table in the launch's table pool */
    uint32_t w;         /* BITPAR: words per column (1..4); WAVEFRONT: rows per lane R (1,2,4) */
    uint32_t stride;    /* BITPAR: words per table entry (1,2,4) */
    uint32_t index;     /* position in the caller's pattern list (counts[] slot) */
};

/* Arguments common to the tiled scan kernels.  All positions are RELATIVE to
 * text[0] (= global position text_off of the shard). */
struct ApmScanArgs {
    const uint8_t *text;   /* device */
    int64_t avail;         /* bytes valid at text[0..avail) */
    int64_t jb, je;        /* window starts to decide: [jb, je), already ∩ [0, n_total-k) */
    int64_t nrel;          /* end of the WHOLE text, relative; window j is full iff j+m <= nrel */
    int64_t tile0;         /* base of tile 0 (<= jb, text+tile0 is 16-byte aligned) */
    const ApmPatDesc *pats;
    const uint8_t *bytes;  /* pattern bytes/codes pool of this launch */
    const uint32_t *tables;/* BITPAR Eq tables of this launch */
    const uint8_t *lut;    /* BITPAR: 256-byte text->code remap of this launch */
    unsigned long long *counts;
    int n_pats;
    int k;
    int tile;              /* window starts per workgroup */
    int halo;              /* m_max - 1 of this launch */
    int table_words;       /* BITPAR: total uint32 in tables */
    int bytes_len;         /* bytes in the pattern pool */
    ApmPosSink pos;
};

struct ApmGenericArgs {
    const uint8_t *text;
    int64_t avail;
    int64_t jb, je;        /* window starts [jb, je), already ∩ [0, n_total-k) */
    int64_t nrel;
    const ApmPatDesc *pats;/* one per blockIdx.y: m, byte_off (into bytes), index */
    const uint8_t *bytes;  /* raw pattern bytes pool */
    int k;
    int mode;              /* 0: full windows only, 1: truncated tail windows only, 2: both */
    int col_stride;        /* m_max + 1 */
    uint16_t *scratch;     /* n_pats * col_stride * (gridDim.x*256) uint16, lane-interleaved */
    unsigned long long *counts;
    ApmPosSink pos;
};

/* NFA launch (apm_nfa.hip): short loose patterns, 32 window starts per lane.  Positions relative to text[0]. */
struct ApmNfaArgs {
    const uint8_t *text;
    int64_t avail;
    int64_t jb, je, nrel;      /* window starts to decide [jb, je); full windows only (truncated ones: the tail kernels) */
    int64_t tile0;             /* first window start of workgroup 0 (<= jb, text + tile0 16-byte aligned) */
    const ApmPatDesc *pats;    /* m, byte_off (into classes: a multiple of 16), index */
    const uint8_t *classes;    /* the patterns as class numbers, one nibble per pattern byte, 16 bytes each (zero padded), cls_len bytes, 16-byte aligned */
    int cls_len;
    uint8_t class_bytes[16];   /* the byte of class c */
    int n_classes;             /* distinct pattern bytes of the launch, <= 16 */
    unsigned long long *counts;
    int n_pats, k;
    ApmPosSink pos;
};

struct ApmTailArgs {       /* truncated tail windows of patterns with m <= 128 */
    const uint8_t *text;
    int64_t jb, je, nrel;
    const ApmPatDesc *pats;
    const uint8_t *bytes;
    unsigned long long *counts;
    int k;
    ApmPosSink pos;
};

/* BANDED (filter + verify) launch.  A key is a KL-byte sub-block of one of the k+1 disjoint
 * pieces of a pattern; `off` is its offset inside the pattern (piece offset + r). */
struct ApmKey {
    uint32_t fp;        /* fingerprint of the KL key bytes (apm_fp8 / apm_fp16) */
    uint16_t pat;       /* pattern slot inside the launch */
    uint16_t off;       /* offset of the key bytes inside the pattern */
    uint16_t piece;     /* index q of the piece it belongs to */
    uint16_t next;      /* 1 + id of the next key with the same fingerprint (0 = end of chain) */
};

#define APM_FILTER_POS 4096   /* text bytes fingerprinted per workgroup tile (16 per lane) */

struct ApmFilterArgs {
    const uint8_t *text;
    int64_t avail;
    int64_t avail_pad;     /* avail rounded up so that text+avail_pad is 16-byte aligned (same allocation granule) */
    int64_t jb, je;
    int64_t nrel;
    int64_t tile0;         /* first window start of tile 0; text+tile0-front is 16-byte aligned */
    int64_t ntiles;
    const ApmPatDesc *pats;/* m, byte_off (into bytes), index, aux_off = first entry in piece_off, w = pieces (k+1) */
    const uint4 *image;    /* launch image, copied verbatim to LDS: pattern bytes at 0, then (all 16-byte aligned)
                              o_tab: nb buckets x 8 16-bit tags (empty 0xffff); o_kid: nb x 8 16-bit key ids
                              (empty 0xffff, bit 15 = head of a chain); o_ovf: n_ovf x {tag, kid16};
                              o_kinfo: nk x (pat | off<<12 | piece<<21); o_pinfo: n_pats x {byte_off | m<<16, aux_off} */
    int image_len, o_tab, o_kid, o_ovf, o_kinfo, o_pinfo, o_next, o_poff; /* o_next: nk x u16 chain links,
                              o_poff: piece offsets a_q (u16), per pattern contiguous */
    int o_pat;             /* pattern bytes inside the image (0 unless a bitmap leads the image) */
    int o_kext;            /* per-position classes, nk x u32: key byte offset in the pattern bytes | piece length << 16
                              | partner length << 24 (31 = longer than 16) | partner side << 29 (0 none, 1 behind the
                              piece, 2 in front of it): all the pair pre-check needs, in one LDS read */
    int o_bmp, code_shift; /* per-position classes: key presence bitmap over the 2-bit byte codes
                              (b >> code_shift) & 3 of the key_len key bytes (2^(2*key_len) bits); bit of code
                              word x lives in byte x & (NB-1), bit x >> log2(NB), NB = 2^(2*key_len-3) */
    unsigned long long *counts;
    int n_pats, nk;
    int nb, lg_nb, n_ovf;  /* hash table geometry */
    int qcap;              /* candidate queue entries */
    int key_len, stride;   /* (16,16), (8,8), (8,1), (6,1) or (4,1) */
    int k, band;           /* band = k/2 */
    int tile_w;            /* window starts per workgroup tile (multiple of 32) */
    int front;             /* bytes staged in front of the first window (0 or 16) */
    int tile_len;          /* bytes staged per tile: APM_FILTER_POS (one 16-byte load per lane) */
#ifdef APM_MEASURE
    int skip_mask;         /* tools/ build only (libapm_hip_measure.so): stages to leave out, results invalid */
#endif
    int use_dma;           /* 1: LDS-DMA tile path (text pointer 16-byte aligned), 0: register-staged path */
    int n_main_blocks;     /* set by the launcher: persistent scan workgroups */
    int n_tail;            /* extra workgroups, one per tail pattern (0: tails launched separately) */
    ApmTailArgs tail;
    int n_cu;              /* compute units of the device (set by the runtime; spreads the verification over the SIMDs) */
};

#define APM_TAG_EMPTY 0x5bd1e995u

/* Stage-skipping switches of the kernels exist only in the measurement build (make measure ->
 * libapm_hip_measure.so, used by tools/): in the product build the test folds to a constant 0 and the
 * branches vanish; no environment variable can change what the shipped library computes. */
#ifdef APM_MEASURE
#define APM_SKIP(a, bits) ((a).skip_mask & (bits))
#else
#define APM_SKIP(a, bits) 0
#endif

/* launchers (apm_kernels.hip) */
hipError_t apm_launch_filter(const ApmFilterArgs &a, int max_blocks, hipStream_t s);
hipError_t apm_launch_tail(const ApmTailArgs &a, int n_pats, hipStream_t s);
hipError_t apm_launch_tail_wide(const ApmTailArgs &a, int n_pats, hipStream_t s); /* 128 < m <= 512 */
size_t apm_filter_lds_bytes(const ApmFilterArgs &a);
int apm_filter_blocks_per_cu(int band, int key_len, int stride, int dma, size_t lds);
hipError_t apm_launch_stream(const ApmFilterArgs &a, int max_blocks, hipStream_t s);
int apm_stream_blocks_per_cu(const ApmFilterArgs &a);
hipError_t apm_launch_bitpar(const ApmScanArgs &a, hipStream_t s);
hipError_t apm_launch_wavefront(const ApmScanArgs &a, hipStream_t s);
hipError_t apm_launch_generic(const ApmGenericArgs &a, int nbx, int n_pats, hipStream_t s);
hipError_t apm_launch_synth(uint8_t *dst, uint64_t global_off, uint64_t len, uint64_t seed, hipStream_t s);
size_t apm_bitpar_lds_bytes(const ApmScanArgs &a);
/* apm_nfa.hip */
hipError_t apm_launch_nfa(const ApmNfaArgs &a, hipStream_t s);
/* apm_bitlong.hip */
hipError_t apm_launch_bitpar_xwide(const ApmScanArgs &a, unsigned n_tiles, size_t lds_bytes, hipStream_t s); /* 512 < m <= 1024 */
hipError_t apm_launch_bitlong(const ApmScanArgs &a, int m, hipStream_t s);                                    /* 1024 < m <= 4096, one pattern */
hipError_t apm_launch_tail_xwide(const ApmTailArgs &a, int n_pats, hipStream_t s);                            /* tails, 512 < m <= 1024 */
size_t apm_wavefront_lds_bytes(const ApmScanArgs &a);

#endif
