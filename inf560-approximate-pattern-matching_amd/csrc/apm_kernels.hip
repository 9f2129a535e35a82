/*
 * apm_kernels.hip -- CDNA4 (gfx950) kernels for the Levenshtein sliding-window
 * DP + per-pattern match count.
 *
 * Path replaced (reference file:line):
 *   levenshtein()            /root/reference/src/utils.c:76-99
 *   per-pattern scan loop    /root/reference/src/sequential.c:105-144
 *   (superseded GPU forms:   src/patterns_over_ranks.cu:19-73 ComputeMatches,
 *                            src/database_over_ranks.cu:20-134 searchPattern)
 *
 * Common shape of the tiled scan kernels: one 256-thread workgroup owns a tile
 * of `tile` consecutive window starts.  It stages tile+halo bytes of text from
 * HBM into LDS with 16-byte coalesced loads ONCE, then runs every pattern of
 * the launch over the LDS copy (1 HBM byte per text position per launch,
 * independent of the number of patterns), reduces matches wave -> workgroup in
 * LDS and issues one 64-bit global atomic per (workgroup, pattern) that has
 * matches.
 */
#include "apm_internal.h"
#include "apm_core.h"
#include <type_traits>
#include "apm_device.h"

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------


// lane l receives lane (l-1)'s value; lane 0 keeps `self` (DPP wave_shr:1, 1 VALU op)
__device__ __forceinline__ int apm_shift_up1(int v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}

// the same with lane 0 receiving zero: one v_mov_b32_dpp without the copy that keeps `self` (lane 0 is a row-0 lane of
// the wavefront kernel: it never uses what it receives)
__device__ __forceinline__ int apm_shift_up1z(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
}

#include "apm_bitpar.h"

size_t apm_bitpar_lds_bytes(const ApmScanArgs &a) {
    const size_t tile_bytes = (size_t)((a.tile + a.halo + APM_TILE_SLACK + 15) & ~15);
    return tile_bytes + 256 + (size_t)((a.table_words + 3) & ~3) * 4 + (size_t)a.n_pats * 4 + 16;
}

hipError_t apm_launch_bitpar(const ApmScanArgs &a, hipStream_t s) {
    const int64_t span = a.je - a.tile0;
    if (span <= 0 || a.n_pats <= 0) return hipSuccess;
    const int64_t nt = (span + a.tile - 1) / a.tile;
    if (nt > 0x7fffffffLL) return hipErrorInvalidValue;
    if (a.halo >= 512) return apm_launch_bitpar_xwide(a, (unsigned)nt, apm_bitpar_lds_bytes(a), s); // 513 .. 1024 bytes (apm_bitlong.hip)
    if (a.halo >= 128) return apm_launch_bitpar_wide(a, (unsigned)nt, apm_bitpar_lds_bytes(a), s); // (the runtime never mixes; apm_bitpar_wide.hip)
    hipLaunchKernelGGL(apm_bitpar_kernel<false>, dim3((unsigned)nt), dim3(APM_BLOCK), apm_bitpar_lds_bytes(a), s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// WAVEFRONT: the anti-diagonal kernel BASELINE.json's north_star names.
//
// A pattern of m rows is laid over Lm = ceil(m/R) lanes (R rows per lane);
// S = 64/Lm windows ("streams") sit side by side in one wave64 and advance in
// lockstep.  At step s the lane holding rows r0+1..r0+R works on text column
// x = s - y0 + 1 (y0 = its index inside the stream): the anti-diagonal.  The
// value cell(x, r0) it needs from the lane above was produced one step earlier
// and arrives through one DPP wave_shr:1 move (the "__shfl_up column passing");
// cell(x-1, r0) is last step's arrival.  All values are kept "+1" so that a
// cell costs v_cmp_eq, v_subb, v_min3, v_add.  Ramp-up (s < y0) is exec-masked;
// ramp-down garbage never flows back up.  The last lane of a stream finishes
// cell(m,m) exactly at the last step, m + Lm - 2.
// ---------------------------------------------------------------------------
// packed 16-bit lanes: every register holds the same DP cell of TWO adjacent windows (low / high
// half), so one v_pk_add_u16 / v_pk_min_u16 advances two cells (distances <= 257 fit 16 bits)
typedef unsigned short wf_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t wf_pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(wf_u16x2, a) + __builtin_bit_cast(wf_u16x2, b));
}
__device__ __forceinline__ uint32_t wf_pk_min(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(wf_u16x2, a), __builtin_bit_cast(wf_u16x2, b)));
}

template <int R>
__device__ __forceinline__ uint32_t wf_scan(const uint8_t *s_tile, const uint8_t *s_pat, int m, int k,
                                            int64_t base, int64_t jb, int64_t je_p, int tile, int wave, int lane) {
    const int Lm = (m + R - 1) / R;
    const int S = 64 / Lm;
    const int sig = lane / Lm;
    const int y0 = lane - sig * Lm;
    const bool live = sig < S;
    const int r0 = y0 * R;
    constexpr uint32_t ONE2 = 0x00010001u;
    uint32_t pch2[R]; // pattern byte of row r0+i in both halves (rows past m: never equal to a text byte)
#pragma unroll
    for (int i = 0; i < R; ++i) pch2[i] = ((r0 + i < m) ? (uint32_t)s_pat[r0 + i] : (uint32_t)(0x100 + i)) * ONE2;
    const bool row0 = (y0 == 0);
    const bool is_res = live && (y0 == Lm - 1);
    const int ires = (m - 1) - (Lm - 1) * R;
    const int nsteps = m + Lm - 1;
    const uint32_t xbase2 = (uint32_t)(2 - y0) * ONE2; // + s*ONE2 = cell(x, 0) + 1 = x + 1 in both halves
    uint32_t one2 = ONE2;
    asm volatile("" : "+v"(one2)); // opaque: keeps min(x, 1) a single v_pk_min_u16 (LLVM would expand it to compares)
    uint32_t cnt = 0;

    for (int g0 = wave * 2 * S; g0 < tile; g0 += (APM_BLOCK / 64) * 2 * S) {
        const int joff = live ? g0 + 2 * sig : g0; // window A = joff (low half), window B = joff + 1 (high half)
        uint32_t cp[R];                            // cell(x-1, r) + 1, packed
#pragma unroll
        for (int i = 0; i < R; ++i) cp[i] = (uint32_t)(r0 + i + 2) * ONE2; // cell(0, r0+i+1) + 1
        uint32_t upprev = (uint32_t)(r0 + 1) * ONE2;                        // cell(0, r0) + 1
        const int cidx = joff - y0;
        // text: the lane keeps the two aligned dwords its column lies in and takes the bytes of windows A and B out of
        // them with ONE v_perm_b32 per step (selector bytes: column & 3 | 0x0c = zero); one aligned ds_read_b32 every
        // four steps, four steps ahead of its use -- no byte reads, no address arithmetic and no LDS wait in the step.
        // (columns in front of the tile occur on masked ramp-up steps only: their dwords are read from offset 0)
        const int cal = cidx & ~3;
        const uint32_t a0 = (uint32_t)(cidx - cal);
        const uint32_t sel0 = 0x0c000c00u | a0 | ((a0 + 1u) << 16); // step i of a block: + i * 0x00010001 (a0 + i + 1 <= 7)
        auto ld = [&](int addr) __attribute__((always_inline)) { return *reinterpret_cast<const uint32_t *>(s_tile + (addr < 0 ? 0 : addr)); };
        uint32_t lo, hi = ld(cal), nx = ld(cal + 4); // (nx: the dword of the NEXT block, on its way while this one is worked on)
        int nxt = cal + 8;

        auto step = [&](int s, uint32_t recv, uint32_t tch2) __attribute__((always_inline)) {
            uint32_t up = row0 ? xbase2 + (uint32_t)s * ONE2 : recv; // cell(x, r0) + 1
            uint32_t dg = upprev;                                    // cell(x-1, r0) + 1
            upprev = up;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const uint32_t left = cp[i];
                const uint32_t neq = wf_pk_min(tch2 ^ pch2[i], one2);            // 0/1 per half
                const uint32_t v = wf_pk_min(wf_pk_add(wf_pk_min(left, up), ONE2), // min(left, up) + 1
                                             wf_pk_add(dg, neq));                 // diag + neq
                dg = left;
                up = v;
                cp[i] = v;
            }
        };
        // MODE 0: every step of the block is a ramp-up step (lanes join one per step); 2: none is; 1: decided per step,
        // and the block may end early (the blocks around s = Lm - 1 and s = nsteps)
        auto block = [&](int s0, auto mode) __attribute__((always_inline)) {
            constexpr int MODE = decltype(mode)::value;
            lo = hi;
            hi = nx;
            nx = ld(nxt);
            nxt += 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int s = s0 + i;
                if (MODE == 1 && s >= nsteps) break;
                const uint32_t tch2 = __builtin_amdgcn_perm(hi, lo, sel0 + (uint32_t)i * 0x00010001u);
                const uint32_t recv = (uint32_t)apm_shift_up1z((int)cp[R - 1]);
                if (MODE == 0 || (MODE == 1 && s < Lm - 1)) {
                    if (s >= y0) step(s, recv, tch2);
                } else {
                    step(s, recv, tch2);
                }
            }
        };
        int s = 0;
        for (; s + 4 <= Lm - 1; s += 4) block(s, std::integral_constant<int, 0>());
        if (s < Lm - 1) { block(s, std::integral_constant<int, 1>()); s += 4; }
        for (; s + 4 <= nsteps; s += 4) block(s, std::integral_constant<int, 2>());
        if (s < nsteps) block(s, std::integral_constant<int, 1>());

        uint32_t res = cp[0];
#pragma unroll
        for (int i = 1; i < R; ++i)
            if (i == ires) res = cp[i];
        const int64_t jA = base + joff, jB = jA + 1;
        const uint32_t thr = (uint32_t)(k + 1);
        cnt += apm_wave_count(is_res && joff < tile && jA >= jb && jA < je_p && (res & 0xffffu) <= thr);
        cnt += apm_wave_count(is_res && joff + 1 < tile && jB >= jb && jB < je_p && (res >> 16) <= thr);
    }
    return cnt;
}

__global__ __launch_bounds__(APM_BLOCK) void apm_wavefront_kernel(ApmScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile_bytes = (a.tile + a.halo + APM_TILE_SLACK + 15) & ~15;
    uint8_t *s_tile = smem;
    uint8_t *s_pat = smem + tile_bytes;
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_pat + ((a.bytes_len + 15) & ~15));
    const int64_t base = a.tile0 + (int64_t)blockIdx.x * a.tile;

    for (int i = tid; i < a.bytes_len; i += APM_BLOCK) s_pat[i] = a.bytes[i];
    for (int i = tid; i < a.n_pats; i += APM_BLOCK) s_cnt[i] = 0u;
    const int nload = (a.tile + a.halo + 31) & ~15;
    for (int i = tid * 16; i < nload; i += APM_BLOCK * 16)
        *reinterpret_cast<uint4 *>(s_tile + i) = apm_load16_guarded(a.text, base + i, a.avail);
    __syncthreads();

    for (int p = 0; p < a.n_pats; ++p) {
        const ApmPatDesc d = a.pats[p];
        const int m = (int)d.m;
        const int64_t je_p = min(a.je, a.nrel - m + 1);
        const uint8_t *pat = s_pat + d.byte_off;
        uint32_t cnt;
        switch (d.w) {
        case 1: cnt = wf_scan<1>(s_tile, pat, m, a.k, base, a.jb, je_p, a.tile, wave, lane); break;
        case 2: cnt = wf_scan<2>(s_tile, pat, m, a.k, base, a.jb, je_p, a.tile, wave, lane); break;
        default: cnt = wf_scan<4>(s_tile, pat, m, a.k, base, a.jb, je_p, a.tile, wave, lane); break;
        }
        if (lane == 0 && cnt) atomicAdd(&s_cnt[p], cnt);
    }
    __syncthreads();
    for (int i = tid; i < a.n_pats; i += APM_BLOCK) {
        const uint32_t c = s_cnt[i];
        if (c) atomicAdd(&a.counts[a.pats[i].index], (unsigned long long)c);
    }
}

size_t apm_wavefront_lds_bytes(const ApmScanArgs &a) {
    const size_t tile_bytes = (size_t)((a.tile + a.halo + APM_TILE_SLACK + 15) & ~15);
    return tile_bytes + (size_t)((a.bytes_len + 15) & ~15) + (size_t)a.n_pats * 4 + 16;
}

hipError_t apm_launch_wavefront(const ApmScanArgs &a, hipStream_t s) {
    const int64_t span = a.je - a.tile0;
    if (span <= 0 || a.n_pats <= 0) return hipSuccess;
    const int64_t nt = (span + a.tile - 1) / a.tile;
    if (nt > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(apm_wavefront_kernel, dim3((unsigned)nt), dim3(APM_BLOCK), apm_wavefront_lds_bytes(a), s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// GENERIC: literal one-column DP per lane (the loop of utils.c:84-97 with the
// column in a per-thread slice of global scratch, interleaved so that lanes
// touch consecutive addresses).  Any m <= 65535; applies the end-of-text
// truncation of sequential.c:131-134.  Used for the <= m-1 truncated tail
// windows of every pattern and for patterns too long for the tiled kernels.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(APM_BLOCK) void apm_generic_kernel(ApmGenericArgs a) {
    const ApmPatDesc d = a.pats[blockIdx.y];
    const int m = (int)d.m;
    const uint8_t *pat = a.bytes + d.byte_off;
    const int64_t nthreads = (int64_t)gridDim.x * APM_BLOCK;
    const int64_t gtid = (int64_t)blockIdx.x * APM_BLOCK + threadIdx.x;
    uint16_t *col = a.scratch + (int64_t)blockIdx.y * a.col_stride * nthreads + gtid;
    const int64_t first_trunc = a.nrel - m + 1; // first window start that runs past the end of the text
    int64_t jb = a.jb, je = a.je;
    if (a.mode == 0) je = min(je, first_trunc);
    if (a.mode == 1) jb = max(jb, first_trunc);
    uint32_t cnt = 0;
    const int64_t span = je - jb;
    const int64_t rounds = span > 0 ? (span + nthreads - 1) / nthreads : 0;
    for (int64_t r = 0; r < rounds; ++r) {
        const int64_t j = jb + r * nthreads + gtid;
        bool hit = false;
        if (j < je) {
            const int64_t rem = a.nrel - j;
            const int size = rem < (int64_t)m ? (int)rem : m; // sequential.c:131-134
            for (int y = 0; y <= size; ++y) col[(int64_t)y * nthreads] = (uint16_t)y;
            for (int x = 1; x <= size; ++x) {
                const uint8_t tc = a.text[j + x - 1];
                int diag = x - 1;
                int up = x;
                col[0] = (uint16_t)x;
                for (int y = 1; y <= size; ++y) {
                    const int left = col[(int64_t)y * nthreads];
                    const int v = apm_min3(left + 1, up + 1, diag + (pat[y - 1] != tc ? 1 : 0));
                    col[(int64_t)y * nthreads] = (uint16_t)v;
                    diag = left;
                    up = v;
                }
            }
            hit = (int)col[(int64_t)size * nthreads] <= a.k;
            if (a.pos.out && hit) apm_push_pos(a.pos, j);
        }
        cnt += apm_wave_count(hit);
    }
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&a.counts[d.index], (unsigned long long)cnt);
}

hipError_t apm_launch_generic(const ApmGenericArgs &a, int nbx, int n_pats, hipStream_t s) {
    if (a.je <= a.jb || n_pats <= 0) return hipSuccess;
    hipLaunchKernelGGL(apm_generic_kernel, dim3((unsigned)nbx, (unsigned)n_pats), dim3(APM_BLOCK), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// synthetic text fill: 16 bytes per lane, 16-byte stores
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(APM_BLOCK) void apm_synth_kernel(uint8_t *dst, uint64_t global_off, uint64_t len,
                                                             uint64_t seed) {
    const uint64_t nthreads = (uint64_t)gridDim.x * APM_BLOCK;
    for (uint64_t t = (uint64_t)blockIdx.x * APM_BLOCK + threadIdx.x; t * 16 < len; t += nthreads) {
        const uint64_t o = t * 16;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int b = 0; b < 16; ++b) w[b >> 2] |= (uint32_t)apm_synth_byte(global_off + o + b, seed) << (8 * (b & 3));
        if (o + 16 <= len && ((reinterpret_cast<uintptr_t>(dst + o) & 15u) == 0)) {
            *reinterpret_cast<uint4 *>(dst + o) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (int b = 0; o + b < len; ++b) dst[o + b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
        }
    }
}

hipError_t apm_launch_synth(uint8_t *dst, uint64_t global_off, uint64_t len, uint64_t seed, hipStream_t s) {
    if (len == 0) return hipSuccess;
    uint64_t nb = (len / 16 + APM_BLOCK - 1) / APM_BLOCK + 1;
    if (nb > 256 * 32) nb = 256 * 32;
    hipLaunchKernelGGL(apm_synth_kernel, dim3((unsigned)nb), dim3(APM_BLOCK), 0, s, dst, global_off, len, seed);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// TAIL: the <= m-1 truncated windows at the very end of the text
// (sequential.c:131-134: window AND pattern cut to size = n - j), m <= 128.
// One 128-lane workgroup per pattern; each lane runs the 4-word bit-vector
// column for `size` steps over an Eq table built in LDS.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(128) void apm_tail_kernel(ApmTailArgs a) {
    __shared__ uint4 s_eq[256 + 8];
    apm_tail_body(a, (int)blockIdx.x, s_eq, (int)threadIdx.x);
}

hipError_t apm_launch_tail(const ApmTailArgs &a, int n_pats, hipStream_t s) {
    if (n_pats <= 0 || a.je <= a.jb) return hipSuccess;
    hipLaunchKernelGGL(apm_tail_kernel, dim3((unsigned)n_pats), dim3(128), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// BANDED: exact shortcut for the predicate dist <= k.
//
// (1) Equal-length global alignment with <= k edits has #ins == #del <= k/2, so
//     only the diagonals |x-y| <= band = k/2 can carry it (k<=1: Hamming).
// (2) Pigeonhole: cut the pattern into k+1 disjoint pieces; <= k edits leave one
//     piece intact, and it sits in the window at its own offset shifted by
//     delta in [-band, band].
// (3) Sampling: a piece of length L >= 2*KL-1 contains a KL-byte block that is
//     KL-ALIGNED in the text whatever the window's alignment; so it is enough to
//     look at every KL-th text position (STRIDE = KL) and to know, per piece,
//     its KL shifted sub-keys pattern[a_q + r : a_q + r + KL), r = 0..KL-1.
//     Shorter pieces (L >= KL) use STRIDE = 1 (every position, r = 0).
// A window is a CANDIDATE only if some sub-key occurs at a sampled text position
// consistent with it:  j = position - (a_q + r) - delta.
//
// Workgroup = persistent, software-pipelined walker over tiles of 4096 text
// bytes (16 per lane): the 16-byte buffer loads of tiles t+G and t+2G are in
// flight while tile t is processed out of LDS.  Per tile:
//   filter   each lane fingerprints its 16/STRIDE sampled positions and looks the
//            fingerprint up in an LDS hash table of all sub-keys of the launch
//            (4-way buckets of 32-bit tags + a short overflow list): ~10 VALU
//            ops and one ds_read_b128 per lookup, independent of the key count;
//   enqueue  a tag match pushes (key, position) into an LDS queue;
//   verify   all lanes pop the queue: byte-compare the key (fingerprints can
//            collide), banded DP with early exit over the candidate window,
//            and count it only from its FIRST true (piece, shift) nominator so a
//            window nominated several times is counted once (stateless dedup).
// A queue overflow (adversarial low-entropy text) falls back to a dense pass of
// the same verification over every (sampled position, key) of the tile.
// Per launch the text is read from HBM exactly once (+ halo per tile).
// ---------------------------------------------------------------------------

__device__ __forceinline__ uint32_t apm_fp8(uint32_t lo, uint32_t hi) {
    return lo + (hi << 3); // v_lshl_add_u32; injective on ACGT 8-mers (no carries between bytes)
}
__device__ __forceinline__ uint32_t apm_fp16(uint32_t f_lo, uint32_t f_hi) {
    return f_lo + __umul24(f_hi, 0x9E3779u); // v_mad_u32_u24: 16 text bytes -> 32 bits
}
__device__ __forceinline__ uint32_t apm_slot_hash(uint32_t f) {
    return __umul24(f, 0x9E3779u) + __umul24(f >> 12, 0x85EBCAu); // bucket = top bits, tag = low 16 bits
}

// bucket = top bits, tag = low 16 bits.  A 16-byte fingerprint is already mixed by its mad24;
// an 8-byte one (lo + hi*8, injective but structured) needs the extra multiply-mix.
template <int KL>
__device__ __forceinline__ uint32_t apm_table_hash(uint32_t f) {
    if constexpr (KL == 16) return f;
    else return apm_slot_hash(f);
}


// mask of the key bytes living in the second dword of an (up to) 8-byte key
template <int KL>
__device__ __forceinline__ constexpr uint32_t apm_hi_mask() {
    return KL >= 8 ? 0xffffffffu : (KL <= 4 ? 0u : ((1u << (8 * ((KL - 4) & 3))) - 1u));
}

template <int KL>
__device__ __forceinline__ bool apm_key_equal(const uint8_t *tb, int toff, const uint8_t *pb, int poff) {
    constexpr int ND = (KL + 3) / 4;
    uint32_t x[ND], y[ND];
    apm_lds_dwords<ND>(tb, toff, x);
    apm_lds_dwords<ND>(pb, poff, y);
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        uint32_t t = x[i] ^ y[i];
        if (i == ND - 1 && (KL & 3)) t &= (1u << (8 * (KL & 3))) - 1u;
        d |= t;
    }
    return d == 0u;
}

// where a candidate window's text bytes come from: the LDS tile (tile kernel) or global memory
// (stream kernel; the bytes were streamed moments ago, so they sit in L2 / Infinity Cache)
struct ApmLdsText {
    const uint8_t *base; // 16-byte aligned LDS buffer
    int off;             // window start inside it
    static constexpr bool kBlocks = false;
    __device__ __forceinline__ bool can16(int) const { return true; }
    __device__ __forceinline__ void load16(uint32_t (&T)[4]) const { apm_lds_dwords<4>(base, off, T); }
    __device__ __forceinline__ int byte(int x) const { return (int)base[off + x]; }
};
struct ApmGlobalText {
    const uint8_t *text; // 16-byte aligned
    int64_t off;         // window start (relative position)
    int64_t limit;       // bytes readable from text (avail_pad)
    static constexpr bool kBlocks = false;
    __device__ __forceinline__ bool can16(int) const { return (off & ~(int64_t)3) + 20 <= limit; }
    __device__ __forceinline__ void load16(uint32_t (&T)[4]) const {
        const uint32_t *a = reinterpret_cast<const uint32_t *>(text + (off & ~(int64_t)3));
        const uint32_t sh = (uint32_t)off & 3u;
        uint32_t w[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) w[i] = a[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) T[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);
    }
    __device__ __forceinline__ int byte(int x) const { return (int)text[off + x]; }
};


// ---- hierarchical verification (per-position key classes) ---------------------------------
// The k+1 pigeonhole pieces are paired (0,1), (2,3), ...  With <= k edits in the window some pair
// carries <= 1 edit (or, for even k, the unpaired last piece is intact), and inside such a pair one
// piece is intact while its partner matches the adjoining text with <= 1 edit, anchored at the
// intact piece and free at the far end.  A key hit therefore only needs the expensive window DP if
// (a) the rest of its piece is intact and (b) its partner extends with <= 1 edit -- two short byte
// comparisons that reject almost every random key hit.
// Core for partner pieces of <= 16 bytes (apm_ext1_core16 below): the one edit is located by the first
// mismatching byte i; the three ways to spend it are checked with shifted compares: substitution (P vs T),
// pattern byte without text counterpart (P vs T<<8), one extra text byte (P vs T>>8).  Longer partners (only
// when long patterns joined a per-position class) take the byte loops of apm_ext_fwd / apm_ext_bwd.
// (apm_ext1_core16 lives in apm_core.h: tests/host_core_test.cpp checks it against the byte loops on the host)
// pattern pb[pp..pp+n) vs text read FORWARD from tb[tp]: <= 1 edit, all of the pattern consumed
__device__ __forceinline__ bool apm_ext_fwd(const uint8_t *tb, int tp, const uint8_t *pb, int pp, int n) {
    int i = 0;
    while (i < n && tb[tp + i] == pb[pp + i]) ++i;
    if (i >= n - 1) return true; // no mismatch, or a single substitution at the last byte
    bool ok = true;              // substitution at i
    for (int j = i + 1; j < n && ok; ++j) ok = tb[tp + j] == pb[pp + j];
    if (ok) return true;
    ok = true;                   // pattern byte i has no text counterpart
    for (int j = i + 1; j < n && ok; ++j) ok = tb[tp + j - 1] == pb[pp + j];
    if (ok) return true;
    ok = true;                   // one extra text byte before pattern byte i
    for (int j = i; j < n && ok; ++j) ok = tb[tp + j + 1] == pb[pp + j];
    return ok;
}
// pattern pb[pp..pp+n) vs text read BACKWARD from tb[te-1] (te exclusive): <= 1 edit
__device__ __forceinline__ bool apm_ext_bwd(const uint8_t *tb, int te, const uint8_t *pb, int pp, int n) {
    int i = 0;
    while (i < n && tb[te - 1 - i] == pb[pp + n - 1 - i]) ++i;
    if (i >= n - 1) return true;
    bool ok = true;
    for (int j = i + 1; j < n && ok; ++j) ok = tb[te - 1 - j] == pb[pp + n - 1 - j];
    if (ok) return true;
    ok = true;
    for (int j = i + 1; j < n && ok; ++j) ok = tb[te - j] == pb[pp + n - 1 - j];
    if (ok) return true;
    ok = true;
    for (int j = i; j < n && ok; ++j) ok = tb[te - 2 - j] == pb[pp + n - 1 - j];
    return ok;
}

// DMA = 1: tiles travel HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR staging), three LDS
// tile buffers (four for the per-position classes, whose candidates of two consecutive tiles are verified in
// one pass), waits placed by hand (vmcnt(1): the younger tile stays in flight).  Needs a 16-byte aligned text
// pointer.  DMA = 0: register-staged buffer loads, two LDS buffers, compiler-placed waits.
typedef __attribute__((address_space(3))) uint8_t apm_lds_u8; // LDS byte, for constant-address accesses

template <int BAND, int KL, int STRIDE, int DMA>
__global__ __launch_bounds__(APM_BLOCK, (BAND == 0 && STRIDE > 1) ? 8 : ((BAND >= 2 && STRIDE == 1 && !DMA) ? 3 : ((BAND >= 3 || (BAND == 2 && STRIDE == 1)) ? 4 : ((BAND >= 1 && STRIDE == 1) ? 4 : 6)))) /* (per-position forms with a band: the LDS-DMA instantiations spilled 8 bytes per lane at 5 waves per SIMD; they are the fallback for unaligned text only since round 2, so they take the registers) */
void apm_filter_kernel(ApmFilterArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= a.n_main_blocks) { // extra workgroups: truncated tail windows (one pattern each)
        apm_tail_body(a.tail, (int)blockIdx.x - a.n_main_blocks, reinterpret_cast<uint4 *>(smem), tid);
        return;
    }
    // LDS: [tile 0 | tile 1 | (tile 2) | launch image (pattern bytes, hash table, key/pattern records) | queue | counts]
    // LDS-DMA: three tile buffers; the per-position classes keep a fourth so that the tile before the current
    // one stays intact and two tiles are verified in ONE pass (see the tile loop)
    constexpr bool TWO_TILES = DMA && STRIDE == 1;
    constexpr int NBUF = DMA ? (TWO_TILES ? 4 : 3) : 2;
    uint8_t *s_tile0 = smem;
    uint8_t *s_tile1 = smem + a.tile_len;
    uint8_t *s_img = smem + NBUF * APM_FILTER_POS;                       // (= a.tile_len, as a constant)
    uint8_t *s_pat = s_img + a.o_pat;                                    // raw pattern bytes
    uint4 *s_tab = reinterpret_cast<uint4 *>(s_img + a.o_tab);           // nb buckets x 8 16-bit tags
    uint4 *s_kid = reinterpret_cast<uint4 *>(s_img + a.o_kid);           // nb x 8 16-bit key ids
    uint32_t *s_ovf = reinterpret_cast<uint32_t *>(s_img + a.o_ovf);     // n_ovf x {tag, kid16}
    uint32_t *s_kinfo = reinterpret_cast<uint32_t *>(s_img + a.o_kinfo); // nk: pat | off<<12 | piece<<21
    uint2 *s_pinfo = reinterpret_cast<uint2 *>(s_img + a.o_pinfo);       // n_pats: {byte_off | m<<16, aux_off}
    const uint16_t *s_next = reinterpret_cast<const uint16_t *>(s_img + a.o_next); // nk: chain links (id+1, 0 = end)
    const uint16_t *s_poff = reinterpret_cast<const uint16_t *>(s_img + a.o_poff); // piece offsets a_q
    uint32_t *s_queue = reinterpret_cast<uint32_t *>(s_img + a.image_len); // 2 x qcap
    uint32_t *s_cnt = s_queue + 2 * a.qcap;
    uint32_t *s_qn = s_cnt + ((a.n_pats + 3) & ~3); // [2] queue counters, [4..7] per-wave survivor counters
    constexpr int SCAP = 128;                       // per-wave list of pre-check survivors (kid | pos << 16)
    uint32_t *s_surv = s_qn + 8;

    // Branch-free tile fetch: ONE raw buffer load of 16 bytes per lane per tile (tile = 4096 bytes),
    // the descriptor's num_records does the bounds check (out-of-range lanes return 0 and move no
    // data).  No other vector-memory operation lives in the tile loop, so the compiler's vmcnt
    // bookkeeping keeps the younger prefetch in flight while the older one is consumed.
    // The descriptor is based on the 16-byte granule that holds text[0] (the bytes in front of an unaligned
    // text pointer lie in the same allocation granule and never belong to a counted window): every lane's
    // 16-byte load then starts on a multiple of 16 from that base, so a load is either wholly in front of the
    // text (-> zeros) or wholly inside -- a load straddling text[0] would be dropped as a whole and lose the
    // first bytes of an unaligned text.
    const int text_sh = (int)(reinterpret_cast<uintptr_t>(a.text) & 15u);
    auto fetch = [&](int t, u32x4 &r0) __attribute__((always_inline)) {
        const int64_t g = a.tile0 + (int64_t)t * a.tile_w - a.front + text_sh; // multiple of 16, >= -32, from the granule base
        const int64_t gb = g > 0 ? g : 0;
        const int64_t lim = a.avail_pad + text_sh - gb;
        const uint32_t nrec = lim <= 0 ? 0u : (lim > 0x7fffffffLL ? 0x7fffffffu : (uint32_t)lim);
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.text) - text_sh + gb, 0, (int)nrec, 0x00020000);
        const uint32_t o0 = (uint32_t)((int)(g - gb) + 16 * tid); // negative wraps -> out of range -> 0
        r0 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o0, 0, 0);
    };
    auto stash = [&](uint8_t *buf, const u32x4 &r0) __attribute__((always_inline)) {
        *reinterpret_cast<u32x4 *>(buf + 16 * tid) = r0;
    };

    // tile indices fit 32 bits (the launcher refuses more than 2^30 tiles = 4 TiB of text per launch);
    // 64-bit loop state would not fit the SGPR budget and spill
    const int G = a.n_main_blocks;
    const int ntiles = (int)a.ntiles;
    int t = (int)blockIdx.x;
    u32x4 ra0 = {0, 0, 0, 0}, rb0 = {0, 0, 0, 0};

    // LDS-DMA of one tile: lane L's 16 bytes land at (buffer + wave*1024) + 16*L.  Lanes outside the
    // text read a clamped in-range address instead (their bytes are never part of a counted window).
    const uint32_t lds0 = __builtin_amdgcn_groupstaticsize();
    const uint32_t wave_off = __builtin_amdgcn_readfirstlane((uint32_t)tid >> 6) * 1024u;
    const uint32_t lane_off = 16u * (uint32_t)tid;
    auto dma = [&](int tt, int buf) __attribute__((always_inline)) {
        const int64_t gt = a.tile0 + (int64_t)tt * a.tile_w - a.front; // tile start (uniform)
        const uint32_t base = lds0 + (uint32_t)buf * (uint32_t)APM_FILTER_POS + wave_off;
        uint32_t keep;
        if (gt >= 0 && gt + APM_FILTER_POS <= a.avail_pad) { // whole tile inside the text: scalar base + lane offset
            const uint8_t *gbase = a.text + gt;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(lane_off), "s"(base), "s"(gbase)
                         : "memory");
        } else { // edge tile: clamp every lane's address into the text
            int64_t g = gt + 16 * tid;
            const int64_t hi = a.avail_pad - 16;
            g = g > hi ? hi : g;
            g = g < 0 ? 0 : g;
            const uint8_t *gp = a.text + g;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(gp), "s"(base)
                         : "memory");
        }
    };

    if constexpr (DMA) {
        for (int b = 0; b < 3; ++b)
            if (t + b * G < ntiles) dma(t + b * G, b);
    } else {
        if (t < ntiles) fetch(t, ra0);
    }
    // the launch image is ONE contiguous blob laid out like its LDS copy: a single round of
    // 16-byte loads, one wait (separate small copies would chain their HBM latencies)
    for (int i = tid; i < (a.image_len >> 4); i += APM_BLOCK)
        reinterpret_cast<uint4 *>(s_img)[i] = a.image[i];
    for (int i = tid; i < a.n_pats; i += APM_BLOCK) s_cnt[i] = 0u;
    if (tid < 2) s_qn[tid] = 0u;
    if constexpr (!DMA) {
        if (t < ntiles) {
            stash(s_tile0, ra0);
            if (t + G < ntiles) fetch(t + G, ra0);         // A: tile t+G   -> lands in buffer 1
            if (t + 2 * G < ntiles) fetch(t + 2 * G, rb0); // B: tile t+2G -> lands in buffer 0
        }
    }
    __syncthreads(); // (drains the prologue loads, DMA included)

    // verification of one (key, sampled text position, shift) nomination; bumps s_cnt
    // does piece q of the pattern (piece offsets at s_poff[aux..], n_pieces of them, length m, bytes at
    // s_pat+poff) found intact at LDS text offset tq pass the pair pre-check?  (see apm_ext_fwd)
    constexpr bool PAIRS = (STRIDE == 1) && (BAND >= 1);
    auto group_check = [&](const uint8_t *s_tile, int q, int tq, int aux, int n_pieces, int m, int poff) __attribute__((always_inline)) {
        if constexpr (!PAIRS) {
            return true;
        } else {
            const int aq = (int)s_poff[aux + q];
            const int aq1 = (q + 1 < n_pieces) ? (int)s_poff[aux + q + 1] : m;
            for (int x = KL; x < aq1 - aq; ++x) // rest of the piece behind its key bytes
                if (s_tile[tq + x] != s_pat[poff + aq + x]) return false;
            const int p = q ^ 1;
            if (p >= n_pieces) return true; // unpaired last piece (even k)
            // partner piece: after this one (p > q, text read forward from the end of the piece) or before it
            // (read backward from its start, both strings byte-reversed); <= 16 bytes -> one 128-bit core
            uint32_t P[4], T[5];
            int n;
            if (p > q) {
                const int ap1 = (p + 1 < n_pieces) ? (int)s_poff[aux + p + 1] : m;
                n = ap1 - aq1;
                if (n > 16) return apm_ext_fwd(s_tile, tq + (aq1 - aq), s_pat, poff + aq1, n);
                apm_lds_dwords<4>(s_pat, poff + aq1, P);
                apm_lds_dwords<5>(s_tile, tq + (aq1 - aq), T);
            } else {
                const int ap = (int)s_poff[aux + p];
                n = aq - ap;
                if (n > 16) return apm_ext_bwd(s_tile, tq, s_pat, poff + ap, n);
                // a window this tile counts has tq >= front + n - band >= 14: nearer the tile start it is none,
                // and the 20 text bytes in front of tq exist only from tq = 20 on (short partners need 9)
                if (tq < 12 || (tq < 20 && n > 8)) return false;
                uint32_t Q[4], W[5];
                apm_lds_dwords<4>(s_pat, poff + aq - 16, Q);
                const int back = tq < 20 ? 12 : 20;
                apm_lds_dwords<5>(s_tile, tq - back, W); // bytes [tq-back, tq-back+20)
#pragma unroll
                for (int z = 0; z < 4; ++z) P[z] = apm_bswap(Q[3 - z]);
                if (back == 20) {
#pragma unroll
                    for (int z = 0; z < 5; ++z) T[z] = apm_bswap(W[4 - z]);
                } else { // the 12 bytes in front of tq, reversed; n <= 8 looks at 9 of them
                    T[0] = apm_bswap(W[2]);
                    T[1] = apm_bswap(W[1]);
                    T[2] = apm_bswap(W[0]);
                    T[3] = 0u;
                    T[4] = 0u;
                }
            }
            return apm_ext1_core16(P, T, n);
        }
    };

    // stage 1 of a nomination (key, sampled text position): key bytes equal + pair pre-check
    auto stage1_item = [&](const uint8_t *s_tile, int kid, int pos) __attribute__((always_inline)) {
        const uint32_t ki = s_kinfo[kid];
        const int kpat = (int)(ki & 0xfffu), koff = (int)((ki >> 12) & 0x1ffu), kpiece = (int)((ki >> 21) & 7u);
        const uint2 pinf = s_pinfo[kpat];
        const int poff = (int)(pinf.x & 0xffffu);
        if (!apm_key_equal<KL>(s_tile, pos, s_pat, poff + koff)) return false; // fingerprint collision
        return (bool)group_check(s_tile, kpiece, pos, (int)pinf.y, a.k + 1, (int)(pinf.x >> 16), poff);
    };
    // the same for the per-position classes out of ONE packed record per key (a.o_kext): the whole piece in a
    // single masked 16-byte compare, then the partner through the 128-bit core; no walk through kinfo / pinfo /
    // piece offsets (four dependent LDS reads less on the path every candidate takes)
    const uint32_t *s_kext = reinterpret_cast<const uint32_t *>(s_img + a.o_kext);
    auto stage1_fast = [&](const uint8_t *s_tile, int kid, int pos) __attribute__((always_inline)) {
        typedef unsigned long long u64;
        const uint32_t kx = s_kext[kid];
        const int at = (int)(kx & 0xffffu), len = (int)((kx >> 16) & 0xffu), n = (int)((kx >> 24) & 31u), side = (int)(kx >> 29);
        uint32_t A[4], B[4];
        apm_lds_dwords<4>(s_tile, pos, A);
        apm_lds_dwords<4>(s_pat, at, B);
        const u64 ml = len >= 8 ? ~0ull : ((1ull << (8 * len)) - 1ull);
        const u64 mh = len <= 8 ? 0ull : (len >= 16 ? ~0ull : ((1ull << (8 * (len - 8))) - 1ull));
        const u64 dl = (((u64)(A[1] ^ B[1]) << 32) | (A[0] ^ B[0])) & ml, dh = (((u64)(A[3] ^ B[3]) << 32) | (A[2] ^ B[2])) & mh;
        if ((dl | dh) != 0ull) return false; // fingerprint collision, or the piece is not intact
        for (int x = 16; x < len; ++x)       // (pieces beyond 16 bytes: only when long patterns joined this class)
            if (s_tile[pos + x] != s_pat[at + x]) return false;
        if (side == 0) return true;          // unpaired last piece (even k)
        if (n == 31) return (bool)stage1_item(s_tile, kid, pos); // partner longer than 16 bytes: the generic walk
        uint32_t P[4], T[5];
        if (side == 1) {
            apm_lds_dwords<4>(s_pat, at + len, P);
            apm_lds_dwords<5>(s_tile, pos + len, T);
        } else {
            if (pos < 12 || (pos < 20 && n > 8)) return false; // see group_check
            uint32_t Q[4], W[5];
            apm_lds_dwords<4>(s_pat, at - 16, Q);
            const int back = pos < 20 ? 12 : 20;
            apm_lds_dwords<5>(s_tile, pos - back, W);
#pragma unroll
            for (int z = 0; z < 4; ++z) P[z] = apm_bswap(Q[3 - z]);
            if (back == 20) {
#pragma unroll
                for (int z = 0; z < 5; ++z) T[z] = apm_bswap(W[4 - z]);
            } else {
                T[0] = apm_bswap(W[2]);
                T[1] = apm_bswap(W[1]);
                T[2] = apm_bswap(W[0]);
                T[3] = 0u;
                T[4] = 0u;
            }
        }
        return apm_ext1_core16(P, T, n);
    };
    // stage 2: banded DP of the window the nomination implies under shift dl + stateless dedup; bumps s_cnt
    auto dp_item = [&](const uint8_t *s_tile, int64_t base, int kid, int pos, int dl) __attribute__((always_inline)) {
        const uint32_t ki = s_kinfo[kid];
        struct { int pat, off, piece; } key = {(int)(ki & 0xfffu), (int)((ki >> 12) & 0x1ffu), (int)((ki >> 21) & 7u)};
        const uint2 pinf = s_pinfo[key.pat];
        struct { int m, aux_off; } d = {(int)(pinf.x >> 16), (int)pinf.y};
        const int poff = (int)(pinf.x & 0xffffu);
        const int m = d.m;
        const int n_pieces = a.k + 1;
        const int64_t je_p = min(a.je, a.nrel - m + 1);
        const int jr = pos - a.front - key.off - dl; // window start relative to base
        const int64_t j = base + jr;
        if (jr < 0 || jr >= a.tile_w || j < a.jb || j >= je_p) return;
        if (!apm_banded_verify<BAND>(ApmLdsText{s_tile, a.front + jr}, s_pat, poff, m, a.k)) return;
        // count the window once: only from its first true (piece, shift) nominator
        for (int qq = 0; qq <= (int)key.piece; ++qq) {
            const int aq = (int)s_poff[d.aux_off + qq];
            for (int dd = -BAND; dd <= BAND; ++dd) {
                if (qq == (int)key.piece && dd >= dl) break;
                const int o = a.front + jr + aq + dd;           // piece start under shift dd
                const int rr = (STRIDE - (o % STRIDE)) % STRIDE; // its sampled (aligned) block
                if (apm_key_equal<KL>(s_tile, o + rr, s_pat, poff + aq + rr) &&
                    group_check(s_tile, qq, o, d.aux_off, n_pieces, m, poff))
                    return;
            }
        }
        atomicAdd(&s_cnt[key.pat], 1u);
    };
    auto verify_item = [&](const uint8_t *s_tile, int64_t base, int kid, int pos, int dl_lo, int dl_hi) __attribute__((always_inline)) {
        if (!stage1_item(s_tile, kid, pos)) return;
        if (APM_SKIP(a, 16)) return; // (measurement build) skip the banded DP
        for (int dl = dl_lo; dl <= dl_hi; ++dl) dp_item(s_tile, base, kid, pos, dl);
    };

    const uint32_t hshift = 32u - (uint32_t)a.lg_nb;
    // resident slot of this workgroup on its CU (workgroups b, b + n_cu, b + 2 n_cu, ... tend to share a CU;
    // a heuristic, only the spread of the verification over the SIMDs depends on it): slots 0,1,2,3 -> 0,2,1,3
    const int slot_on_cu = (int)blockIdx.x / (a.n_cu > 0 ? a.n_cu : 256);
    const int rot0 = 2 * slot_on_cu + (slot_on_cu >> 1);

    // filter + enqueue, barrier, cooperative verification of tile t held in s_tile
    // filter + enqueue of the tile held in s_tile: pushes (tag, position) into queue array `qa`
    // (0/1) under counter s_qn[qc]
    auto filter_tile = [&](const uint8_t *s_tile, int qa, int qc, int tile_bit) __attribute__((always_inline)) {
        const int p0 = tid * 16; // LDS offset of this lane's first position
        uint32_t *queue = s_queue + qa * a.qcap;

        auto probe = [&](uint32_t fi, int pos) __attribute__((always_inline)) { // one hash-table probe
            const uint32_t h = apm_table_hash<KL>(fi);
            const uint32_t slot = h >> hshift;
            const uint32_t tag = h & 0xffffu;
            const uint32_t rep = tag | (tag << 16);
            const uint4 tg = s_tab[slot]; // 8 x 16-bit tags
            const u16x2 m01 = __builtin_elementwise_min(apm_as_u16x2(tg.x ^ rep), apm_as_u16x2(tg.y ^ rep));
            const u16x2 m23 = __builtin_elementwise_min(apm_as_u16x2(tg.z ^ rep), apm_as_u16x2(tg.w ^ rep));
            const u16x2 mm = __builtin_elementwise_min(m01, m23); // v_pk_min_u16: a zero half = tag match
            bool hit = (mm.x == 0) | (mm.y == 0);
            for (int o = 0; o < a.n_ovf; ++o) hit |= (s_ovf[2 * o] == tag);
            if (hit) { // rare with long keys: defer the bucket walk to the cooperative phase
                const uint32_t idx = atomicAdd(&s_qn[qc], 1u);
                if (idx < (uint32_t)a.qcap) queue[idx] = (tag << 16) | (uint32_t)pos;
            }
        };
        if (APM_SKIP(a, 1)) return; // (measurement build) skip the probes
        const uint4 va = *reinterpret_cast<const uint4 *>(s_tile + p0);
        if constexpr (STRIDE == 16) {
            probe(apm_fp16(apm_fp8(va.x, va.y), apm_fp8(va.z, va.w)), p0);
        } else if constexpr (STRIDE == 8) {
            probe(apm_fp8(va.x, va.y), p0);
            probe(apm_fp8(va.z, va.w), p0 + 8);
        } else {
            // every position (KL = 8, 6 or 4 key bytes).  First level = a presence bitmap over the 2-bit byte
            // codes (b >> code_shift) & 3: the lane packs the codes of its 24 bytes into 48 bits once, the
            // key code word of position i is a 2i-bit funnel shift of them, and its bit sits in byte
            // x & (NB-1) of the bitmap (ds_read_u8), bit x >> log2(NB).  ~6 instructions per position instead
            // of a fingerprint + bucket probe; the bucket walk happens in the cooperative phase for the hits.
            const uint2 vb = *reinterpret_cast<const uint2 *>(s_tile + p0 + 16);
            const uint32_t w[6] = {va.x, va.y, va.z, va.w, vb.x, vb.y};
            uint32_t pk[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const uint32_t c = (w[i] >> a.code_shift) & 0x03030303u;
                const uint32_t u = c | (c >> 6);
                pk[i] = (u | (u >> 12)) & 0xffu;
            }
            const uint32_t clo = pk[0] | (pk[1] << 8) | (pk[2] << 16) | (pk[3] << 24);
            const uint32_t chi = pk[4] | (pk[5] << 8);
            constexpr int LB = 13; // log2 of the bitmap's byte count: 8 KiB over 8-byte code words for every key length
            // the bitmap leads the image, and these kernels own no static LDS (tests check the build's
            // resource digest): its LDS address is a compile-time constant -> no address add per probe
            const apm_lds_u8 *bmp0 = (const apm_lds_u8 *)(uintptr_t)(NBUF * APM_FILTER_POS);
            uint32_t hits = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) { // and, ds_read_u8, 2 x v_bfe_u32, v_lshl_or_b32 per position
                const uint32_t x = i ? __builtin_amdgcn_alignbit(chi, clo, 2u * (uint32_t)i) : clo;
                const uint32_t byte = bmp0[x & ((1u << LB) - 1u)];
                hits |= __builtin_amdgcn_ubfe(byte, __builtin_amdgcn_ubfe(x, LB, 3), 1) << i;
            }
            while (hits) {
                const int i = __builtin_ctz(hits);
                hits &= hits - 1u;
                const uint32_t idx = atomicAdd(&s_qn[qc], 1u);
                if (idx < (uint32_t)a.qcap) queue[idx] = (uint32_t)(p0 + i) | ((uint32_t)tile_bit << 12);
            }
        }
    };

    // cooperative verification of the candidates queued for tile t (held in s_tile); the queue must be
    // complete (a barrier since its filter)
    // (the queue may hold the candidates of TWO tiles: bit 12 of an entry's position says which one -- the
    // "even" tile s_tile/t or the "odd" tile s_tile1/t1; callers with one tile pass it twice)
    auto verify_tile = [&](const uint8_t *s_tile, int t, const uint8_t *s_tile1, int t1, int qa, int qc, int rot) __attribute__((always_inline)) {
        // queue entries are dealt to the waves starting at wave `rot` (rotates per tile and workgroup): a short
        // queue keeps one wave busy, and wave i of every resident workgroup sits on SIMD i -- without the
        // rotation the verification of the whole CU would pile up on SIMD 0
        const uint32_t vtid = (uint32_t)(tid - 64 * rot) & (APM_BLOCK - 1);
        const int64_t base = a.tile0 + (int64_t)t * a.tile_w; // first window start of the tile
        const int64_t base1 = a.tile0 + (int64_t)t1 * a.tile_w;
        const uint8_t *const s_tile0v = s_tile;
        const int64_t base0v = base;
        const int p0 = tid * 16;
        constexpr int NF = 16 / STRIDE;
        const uint32_t *queue = s_queue + qa * a.qcap;
        const uint32_t qn = s_qn[qc];
        constexpr int NSH = 2 * BAND + 1;
        // all keys whose tag matches at this sampled position (bucket ways, overflow list, chains);
        // one runtime loop = ONE inlined copy of the verification code
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        auto for_each_key = [&](uint32_t tag, int pos13, int dl_lo, int dl_hi) __attribute__((always_inline)) {
            const int pos = pos13 & 0xfff;
            const uint8_t *s_tile = (pos13 & 0x1000) ? s_tile1 : s_tile0v; // (shadows: this entry's tile)
            const int64_t base = (pos13 & 0x1000) ? base1 : base0v;
            auto handle = [&](int kid) __attribute__((always_inline)) {
                if constexpr (PAIRS) {
                    // pre-check here, banded DP later: the survivors (few per wave) go to a wave-private list so
                    // that the DP runs on dense lanes, one (survivor, shift) each, instead of inside this
                    // divergent walk with the three shifts in sequence
                    if (!stage1_fast(s_tile, kid, pos)) return;
                    if (APM_SKIP(a, 16)) return; // (measurement build) skip the banded DP
                    const uint32_t idx = atomicAdd(&s_qn[4 + wv], 1u);
                    if (idx < (uint32_t)SCAP) s_surv[wv * SCAP + idx] = (uint32_t)kid | ((uint32_t)pos13 << 16);
                    else
                        for (int dl = dl_lo; dl <= dl_hi; ++dl) dp_item(s_tile, base, kid, pos, dl); // list full (rare)
                } else {
                    verify_item(s_tile, base, kid, pos, dl_lo, dl_hi);
                }
            };
            uint32_t fw[(KL + 3) / 4];
            apm_lds_dwords<(KL + 3) / 4>(s_tile, pos, fw);
            uint32_t fi;
            if constexpr (KL == 16) fi = apm_fp16(apm_fp8(fw[0], fw[1]), apm_fp8(fw[2], fw[3]));
            else if constexpr (KL > 4) fi = apm_fp8(fw[0], fw[1] & apm_hi_mask<KL>());
            else fi = fw[0];
            const uint32_t hh = apm_table_hash<KL>(fi);
            const uint32_t slot = hh >> hshift;
            if constexpr (STRIDE == 1) tag = hh & 0xffffu; // (the bitmap filter queues bare positions)
            const uint16_t *tag16 = reinterpret_cast<const uint16_t *>(s_tab + slot);
            const uint16_t *kid16p = reinterpret_cast<const uint16_t *>(s_kid + slot);
            // Find this lane's matching way first (one 16-byte read of the tags, one of the key ids), THEN
            // verify: all lanes of the wave run the checks together instead of once per way index.
            uint32_t first = 0xffffffffu;
            int n_match = 0;
            {
                const uint4 tg = s_tab[slot], kd = s_kid[slot];
                const uint32_t tw[4] = {tg.x, tg.y, tg.z, tg.w}, kw[4] = {kd.x, kd.y, kd.z, kd.w};
#pragma unroll
                for (int wv = 3; wv >= 0; --wv) { // descending: `first` ends up as the lowest matching way
                    const uint32_t khi = kw[wv] >> 16, klo = kw[wv] & 0xffffu;
                    if ((tw[wv] >> 16) == tag && khi != 0xffffu) { first = khi; ++n_match; }
                    if ((tw[wv] & 0xffffu) == tag && klo != 0xffffu) { first = klo; ++n_match; }
                }
                for (int o = 0; o < a.n_ovf; ++o) // ascending: same "first" as the walk below
                    if (s_ovf[2 * o] == tag) {
                        if (n_match == 0) first = s_ovf[2 * o + 1];
                        ++n_match;
                    }
            }
            if (n_match > 0) {
                uint32_t kid = first & 0x7fffu;
                const bool more = (first & 0x8000u) != 0; // heads a chain of keys with the same tag
                for (;;) {
                    handle((int)kid);
                    if (!more) break;
                    const uint32_t nxt = s_next[kid];
                    if (!nxt) break;
                    kid = nxt - 1u;
                }
            }
            if (n_match > 1) { // several ways carry this tag (rare): walk the remaining ones
                int seen = 0;
#pragma unroll 1
                for (int c = 0; c < 8 + a.n_ovf; ++c) {
                    uint32_t t16, kid16;
                    if (c < 8) {
                        t16 = tag16[c];
                        kid16 = kid16p[c];
                    } else {
                        t16 = s_ovf[2 * (c - 8)];
                        kid16 = s_ovf[2 * (c - 8) + 1];
                    }
                    if (t16 != tag || kid16 == 0xffffu) continue;
                    if (seen++ == 0) continue;
                    uint32_t kid = kid16 & 0x7fffu;
                    const bool more = (kid16 & 0x8000u) != 0;
                    for (;;) {
                        handle((int)kid);
                        if (!more) break;
                        const uint32_t nxt = s_next[kid];
                        if (!nxt) break;
                        kid = nxt - 1u;
                    }
                }
            }
        };
        if (APM_SKIP(a, 8)) { // (measurement build) skip verification
        } else if (qn <= (uint32_t)a.qcap) {
            if constexpr (PAIRS) { // work item = queue entry: the cheap pair pre-check runs once per entry ...
                if ((tid & 63) == 0) s_qn[4 + wv] = 0u; // (wave-private: LDS operations of one wave stay in order)
                for (uint32_t wi = vtid; wi < qn; wi += APM_BLOCK) {
                    const uint32_t ent = queue[wi];
                    for_each_key(ent >> 16, (int)(ent & 0xffffu), -BAND, BAND);
                }
                // ... then work item = (survivor, shift) of this wave's list
                __builtin_amdgcn_wave_barrier();
                const uint32_t ns = min(s_qn[4 + wv], (uint32_t)SCAP);
                for (uint32_t wi = (uint32_t)(tid & 63); wi < ns * NSH; wi += 64) {
                    const uint32_t e = s_surv[wv * SCAP + wi / NSH];
                    const bool odd = (e >> 28) & 1u;
                    dp_item(odd ? s_tile1 : s_tile, odd ? base1 : base, (int)(e & 0xffffu), (int)((e >> 16) & 0xfffu), (int)(wi % NSH) - BAND);
                }
            } else { // work item = (queue entry, shift): keeps all lanes busy
                for (uint32_t wi = vtid; wi < qn * NSH; wi += APM_BLOCK) {
                    const uint32_t ent = queue[wi / NSH];
                    const int dl = (int)(wi % NSH) - BAND;
                    for_each_key(ent >> 16, (int)(ent & 0xffffu), dl, dl);
                }
            }
        } else { // queue overflow: dense pass over every (sampled position, key) of the tile(s)
            for (int half = 0; half < (t1 != t ? 2 : 1); ++half)
                for (int i = 0; i < NF; ++i)
                    for (int kid = 0; kid < a.nk; ++kid)
                        verify_item(half ? s_tile1 : s_tile, half ? base1 : base, kid, p0 + i * STRIDE, -BAND, BAND);
        }
    };

    if constexpr (DMA) {
        for (int it = 0; t < ntiles; ++it, t += G) {
            // tile t (DMA issued two iterations ago) must have landed; the DMA of tile t+G stays in flight
            if (t + G < ntiles) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // every wave is past tile t-G now: its buffer takes tile t+2G
            if (it >= 1 && t + 2 * G < ntiles && !APM_SKIP(a, 2)) dma(t + 2 * G, (it + 2) % NBUF);
            const uint8_t *s_tile = smem + (it % NBUF) * APM_FILTER_POS;
            if constexpr (TWO_TILES) {
                // Dense per-position classes: the verification pass is latency bound and its lanes mostly
                // empty, so the candidates of two consecutive tiles share ONE queue and ONE pass (the fourth
                // buffer keeps the even tile intact while the odd one is filtered).
                const int q = (it >> 1) & 1;
                filter_tile(s_tile, q, q, it & 1);
                if ((it & 1) || t + G >= ntiles) {
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // A: queue complete (LDS only)
                    if (tid == 0) s_qn[q ^ 1] = 0u; // the next pair's counter (pushed to only after the next barrier)
                    if (it & 1) verify_tile(smem + ((it - 1) % NBUF) * APM_FILTER_POS, t - G, s_tile, t, q, q, ((it >> 1) + rot0) & 3);
                    else verify_tile(s_tile, t, s_tile, t, q, q, ((it >> 1) + rot0) & 3);
                }
            } else {
                filter_tile(s_tile, it & 1, it & 1, 0);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // A: queue complete (LDS only)
                if (tid == 0) s_qn[(it + 1) & 1] = 0u; // next tile's counter (read again only after the next barrier)
                verify_tile(s_tile, t, s_tile, t, it & 1, it & 1, (it + rot0) & 3);
            }
        }
    } else {
        // one iteration: filter tile t, barrier, verify it, then land the registers `r` (tile t+G) in the
        // other buffer and refill them with tile t+3G
        auto iteration = [&](int it, int t, const uint8_t *s_tile, uint8_t *s_other, u32x4 &r0) __attribute__((always_inline)) {
            filter_tile(s_tile, it & 1, it & 1, 0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // A: queue complete (LDS only)
            if (tid == 0) s_qn[(it + 1) & 1] = 0u; // next iteration's counter (nobody reads it before barrier B)
            verify_tile(s_tile, t, s_tile, t, it & 1, it & 1, (it + rot0) & 3);
            if (t + G < ntiles) {
                stash(s_other, r0);
                if (t + 3 * G < ntiles && !APM_SKIP(a, 2)) fetch(t + 3 * G, r0);
            }
            __syncthreads(); // B
        };
        for (int it = 0; t < ntiles; it += 2, t += 2 * G) {
            iteration(it, t, s_tile0, s_tile1, ra0);
            if (t + G < ntiles) iteration(it + 1, t + G, s_tile1, s_tile0, rb0);
        }
    }
    __syncthreads();

    for (int i = tid; i < a.n_pats; i += APM_BLOCK) {
        const uint32_t c = s_cnt[i];
        if (c) atomicAdd(&a.counts[a.pats[i].index], (unsigned long long)c);
    }
}

// ---------------------------------------------------------------------------
// STREAM form of the BANDED filter: every WAVE is autonomous -- no LDS text tile, no workgroup
// barrier in the loop.  A wave walks 1 KiB chunks (16 bytes per lane; chunks c, c+W, c+2W, ... for W
// waves in flight) and keeps four buffer loads per lane in flight.  Sampled classes (STRIDE == KL)
// fingerprint the lane's own 16 bytes; per-position classes (STRIDE == 1) also fetch the 8 bytes
// that follow them.  Fingerprints probe the LDS hash table; the hits go to a wave-private LDS queue
// (ballot + mbcnt, no atomics).  When the queue holds a wave's worth of work it is verified on the
// spot against global memory (the bytes were streamed moments ago: L2 / Infinity Cache hits): key
// compare, pair pre-check (per-position classes), banded DP, stateless dedup.
// Requires a 16-byte aligned text pointer (sampling grid = address grid).
// ---------------------------------------------------------------------------
// N dwords of text starting at relative position off (any alignment), zero beyond [0, limit)
template <int N>
__device__ __forceinline__ void apm_gdwords(const uint8_t *text, int64_t limit, int64_t off, uint32_t (&out)[N]) {
    const int64_t a0 = off & ~(int64_t)3;
    const uint32_t sh = (uint32_t)off & 3u;
    uint32_t w[N + 1];
#pragma unroll
    for (int i = 0; i <= N; ++i) {
        const int64_t q = a0 + 4 * i;
        w[i] = (q >= 0 && q + 4 <= limit) ? *reinterpret_cast<const uint32_t *>(text + q) : 0u;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);
}
__device__ __forceinline__ int apm_gbyte(const uint8_t *text, int64_t limit, int64_t off) {
    return (off >= 0 && off < limit) ? (int)text[off] : 0x100;
}
// global-text versions of apm_ext_fwd / apm_ext_bwd (see there)
__device__ __forceinline__ bool apm_ext_fwd_g(const uint8_t *text, int64_t limit, int64_t tp, const uint8_t *pb, int pp, int n) {
    int i = 0;
    while (i < n && apm_gbyte(text, limit, tp + i) == (int)pb[pp + i]) ++i;
    if (i >= n - 1) return true;
    bool ok = true;
    for (int j = i + 1; j < n && ok; ++j) ok = apm_gbyte(text, limit, tp + j) == (int)pb[pp + j];
    if (ok) return true;
    ok = true;
    for (int j = i + 1; j < n && ok; ++j) ok = apm_gbyte(text, limit, tp + j - 1) == (int)pb[pp + j];
    if (ok) return true;
    ok = true;
    for (int j = i; j < n && ok; ++j) ok = apm_gbyte(text, limit, tp + j + 1) == (int)pb[pp + j];
    return ok;
}
__device__ __forceinline__ bool apm_ext_bwd_g(const uint8_t *text, int64_t limit, int64_t te, const uint8_t *pb, int pp, int n) {
    int i = 0;
    while (i < n && apm_gbyte(text, limit, te - 1 - i) == (int)pb[pp + n - 1 - i]) ++i;
    if (i >= n - 1) return true;
    bool ok = true;
    for (int j = i + 1; j < n && ok; ++j) ok = apm_gbyte(text, limit, te - 1 - j) == (int)pb[pp + n - 1 - j];
    if (ok) return true;
    ok = true;
    for (int j = i + 1; j < n && ok; ++j) ok = apm_gbyte(text, limit, te - j) == (int)pb[pp + n - 1 - j];
    if (ok) return true;
    ok = true;
    for (int j = i; j < n && ok; ++j) ok = apm_gbyte(text, limit, te - 2 - j) == (int)pb[pp + n - 1 - j];
    return ok;
}

template <int BAND, int KL, int STRIDE>
__global__ __launch_bounds__(APM_BLOCK, (BAND == 0 && STRIDE > 1) ? 7 : ((BAND >= 1 && STRIDE == 1) ? 3 : ((BAND >= 2 || STRIDE == 1) ? 4 : 5)))
void apm_stream_kernel(ApmFilterArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6); // provably wave-uniform: descriptors stay in SGPRs
    if ((int)blockIdx.x >= a.n_main_blocks) { // extra workgroups: truncated tail windows (one pattern each)
        apm_tail_body(a.tail, (int)blockIdx.x - a.n_main_blocks, reinterpret_cast<uint4 *>(smem), tid);
        return;
    }
    constexpr int NSH = 2 * BAND + 1;
    constexpr bool PAIRS = (STRIDE == 1) && (BAND >= 1);
    constexpr int GRP = 4;                        // probes between two flush checks
    constexpr int QW = 64 * GRP + 64;             // wave queue entries: flushed as soon as it holds >= 64
    uint8_t *s_img = smem;
    uint8_t *s_pat = s_img + a.o_pat;
    const uint4 *s_tab = reinterpret_cast<const uint4 *>(s_img + a.o_tab);
    const uint4 *s_kid = reinterpret_cast<const uint4 *>(s_img + a.o_kid);
    const uint32_t *s_ovf = reinterpret_cast<const uint32_t *>(s_img + a.o_ovf);
    const uint32_t *s_kinfo = reinterpret_cast<const uint32_t *>(s_img + a.o_kinfo);
    const uint2 *s_pinfo = reinterpret_cast<const uint2 *>(s_img + a.o_pinfo);
    const uint16_t *s_next = reinterpret_cast<const uint16_t *>(s_img + a.o_next);
    const uint16_t *s_poff = reinterpret_cast<const uint16_t *>(s_img + a.o_poff);
    uint2 *s_queue = reinterpret_cast<uint2 *>(s_img + a.image_len) + wv * QW; // this wave's queue
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_img + a.image_len + 4 * QW * 8);

    apm_stage_image(reinterpret_cast<uint4 *>(s_img), a.image, a.image_len >> 4, tid, APM_BLOCK);
    for (int i = tid; i < a.n_pats; i += APM_BLOCK) s_cnt[i] = 0u;
    __syncthreads(); // the only workgroup barrier before the final count flush

    const uint32_t hshift = 32u - (uint32_t)a.lg_nb;
    const int64_t W = (int64_t)a.n_main_blocks * (APM_BLOCK / 64);
    const int64_t nch = a.ntiles; // 1 KiB chunks from relative position a.tile0 (multiple of 16)
    const uint8_t *text = a.text;
    const int64_t limit = a.avail_pad;

    auto load_chunk = [&](int64_t cc, u32x4 &r, uint2 &e) __attribute__((always_inline)) {
        const int64_t g = a.tile0 + cc * 1024;
        const int64_t lim = cc < nch ? a.avail_pad - g : 0; // chunks past the end: zero records -> zeros, no traffic
        const uint32_t nrec = lim <= 0 ? 0u : (lim > 1040 ? 1040u : (uint32_t)lim);
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.text) + (cc < nch ? g : 0), 0, (int)nrec, 0x00020000);
        r = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * lane, 0, 0); // (nt / sc0 / sc1 cache-policy bits measured: no gain, nt loses 4 %)
        if constexpr (STRIDE == 1) { // the 8 bytes behind the lane's 16 (next lane's / next chunk's head)
            typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
            const u32x2v x = __builtin_amdgcn_raw_buffer_load_b64(rs, 16 * lane + 16, 0, 0);
            e = make_uint2(x.x, x.y);
        }
    };

    // text[tpos..+KL) == pattern bytes [ppos..+KL) of the pattern stored at s_pat+poff ?
    auto key_at = [&](int64_t tpos, int poff, int ppos) __attribute__((always_inline)) {
        constexpr int ND = (KL + 3) / 4;
        uint32_t x[ND], y[ND];
        apm_lds_dwords<ND>(s_pat, poff + ppos, y);
        apm_gdwords<ND>(text, limit, tpos, x);
        uint32_t dd = 0;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            uint32_t tt = x[i] ^ y[i];
            if (i == ND - 1 && (KL & 3)) tt &= (1u << (8 * (KL & 3))) - 1u;
            dd |= tt;
        }
        return dd == 0u;
    };
    // pair pre-check of piece q found intact at text position tq (see apm_ext_fwd)
    auto group_check = [&](int q, int64_t tq, int aux, int n_pieces, int m, int poff) __attribute__((always_inline)) {
        if constexpr (!PAIRS) {
            return true;
        } else {
            const int aq = (int)s_poff[aux + q];
            const int aq1 = (q + 1 < n_pieces) ? (int)s_poff[aux + q + 1] : m;
            for (int x = KL; x < aq1 - aq; ++x) // rest of the piece behind its key bytes
                if (apm_gbyte(text, limit, tq + x) != (int)s_pat[poff + aq + x]) return false;
            const int p = q ^ 1;
            if (p >= n_pieces) return true; // unpaired last piece (even k)
            uint32_t P[4], T[5]; // (see apm_filter_kernel: one 128-bit core for both directions)
            int n;
            if (p > q) {
                const int ap1 = (p + 1 < n_pieces) ? (int)s_poff[aux + p + 1] : m;
                n = ap1 - aq1;
                if (n > 16) return apm_ext_fwd_g(text, limit, tq + (aq1 - aq), s_pat, poff + aq1, n);
                apm_lds_dwords<4>(s_pat, poff + aq1, P);
                apm_gdwords<5>(text, limit, tq + (aq1 - aq), T);
            } else {
                const int ap = (int)s_poff[aux + p];
                n = aq - ap;
                if (n > 16) return apm_ext_bwd_g(text, limit, tq, s_pat, poff + ap, n);
                uint32_t Q[4], W[5];
                apm_lds_dwords<4>(s_pat, poff + aq - 16, Q);
                apm_gdwords<5>(text, limit, tq - 20, W);
#pragma unroll
                for (int z = 0; z < 4; ++z) P[z] = apm_bswap(Q[3 - z]);
#pragma unroll
                for (int z = 0; z < 5; ++z) T[z] = apm_bswap(W[4 - z]);
            }
            return apm_ext1_core16(P, T, n);
        }
    };

    // ---- verification of one (key, sampled position) nomination over shifts [dl_lo, dl_hi] ----
    auto verify_item = [&](int kid, int64_t pos, int dl_lo, int dl_hi) __attribute__((always_inline)) {
        const uint32_t ki = s_kinfo[kid];
        const int kpat = (int)(ki & 0xfffu), koff = (int)((ki >> 12) & 0x1ffu), kpiece = (int)((ki >> 21) & 7u);
        const uint2 pinf = s_pinfo[kpat];
        const int m = (int)(pinf.x >> 16), aux = (int)pinf.y, poff = (int)(pinf.x & 0xffffu);
        const int n_pieces = a.k + 1;
        if (!key_at(pos, poff, koff)) return; // fingerprint / tag collision
        if (!group_check(kpiece, pos, aux, n_pieces, m, poff)) return;
        const int64_t je_p = min(a.je, a.nrel - m + 1);
        for (int dl = dl_lo; dl <= dl_hi; ++dl) {
            const int64_t j = pos - koff - dl; // candidate window start
            if (j < a.jb || j >= je_p) continue;
            if (!apm_banded_verify<BAND>(ApmGlobalText{text, j, limit}, s_pat, poff, m, a.k)) continue;
            // count the window once: only from its first true (piece, shift) nominator
            bool first = true;
            for (int qq = 0; qq <= kpiece && first; ++qq) {
                const int aq = (int)s_poff[aux + qq];
                for (int dd = -BAND; dd <= BAND; ++dd) {
                    if (qq == kpiece && dd >= dl) break;
                    const int64_t o = j + aq + dd;                                   // piece start under shift dd
                    const int rr = (int)((STRIDE - (o & (STRIDE - 1))) & (STRIDE - 1)); // its sampled (aligned) block
                    if (o + rr >= 0 && key_at(o + rr, poff, aq + rr) && group_check(qq, o, aux, n_pieces, m, poff)) {
                        first = false;
                        break;
                    }
                }
            }
            if (first) atomicAdd(&s_cnt[kpat], 1u);
        }
    };

    auto flush = [&](uint32_t qcount) __attribute__((always_inline)) {
        const uint32_t nitems = PAIRS ? qcount : qcount * NSH; // PAIRS: the cheap pre-check runs once per entry
        for (uint32_t wi = lane; wi < nitems; wi += 64) {
            const uint2 ent = s_queue[PAIRS ? wi : wi / NSH];
            const int dl_lo = PAIRS ? -BAND : (int)(wi % NSH) - BAND;
            const int dl_hi = PAIRS ? BAND : dl_lo;
            const int64_t pos = (int64_t)ent.x | ((int64_t)ent.y << 32);
            if (pos + KL > a.avail) continue;
            uint32_t fw[(KL + 3) / 4];
            apm_gdwords<(KL + 3) / 4>(text, limit, pos, fw);
            uint32_t fi;
            if constexpr (KL == 16) fi = apm_fp16(apm_fp8(fw[0], fw[1]), apm_fp8(fw[2], fw[3]));
            else if constexpr (KL > 4) fi = apm_fp8(fw[0], fw[1] & apm_hi_mask<KL>());
            else fi = fw[0];
            const uint32_t hh = apm_table_hash<KL>(fi);
            const uint32_t slot = hh >> hshift;
            const uint32_t tag = hh & 0xffffu; // (the bitmap filter queues bare positions)
            uint32_t first = 0xffffffffu; // this lane's matching way, verified with the whole wave
            int n_match = 0;
            {
                const uint4 tg = s_tab[slot], kd = s_kid[slot];
                const uint32_t tw[4] = {tg.x, tg.y, tg.z, tg.w}, kw[4] = {kd.x, kd.y, kd.z, kd.w};
#pragma unroll
                for (int w4 = 3; w4 >= 0; --w4) { // descending: `first` ends up as the lowest matching way
                    const uint32_t khi = kw[w4] >> 16, klo = kw[w4] & 0xffffu;
                    if ((tw[w4] >> 16) == tag && khi != 0xffffu) { first = khi; ++n_match; }
                    if ((tw[w4] & 0xffffu) == tag && klo != 0xffffu) { first = klo; ++n_match; }
                }
                for (int o = 0; o < a.n_ovf; ++o)
                    if (s_ovf[2 * o] == tag) {
                        if (n_match == 0) first = s_ovf[2 * o + 1];
                        ++n_match;
                    }
            }
            if (n_match > 0) {
                uint32_t kid = first & 0x7fffu;
                const bool more = (first & 0x8000u) != 0;
                for (;;) {
                    verify_item((int)kid, pos, dl_lo, dl_hi);
                    if (!more) break;
                    const uint32_t nxt = s_next[kid];
                    if (!nxt) break;
                    kid = nxt - 1u;
                }
            }
            if (n_match > 1) { // several ways carry this tag (rare)
                const uint16_t *tag16 = reinterpret_cast<const uint16_t *>(s_tab + slot);
                const uint16_t *kid16p = reinterpret_cast<const uint16_t *>(s_kid + slot);
                int seen = 0;
#pragma unroll 1
                for (int c = 0; c < 8 + a.n_ovf; ++c) {
                    uint32_t t16, kid16;
                    if (c < 8) {
                        t16 = tag16[c];
                        kid16 = kid16p[c];
                    } else {
                        t16 = s_ovf[2 * (c - 8)];
                        kid16 = s_ovf[2 * (c - 8) + 1];
                    }
                    if (t16 != tag || kid16 == 0xffffu) continue;
                    if (seen++ == 0) continue;
                    uint32_t kid = kid16 & 0x7fffu;
                    const bool more = (kid16 & 0x8000u) != 0;
                    for (;;) {
                        verify_item((int)kid, pos, dl_lo, dl_hi);
                        if (!more) break;
                        const uint32_t nxt = s_next[kid];
                        if (!nxt) break;
                        kid = nxt - 1u;
                    }
                }
            }
        }
    };

    uint32_t qcount = 0; // wave-uniform
    auto drain = [&]() __attribute__((always_inline)) {
        if (qcount >= 64u) {
            flush(qcount);
            qcount = 0;
        }
    };
    // First-level filter of the per-position classes: a presence bitmap over the 2-bit byte codes
    // (b >> code_shift) & 3 of the key bytes (built by the host next to the hash table).  The lane packs
    // the codes of its 24 bytes into a bit string once; the code word of position i is a 2i-bit funnel
    // shift of it and its bit sits in byte x & (NB-1) of the bitmap, bit x >> log2(NB).  Returns the
    // lane's 16-bit hit mask (bit i = position i of its 16 bytes).
    auto pack4 = [&](uint32_t wv4) __attribute__((always_inline)) { // 4 bytes -> 8 code bits
        const uint32_t cd = (wv4 >> a.code_shift) & 0x03030303u;
        const uint32_t u = cd | (cd >> 6);
        return (u | (u >> 12)) & 0xffu;
    };
    constexpr int LB = 13; // log2 of the bitmap's byte count: 8 KiB over 8-byte code words for every key length
    // the bitmap leads the image = the start of dynamic LDS, and this kernel owns no static LDS (tests
    // check the build's resource digest): LDS address 0, a compile-time constant -> no address add per probe
    const apm_lds_u8 *bmp0 = (const apm_lds_u8 *)(uintptr_t)0;
    auto bmp_bit = [&](uint32_t x) __attribute__((always_inline)) { // and, ds_read_u8, 2 x v_bfe_u32
        const uint32_t byte = bmp0[x & ((1u << LB) - 1u)];
        return __builtin_amdgcn_ubfe(byte, __builtin_amdgcn_ubfe(x, LB, 3), 1);
    };
    auto hit_bits = [&](const u32x4 &v, const uint2 &e, int64_t cc) __attribute__((always_inline)) {
        if (APM_SKIP(a, 32)) return (v.x ^ e.x) == 0x12345u ? 1u : 0u; // (measurement build) streaming skeleton only
        uint32_t hits = 0;
        const uint32_t clo = pack4(v.x) | (pack4(v.y) << 8) | (pack4(v.z) << 16) | (pack4(v.w) << 24);
        const uint32_t chi = pack4(e.x) | (pack4(e.y) << 8);
#pragma unroll
        for (int i = 15; i >= 0; --i) // descending: hits = hits << 1 | bit (one v_lshl_or_b32 each)
            hits = (hits << 1) | bmp_bit(i ? __builtin_amdgcn_alignbit(chi, clo, 2u * (uint32_t)i) : clo);
        return (cc < nch && !APM_SKIP(a, 1)) ? hits : 0u;
    };
    // the hit positions of one chunk go to the wave queue, one bit per lane and round (<= 16 rounds of
    // <= 64 positions); the queue is verified as soon as it holds a wave's worth
    auto push_hits = [&](uint32_t hits, int64_t cc) __attribute__((always_inline)) {
        const int64_t pos = a.tile0 + cc * 1024 + 16 * lane;
        while (__builtin_amdgcn_ballot_w64(hits != 0)) {
            const bool has = hits != 0;
            const int i = has ? __builtin_ctz(hits) : 0;
            hits &= hits - 1u; // (0 stays 0)
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(has);
            const uint32_t idx = qcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            const int64_t pp = pos + i;
            if (has) s_queue[idx] = make_uint2((uint32_t)pp, (uint32_t)(pp >> 32));
            qcount += (uint32_t)__builtin_popcountll(mask);
            drain();
        }
    };

    // sampled classes: one (stride 16) or two (stride 8) fingerprints per lane and chunk probe the hash
    // table directly (8 x 16-bit tags per bucket, v_pk_min_u16 match); hits are queued on the spot.
    // (A bitmap first level was measured here too: no gain at one or two probes per lane.)
    auto probe = [&](uint32_t fi, int64_t pos, bool valid) __attribute__((always_inline)) {
        const uint32_t h = apm_table_hash<KL>(fi);
        const uint32_t slot = h >> hshift;
        const uint32_t tag = h & 0xffffu;
        const uint32_t rep = tag | (tag << 16);
        const uint4 tg = s_tab[slot];
        const u16x2 m01 = __builtin_elementwise_min(apm_as_u16x2(tg.x ^ rep), apm_as_u16x2(tg.y ^ rep));
        const u16x2 m23 = __builtin_elementwise_min(apm_as_u16x2(tg.z ^ rep), apm_as_u16x2(tg.w ^ rep));
        const u16x2 mm = __builtin_elementwise_min(m01, m23);
        bool hit = (mm.x == 0) | (mm.y == 0);
        for (int o = 0; o < a.n_ovf; ++o) hit |= (s_ovf[2 * o] == tag);
        hit &= valid;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
        if (mask) { // rare with long keys
            const uint32_t idx = qcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (hit) s_queue[idx] = make_uint2((uint32_t)pos, (uint32_t)(pos >> 32));
            qcount += (uint32_t)__builtin_popcountll(mask);
        }
    };
    auto process = [&](const u32x4 &v, int64_t cc) __attribute__((always_inline)) {
        const int64_t pos = a.tile0 + cc * 1024 + 16 * lane;
        const bool valid = cc < nch && !APM_SKIP(a, 1);
        if constexpr (STRIDE == 16) {
            probe(apm_fp16(apm_fp8(v.x, v.y), apm_fp8(v.z, v.w)), pos, valid);
        } else if constexpr (STRIDE == 8) {
            probe(apm_fp8(v.x, v.y), pos, valid);
            probe(apm_fp8(v.z, v.w), pos + 8, valid);
        }
    };

    // four chunks in flight per lane.  Per-position classes: the four bitmap passes only produce hit
    // masks and ONE runtime loop queues them, so the kernel holds a single copy of the verification code.
    int64_t c = (int64_t)blockIdx.x * (APM_BLOCK / 64) + wv;
    if constexpr (STRIDE > 1) {
        // the four chunks a wave has in flight are neighbours (4 KiB contiguous per wave, 16 KiB per workgroup;
        // measured against chunks W apart: same on the HBM-bound cfg2, 7 % faster on cfg4)
        constexpr int64_t CS = 1;
        c *= 4;
        u32x4 r0, r1, r2, r3;
        uint2 e0 = make_uint2(0, 0), e1 = e0, e2 = e0, e3 = e0;
        load_chunk(c, r0, e0);
        load_chunk(c + CS, r1, e1);
        load_chunk(c + 2 * CS, r2, e2);
        load_chunk(c + 3 * CS, r3, e3);
        for (; c < nch; c += 4 * W) {
            { const u32x4 v = r0; load_chunk(c + 4 * W, r0, e0); process(v, c); }
            { const u32x4 v = r1; load_chunk(c + 4 * W + CS, r1, e1); process(v, c + CS); }
            if constexpr (STRIDE == 8) drain(); // four positions queued: keeps the wave queue (LDS) small
            { const u32x4 v = r2; load_chunk(c + 4 * W + 2 * CS, r2, e2); process(v, c + 2 * CS); }
            { const u32x4 v = r3; load_chunk(c + 4 * W + 3 * CS, r3, e3); process(v, c + 3 * CS); }
            drain(); // few verification sites: keeps the loop small and its uniform state in SGPRs
        }
    } else {
        u32x4 r0, r1, r2, r3;
        uint2 e0 = make_uint2(0, 0), e1 = e0, e2 = e0, e3 = e0;
        c *= 4; // (four neighbouring chunks per wave, as above)
        load_chunk(c, r0, e0);
        load_chunk(c + 1, r1, e1);
        load_chunk(c + 2, r2, e2);
        load_chunk(c + 3, r3, e3);
        for (; c < nch; c += 4 * W) {
            uint32_t h0, h1, h2, h3;
            { const u32x4 v = r0; const uint2 e = e0; load_chunk(c + 4 * W, r0, e0); h0 = hit_bits(v, e, c); }
            { const u32x4 v = r1; const uint2 e = e1; load_chunk(c + 4 * W + 1, r1, e1); h1 = hit_bits(v, e, c + 1); }
            { const u32x4 v = r2; const uint2 e = e2; load_chunk(c + 4 * W + 2, r2, e2); h2 = hit_bits(v, e, c + 2); }
            { const u32x4 v = r3; const uint2 e = e3; load_chunk(c + 4 * W + 3, r3, e3); h3 = hit_bits(v, e, c + 3); }
            if (__builtin_amdgcn_ballot_w64((h0 | h1 | h2 | h3) != 0)) {
#pragma unroll 1
                for (int j = 0; j < 4; ++j) push_hits(j == 0 ? h0 : (j == 1 ? h1 : (j == 2 ? h2 : h3)), c + j);
            }
        }
    }
    flush(qcount); // the last partial queue (same single verification site)

    __syncthreads();
    for (int i = tid; i < a.n_pats; i += APM_BLOCK) {
        const uint32_t cnt = s_cnt[i];
        if (cnt) atomicAdd(&a.counts[a.pats[i].index], (unsigned long long)cnt);
    }
}

static size_t apm_stream_lds_bytes(const ApmFilterArgs &a) {
    const size_t qw = (size_t)(64 * 4 + 64); // = QW of the kernel
    size_t b = (size_t)a.image_len + 4 * qw * 8 + (size_t)((a.n_pats + 3) & ~3) * 4 + 16;
    return b < 4608 ? 4608 : b; // the tail workgroups need 256 uint4 + 128 bytes
}

template <int BAND>
static const void *apm_stream_fn_kl(int kl, int stride) {
    if (kl == 16 && stride == 16) return (const void *)apm_stream_kernel<BAND, 16, 16>;
    if (kl == 8 && stride == 8) return (const void *)apm_stream_kernel<BAND, 8, 8>;
    // Per-position classes: the host picks this form only for launches whose keys are expected to hit
    // rarely (few keys per 4^key_len codes).  With frequent candidates the LDS-tile kernel wins clearly
    // (verification against global text: cfg5 20 ms vs 3.6 ms per GiB on MI355X).
    // Wider bands (k >= 4) would spill there and stay on the tile kernel.
    if constexpr (BAND <= 1) {
        if (kl == 8 && stride == 1) return (const void *)apm_stream_kernel<BAND, 8, 1>;
        if (kl == 6 && stride == 1) return (const void *)apm_stream_kernel<BAND, 6, 1>;
        if (kl == 4 && stride == 1) return (const void *)apm_stream_kernel<BAND, 4, 1>;
    }
    return nullptr;
}
static const void *apm_stream_fn(int band, int kl, int stride) {
    switch (band) {
    case 0: return apm_stream_fn_kl<0>(kl, stride);
    case 1: return apm_stream_fn_kl<1>(kl, stride);
    case 2: return apm_stream_fn_kl<2>(kl, stride);
    case 3: return apm_stream_fn_kl<3>(kl, stride);
    default: return nullptr;
    }
}

int apm_stream_blocks_per_cu(const ApmFilterArgs &a) {
    int per_cu = 0;
    const void *fn = apm_stream_fn(a.band, a.key_len, a.stride);
    if (!fn || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, APM_BLOCK, apm_stream_lds_bytes(a)) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 2;
    }
    return per_cu > 8 ? 8 : per_cu;
}

// a.tile0 = first scanned relative position (multiple of 16), a.ntiles = number of 1 KiB chunks
hipError_t apm_launch_stream(const ApmFilterArgs &a, int max_blocks, hipStream_t s) {
    if (a.ntiles <= 0 || a.n_pats <= 0) return hipSuccess;
    const void *fn = apm_stream_fn(a.band, a.key_len, a.stride);
    if (!fn) return hipErrorInvalidValue;
    const int64_t want = (a.ntiles + 3) / 4;
    const int64_t cap = max_blocks < 1 ? 1 : max_blocks;
    const int64_t nb = want < cap ? want : cap;
    ApmFilterArgs args = a;
    args.n_main_blocks = (int)nb;
#ifdef APM_MEASURE
    if (const char *e = getenv("APM_MEASURE_SKIP")) args.skip_mask = atoi(e);
#endif
    void *kargs[] = {&args};
    return hipLaunchKernel(fn, dim3((unsigned)(nb + a.n_tail)), dim3(APM_BLOCK), kargs, apm_stream_lds_bytes(a), s);
}

size_t apm_filter_lds_bytes(const ApmFilterArgs &a) { // always >= 4352 B, which the tail workgroups need
    return (size_t)(a.use_dma ? (a.stride == 1 ? 4 : 3) : 2) * (size_t)a.tile_len + (size_t)a.image_len + 2 * (size_t)a.qcap * 4 +
           (size_t)((a.n_pats + 3) & ~3) * 4 + 32 + 4 * 128 * 4 + 16; // counters [8] + 4 survivor lists (SCAP = 128)
}

template <int BAND, int DMA>
static const void *apm_filter_fn_kl(int kl, int stride) {
    if (kl == 16 && stride == 16) return (const void *)apm_filter_kernel<BAND, 16, 16, DMA>;
    if (kl == 8 && stride == 8) return (const void *)apm_filter_kernel<BAND, 8, 8, DMA>;
    if (kl == 8 && stride == 1) return (const void *)apm_filter_kernel<BAND, 8, 1, DMA>;
    if (kl == 6 && stride == 1) return (const void *)apm_filter_kernel<BAND, 6, 1, DMA>;
    if (kl == 4 && stride == 1) return (const void *)apm_filter_kernel<BAND, 4, 1, DMA>;
    return nullptr;
}

static const void *apm_filter_fn(int band, int kl, int stride, int dma) {
    switch (band * 2 + (dma ? 1 : 0)) {
    case 0: return apm_filter_fn_kl<0, 0>(kl, stride);
    case 1: return apm_filter_fn_kl<0, 1>(kl, stride);
    case 2: return apm_filter_fn_kl<1, 0>(kl, stride);
    case 3: return apm_filter_fn_kl<1, 1>(kl, stride);
    case 4: return apm_filter_fn_kl<2, 0>(kl, stride);
    case 5: return apm_filter_fn_kl<2, 1>(kl, stride);
    case 6: return apm_filter_fn_kl<3, 0>(kl, stride);
    case 7: return apm_filter_fn_kl<3, 1>(kl, stride);
    default: return nullptr;
    }
}

// workgroups of the filter kernel resident per CU for this LDS budget (queried once per plan)
int apm_filter_blocks_per_cu(int band, int kl, int stride, int dma, size_t lds) {
    int per_cu = 0;
    const void *fn = apm_filter_fn(band, kl, stride, dma);
    if (!fn || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, APM_BLOCK, lds) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 2;
    }
    return per_cu > 8 ? 8 : per_cu;
}

hipError_t apm_launch_filter(const ApmFilterArgs &a, int max_blocks, hipStream_t s) {
    if (a.ntiles <= 0 || a.n_pats <= 0) return hipSuccess;
    if (a.ntiles > ((int64_t)1 << 30)) return hipErrorInvalidValue; // 32-bit tile indices in the kernel
    const void *fn = apm_filter_fn(a.band, a.key_len, a.stride, a.use_dma);
    if (!fn) return hipErrorInvalidValue;
    const size_t lds = apm_filter_lds_bytes(a);
    const int64_t cap = max_blocks < 1 ? 1 : max_blocks; // persistent grid = resident workgroups
    const int64_t nb = a.ntiles < cap ? a.ntiles : cap;
    ApmFilterArgs args = a;
    args.n_main_blocks = (int)nb;
#ifdef APM_MEASURE
    if (const char *e = getenv("APM_MEASURE_SKIP")) args.skip_mask = atoi(e);
#endif
    void *kargs[] = {&args};
    return hipLaunchKernel(fn, dim3((unsigned)(nb + a.n_tail)), dim3(APM_BLOCK), kargs, lds, s);
}

