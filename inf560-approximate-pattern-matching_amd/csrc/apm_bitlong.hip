/*
 * apm_bitlong.hip -- the bit-vector (BITPAR) kernels for LONG patterns: 513 .. 4096 bytes.  The reference takes any
 * strlen (/root/reference/src/sequential.c:106-112); round 2 sent everything beyond 512 bytes to the one-column GENERIC
 * kernel (a DP column in global memory: 1.5e6 ms per GiB for two patterns of 600 and 700 bytes).  Same arithmetic as the
 * short forms -- the Myers / Hyyro column of apm_core.h, cell recurrence of /root/reference/src/utils.c:84-97 --
 * in two layouts:
 *
 *   apm_bitpar_xwide_kernel   513 .. 1024 bytes: still one window per LANE, columns of 24 or 32 words in registers; the
 *                             column step runs word by word in ONE pass (carry, shifted-in bits and the Eq words, four at
 *                             a time out of LDS, are carried along), so it needs 2 W + a few registers where the
 *                             three-pass form of apm_core.h needs 5 W.
 *   apm_bitlong_kernel        1025 .. 4096 bytes: one window per WAVE.  Lane l holds rows 32 l .. 32 l + 31 (64 l .. for
 *                             the two-word form) of the column; the carry of the column's one addition crosses the lanes
 *                             through two ballots and a 64-bit scalar add -- with g = "lane generates a carry", p = "lane
 *                             propagates one", the carries into the lanes are the carry bits of (g | p) + g --, the
 *                             shifted-in delta bits through one DPP move each.  ~35 instructions advance 2048 (4096)
 *                             cells.  It evaluates the truncated tail windows too (per-window size), so GENERIC is left
 *                             with patterns beyond 4096 bytes only.
 */
#include "apm_internal.h"
#include "apm_core.h"
#include "apm_device.h"

// ---------------------------------------------------------------------------
// one pass over the W words of a column: pv / mv in registers, the text code's Eq row in LDS (16-byte aligned)
// ---------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void bp_step_seq(uint32_t (&pv)[W], uint32_t (&mv)[W], const uint32_t *row) {
    uint32_t carry = 0, pin = 1u, min_ = 0u; // the horizontal delta entering row 1 is +1 (apm_core.h)
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
        const uint4 v = reinterpret_cast<const uint4 *>(row)[q];
        const uint32_t e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int z = 0; z < 4; ++z) {
            const int w = 4 * q + z;
            const uint32_t eq = e4[z], p = pv[w], mm = mv[w];
            const uint32_t t = eq & p;
            const uint64_t s = (uint64_t)t + p + carry;
            carry = (uint32_t)(s >> 32);
            const uint32_t xh = (((uint32_t)s) ^ p) | eq;
            const uint32_t ph = mm | ~(xh | p), mh = p & xh;
            const uint32_t phs = (ph << 1) | pin, mhs = (mh << 1) | min_;
            pin = ph >> 31;
            min_ = mh >> 31;
            const uint32_t xv = eq | mm;
            pv[w] = mhs | ~(xv | phs);
            mv[w] = phs & xv;
        }
    }
}

template <int W>
__device__ __forceinline__ uint32_t bpx_scan(const uint8_t *s_tile, const uint32_t *tab, int m, int k, int64_t base, int64_t jb,
                                             int64_t je_p, int tile, int tid, const ApmPosSink &ps) {
    uint32_t cnt = 0;
    for (int it = 0; it < tile; it += APM_BLOCK) {
        const int joff = it + tid;
        const int64_t j = base + joff;
        uint32_t pv[W], mv[W];
        bp_init<W>(pv, mv);
        for (int x = 0; x < m; ++x) bp_step_seq<W>(pv, mv, tab + (uint32_t)s_tile[joff + x] * W);
        const bool hit = j >= jb && j < je_p && bp_distance<W>(pv, mv, m, m) <= k;
        cnt += apm_wave_count(hit);
        if (ps.out && hit) apm_push_pos(ps, j);
    }
    return cnt;
}

// the frame of apm_bitpar_kernel (apm_bitpar.h): text tile remapped to codes in LDS, per-pattern Eq tables, counts
__global__ __launch_bounds__(APM_BLOCK) void apm_bitpar_xwide_kernel(ApmScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int tile_bytes = (a.tile + a.halo + APM_TILE_SLACK + 15) & ~15;
    uint8_t *s_tile = smem;
    uint8_t *s_lut = smem + tile_bytes;
    uint32_t *s_tab = reinterpret_cast<uint32_t *>(s_lut + 256);
    uint32_t *s_cnt = s_tab + ((a.table_words + 3) & ~3);
    const int64_t base = a.tile0 + (int64_t)blockIdx.x * a.tile;

    s_lut[tid] = a.lut[tid];
    for (int i = tid; i < a.table_words; i += APM_BLOCK) s_tab[i] = a.tables[i];
    for (int i = tid; i < a.n_pats; i += APM_BLOCK) s_cnt[i] = 0u;
    __syncthreads();
    const int nload = (a.tile + a.halo + 31) & ~15;
    for (int i = tid * 16; i < nload; i += APM_BLOCK * 16) {
        const uint4 v = apm_load16_guarded(a.text, base + i, a.avail);
        const uint32_t in[4] = {v.x, v.y, v.z, v.w};
        uint32_t out[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            out[q] = (uint32_t)s_lut[in[q] & 0xffu] | ((uint32_t)s_lut[(in[q] >> 8) & 0xffu] << 8) |
                     ((uint32_t)s_lut[(in[q] >> 16) & 0xffu] << 16) | ((uint32_t)s_lut[in[q] >> 24] << 24);
        *reinterpret_cast<uint4 *>(s_tile + i) = make_uint4(out[0], out[1], out[2], out[3]);
    }
    __syncthreads();
    for (int p = 0; p < a.n_pats; ++p) {
        const ApmPatDesc d = a.pats[p];
        const int m = (int)d.m;
        const int64_t je_p = min(a.je, a.nrel - m + 1);
        const uint32_t *tab = s_tab + d.aux_off;
        uint32_t cnt;
        if (d.w == 24) cnt = bpx_scan<24>(s_tile, tab, m, a.k, base, a.jb, je_p, a.tile, tid, a.pos);
        else cnt = bpx_scan<32>(s_tile, tab, m, a.k, base, a.jb, je_p, a.tile, tid, a.pos);
        if (lane == 0 && cnt) atomicAdd(&s_cnt[p], cnt);
    }
    __syncthreads();
    for (int i = tid; i < a.n_pats; i += APM_BLOCK) {
        const uint32_t c = s_cnt[i];
        if (c) atomicAdd(&a.counts[a.pats[i].index], (unsigned long long)c);
    }
}

hipError_t apm_launch_bitpar_xwide(const ApmScanArgs &a, unsigned n_tiles, size_t lds_bytes, hipStream_t s) {
    hipLaunchKernelGGL(apm_bitpar_xwide_kernel, dim3(n_tiles), dim3(APM_BLOCK), lds_bytes, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// one window per wave: T = uint32_t (m <= 2048) or uint64_t (m <= 4096) rows per lane
// ---------------------------------------------------------------------------
#define APM_LONG_TILE 64 /* window starts per workgroup (four waves, sixteen windows each) */

template <typename T>
__device__ __forceinline__ int apm_popc(T v) {
    if constexpr (sizeof(T) == 8) return __popcll((unsigned long long)v);
    else return __popc((unsigned int)v);
}

template <typename T>
__global__ __launch_bounds__(APM_BLOCK) void apm_bitlong_kernel(ApmScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int BITS = 8 * (int)sizeof(T), WT = 64 * (int)sizeof(T) / 4; // rows per lane; words per Eq row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const ApmPatDesc d = a.pats[0]; // one pattern per launch
    const int m = (int)d.m;
    const int tile_bytes = (APM_LONG_TILE + m + 15 + 16) & ~15;
    uint8_t *s_tile = smem;
    uint8_t *s_lut = smem + tile_bytes;
    const T *s_tab = reinterpret_cast<const T *>(s_lut + 256);
    const int64_t base = a.tile0 + (int64_t)blockIdx.x * APM_LONG_TILE;

    s_lut[tid] = a.lut[tid];
    for (int i = tid; i < a.table_words; i += APM_BLOCK) reinterpret_cast<uint32_t *>(s_lut + 256)[i] = a.tables[i];
    __syncthreads();
    const int nload = (APM_LONG_TILE + m + 15) & ~15;
    for (int i = tid * 16; i < nload; i += APM_BLOCK * 16) {
        const uint4 v = apm_load16_guarded(a.text, base + i, a.avail);
        const uint32_t in[4] = {v.x, v.y, v.z, v.w};
        uint32_t out[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            out[q] = (uint32_t)s_lut[in[q] & 0xffu] | ((uint32_t)s_lut[(in[q] >> 8) & 0xffu] << 8) |
                     ((uint32_t)s_lut[(in[q] >> 16) & 0xffu] << 16) | ((uint32_t)s_lut[in[q] >> 24] << 24);
        *reinterpret_cast<uint4 *>(s_tile + i) = make_uint4(out[0], out[1], out[2], out[3]);
    }
    __syncthreads();

    uint32_t cnt = 0; // (lane 0)
    for (int it = wv; it < APM_LONG_TILE; it += APM_BLOCK / 64) {
        const int64_t j = base + it;
        if (j < a.jb || j >= a.je) continue; // (wave-uniform)
        const int64_t rest = a.nrel - j;
        const int size = rest < (int64_t)m ? (int)rest : m; // the reference cuts pattern AND window at the end of the text (sequential.c:131-134)
        T pv = ~(T)0, mv = 0;
        for (int x = 0; x < size; ++x) {
            const uint32_t c = s_tile[it + x]; // (one address for the wave)
            const T eq = s_tab[c * 64u + (uint32_t)lane];
            const T xv = eq | mv, t = eq & pv;
            const T s0 = t + pv;
            const unsigned long long G = __builtin_amdgcn_ballot_w64(s0 < t), P = __builtin_amdgcn_ballot_w64(s0 == ~(T)0);
            const unsigned long long U = G | P, C = (U + G) ^ U ^ G; // bit l = carry into lane l
            const T sum = s0 + (T)((C >> lane) & 1ull);
            const T xh = (sum ^ pv) | eq;
            const T ph = mv | ~(xh | pv), mh = pv & xh;
            // the delta bits that leave a lane at the top enter the next one at the bottom; lane 0: row 0's horizontal delta, +1
            const uint32_t pin = (uint32_t)__builtin_amdgcn_update_dpp(1, (int)(uint32_t)(ph >> (BITS - 1)), 0x138, 0xf, 0xf, false);
            const uint32_t min_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(mh >> (BITS - 1)), 0x138, 0xf, 0xf, false);
            const T phs = (ph << 1) | (T)pin, mhs = (mh << 1) | (T)min_;
            pv = mhs | ~(xv | phs);
            mv = phs & xv;
        }
        // cell(size, size) = size + sum over the rows < size of the vertical deltas
        const int r0 = lane * BITS;
        T mask = 0;
        if (size >= r0 + BITS) mask = ~(T)0;
        else if (size > r0) mask = (((T)1) << (size - r0)) - 1;
        int dsum = apm_popc<T>(pv & mask) - apm_popc<T>(mv & mask);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) dsum += __shfl_xor(dsum, o, 64);
        const bool hit = size + dsum <= a.k;
        if (lane == 0 && hit) {
            ++cnt;
            if (a.pos.out) apm_push_pos(a.pos, j);
        }
    }
    if (lane == 0 && cnt) atomicAdd(&a.counts[d.index], (unsigned long long)cnt);
    (void)WT;
}

size_t apm_bitlong_lds_bytes(const ApmScanArgs &a, int m) {
    return (size_t)((APM_LONG_TILE + m + 15 + 16) & ~15) + 256 + (size_t)((a.table_words + 3) & ~3) * 4 + 16;
}

// a.pats: ONE pattern (m <= 4096); a.tables: its Eq rows, n_codes x 64 (m <= 2048) or x 128 words; windows [jb, je) -- full and
// truncated ones alike
hipError_t apm_launch_bitlong(const ApmScanArgs &a, int m, hipStream_t s) {
    const int64_t span = a.je - a.tile0;
    if (span <= 0 || a.n_pats != 1) return hipSuccess;
    const int64_t nt = (span + APM_LONG_TILE - 1) / APM_LONG_TILE;
    if (nt > 0x7fffffffLL || m > 4096) return hipErrorInvalidValue;
    const size_t lds = apm_bitlong_lds_bytes(a, m);
    if (m <= 2048) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)apm_bitlong_kernel<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(apm_bitlong_kernel<uint32_t>, dim3((unsigned)nt), dim3(APM_BLOCK), lds, s, a);
    } else {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)apm_bitlong_kernel<uint64_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(apm_bitlong_kernel<uint64_t>, dim3((unsigned)nt), dim3(APM_BLOCK), lds, s, a);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// truncated tail windows of the 513 .. 1024-byte patterns: 32-word columns, one window per thread of a 1024-thread
// workgroup (the form of apm_tail_wide_kernel, apm_bitpar_wide.hip)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void apm_tail_xwide_kernel(ApmTailArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_eq[256 * 32];
    __shared__ uint8_t s_txt[1024 + 16]; // the last <= 1024 text bytes
    constexpr int W = 32, NT = 1024;
    const int tid = (int)threadIdx.x;
    const ApmPatDesc d = a.pats[blockIdx.x];
    const int m = (int)d.m;
    const uint8_t *pat = a.bytes + d.byte_off;
    for (int i = tid; i < 256 * W; i += NT) s_eq[i] = 0u;
    const int64_t first_trunc = max(a.jb, a.nrel - m + 1);
    const int64_t t0 = a.nrel - NT > 0 ? a.nrel - NT : 0;
    s_txt[tid] = (t0 + tid < a.nrel) ? a.text[t0 + tid] : (uint8_t)0;
    __syncthreads();
    if (tid < m) atomicOr(&s_eq[(int)pat[tid] * W + (tid >> 5)], 1u << (tid & 31));
    __syncthreads();
    const int64_t j = first_trunc + tid;
    const bool valid = j < a.je;
    const int size = valid ? (int)(a.nrel - j) : 0; // 1 .. m-1
    const int lo = valid ? (int)(j - t0) : 0;
    uint32_t pv[W], mv[W];
    bp_init<W>(pv, mv);
    for (int x = 0; x < m - 1; ++x)
        if (x < size) bp_step_seq<W>(pv, mv, s_eq + (int)s_txt[lo + x] * W);
    const bool hit = valid && bp_distance<W>(pv, mv, size, size) <= a.k;
    if (a.pos.out && hit) apm_push_pos(a.pos, j);
    const uint32_t cnt = apm_wave_count(hit);
    if ((tid & 63) == 0 && cnt) atomicAdd(&a.counts[d.index], (unsigned long long)cnt);
}

hipError_t apm_launch_tail_xwide(const ApmTailArgs &a, int n_pats, hipStream_t s) {
    if (n_pats <= 0 || a.je <= a.jb) return hipSuccess;
    hipLaunchKernelGGL(apm_tail_xwide_kernel, dim3((unsigned)n_pats), dim3(1024), 0, s, a);
    return hipGetLastError();
}
