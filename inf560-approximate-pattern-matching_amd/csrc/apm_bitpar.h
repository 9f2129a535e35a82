/*
 * apm_bitpar.h -- BITPAR device code shared by apm_kernels.hip (columns of 1..4 words) and apm_bitpar_wide.hip (8 and 16
 * words: compiled as a unit of its own, beside the other, because the wide instantiations take minutes to compile).
 */
#ifndef APM_BITPAR_H
#define APM_BITPAR_H

#include "apm_device.h"

// ---------------------------------------------------------------------------
// BITPAR: one window per lane, bit-vector columns (apm_core.h), exact distance.
// The text tile is stored in LDS as CODES (byte -> small alphabet index through
// a 256-entry LUT built from the launch's patterns; code 0 = "occurs in no
// pattern"), so a pattern's Eq table is n_codes * stride words instead of 256.
// ---------------------------------------------------------------------------
template <int W, int STRIDE>
__device__ __forceinline__ void bp_load_eq(const uint32_t *tab, uint32_t c, uint32_t (&eq)[W]) {
    if constexpr (STRIDE == 1) {
        eq[0] = tab[c];
    } else if constexpr (STRIDE == 2) {
        const uint2 v = reinterpret_cast<const uint2 *>(tab)[c];
        eq[0] = v.x;
        eq[1] = v.y;
    } else if constexpr (STRIDE == 4) {
        const uint4 v = reinterpret_cast<const uint4 *>(tab)[c];
        eq[0] = v.x;
        eq[1] = v.y;
        eq[2] = v.z;
        if constexpr (W == 4) eq[3] = v.w;
    } else { // W = STRIDE = 8 or 16 words (patterns of 129 .. 512 bytes)
        static_assert(W == STRIDE && (STRIDE == 8 || STRIDE == 16), "bit-vector widths beyond 4 words come in 8 and 16");
#pragma unroll
        for (int q = 0; q < STRIDE / 4; ++q) {
            const uint4 v = reinterpret_cast<const uint4 *>(tab)[c * (STRIDE / 4) + q];
            eq[4 * q] = v.x;
            eq[4 * q + 1] = v.y;
            eq[4 * q + 2] = v.z;
            eq[4 * q + 3] = v.w;
        }
    }
}

template <int W, int STRIDE>
__device__ __forceinline__ int bp_window(const uint8_t *s_tile, int joff, const uint32_t *tab, int m) {
    uint32_t pv[W], mv[W];
    bp_init<W>(pv, mv);
    const uint32_t *t32 = reinterpret_cast<const uint32_t *>(s_tile + (joff & ~3));
    const uint32_t sh = (uint32_t)joff & 3u;
    uint32_t lo = t32[0];
    int x = 0, q = 1;
    for (; x + 4 <= m; x += 4, ++q) {
        const uint32_t hi = t32[q];
        const uint32_t w4 = __builtin_amdgcn_alignbyte(hi, lo, sh); // codes of t[j+x .. j+x+3]
        lo = hi;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            uint32_t eq[W];
            bp_load_eq<W, STRIDE>(tab, (w4 >> (8 * b)) & 0xffu, eq);
            bp_step<W>(pv, mv, eq);
        }
    }
    if (x < m) {
        uint32_t w4 = __builtin_amdgcn_alignbyte(t32[q], lo, sh);
        for (; x < m; ++x) {
            uint32_t eq[W];
            bp_load_eq<W, STRIDE>(tab, w4 & 0xffu, eq);
            bp_step<W>(pv, mv, eq);
            w4 >>= 8;
        }
    }
    return bp_distance<W>(pv, mv, m, m);
}

template <int W, int STRIDE>
__device__ __forceinline__ uint32_t bp_scan(const uint8_t *s_tile, const uint32_t *tab, int m, int k,
                                            int64_t base, int64_t jb, int64_t je_p, int tile, int tid,
                                            const ApmPosSink &ps) {
    uint32_t cnt = 0;
    for (int it = 0; it < tile; it += APM_BLOCK) {
        const int joff = it + tid;
        const int64_t j = base + joff;
        const int dist = bp_window<W, STRIDE>(s_tile, joff, tab, m);
        const bool hit = j >= jb && j < je_p && dist <= k;
        cnt += apm_wave_count(hit);
        if (ps.out && hit) apm_push_pos(ps, j);
    }
    return cnt;
}

// WIDE: the launch's patterns are all longer than 128 bytes (columns of 8 or 16 words: ~150 VGPRs); their own
// instantiation, so that the short-column launches keep their occupancy
template <bool WIDE>
__global__ __launch_bounds__(APM_BLOCK) void apm_bitpar_kernel(ApmScanArgs a) {
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
        for (int q = 0; q < 4; ++q) {
            out[q] = (uint32_t)s_lut[in[q] & 0xffu] | ((uint32_t)s_lut[(in[q] >> 8) & 0xffu] << 8) |
                     ((uint32_t)s_lut[(in[q] >> 16) & 0xffu] << 16) | ((uint32_t)s_lut[in[q] >> 24] << 24);
        }
        *reinterpret_cast<uint4 *>(s_tile + i) = make_uint4(out[0], out[1], out[2], out[3]);
    }
    __syncthreads();

    for (int p = 0; p < a.n_pats; ++p) {
        const ApmPatDesc d = a.pats[p];
        const int m = (int)d.m;
        const int64_t je_p = min(a.je, a.nrel - m + 1);
        const uint32_t *tab = s_tab + d.aux_off;
        uint32_t cnt;
        if constexpr (WIDE) {
            if (d.w == 8) cnt = bp_scan<8, 8>(s_tile, tab, m, a.k, base, a.jb, je_p, a.tile, tid, a.pos);
            else cnt = bp_scan<16, 16>(s_tile, tab, m, a.k, base, a.jb, je_p, a.tile, tid, a.pos);
        } else {
            switch (d.w) {
            case 1: cnt = bp_scan<1, 1>(s_tile, tab, m, a.k, base, a.jb, je_p, a.tile, tid, a.pos); break;
            case 2: cnt = bp_scan<2, 2>(s_tile, tab, m, a.k, base, a.jb, je_p, a.tile, tid, a.pos); break;
            case 3: cnt = bp_scan<3, 4>(s_tile, tab, m, a.k, base, a.jb, je_p, a.tile, tid, a.pos); break;
            default: cnt = bp_scan<4, 4>(s_tile, tab, m, a.k, base, a.jb, je_p, a.tile, tid, a.pos); break;
            }
        }
        if (lane == 0 && cnt) atomicAdd(&s_cnt[p], cnt);
    }
    __syncthreads();
    for (int i = tid; i < a.n_pats; i += APM_BLOCK) {
        const uint32_t c = s_cnt[i];
        if (c) atomicAdd(&a.counts[a.pats[i].index], (unsigned long long)c);
    }
}


hipError_t apm_launch_bitpar_wide(const ApmScanArgs &a, unsigned n_tiles, size_t lds_bytes, hipStream_t s); /* apm_bitpar_wide.hip */

#endif /* APM_BITPAR_H */
