/*
 * apm_device.h -- device helpers shared by the kernel translation units (apm_kernels.hip, apm_sieve.hip).
 */
#ifndef APM_DEVICE_H
#define APM_DEVICE_H

#include "apm_internal.h"
#include "apm_core.h"

__device__ __forceinline__ uint32_t apm_wave_count(bool pred) {
    return (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(pred));
}

__device__ __forceinline__ int apm_min3(int a, int b, int c) { return min(min(a, b), c); }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 apm_as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }

// Stage n16 x 16 bytes of a launch image from global memory into LDS with FOUR loads in flight per thread: the plain
// copy loop compiles to load, wait, store, and a 17..42 KiB image then costs a workgroup four to six memory round trips
// one after the other -- most of an otherwise empty launch's 17 microseconds (round 3 measurement).
__device__ __forceinline__ void apm_stage_image(uint4 *dst, const uint4 *src, int n16, int tid, int threads) {
    for (int i = tid; i < n16; i += 4 * threads) {
        const int i1 = i + threads, i2 = i + 2 * threads, i3 = i + 3 * threads;
        const uint4 v0 = src[i];
        const uint4 v1 = src[i1 < n16 ? i1 : i]; // (in range whatever n16 is; stored only when its slot exists)
        const uint4 v2 = src[i2 < n16 ? i2 : i];
        const uint4 v3 = src[i3 < n16 ? i3 : i];
        dst[i] = v0;
        if (i1 < n16) dst[i1] = v1;
        if (i2 < n16) dst[i2] = v2;
        if (i3 < n16) dst[i3] = v3;
    }
}

// N dwords of bytes starting at (16-byte aligned LDS base) + off, any alignment of off:
// N+1 aligned ds_read_b32 + N v_alignbyte -- no dependent byte loads
template <int N>
__device__ __forceinline__ void apm_lds_dwords(const uint8_t *base, int off, uint32_t (&out)[N]) {
    const uint32_t *a = reinterpret_cast<const uint32_t *>(base) + (off >> 2);
    const uint32_t sh = (uint32_t)off & 3u;
    uint32_t w[N + 1];
#pragma unroll
    for (int i = 0; i <= N; ++i) w[i] = a[i];
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);
}

// Banded DP (|x-y| <= BAND) with early exit over a window of m text bytes vs pattern pb[poff..poff+m).
// Columns 1..16 run out of registers (bytes fetched as dwords up front, statically indexed); most
// candidates die there.  Needs m >= 16 for the register phase, otherwise byte loop only.
template <int BAND, typename Text>
__device__ __forceinline__ bool apm_banded_verify(const Text &tx, const uint8_t *pb, int poff, int m, int k) {
    const uint8_t *p = pb + poff;
    if constexpr (BAND == 0 && Text::kBlocks) {
        // Hamming distance, 16 bytes per step: nonzero bytes of text ^ pattern counted with the carry trick
        int mism = 0;
        for (int xb = 0; xb < m; xb += 16) {
            uint32_t Tb[4], Pb[4];
            tx.load16_at(xb, Tb);
            apm_lds_dwords<4>(pb, poff + xb, Pb);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int valid = m - xb - 4 * i; // bytes of this dword inside the window
                const uint32_t mask = valid >= 4 ? 0xffffffffu : (valid <= 0 ? 0u : ((1u << (8 * valid)) - 1u));
                const uint32_t x = (Tb[i] ^ Pb[i]) & mask;
                mism += __builtin_popcount((x | ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u);
            }
            if (mism > k) return false;
        }
        return true;
    } else if constexpr (BAND == 0) {
        int mism = 0;
        for (int x = 0; x < m; ++x) {
            mism += (tx.byte(x) != (int)p[x]) ? 1 : 0;
            if (mism > k) return false;
        }
        return true;
    } else {
        constexpr int NB = 2 * BAND + 1;
        constexpr int INF = 1 << 20;
        int e[NB]; // e[d+BAND] = cell(x, x+d)
#pragma unroll
        for (int i = 0; i < NB; ++i) e[i] = (i >= BAND) ? (i - BAND) : INF; // cell(0, d) = d
        int x0 = 1;
        if (m >= 16 && tx.can16(m)) {
            uint32_t T[4], P[5];
            tx.load16(T);
            apm_lds_dwords<5>(pb, poff, P); // pattern bytes 0..19 (>= 16 + BAND - 1)
#pragma unroll
            for (int x = 1; x <= 16; ++x) {
                const int tc = (int)((T[(x - 1) >> 2] >> (8 * ((x - 1) & 3))) & 0xffu);
                int up = INF, best = INF;
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int y = x + i - BAND; // static
                    int nv;
                    if (y < 1) {
                        nv = (y == 0) ? x : INF;
                    } else if (y > 16 && y > m) { // only reachable for 16 <= m < 16 + BAND
                        nv = INF;
                    } else {
                        const int pc = (int)((P[(y - 1) >> 2] >> (8 * ((y - 1) & 3))) & 0xffu);
                        const int diag = e[i] + ((pc != tc) ? 1 : 0);
                        const int left = (i + 1 < NB) ? e[i + 1] + 1 : INF;
                        nv = apm_min3(diag, left, up + 1);
                    }
                    e[i] = nv;
                    up = nv;
                    best = min(best, nv);
                }
                if ((x & 3) == 0 && best > k) return false;
            }
            x0 = 17;
            if constexpr (Text::kBlocks) {
                // the remaining columns 16 at a time, text and pattern bytes fetched as dwords up front (a byte load
                // per column from global memory would chain their latencies): block b covers columns xb .. xb+15,
                // its pattern window starts at byte xb - 1 - BAND (cell (xb + xi, xb + xi + i - BAND) reads byte xi + i)
                for (int xb = 17; xb <= m; xb += 16) {
                    uint32_t Tb[4], Pb[6];
                    tx.load16_at(xb - 1, Tb);
                    apm_lds_dwords<6>(pb, poff + xb - 1 - BAND, Pb);
#pragma unroll
                    for (int xi = 0; xi < 16; ++xi) {
                        const int x = xb + xi;
                        const int tc = (int)((Tb[xi >> 2] >> (8 * (xi & 3))) & 0xffu);
                        int up = INF, best = INF;
                        if (x <= m) {
#pragma unroll
                            for (int i = 0; i < NB; ++i) {
                                const int y = x + i - BAND; // >= 17 - BAND >= 1
                                int nv = INF;
                                if (y <= m) {
                                    const int pc = (int)((Pb[(xi + i) >> 2] >> (8 * ((xi + i) & 3))) & 0xffu);
                                    const int diag = e[i] + ((pc != tc) ? 1 : 0);
                                    const int left = (i + 1 < NB) ? e[i + 1] + 1 : INF;
                                    nv = apm_min3(diag, left, up + 1);
                                }
                                e[i] = nv;
                                up = nv;
                                best = min(best, nv);
                            }
                            if ((xi & 3) == 3 && best > k) return false;
                        }
                    }
                }
                return e[BAND] <= k;
            }
        }
        for (int x = x0; x <= m; ++x) {
            const int tc = tx.byte(x - 1);
            int up = INF; // cell(x, y-1) of the previous diagonal at this x
            int best = INF;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int y = x + i - BAND;
                int nv;
                if (y < 1) {
                    nv = (y == 0) ? x : INF;
                } else if (y > m) {
                    nv = INF;
                } else {
                    const int diag = e[i] + (((int)p[y - 1] != tc) ? 1 : 0);
                    const int left = (i + 1 < NB) ? e[i + 1] + 1 : INF;
                    nv = apm_min3(diag, left, up + 1);
                }
                e[i] = nv;
                up = nv;
                best = min(best, nv);
            }
            if (best > k) return false;
        }
        return e[BAND] <= k;
    }
}

__device__ __forceinline__ uint32_t apm_bswap(uint32_t v) { return __builtin_bswap32(v); }

// match-position sink (apm_find_buffer): rare, unordered; the host sorts
__device__ __forceinline__ void apm_push_pos(const ApmPosSink &ps, int64_t j_rel) {
    const unsigned long long idx = atomicAdd(ps.count, 1ull);
    if (idx < ps.cap) ps.out[idx] = ps.text_off + (unsigned long long)j_rel;
}


// body shared by apm_tail_kernel and by the extra workgroups of the BANDED launch;
// needs >= 128 threads, uses lanes 0..127; s_eq = 256 uint4 of LDS
__device__ __forceinline__ void apm_tail_body(const ApmTailArgs &a, int pat_slot, uint4 *s_eq, int tid) {
    const ApmPatDesc d = a.pats[pat_slot];
    const int m = (int)d.m;
    const uint8_t *pat = a.bytes + d.byte_off;
    uint32_t *eqw = reinterpret_cast<uint32_t *>(s_eq);
    uint8_t *s_txt = reinterpret_cast<uint8_t *>(s_eq + 256); // last <= 128 text bytes
    for (int i = tid; i < 1024; i += 128) eqw[i] = 0u;
    const int64_t first_trunc = max(a.jb, a.nrel - m + 1);
    const int64_t t0 = a.nrel - 128 > 0 ? a.nrel - 128 : 0; // stage the end of the text once
    if (tid < 128) s_txt[tid] = (t0 + tid < a.nrel) ? a.text[t0 + tid] : (uint8_t)0;
    __syncthreads();
    if (tid < m) atomicOr(&eqw[(int)pat[tid] * 4 + (tid >> 5)], 1u << (tid & 31));
    __syncthreads();
    if (tid < 128) {
        const int64_t j = first_trunc + tid;
        const bool valid = j < a.je;
        const int size = valid ? (int)(a.nrel - j) : 0; // 1 .. m-1 (<= 127)
        const int lo = valid ? (int)(j - t0) : 0;
        uint32_t pv[4], mv[4];
        bp_init<4>(pv, mv);
        for (int x = 0; x < m - 1; ++x) {
            if (x < size) {
                const uint4 v = s_eq[s_txt[lo + x]];
                const uint32_t eq[4] = {v.x, v.y, v.z, v.w};
                bp_step<4>(pv, mv, eq);
            }
        }
        const bool hit = valid && bp_distance<4>(pv, mv, size, size) <= a.k;
        if (a.pos.out && hit) apm_push_pos(a.pos, j);
        const uint32_t cnt = apm_wave_count(hit);
        if ((tid & 63) == 0 && cnt) atomicAdd(&a.counts[d.index], (unsigned long long)cnt);
    }
}


// 16 text bytes at pos, zeros outside [0, avail)
__device__ __forceinline__ uint4 apm_load16_guarded(const uint8_t *text, int64_t pos, int64_t avail) {
    if (pos >= 0 && pos + 16 <= avail) return *reinterpret_cast<const uint4 *>(text + pos);
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const int64_t q = pos + b;
        if (q >= 0 && q < avail) w[b >> 2] |= (uint32_t)text[q] << (8 * (b & 3));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

#endif /* APM_DEVICE_H */
