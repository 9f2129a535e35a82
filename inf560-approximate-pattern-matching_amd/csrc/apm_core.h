/*
 * apm_core.h -- arithmetic cores shared by the HIP kernels and by the host-side
 * unit test (tests/host_core_test.cpp compiles this with g++, no GPU needed).
 *
 * Everything here restates the cell recurrence of
 *   levenshtein()  /root/reference/src/utils.c:84-97
 *     cell(x,y) = min(cell(x-1,y)+1, cell(x,y-1)+1, cell(x-1,y-1)+(p[y-1]!=t[x-1]))
 *     cell(0,y) = y, cell(x,0) = x, answer cell(len,len)
 * in forms that map well onto CDNA4 integer VALU.
 */
#ifndef APM_CORE_H
#define APM_CORE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define APM_HD __host__ __device__ __forceinline__
#else
#define APM_HD inline
#endif

/* ---------------------------------------------------------------------------
 * Bit-vector column (Myers 1999 / Hyyro 2003, global-distance boundary).
 *
 * One DP column cell(x, 1..m) is held as vertical deltas
 *   pv bit (y-1) = 1  <=>  cell(x,y) - cell(x,y-1) = +1
 *   mv bit (y-1) = 1  <=>  cell(x,y) - cell(x,y-1) = -1
 * in W 32-bit words (m <= 32*W).  Column 0 is cell(0,y)=y: pv = all ones.
 * Row 0 is cell(x,0)=x, i.e. the horizontal delta entering row 1 is always +1
 * (the "| 1" below); that is the only difference to the text-search form.
 * Advancing one text byte costs ~13 integer ops per word and updates 32 cells.
 * Bits >= m never influence bits < m (all carries/shifts move upward), so no
 * masking is needed until the end:  cell(x,m) = x + popc(pv&mask) - popc(mv&mask).
 * ------------------------------------------------------------------------- */
template <int W>
APM_HD void bp_init(uint32_t (&pv)[W], uint32_t (&mv)[W]) {
#pragma unroll
    for (int w = 0; w < W; ++w) {
        pv[w] = 0xffffffffu;
        mv[w] = 0u;
    }
}

template <int W>
APM_HD void bp_step(uint32_t (&pv)[W], uint32_t (&mv)[W], const uint32_t (&eq)[W]) {
    uint32_t xh[W], ph[W], mh[W];
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const uint32_t t = eq[w] & pv[w];
        const uint64_t s = (uint64_t)t + pv[w] + carry; /* v_add_co / v_addc_co */
        carry = (uint32_t)(s >> 32);
        xh[w] = (((uint32_t)s) ^ pv[w]) | eq[w];
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
        ph[w] = mv[w] | ~(xh[w] | pv[w]);
        mh[w] = pv[w] & xh[w];
    }
    uint32_t pin = 1u, min_ = 0u; /* horizontal delta at row 0 is +1 */
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const uint32_t phs = (ph[w] << 1) | pin;
        const uint32_t mhs = (mh[w] << 1) | min_;
        pin = ph[w] >> 31;
        min_ = mh[w] >> 31;
        const uint32_t xv = eq[w] | mv[w];
        pv[w] = mhs | ~(xv | phs);
        mv[w] = phs & xv;
    }
}

/* cell(x, m) after x steps */
template <int W>
APM_HD int bp_distance(const uint32_t (&pv)[W], const uint32_t (&mv)[W], int m, int x) {
    int d = x;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int lo = 32 * w;
        uint32_t mask;
        if (m >= lo + 32) mask = 0xffffffffu;
        else if (m <= lo) mask = 0u;
        else mask = (1u << (m - lo)) - 1u;
#if defined(__HIP_DEVICE_COMPILE__)
        d += __popc(pv[w] & mask) - __popc(mv[w] & mask);
#else
        d += __builtin_popcount(pv[w] & mask) - __builtin_popcount(mv[w] & mask);
#endif
    }
    return d;
}

/* ---------------------------------------------------------------------------
 * splitmix64 counter-based DNA generator:
 *   byte i = "ACGT"[(splitmix64(seed ^ (i >> 5)) >> (2*(i & 31))) & 3]
 * ------------------------------------------------------------------------- */
APM_HD uint64_t apm_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

APM_HD uint8_t apm_synth_byte(uint64_t i, uint64_t seed) {
    const uint64_t r = apm_splitmix64(seed ^ (i >> 5));
    const uint32_t c = (uint32_t)(r >> (2 * (i & 31))) & 3u;
    return (uint8_t)((0x54474341u >> (8 * c)) & 0xffu); /* "ACGT" little-endian */
}

/* ---------------------------------------------------------------------------
 * One-edit extension (pair pre-check of the BANDED path): does the pattern piece P[0..n) match the text
 * read from T[0] with at most ONE edit, all of P consumed, the far end of the text free (n-1, n or n+1
 * text bytes used)?  1 <= n <= 16, on 128-bit values held as two 64-bit halves: p = 16 pattern bytes
 * (byte i in bits 8i..), t = 20 text bytes.  The edit is located by the first mismatching byte i; the
 * three ways to spend it are checked with masked compares against T, T<<8 and T>>8.
 * Definition = the byte loops apm_ext_fwd in apm_kernels.hip (and ext1_loop in tests/host_core_test.cpp).
 * ------------------------------------------------------------------------- */
APM_HD int apm_ctz64(unsigned long long v) { return __builtin_ctzll(v); }
APM_HD bool apm_ext1_core16(const uint32_t (&p)[4], const uint32_t (&t)[5], int n) {
    typedef unsigned long long u64;
    const u64 Pl = ((u64)p[1] << 32) | p[0], Ph = ((u64)p[3] << 32) | p[2];
    const u64 Tl = ((u64)t[1] << 32) | t[0], Th = ((u64)t[3] << 32) | t[2];
    const u64 nl = n >= 8 ? ~0ull : ((1ull << (8 * n)) - 1ull);                          // bytes 0..min(n,8)-1
    const u64 nh = n <= 8 ? 0ull : (n >= 16 ? ~0ull : ((1ull << (8 * (n - 8))) - 1ull)); // bytes 8..n-1
    const u64 x0l = (Pl ^ Tl) & nl, x0h = (Ph ^ Th) & nh;
    if ((x0l | x0h) == 0ull) return true;
    const int i = x0l ? (apm_ctz64(x0l) >> 3) : 8 + (apm_ctz64(x0h) >> 3); // first mismatching byte
    if (i >= n - 1) return true;
    // masks of the bytes above i (for the substitution / missing-text-byte cases) and from i on (extra text byte)
    const u64 gl = i + 1 < 8 ? (~0ull << (8 * (i + 1))) : 0ull, gh = i + 1 < 8 ? ~0ull : (~0ull << (8 * (i + 1 - 8)));
    const u64 el = i < 8 ? (~0ull << (8 * i)) : 0ull, eh = i < 8 ? ~0ull : (~0ull << (8 * (i - 8)));
    if (((x0l & gl) | (x0h & gh)) == 0ull) return true; // substitution at i
    const u64 Tdl = Tl << 8, Tdh = (Th << 8) | (Tl >> 56); // text shifted up by one byte
    if ((((Pl ^ Tdl) & nl & gl) | ((Ph ^ Tdh) & nh & gh)) == 0ull) return true; // pattern byte i has no text counterpart
    const u64 Tul = (Tl >> 8) | (Th << 56), Tuh = (Th >> 8) | ((u64)t[4] << 56); // text shifted down by one byte
    return (((Pl ^ Tul) & nl & el) | ((Ph ^ Tuh) & nh & eh)) == 0ull; // one extra text byte before pattern byte i
}

#endif /* APM_CORE_H */
