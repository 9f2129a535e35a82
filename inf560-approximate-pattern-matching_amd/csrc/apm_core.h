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

/* ---------------------------------------------------------------------------
 * The same one-edit extension on strings of 2-bit CODES (code i in bits 2i..2i+1), n <= 15 codes of the partner
 * against 16 codes of text: the sieve's second stage (apm_sieve.hip, "code filter") runs it on the codes it has
 * in hand.  Equal bytes have equal codes, so every byte-level alignment with <= 1 edit is one of the code
 * strings too, and a prefix of such an alignment has <= 1 edit: the result is a superset of what
 * apm_ext1_core16 accepts for the whole partner -- a filter, never a decision.
 * ------------------------------------------------------------------------- */
/* Branch-free: with i = the first mismatching code (bit index i2 = 2 i), g = the bits above code i, e = the bits from code
 * i on, the three ways to spend the edit leave these mismatch bits, and the answer is "one of them is empty":
 *   substitution at i                          x0 & g            (also empty when there is no mismatch at all, and when
 *                                                                 i is the last code: g then lies beyond the n codes)
 *   pattern code i has no text counterpart     (p ^ t << 2) & g
 *   one extra text code before pattern code i  (p ^ t >> 2) & e                                                        */
APM_HD uint32_t apm_ext1_codes_bits(uint32_t p, uint32_t t, int n) {
    const uint32_t maskn = (1u << (2 * n)) - 1u;
    const uint32_t x0 = (p ^ t) & maskn;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t i2 = (uint32_t)__builtin_ctzg(x0, 32) & 30u; /* v_ffbl_b32 + v_and (x0 == 0: any value will do) */
#else
    const uint32_t i2 = x0 ? ((uint32_t)__builtin_ctz(x0) & 30u) : 0u;
#endif
    const uint32_t g = 0xfffffffcu << i2, e = 0xffffffffu << i2;
    const uint32_t sub = x0 & g, del = (p ^ (t << 2)) & maskn & g, ins = (p ^ (t >> 2)) & maskn & e;
    const uint32_t m1 = sub < del ? sub : del;
    return m1 < ins ? m1 : ins; /* v_min3_u32 */
}
APM_HD bool apm_ext1_codes(uint32_t p, uint32_t t, int n) { return apm_ext1_codes_bits(p, t, n) == 0u; }

/* Code-filter record of a nomination unit (two dwords per key, built by apm_cf_record below):
 *   rx = first np <= 15 partner codes, read AWAY from the exact part (side 2: the partner's last byte first) | side << 30
 *   ry = codes of the exact part's bytes 8..15 (16 bits) | np << 16 | exact length << 20 (8 bits)
 * apm_cf_pass: can the unit's nomination predicate (apm_sieve.hip, stage1) hold at a text position, judged by codes
 * alone?  c0 = codes of the 16 text bytes from the position on; tw = codes of the 16 text bytes next to the exact part on
 * the partner's side, read away from it (side 1: from position + exact length on; side 2: the bytes in front of the
 * position, last one first); `visible` false = those bytes are out of the caller's reach, the partner is not judged.
 * Never false when stage1 is true.  (Branch-free: the kernels run it on partly filled waves.) */
APM_HD bool apm_cf_pass(uint32_t rx, uint32_t ry, uint32_t c0, uint32_t tw, bool visible) {
    const uint32_t side = rx >> 30, len = (ry >> 20) & 0xffu; /* (bit 31 of ry: the list flag of the sieve's tables) */
    const int np = (int)((ry >> 16) & 15u);
    const uint32_t el = len < 16u ? len : 16u;
    const uint32_t e2mask = el > 8u ? ((1u << (2u * (el - 8u))) - 1u) : 0u;
    const uint32_t e2bits = ((c0 >> 16) ^ ry) & e2mask; /* bytes 8.. of the exact part */
    const uint32_t pbits = (side != 0u && visible) ? apm_ext1_codes_bits(rx & 0x3fffffffu, tw, np) : 0u;
    return (e2bits | pbits) == 0u;
}
/* 16 codes in reverse order (code i <-> code 15 - i) */
APM_HD uint32_t apm_rev_codes(uint32_t w) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t r = __builtin_bitreverse32(w);
#else
    uint32_t r = 0;
    for (int i = 0; i < 32; ++i) r |= ((w >> i) & 1u) << (31 - i);
#endif
    return ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
}

/* ---------------------------------------------------------------------------
 * The k-error automaton of the window DP over 32 window starts at once (apm_nfa.hip; tests/host_core_test.cpp runs the
 * same two functions against the literal window DP on the host).  R[e][i]: bit b = "for window start j0 + b, cell(x, x + i - B)
 * <= e" after x pattern bytes, B = K / 2 (equal lengths: an alignment with <= K edits stays on the diagonals |y - x| <= B).
 * M[i]: bit b = (text[j0 + b + x + i - B] == pattern[x]).  cell(0, y) = y; cells outside the m x m square are empty; the
 * window matches iff R[K][B] holds after m bytes.
 * ------------------------------------------------------------------------- */
/* The cells that can lie on a path to the accepting cell (K, 0): reaching diagonal d costs |d| errors and so does coming
   back from it -- |d| <= e and e + |d| <= K.  The others are never set (|d| > e) or never used (e + |d| > K); kept at a
   constant 0 they cost nothing: 8 of 12 cells at K = 3, 32 of 56 at K = 7. */
APM_HD constexpr bool apm_nfa_live(int K, int e, int d) { return (d < 0 ? -d : d) <= e && e + (d < 0 ? -d : d) <= K; }
template <int K>
APM_HD void apm_nfa_init(uint32_t (&R)[K + 1][2 * (K / 2) + 1]) {
    constexpr int B = K / 2, ND = 2 * B + 1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int e = 0; e <= K; ++e)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i < ND; ++i) R[e][i] = (i - B >= 0 && i - B <= e && apm_nfa_live(K, e, i - B)) ? 0xffffffffu : 0u;
}
/* (a & b) | c and a | b | c in ONE instruction of the 2-cycle class on gfx950 (v_bitop3_b32; the compiler's own choice,
   v_and_or_b32 / v_or3_b32, issues in 4: tools/valu_probe.hip) */
APM_HD uint32_t apm_and_or(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xEA);
#else
    return (a & b) | c;
#endif
}
APM_HD uint32_t apm_or3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xFE);
#else
    return a | b | c;
#endif
}
/* EDGE: the column has cells outside the square (the first and the last B columns): they are emptied */
template <int K, bool EDGE>
APM_HD void apm_nfa_step(const uint32_t (&Rin)[K + 1][2 * (K / 2) + 1], uint32_t (&Rout)[K + 1][2 * (K / 2) + 1],
                         const uint32_t (&M)[2 * (K / 2) + 1], int x, int m) {
    constexpr int B = K / 2, ND = 2 * B + 1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int e = 0; e <= K; ++e)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i < ND; ++i) {
            uint32_t v;
            if (!apm_nfa_live(K, e, i - B)) {
                Rout[e][i] = 0u;
                continue;
            }
            if (e == 0) v = Rin[e][i] & M[i];                                      /* match on the diagonal */
            else {
                /* the live ones among: substitution, the cell above (pattern byte without a text byte), the cell to the left
                   (text byte without a pattern byte, same column); explicit three-input forms -- left to itself the compiler
                   picks v_and_or_b32, which issues at half the rate of v_bitop3_b32 */
                const bool has_sub = apm_nfa_live(K, e - 1, i - B);
                const bool has_up = i + 1 < ND && apm_nfa_live(K, e - 1, i + 1 - B), has_left = i > 0 && apm_nfa_live(K, e - 1, i - 1 - B);
                uint32_t o[3] = {0u, 0u, 0u};
                int n = 0;
                if (has_sub) o[n++] = Rin[e - 1][i];
                if (has_up) o[n++] = Rin[e - 1][i + 1];
                if (has_left) o[n++] = Rout[e - 1][i - 1];
                v = n == 0 ? (Rin[e][i] & M[i]) : apm_and_or(Rin[e][i], M[i], o[0]);
                if (n == 2) v |= o[1];
                else if (n == 3) v = apm_or3(v, o[1], o[2]);
            }
            if (EDGE) {
                const int y = x + 1 + i - B;                                        /* the cell's text offset: inside [0, m] */
                v = (y >= 0 && y <= m) ? v : 0u;
            }
            Rout[e][i] = v;
        }
}

/* ---------------------------------------------------------------------------
 * Host side of the presence bitmaps (plan builder in apm_runtime.hip; tests/host_core_test.cpp checks the
 * constructive enumeration against the brute-force definition).
 *
 * A nomination UNIT of a pattern is an exact part pat[off, off+len) followed (side 1) or preceded (side 2) by a
 * partner pat[poff, poff+plen) that must match the adjoining text within ONE edit, anchored at the exact part and
 * free at its far end (apm_ext1_core16 / apm_ext_fwd semantics); side 0 = no partner.  len may be 0 with side 1:
 * the whole unit is then "partner within one edit from the unit's text position on" (a pair of short pieces).
 * apm_enum_unit_windows calls fn(x) for every word x of 2-bit codes (byte z of the window in bits 2z..,
 * code = (byte >> shift) & 3) that the W text bytes at the unit's position may show when the unit's predicate holds
 * (W = 8: the 16-bit words of the verify image; W = 9: the 18-bit words of the sieve's even alignment):
 * a superset is fine (the bitmap is a filter), a missing word would lose matches.
 * ------------------------------------------------------------------------- */
#if 1 /* host functions (parsed in both passes of a .hip unit) */
struct ApmUnit {
    int off, len;   /* exact part */
    int poff, plen; /* partner (side != 0) */
    int side;       /* 0 none, 1 partner behind the exact part, 2 in front of it */
};

/* the unit's code-filter record (apm_cf_pass) */
inline void apm_cf_record(const uint8_t *pat, const ApmUnit &u, int shift, uint32_t *rx, uint32_t *ry) {
    auto code = [&](int y) { return (uint32_t)((pat[y] >> shift) & 3); };
    /* (an exact part beyond 255 bytes -- patterns beyond 256 bytes with k = 0 -- does not fit the record's length field:
       the partner, whose position it gives, is then not judged) */
    const int side = u.len > 255 ? 0 : u.side;
    const int np = side ? (u.plen < 15 ? u.plen : 15) : 0;
    uint32_t p = 0, e2 = 0;
    for (int i = 0; i < np; ++i) p |= code(side == 1 ? u.poff + i : u.poff + u.plen - 1 - i) << (2 * i);
    for (int i = 8; i < u.len && i < 16; ++i) e2 |= code(u.off + i) << (2 * (i - 8));
    *rx = p | ((uint32_t)side << 30);
    *ry = e2 | ((uint32_t)np << 16) | ((uint32_t)(u.len < 255 ? u.len : 255) << 20);
}

/* can the text codes t[0..vis) (the first vis text bytes behind the exact part) still belong to a text that matches
   the partner codes c[0..n) within one edit?  (bytes beyond vis are unknown = wildcards) */
inline bool apm_ext1_visible_ok(const uint8_t *c, int n, const uint8_t *t, int vis) {
    int i = 0;
    while (i < n && i < vis && t[i] == c[i]) ++i;
    if (i >= vis || i >= n - 1) return true;
    bool ok = true; /* substitution at i */
    for (int j = i + 1; j < n && j < vis && ok; ++j) ok = t[j] == c[j];
    if (ok) return true;
    ok = true; /* pattern byte i has no text counterpart */
    for (int j = i + 1; j < n && j - 1 < vis && ok; ++j) ok = t[j - 1] == c[j];
    if (ok) return true;
    ok = true; /* one extra text byte before pattern byte i */
    for (int j = i; j < n && j + 1 < vis && ok; ++j) ok = t[j + 1] == c[j];
    return ok;
}

/* definition: every continuation of the visible exact part, filtered by apm_ext1_visible_ok */
template <typename F>
inline void apm_enum_unit_windows_bruteforce(const uint8_t *pat, const ApmUnit &u, int shift, F fn, int W = 8) {
    const int vis = u.len < W ? u.len : W, ext = W - vis;
    uint32_t x = 0;
    for (int z = 0; z < vis; ++z) x |= (uint32_t)((pat[u.off + z] >> shift) & 3) << (2 * z);
    uint8_t c[64], t[9];
    const int n = u.side == 1 ? (u.plen < 64 ? u.plen : 64) : 0;
    for (int i = 0; i < n; ++i) c[i] = (uint8_t)((pat[u.poff + i] >> shift) & 3);
    for (uint32_t p = 0; p < (1u << (2 * ext)); ++p) {
        for (int j = 0; j < ext; ++j) t[j] = (uint8_t)((p >> (2 * j)) & 3u);
        if (u.side != 1 || ext == 0 || apm_ext1_visible_ok(c, n, t, ext)) fn(x | (p << (2 * vis)));
    }
}

/* the same set (or a superset), generated from the edit neighbourhood instead of filtered out of 4^ext words */
template <typename F>
inline void apm_enum_unit_windows(const uint8_t *pat, const ApmUnit &u, int shift, F fn, int W = 8) {
    const int vis = u.len < W ? u.len : W, ext = W - vis;
    if (u.side != 1 || ext <= 4) { /* at most 256 continuations: the definition is cheap enough */
        apm_enum_unit_windows_bruteforce(pat, u, shift, fn, W);
        return;
    }
    uint32_t x = 0;
    for (int z = 0; z < vis; ++z) x |= (uint32_t)((pat[u.off + z] >> shift) & 3) << (2 * z);
    uint8_t c[64];
    const int n = u.plen < 64 ? u.plen : 64;
    for (int i = 0; i < n; ++i) c[i] = (uint8_t)((pat[u.poff + i] >> shift) & 3);
    /* emit the words whose visible part is w[0..ext) with `wild` = bitmask of positions that may hold any code */
    auto emit = [&](const uint8_t *w, uint32_t wild) {
        int wp[9], nw = 0;
        uint32_t base = 0;
        for (int j = 0; j < ext; ++j) {
            if ((wild >> j) & 1u) wp[nw++] = j;
            else base |= (uint32_t)w[j] << (2 * j);
        }
        for (uint32_t q = 0; q < (1u << (2 * nw)); ++q) {
            uint32_t v = base;
            for (int z = 0; z < nw; ++z) v |= ((q >> (2 * z)) & 3u) << (2 * wp[z]);
            fn(x | (v << (2 * vis)));
        }
    };
    uint8_t w[9];
    /* no edit inside the visible part / a substitution at i (any code there) */
    for (int i = -1; i < ext && i < n; ++i) {
        uint32_t wild = 0;
        for (int j = 0; j < ext; ++j) {
            if (j >= n || j == i) wild |= 1u << j;
            else w[j] = c[j];
        }
        emit(w, wild);
    }
    /* pattern byte i has no text counterpart: text = c[0..i) c[i+1..] */
    for (int i = 0; i < n && i <= ext; ++i) {
        uint32_t wild = 0;
        for (int j = 0; j < ext; ++j) {
            const int src = j < i ? j : j + 1;
            if (src >= n) wild |= 1u << j;
            else w[j] = c[src];
        }
        emit(w, wild);
    }
    /* one extra text byte before pattern byte i: text = c[0..i) ? c[i..] */
    for (int i = 0; i < n && i < ext; ++i) {
        uint32_t wild = 0;
        for (int j = 0; j < ext; ++j) {
            const int src = j < i ? j : j - 1;
            if (j == i || src >= n) wild |= 1u << j;
            else w[j] = c[src];
        }
        emit(w, wild);
    }
}
#endif /* host */

#endif /* APM_CORE_H */
