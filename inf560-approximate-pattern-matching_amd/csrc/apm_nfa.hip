/*
 * apm_nfa.hip -- short, loose patterns (m + k/2 <= 32, k <= 7, pieces too short for the BANDED filter): the k-error
 * automaton of the window DP, evaluated for 32 CONSECUTIVE WINDOW STARTS per lane at once.
 *
 * The reference decides dist(p, t[j .. j+m)) <= k per window start j with the full m x m DP
 * (/root/reference/src/utils.c:76-99 called from /root/reference/src/sequential.c:121-141).  Pattern and window have the
 * same length, so an alignment with <= k edits has as many insertions as deletions and never leaves the diagonals
 * |y - x| <= B = k/2.  For each error level e <= k and diagonal d the lane keeps a 32-bit word R[e][d]: bit b says
 * "for window start j0 + b, cell(x, x + d) <= e" after x pattern bytes.  One pattern byte c advances ALL 32 windows:
 *
 *     N[e][d] = (R[e][d] & M_d)            match: t[j + x + d] == c, M_d = T_c >> (x + d), T_c = the positions of byte c
 *             |  R[e-1][d]                  substitution
 *             |  R[e-1][d+1]                pattern byte without a text byte (the cell above)
 *             |  N[e-1][d-1]                text byte without a pattern byte (the cell to the left, same column)
 *
 * cells outside the m x m square (y < 0, y > m) are empty; x = 0 starts with cell(0, y) = y; the window matches iff
 * R[k][0] holds after m bytes.  Only the cells that can lie on a path to the accepting one are kept -- |d| <= e and
 * e + |d| <= k (apm_nfa_live): 8 of the 12 at k = 3, 32 of 56 at k = 7 -- each one or two three-input logic instructions
 * (v_bitop3_b32, the 2-cycle class), plus 2B+1 funnel shifts, one LDS read and one address add per pattern byte for 32
 * windows: 12 + 3 + 2 instructions per byte at k = 3, ~8 per window and pattern at m = 14, where the bit-vector column of
 * BITPAR (one window per lane, 13.7 instructions per column) needs ~190.  T_c, the per-lane bitmask of the text positions
 * holding byte c (64 positions: 32 window starts + m + B), is built once per lane and launch class (<= 16 distinct
 * pattern bytes per launch) from the text bytes and parked in LDS; every pattern of the launch then reads it.  The
 * pattern itself is a string of class numbers, a nibble per byte, that stays on the scalar unit (one 16-byte scalar
 * load per pattern, a 64-bit shift per two columns).
 * Exact for the predicate dist <= k (what the reference's `if (distance <= approx_factor)` consumes), not for the distance.
 */
#include "apm_internal.h"
#include "apm_core.h"
#include "apm_device.h"

#define APM_NFA_TILE (APM_BLOCK * 32) /* window starts per workgroup: 32 per lane */

template <int K>
__global__ __launch_bounds__(APM_BLOCK) void apm_nfa_kernel(ApmNfaArgs a) {
    constexpr int B = K / 2, ND = 2 * B + 1, NE = K + 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *s_T = reinterpret_cast<uint32_t *>(smem);                    // [class][word 0 / 1][thread]
    uint32_t *s_cnt = s_T + (size_t)a.n_classes * 2 * APM_BLOCK;           // [n_pats]
    const int tid = threadIdx.x;
    for (int i = tid; i < a.n_pats; i += APM_BLOCK) s_cnt[i] = 0u;

    // the lane's 64 text bytes: window starts j0 .. j0 + 31 and what their windows reach (m + B <= 32 bytes further)
    const int64_t j0 = a.tile0 + (int64_t)blockIdx.x * APM_NFA_TILE + (int64_t)tid * 32;
    uint32_t w[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 v = apm_load16_guarded(a.text, j0 + 16 * q, a.avail);
        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
    // T_c for every class of the launch: bit i of the 64-bit mask = (text[j0 + i] == byte of class c)
    for (int c = 0; c < a.n_classes; ++c) {
        const uint32_t cb = (uint32_t)a.class_bytes[c] * 0x01010101u;
        uint32_t tlo = 0, thi = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t x = w[q] ^ cb;
            const uint32_t nz = __builtin_amdgcn_udot4(((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) >> 7) & 0x01010101u, 0x08040201u, 0u, false); // 4 bits: byte != c
            const uint32_t eq = nz ^ 0xfu;
            if (q < 8) tlo |= eq << (4 * q);
            else thi |= eq << (4 * (q - 8));
        }
        s_T[(size_t)(2 * c) * APM_BLOCK + tid] = tlo;
        s_T[(size_t)(2 * c + 1) * APM_BLOCK + tid] = thi;
    }
    __syncthreads(); // (s_cnt; the T words are read by their own lane only)

    // one pattern byte: Rout = the column after it (see the header); Rin is left as it was.  c = its class.
    auto step = [&](const uint32_t (&Rin)[NE][ND], uint32_t (&Rout)[NE][ND], int x, int m, uint32_t c) __attribute__((always_inline)) {
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(c * 2u * APM_BLOCK)); // (kept on the scalar unit: one v_add for the address)
        const uint32_t tlo = s_T[base + (uint32_t)tid], thi = s_T[base + APM_BLOCK + (uint32_t)tid];
        uint32_t M[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int sh = x + i - B; // text offset of pattern byte x on diagonal i - B (< 32)
            M[i] = sh >= 0 ? __builtin_amdgcn_alignbit(thi, tlo, (uint32_t)sh) : 0u;
        }
        if (x + 1 - B < 0 || x + 1 + B > m) apm_nfa_step<K, true>(Rin, Rout, M, x, m); // (wave-uniform: the first / last B columns)
        else apm_nfa_step<K, false>(Rin, Rout, M, x, m);
    };

    // the lane's window starts against the range [jb, min(je, nrel - m + 1)), clamped once to what 32 bits hold
    const int64_t lo64 = a.jb - j0, je64 = a.je - j0, end64 = a.nrel + 1 - j0;
    const uint32_t lo_mask = lo64 <= 0 ? 0xffffffffu : (lo64 >= 32 ? 0u : (0xffffffffu << (int)lo64));
    const int je_rel = (int)(je64 < -64 ? -64 : (je64 > 64 ? 64 : je64)), end_rel = (int)(end64 < -64 ? -64 : (end64 > 128 ? 128 : end64));
    for (int p = 0; p < a.n_pats; ++p) {
        const ApmPatDesc d = a.pats[p];
        const int m = (int)d.m;
        // the pattern as class numbers, a nibble per byte: one scalar 16-byte load, the next class shifted out per column --
        // no LDS read in front of the T words' address
        const uint4 cw = *reinterpret_cast<const uint4 *>(a.classes + d.byte_off);
        unsigned long long clo = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)cw.y) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)cw.x);
        unsigned long long chi = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)cw.w) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)cw.z);
        uint32_t R[NE][ND], N[NE][ND];
        apm_nfa_init<K>(R); // cell(0, y) = y
        int x = 0;
        // (two bytes per trip: the columns swap roles instead of being copied.  Fully unrolled over 32 bytes the loop
        // measured slower, 9.8 against 9.0 ms per GiB, and so did T words kept in registers for <= 4 classes, picked by
        // uniform selects: 10.9)
        for (; x + 1 < m; x += 2) {
            const uint32_t c2 = (uint32_t)clo & 0xffu;
            clo = (clo >> 8) | (chi << 56);
            chi >>= 8;
            step(R, N, x, m, c2 & 0xfu);
            step(N, R, x + 1, m, c2 >> 4);
        }
        if (x < m) step(R, N, x, m, (uint32_t)clo & 0xfu);
        const uint32_t fin = (m & 1) ? N[K][B] : R[K][B];
        // windows j0 + b that are full windows of this shard's range: bits [lo_rel, hi)
        const int hi = min(je_rel, end_rel - m);
        uint32_t valid = lo_mask;
        if (hi < 32) valid = hi <= 0 ? 0u : (valid & ((1u << hi) - 1u));
        uint32_t hits = fin & valid;
        if (a.pos.out)
            for (uint32_t h = hits; h; h &= h - 1u) apm_push_pos(a.pos, j0 + (int64_t)__builtin_ctz(h));
        if (hits) atomicAdd(&s_cnt[p], (uint32_t)__builtin_popcount(hits)); // (matches are rare: a handful of lanes, one LDS atomic)
    }
    __syncthreads();
    for (int i = tid; i < a.n_pats; i += APM_BLOCK) {
        const uint32_t c = s_cnt[i];
        if (c) atomicAdd(&a.counts[a.pats[i].index], (unsigned long long)c);
    }
}

size_t apm_nfa_lds_bytes(const ApmNfaArgs &a) { return (size_t)a.n_classes * 2 * APM_BLOCK * 4 + (size_t)((a.n_pats + 3) & ~3) * 4 + 16; }

hipError_t apm_launch_nfa(const ApmNfaArgs &a, hipStream_t s) {
    const int64_t span = a.je - a.tile0;
    if (span <= 0 || a.n_pats <= 0) return hipSuccess;
    const int64_t nt = (span + APM_NFA_TILE - 1) / APM_NFA_TILE;
    if (nt > 0x7fffffffLL || a.k < 0 || a.k > 7 || a.n_classes < 1 || a.n_classes > 16) return hipErrorInvalidValue;
    const size_t lds = apm_nfa_lds_bytes(a);
    const dim3 g((unsigned)nt), b(APM_BLOCK);
    switch (a.k) {
    case 0: hipLaunchKernelGGL(apm_nfa_kernel<0>, g, b, lds, s, a); break;
    case 1: hipLaunchKernelGGL(apm_nfa_kernel<1>, g, b, lds, s, a); break;
    case 2: hipLaunchKernelGGL(apm_nfa_kernel<2>, g, b, lds, s, a); break;
    case 3: hipLaunchKernelGGL(apm_nfa_kernel<3>, g, b, lds, s, a); break;
    case 4: hipLaunchKernelGGL(apm_nfa_kernel<4>, g, b, lds, s, a); break;
    case 5: hipLaunchKernelGGL(apm_nfa_kernel<5>, g, b, lds, s, a); break;
    case 6: hipLaunchKernelGGL(apm_nfa_kernel<6>, g, b, lds, s, a); break;
    default: hipLaunchKernelGGL(apm_nfa_kernel<7>, g, b, lds, s, a); break;
    }
    return hipGetLastError();
}
