// Host-side check of the arithmetic cores in csrc/apm_core.h against the oracle.
// Built and run by tests/test_host_logic.py (g++, no GPU).
#include "apm_core.h"
#include "apm_oracle.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <type_traits>
#include <vector>

template <int W>
static int bp_dist(const unsigned char *p, const unsigned char *t, int m) {
    static uint32_t peq[256][W];
    for (int c = 0; c < 256; c++)
        for (int w = 0; w < W; w++) peq[c][w] = 0;
    for (int y = 0; y < m; y++) peq[p[y]][y / 32] |= 1u << (y % 32);
    uint32_t pv[W], mv[W];
    bp_init<W>(pv, mv);
    for (int x = 0; x < m; x++) {
        uint32_t eq[W];
        for (int w = 0; w < W; w++) eq[w] = peq[t[x]][w];
        bp_step<W>(pv, mv, eq);
    }
    return bp_distance<W>(pv, mv, m, m);
}

int main() {
    srand(1);
    int bad = 0;
    std::vector<int> col(200);
    for (int it = 0; it < 100000; it++) {
        const int m = 1 + rand() % 128;
        const int alpha = 2 + rand() % 3;
        unsigned char p[128], t[128];
        for (int i = 0; i < m; i++) {
            p[i] = 'a' + rand() % alpha;
            t[i] = (rand() % 4) ? p[i] : 'a' + rand() % alpha;
        }
        if (rand() % 3 == 0) {
            const int s = rand() % 3;
            for (int i = 0; i + s < m; i++) t[i] = p[i + s];
        }
        const int ref = oracle_window_distance(p, t, m, col.data());
        const int W = (m + 31) / 32;
        const int d = W == 1 ? bp_dist<1>(p, t, m) : W == 2 ? bp_dist<2>(p, t, m) : W == 3 ? bp_dist<3>(p, t, m) : bp_dist<4>(p, t, m);
        const int d4 = bp_dist<4>(p, t, m);
        if (d != ref || d4 != ref) {
            bad++;
            if (bad < 5) printf("m=%d ref=%d d=%d d4=%d\n", m, ref, d, d4);
        }
    }
    // one-edit extension core vs its byte-loop definition (forward direction; the kernels feed the backward
    // case byte-reversed into the same core)
    {
        auto ext1_loop = [](const unsigned char *pb, const unsigned char *tb, int n) {
            int i = 0;
            while (i < n && tb[i] == pb[i]) ++i;
            if (i >= n - 1) return true; // no mismatch, or a single substitution at the last byte
            bool ok = true;              // substitution at i
            for (int j = i + 1; j < n && ok; ++j) ok = tb[j] == pb[j];
            if (ok) return true;
            ok = true;                   // pattern byte i has no text counterpart
            for (int j = i + 1; j < n && ok; ++j) ok = tb[j - 1] == pb[j];
            if (ok) return true;
            ok = true;                   // one extra text byte before pattern byte i
            for (int j = i; j < n && ok; ++j) ok = tb[j + 1] == pb[j];
            return ok;
        };
        long n_true = 0;
        for (int it = 0; it < 2000000; it++) {
            const int n = 1 + rand() % 16;
            const int alpha = 2 + rand() % 3;
            unsigned char pb[16], tb[20];
            for (int i = 0; i < 16; i++) pb[i] = 'a' + rand() % alpha;
            for (int i = 0; i < 20; i++) tb[i] = 'a' + rand() % alpha;
            const int mode = rand() % 5; // bias towards near matches: copy, substitution, deletion, insertion
            if (mode >= 1) {
                const int e = rand() % n;
                int w = 0;
                for (int i = 0; i < n && w < 20; i++) {
                    if (mode == 2 && i == e) { tb[w++] = 'a' + rand() % alpha; continue; }
                    if (mode == 3 && i == e) continue;
                    if (mode == 4 && i == e) tb[w++] = 'a' + rand() % alpha;
                    if (w < 20) tb[w++] = pb[i];
                }
                if (rand() % 4 == 0) tb[rand() % 20] = 'a' + rand() % alpha; // sometimes a second edit
            }
            uint32_t P[4], T[5];
            memcpy(P, pb, 16);
            memcpy(T, tb, 20);
            const bool got = apm_ext1_core16(P, T, n), want = ext1_loop(pb, tb, n);
            n_true += want;
            if (got != want) {
                bad++;
                if (bad < 5) printf("ext1 n=%d got=%d want=%d\n", n, (int)got, (int)want);
            }
        }
        if (n_true < 100000) { printf("ext1 test saw too few positives\n"); bad++; }
    }
    // presence-bitmap windows of a nomination unit (apm_core.h): the constructive enumeration must contain the
    // brute-force definition, stay close to it in size, and -- the property the sieve relies on -- contain the code word
    // of ANY text window for which the unit's byte-level predicate holds (exact part intact, partner within one edit)
    {
        auto ext1_loop = [](const unsigned char *pb, const unsigned char *tb, int n) {
            int i = 0;
            while (i < n && tb[i] == pb[i]) ++i;
            if (i >= n - 1) return true;
            bool ok = true;
            for (int j = i + 1; j < n && ok; ++j) ok = tb[j] == pb[j];
            if (ok) return true;
            ok = true;
            for (int j = i + 1; j < n && ok; ++j) ok = tb[j - 1] == pb[j];
            if (ok) return true;
            ok = true;
            for (int j = i; j < n && ok; ++j) ok = tb[j + 1] == pb[j];
            return ok;
        };
        long words_def = 0, words_gen = 0, positives = 0, words_def9 = 0, words_gen9 = 0;
        std::vector<unsigned char> in_def(65536), in_gen(65536), in_def9(262144), in_gen9(262144); // W = 8 / W = 9 (the sieve's even alignment)
        for (int it = 0; it < 3000; it++) {
            const char *alphabet = (it % 3 == 0) ? "ACGT" : ((it % 3 == 1) ? "ACGTN\n" : "abcdefgh");
            const int na = (int)strlen(alphabet);
            unsigned char pat[40];
            for (int i = 0; i < 40; i++) pat[i] = (unsigned char)alphabet[rand() % na];
            ApmUnit u;
            u.off = rand() % 4;
            u.len = (it % 4 == 0) ? 0 : 1 + rand() % 12;
            u.side = (u.len == 0) ? 1 : rand() % 3;
            u.plen = 1 + rand() % 16;
            u.poff = u.side == 2 ? 0 : u.off + u.len;
            if (u.side == 2) { u.off = u.plen; }
            const int shift = (it % 3 == 2) ? 0 : 1;
            std::fill(in_def.begin(), in_def.end(), 0);
            std::fill(in_gen.begin(), in_gen.end(), 0);
            apm_enum_unit_windows_bruteforce(pat, u, shift, [&](uint32_t x) { in_def[x & 0xffffu] = 1; });
            apm_enum_unit_windows(pat, u, shift, [&](uint32_t x) { in_gen[x & 0xffffu] = 1; });
            std::fill(in_def9.begin(), in_def9.end(), 0);
            std::fill(in_gen9.begin(), in_gen9.end(), 0);
            apm_enum_unit_windows_bruteforce(pat, u, shift, [&](uint32_t x) { in_def9[x & 0x3ffffu] = 1; }, 9);
            apm_enum_unit_windows(pat, u, shift, [&](uint32_t x) { in_gen9[x & 0x3ffffu] = 1; }, 9);
            for (int x = 0; x < 262144; x++) {
                words_def9 += in_def9[x];
                words_gen9 += in_gen9[x];
                if (in_def9[x] && !in_gen9[x]) { bad++; if (bad < 5) printf("enum (W=9) misses word %x (len %d plen %d side %d)\n", x, u.len, u.plen, u.side); break; }
                if (in_gen9[x] && !in_gen[x & 0xffff]) { bad++; if (bad < 5) printf("a nine-byte word whose first eight bytes are no eight-byte word: %x\n", x); break; }
            }
            long nd = 0, ng = 0;
            for (int x = 0; x < 65536; x++) {
                nd += in_def[x];
                ng += in_gen[x];
                if (in_def[x] && !in_gen[x]) { bad++; if (bad < 5) printf("enum misses word %x (len %d plen %d side %d)\n", x, u.len, u.plen, u.side); break; }
            }
            words_def += nd;
            words_gen += ng;
            // texts satisfying the predicate: exact part + partner with at most one edit, random bytes elsewhere
            for (int rep = 0; rep < 40; rep++) {
                unsigned char text[64];
                for (int i = 0; i < 64; i++) text[i] = (unsigned char)alphabet[rand() % na];
                const int s = 24; // unit position
                for (int i = 0; i < u.len; i++) text[s + i] = pat[u.off + i];
                if (u.side == 1) {
                    const int mode = rand() % 4, e = rand() % u.plen;
                    int w = s + u.len;
                    for (int i = 0; i < u.plen && w < 60; i++) {
                        if (mode == 1 && i == e) { text[w++] = (unsigned char)alphabet[rand() % na]; continue; }
                        if (mode == 2 && i == e) continue;
                        if (mode == 3 && i == e) text[w++] = (unsigned char)alphabet[rand() % na];
                        text[w++] = pat[u.poff + i];
                    }
                    if (u.plen <= 16 && !ext1_loop(pat + u.poff, text + s + u.len, u.plen)) continue; // (the edit fell badly)
                }
                uint32_t x = 0;
                for (int z = 0; z < 8; z++) x |= (uint32_t)((text[s + z] >> shift) & 3) << (2 * z);
                positives++;
                uint32_t x9 = x | ((uint32_t)((text[s + 8] >> shift) & 3) << 16);
                if (!in_gen9[x9]) { bad++; if (bad < 5) printf("nine-byte window of a true unit occurrence not enumerated (len %d plen %d side %d)\n", u.len, u.plen, u.side); }
                if (!in_gen[x]) { bad++; if (bad < 5) printf("window of a true unit occurrence not enumerated (len %d plen %d side %d)\n", u.len, u.plen, u.side); }
            }
        }
        if (words_gen9 > words_def9 * 3 / 2 + 4000) { printf("constructive enumeration (W=9) too loose: %ld vs %ld\n", words_gen9, words_def9); bad++; }
        if (words_gen > words_def * 3 / 2 + 1000) { printf("constructive enumeration too loose: %ld vs %ld\n", words_gen, words_def); bad++; }
        if (positives < 50000) { printf("unit window test saw too few positives\n"); bad++; }
    }
    // the sieve's code filter (apm_cf_record / apm_cf_pass / apm_ext1_codes in apm_core.h): never false where the
    // byte-level nomination predicate holds, whatever the alphabet; on ACGT (codes = bytes) with the whole unit inside
    // the record (exact part <= 16 bytes, partner <= 15) it IS the predicate
    {
        auto ext1_loop = [](const unsigned char *pb, const unsigned char *tb, int n, int dir) { // dir -1: both read backwards
            auto P = [&](int i) { return pb[dir * i]; };
            auto T = [&](int i) { return tb[dir * i]; };
            int i = 0;
            while (i < n && T(i) == P(i)) ++i;
            if (i >= n - 1) return true;
            bool ok = true;
            for (int j = i + 1; j < n && ok; ++j) ok = T(j) == P(j);
            if (ok) return true;
            ok = true;
            for (int j = i + 1; j < n && ok; ++j) ok = T(j - 1) == P(j);
            if (ok) return true;
            ok = true;
            for (int j = i; j < n && ok; ++j) ok = T(j + 1) == P(j);
            return ok;
        };
        long pos = 0, neg_exact = 0, rejected = 0;
        for (int it = 0; it < 400000; it++) {
            const char *alphabet = (it % 3 != 1) ? "ACGT" : ((it % 2) ? "ACGTN\n" : "abcdefgh");
            const bool dna = it % 3 != 1;
            const int na = (int)strlen(alphabet), shift = dna ? 1 : (it % 2 ? 1 : 0);
            unsigned char pat[64];
            for (int i = 0; i < 64; i++) pat[i] = (unsigned char)alphabet[rand() % na];
            ApmUnit u;
            u.len = (it % 5 == 0) ? 0 : 1 + rand() % 20;
            u.side = (u.len == 0) ? 1 : rand() % 3;
            u.plen = 1 + rand() % 20;
            if (u.side == 2) { u.poff = 0; u.off = u.plen; }
            else { u.off = rand() % 4; u.poff = u.off + u.len; }
            unsigned char text[96];
            for (int i = 0; i < 96; i++) text[i] = (unsigned char)alphabet[rand() % na];
            const int s = 40;
            if (rand() % 4) { // plant the unit with up to one edit in the partner (and sometimes one more edit anywhere)
                for (int i = 0; i < u.len; i++) text[s + i] = pat[u.off + i];
                const int mode = rand() % 4, e = rand() % u.plen;
                if (u.side == 1) {
                    int w = s + u.len;
                    for (int i = 0; i < u.plen; i++) {
                        if (mode == 1 && i == e) { text[w++] = (unsigned char)alphabet[rand() % na]; continue; }
                        if (mode == 2 && i == e) continue;
                        if (mode == 3 && i == e) text[w++] = (unsigned char)alphabet[rand() % na];
                        text[w++] = pat[u.poff + i];
                    }
                } else if (u.side == 2) {
                    int w = s - 1;
                    for (int i = u.plen - 1; i >= 0; i--) {
                        if (mode == 1 && i == e) { text[w--] = (unsigned char)alphabet[rand() % na]; continue; }
                        if (mode == 2 && i == e) continue;
                        if (mode == 3 && i == e) text[w--] = (unsigned char)alphabet[rand() % na];
                        text[w--] = pat[u.poff + i];
                    }
                }
                if (rand() % 3 == 0) text[s - 20 + rand() % 50] = (unsigned char)alphabet[rand() % na];
            }
            bool pred = memcmp(text + s, pat + u.off, (size_t)u.len) == 0;
            if (pred && u.side == 1) pred = ext1_loop(pat + u.poff, text + s + u.len, u.plen, 1);
            if (pred && u.side == 2) pred = ext1_loop(pat + u.poff + u.plen - 1, text + s - 1, u.plen, -1);
            uint32_t rx, ry, c0 = 0, tw = 0;
            apm_cf_record(pat, u, shift, &rx, &ry);
            auto code = [&](int p) { return (uint32_t)((text[p] >> shift) & 3); };
            for (int i = 0; i < 16; i++) c0 |= code(s + i) << (2 * i);
            for (int i = 0; i < 16; i++) tw |= code(u.side == 2 ? s - 16 + i : s + u.len + i) << (2 * i);
            if (u.side == 2) tw = apm_rev_codes(tw);
            const bool got = apm_cf_pass(rx, ry, c0, tw, true);
            // the 8-byte window word is the sieve's business: here only the filter proper
            if (pred) {
                pos++;
                if (!got) { bad++; if (bad < 5) printf("code filter rejects a true nomination (len %d plen %d side %d)\n", u.len, u.plen, u.side); }
                if (!apm_cf_pass(rx, ry, c0, 0u, false)) { bad++; if (bad < 5) printf("code filter (partner out of reach) rejects a true nomination\n"); }
            } else if (dna && u.len <= 16 && (u.side == 0 || u.plen <= 15)) {
                // (a mismatch inside the first 8 bytes of the exact part is the sieve bitmap's to see, not the filter's)
                if (memcmp(text + s, pat + u.off, (size_t)std::min(u.len, 8)) == 0) {
                    neg_exact++;
                    if (got) { bad++; if (bad < 5) printf("code filter passes a false nomination on ACGT (len %d plen %d side %d)\n", u.len, u.plen, u.side); }
                    else rejected++;
                }
            }
        }
        if (pos < 100000 || rejected < 10000) { printf("code filter test saw too few cases (%ld positives, %ld exact negatives, %ld rejected)\n", pos, neg_exact, rejected); bad++; }
    }
    // the k-error automaton over 32 window starts (apm_core.h, apm_nfa_init / apm_nfa_step: the kernel of apm_nfa.hip runs the
    // same two functions): bit b of R[K][B] after m bytes == (oracle window distance of start j0 + b <= K)
    {
        long checked = 0, positives = 0;
        auto run = [&](auto kc, const unsigned char *pat, int m, const unsigned char *text /* 64 bytes */) -> uint32_t {
            constexpr int K = decltype(kc)::value, B = K / 2, ND = 2 * B + 1;
            uint32_t R[K + 1][ND], N[K + 1][ND];
            apm_nfa_init<K>(R);
            for (int x = 0; x < m; ++x) {
                uint64_t T = 0;
                for (int i = 0; i < 64; ++i) T |= (uint64_t)(text[i] == pat[x]) << i;
                uint32_t M[ND];
                for (int i = 0; i < ND; ++i) { const int sh = x + i - B; M[i] = sh >= 0 ? (uint32_t)(T >> sh) : 0u; }
                if (x + 1 - B < 0 || x + 1 + B > m) apm_nfa_step<K, true>(R, N, M, x, m); else apm_nfa_step<K, false>(R, N, M, x, m);
                memcpy(R, N, sizeof R);
            }
            return R[K][B];
        };
        for (int it = 0; it < 60000; it++) {
            const int k = rand() % 8, B = k / 2;
            const int m = 1 + rand() % (32 - B);
            const int alpha = 2 + rand() % 3;
            unsigned char text[64 + 40], pat[32];
            for (int i = 0; i < 104; i++) text[i] = (unsigned char)('a' + rand() % alpha);
            const int o = rand() % 32;
            for (int i = 0; i < m; i++) pat[i] = (rand() % 5) ? text[o + i] : (unsigned char)('a' + rand() % alpha);
            if (rand() % 3 == 0 && m > 2) { memmove(pat + 1, pat, (size_t)m - 1); pat[0] = (unsigned char)('a' + rand() % alpha); } // an indel pair
            uint32_t got;
            switch (k) {
            case 0: got = run(std::integral_constant<int, 0>(), pat, m, text); break;
            case 1: got = run(std::integral_constant<int, 1>(), pat, m, text); break;
            case 2: got = run(std::integral_constant<int, 2>(), pat, m, text); break;
            case 3: got = run(std::integral_constant<int, 3>(), pat, m, text); break;
            case 4: got = run(std::integral_constant<int, 4>(), pat, m, text); break;
            case 5: got = run(std::integral_constant<int, 5>(), pat, m, text); break;
            case 6: got = run(std::integral_constant<int, 6>(), pat, m, text); break;
            default: got = run(std::integral_constant<int, 7>(), pat, m, text); break;
            }
            for (int b = 0; b < 32; b++) {
                const bool want = oracle_window_distance(pat, text + b, m, col.data()) <= k;
                checked++;
                positives += want;
                if (want != (bool)((got >> b) & 1u)) { bad++; if (bad < 5) printf("nfa m=%d k=%d start %d: got %d want %d\n", m, k, b, (int)((got >> b) & 1u), (int)want); }
            }
        }
        if (positives < 20000 || checked - positives < 20000) { printf("nfa test saw too few cases (%ld of %ld positive)\n", positives, checked); bad++; }
    }
    // synthetic generator: bytes are ACGT, deterministic
    for (uint64_t i = 0; i < 1000; i++) {
        const uint8_t b = apm_synth_byte(i, 0x5EED0002ull);
        if (b != 'A' && b != 'C' && b != 'G' && b != 'T') bad++;
    }
    printf("bad=%d\n", bad);
    return bad != 0;
}
