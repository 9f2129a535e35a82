/*
 * oracle/apm_oracle.c -- CPU restatement of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see apm_oracle.h).  Parity status: PINNED against
 * the reference binary built into oracle/_ref/ and against README.md:58-63.
 *
 * Written from the algorithm's definition, not transcribed: the reference
 * keeps one int column and two diagonal temporaries; this file does the same
 * arithmetic with its own loop shape and 64-bit positions.
 */
#include "apm_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int min3i(int a, int b, int c) {
    int m = a < b ? a : b;
    return m < c ? m : c;
}

/* /root/reference/src/utils.c:76-99.  cell(x,y) = min(cell(x-1,y)+1,
 * cell(x,y-1)+1, cell(x-1,y-1) + (s1[y-1]!=s2[x-1])), boundary cell(0,y)=y,
 * cell(x,0)=x; answer cell(len,len).  x indexes the text window (s2), y the
 * pattern (s1), exactly as utils.c:87-95. */
int oracle_window_distance(const unsigned char *s1, const unsigned char *s2,
                           int len, int *column) {
    for (int y = 0; y <= len; ++y) column[y] = y; /* utils.c:84-86 (+[0]) */
    for (int x = 1; x <= len; ++x) {
        int diag = column[0];  /* cell(x-1, 0) == x-1   utils.c:89 */
        column[0] = x;         /* cell(x, 0)            utils.c:88 */
        const unsigned char tc = s2[x - 1];
        for (int y = 1; y <= len; ++y) {
            const int left = column[y];      /* cell(x-1, y) */
            const int up = column[y - 1];    /* cell(x, y-1) */
            column[y] = min3i(left + 1, up + 1, diag + (s1[y - 1] != tc));
            diag = left;                     /* utils.c:94 */
        }
    }
    return column[len]; /* utils.c:97 */
}

/* /root/reference/src/sequential.c:121-141 for one pattern. */
static int64_t count_literal(const unsigned char *text, uint64_t n,
                             const unsigned char *pattern, int m, int k,
                             uint64_t j_begin, uint64_t j_end, int *column) {
    int64_t hits = 0;
    if (k < 0 || n <= (uint64_t)k) return 0;     /* loop bound n-k, :121 */
    const uint64_t limit = n - (uint64_t)k;
    if (j_end > limit) j_end = limit;
    for (uint64_t j = j_begin; j < j_end; ++j) {
        int size = m;                               /* :131 */
        if (n - j < (uint64_t)m) size = (int)(n - j); /* :132-134 */
        const int d = oracle_window_distance(pattern, text + j, size, column); /* :136 */
        if (d <= k) ++hits;                         /* :138-140 */
    }
    return hits;
}

int64_t oracle_count_range(const unsigned char *text, uint64_t n,
                           const unsigned char *pattern, int m, int k,
                           uint64_t j_begin, uint64_t j_end) {
    int *column = (int *)malloc(((size_t)m + 1) * sizeof(int)); /* :112 */
    if (!column) return -1;
    const int64_t r = count_literal(text, n, pattern, m, k, j_begin, j_end, column);
    free(column);                                               /* :143 */
    return r;
}

int64_t oracle_count(const unsigned char *text, uint64_t n,
                     const unsigned char *pattern, int m, int k) {
    return oracle_count_range(text, n, pattern, m, k, 0, n);
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int64_t oracle_count_range_mt(const unsigned char *text, uint64_t n,
                              const unsigned char *pattern, int m, int k,
                              uint64_t j_begin, uint64_t j_end, int threads) {
    if (k < 0 || n <= (uint64_t)k) return 0;
    if (j_end > n - (uint64_t)k) j_end = n - (uint64_t)k;
    if (j_begin >= j_end) return 0;
    if (threads <= 0) threads = oracle_max_threads();
    int64_t total = 0;
    int failed = 0;
#pragma omp parallel num_threads(threads) reduction(+ : total) reduction(| : failed)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        const int t = 0, nt = 1;
#endif
        int *column = (int *)malloc(((size_t)m + 1) * sizeof(int));
        if (!column) {
            failed = 1;
        } else {
            const uint64_t span = j_end - j_begin;
            const uint64_t lo = j_begin + span * (uint64_t)t / (uint64_t)nt;
            const uint64_t hi = j_begin + span * (uint64_t)(t + 1) / (uint64_t)nt;
            total += count_literal(text, n, pattern, m, k, lo, hi, column);
            free(column);
        }
    }
    return failed ? -1 : total;
}

/* Banded window predicate: dist(pattern[0:len], win[0:len]) <= k ?
 * Only diagonals d = y - x with |d| <= w = k/2 are evaluated (everything
 * outside is +inf), rows abandon as soon as the whole band exceeds k. */
static int banded_le_k(const unsigned char *pat, const unsigned char *win,
                       int len, int k, int *prev, int *cur) {
    const int w = k / 2;
    const int INF = 1 << 20;
    if (len <= 0) return 0 <= k;
    /* prev[d+w] = cell(x-1, x-1+d); start with x-1 = 0: cell(0, d) = d for d>=0 */
    for (int d = -w; d <= w; ++d) prev[d + w] = d >= 0 ? d : INF;
    for (int x = 1; x <= len; ++x) {
        int best = INF;
        const unsigned char tc = win[x - 1];
        for (int d = -w; d <= w; ++d) {
            const int y = x + d;
            int v;
            if (y < 0 || y > len) {
                v = INF;
            } else if (y == 0) {
                v = x;
            } else {
                const int diag = prev[d + w] + (pat[y - 1] != tc);     /* (x-1,y-1) */
                const int left = d + 1 <= w ? prev[d + 1 + w] + 1 : INF; /* (x-1,y)  */
                const int up = d - 1 >= -w ? cur[d - 1 + w] + 1 : INF;   /* (x,y-1)  */
                v = min3i(diag, left, up);
            }
            cur[d + w] = v;
            if (v < best) best = v;
        }
        if (best > k) return 0;
        int *t = prev; prev = cur; cur = t;
    }
    return prev[w] <= k;
}

int64_t oracle_count_range_banded_mt(const unsigned char *text, uint64_t n,
                                     const unsigned char *pattern, int m, int k,
                                     uint64_t j_begin, uint64_t j_end,
                                     int threads) {
    if (k < 0 || n <= (uint64_t)k) return 0;
    if (j_end > n - (uint64_t)k) j_end = n - (uint64_t)k;
    if (j_begin >= j_end) return 0;
    if (threads <= 0) threads = oracle_max_threads();
    int64_t total = 0;
    const int bw = 2 * (k / 2) + 1;
#pragma omp parallel num_threads(threads) reduction(+ : total)
    {
        int *a = (int *)malloc(sizeof(int) * (size_t)bw * 2);
        int *b = a + bw;
        int64_t local = 0;
#pragma omp for schedule(static)
        for (int64_t jj = (int64_t)j_begin; jj < (int64_t)j_end; ++jj) {
            const uint64_t j = (uint64_t)jj;
            int size = m;
            if (n - j < (uint64_t)m) size = (int)(n - j);
            local += banded_le_k(pattern, text + j, size, k, a, b);
        }
        total += local;
        free(a);
    }
    return total;
}
