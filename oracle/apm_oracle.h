/*
 * oracle/apm_oracle.h -- CPU restatement of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (include/apm.h,
 * inf560-approximate-pattern-matching_amd/) may include, link or call this.
 * Allowed users: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 *
 * Parity status: PINNED.  oracle_window_distance / oracle_count are checked
 *   (a) against the reference's own published counts (README.md:58-63,
 *       scripts/run_tests:31 inputs) and
 *   (b) against the reference binary itself, compiled from
 *       /root/reference/src/{utils,sequential}.c into oracle/_ref/ by
 *       oracle/Makefile (golden vectors in tests/golden/golden.json, produced
 *       by oracle/gen_golden.py).
 */
#ifndef APM_ORACLE_H
#define APM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Restates levenshtein(), /root/reference/src/utils.c:76-99.
 * Square global edit distance between s1[0:len] (pattern prefix) and
 * s2[0:len] (text window).  column must hold len+1 ints. */
int oracle_window_distance(const unsigned char *s1, const unsigned char *s2,
                           int len, int *column);

/* Restates the per-pattern scan, /root/reference/src/sequential.c:105-144,
 * with 64-bit positions:  #{ j in [j_begin, j_end) ∩ [0, n-k) :
 *   dist(pattern[0:size], text[j:j+size]) <= k, size = min(m, n-j) }.
 * Single thread, literal loop order.  Returns -1 on allocation failure. */
int64_t oracle_count_range(const unsigned char *text, uint64_t n,
                           const unsigned char *pattern, int m, int k,
                           uint64_t j_begin, uint64_t j_end);

/* Whole-text count for one pattern == sequential.c's n_matches[i]. */
int64_t oracle_count(const unsigned char *text, uint64_t n,
                     const unsigned char *pattern, int m, int k);

/* Same result as oracle_count_range, positions split over `threads` OpenMP
 * threads (each with a private column).  threads<=0: all cores. */
int64_t oracle_count_range_mt(const unsigned char *text, uint64_t n,
                              const unsigned char *pattern, int m, int k,
                              uint64_t j_begin, uint64_t j_end, int threads);

/* Exact-for-the-predicate variant: DP restricted to the diagonals
 * |x-y| <= k/2 with early exit.  Equal strings lengths => #ins == #del, so an
 * alignment of cost <= k never leaves that band; the test-suite checks
 * banded == literal on every fixture.  Used only to make big CPU checks
 * finish in seconds. */
int64_t oracle_count_range_banded_mt(const unsigned char *text, uint64_t n,
                                     const unsigned char *pattern, int m, int k,
                                     uint64_t j_begin, uint64_t j_end,
                                     int threads);

int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
