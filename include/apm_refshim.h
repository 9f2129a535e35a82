/*
 * include/apm_refshim.h -- the reference's OWN six GPU entry points, exported by libapm_hip.so
 * (csrc/apm_refshim.c, plain C over include/apm.h) so that the reference's host sources link unmodified.
 * The reference has no header for them: each caller re-declares the prototypes by hand.  They are
 * repeated here verbatim (same names, argument order and types) with the place they come from:
 *
 *   /root/reference/src/main.c:18-19                  getDeviceCount, setDevice
 *   /root/reference/src/patterns_over_ranks.c:33-36   invoke_kernel, write_kernel_result
 *   /root/reference/src/database_over_ranks.c:18-22   initializeGPU, getGPUResult
 *
 * New code should use include/apm.h; these exist for link-level compatibility only
 * ("Link the reference unmodified").
 */
#ifndef APM_REFSHIM_H
#define APM_REFSHIM_H

#ifdef __cplusplus
extern "C" {
#endif

/* replaces src/cuda_utils.cu:10-20 -- number of HIP devices; prints and exit(EXIT_FAILURE)s on a runtime error */
void getDeviceCount(int *deviceCountPtr);
/* replaces src/cuda_utils.cu:22-35 -- selects device 0; prints one line if deviceCount == 0 */
void setDevice(int rank, int deviceCount);

/* replaces src/patterns_over_ranks.cu:75-113 -- enqueues the scan of buf[0:n_bytes) for one pattern
 * (window starts [0, n_bytes - approx_factor), windows cut at n_bytes); returns an opaque handle the
 * caller hands back to write_kernel_result (the reference returns a device pointer) */
int *invoke_kernel(char *buf, int n_bytes, char *my_pattern, int pattern_length, int approx_factor, int *local_matches);
/* replaces src/patterns_over_ranks.cu:115-134 -- waits, writes *local_matches (initial value + matches), frees the handle */
void write_kernel_result(int *local_matches, int *d_local_matches);

/* replaces src/database_over_ranks.cu:137-192 -- scans patterns [0, lastPatternAnalyzedByGPU) over the rank's piece
 * exactly as the reference's searchPattern kernel does; returns 1 */
int initializeGPU(char *buf, int n_bytes, char **pattern, int nb_patterns, int lastPatternAnalyzedByGPU, int *sizePatterns,
                  int indexFinishMyPieceWithoutExtra, int myRank, int numberProcesses, int indexStartMyPiece,
                  int approx_factor, int *numberOfMatchesInitialized);
/* replaces src/database_over_ranks.cu:194-205 -- malloc()ed int[nb_patterns], caller-owned */
int *getGPUResult(int nb_patterns);

#ifdef __cplusplus
}
#endif
#endif /* APM_REFSHIM_H */
