/* Test program for the link-level shim (include/apm_refshim.h): calls the reference's six GPU entry points
 * the way its host files do (src/main.c:70-73, src/patterns_over_ranks.c:316-327,380, src/database_over_ranks.c:273-279,561)
 * and prints what they return; tests/test_gpu_parity.py compares with the golden counts / the oracle.
 *   usage: refshim_test <k> <file> <pattern...>   */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "apm_refshim.h"

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const int k = atoi(argv[1]);
    FILE *f = fopen(argv[2], "rb");
    if (!f) return 3;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)n + 1);
    if (fread(buf, 1, (size_t)n, f) != (size_t)n) return 4;
    fclose(f);
    const int P = argc - 3;
    char **pat = argv + 3;

    int ndev = -1;
    getDeviceCount(&ndev);
    setDevice(0, ndev);
    printf("devices %d\n", ndev > 0 ? 1 : 0);

    /* patterns_over_ranks: one pattern at a time, initial count carried through */
    printf("invoke");
    for (int i = 0; i < P; ++i) {
        int result = 1000 * (i + 1);
        int *handle = invoke_kernel(buf, (int)n, pat[i], (int)strlen(pat[i]), k, &result);
        write_kernel_result(&result, handle);
        printf(" %d", result - 1000 * (i + 1));
    }
    printf("\n");
    /* the reference hands the GPU a prefix "3n/4 + m - 1" of the text (patterns_over_ranks.c:316-326) */
    printf("invoke34");
    for (int i = 0; i < P; ++i) {
        int result = 0;
        long part = 3 * n / 4 + (long)strlen(pat[i]) - 1;
        if (part > n) part = n;
        int *handle = invoke_kernel(buf, (int)part, pat[i], (int)strlen(pat[i]), k, &result);
        write_kernel_result(&result, handle);
        printf(" %d", result);
    }
    printf("\n");

    /* database_over_ranks: ranks 0 and 1 of 2, GPU takes the first `last` patterns */
    int *sizes = (int *)malloc(sizeof(int) * (size_t)P), *zeros = (int *)calloc((size_t)P, sizeof(int));
    for (int i = 0; i < P; ++i) sizes[i] = (int)strlen(pat[i]);
    const int last = P > 1 ? P - 1 : P;
    for (int rank = 0; rank < 2; ++rank) {
        const int start = rank == 0 ? 0 : (int)(n / 2), end = rank == 0 ? (int)(n / 2) : (int)n;
        zeros[0] = 7; /* initial values are carried */
        initializeGPU(buf, (int)n, pat, P, last, sizes, end, rank, 2, start, k, zeros);
        int *res = getGPUResult(P);
        printf("db%d", rank);
        for (int i = 0; i < P; ++i) printf(" %d", res[i] - (i == 0 ? 7 : 0));
        printf("\n");
        free(res);
    }
    free(buf);
    return 0;
}
