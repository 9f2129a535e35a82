/*
 * apm_parallel -- command-line host of the MI355X engine, plain C over the C ABI
 * (include/apm.h); no HIP headers, no MPI, no OpenMP.
 *
 * Keeps the reference's process contract:
 *   apm_parallel <distance> <text_file> <pattern_1> ... <pattern_P>
 *                [DB_OVER_RANKS|PATTERNS_OVER_RANKS] [--gpus N] [--kernel NAME] [--positions]
 *   argv grammar + usage line        /root/reference/src/sequential.c:35-77
 *   optional trailing approach flag  /root/reference/src/main.c:66-86 (accepted, ignored:
 *                                    the text is always sharded over the GPUs)
 *   banner  "Approximate Pattern Mathing: ..." (sic)   src/sequential.c:79-82,
 *                                                       src/database_over_ranks.c:97-100
 *   timing  "APM done in %lf s"                         src/sequential.c:151
 *   result  "Number of matches for pattern <%s>: %d"    src/sequential.c:157-160
 *   errors on stderr, exit code 1                       src/utils.c:20-23, src/sequential.c:65-68
 * Replaces the MPI/OpenMP dispatch of src/main.c, src/patterns_over_ranks.c and
 * src/database_over_ranks.c with one process driving all GPUs of the node.
 * Counts are printed as 64-bit values (identical text whenever they fit an int).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include "apm.h"

static int kernel_by_name(const char *s) {
    if (!strcmp(s, "auto")) return APM_KERNEL_AUTO;
    if (!strcmp(s, "generic")) return APM_KERNEL_GENERIC;
    if (!strcmp(s, "wavefront")) return APM_KERNEL_WAVEFRONT;
    if (!strcmp(s, "bitpar")) return APM_KERNEL_BITPAR;
    if (!strcmp(s, "banded")) return APM_KERNEL_BANDED;
    if (!strcmp(s, "nfa")) return APM_KERNEL_NFA;
    return -1;
}

int main(int argc, char **argv) {
    int n_gpus = 0; /* 0 = all visible */
    int kernel = APM_KERNEL_AUTO;
    int verbose = 0;
    int want_positions = 0; /* extension: also print the matching offsets */

    /* strip our own options (anywhere after the pattern list starts is fine:
       the reference has none, so nothing is taken away from its grammar) */
    int w = 1;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--gpus") && i + 1 < argc) {
            n_gpus = atoi(argv[++i]);
        } else if (!strcmp(argv[i], "--kernel") && i + 1 < argc) {
            kernel = kernel_by_name(argv[++i]);
            if (kernel < 0) {
                fprintf(stderr, "Unknown kernel variant <%s>\n", argv[i]);
                return 1;
            }
        } else if (!strcmp(argv[i], "--verbose")) {
            verbose = 1;
        } else if (!strcmp(argv[i], "--positions")) {
            want_positions = 1;
        } else {
            argv[w++] = argv[i];
        }
    }
    argc = w;

    /* trailing approach flag of the reference's apm_parallel (src/main.c:66-86) */
    int partition = APM_PARTITION_TEXT; /* DB_OVER_RANKS, and the default: the text is sharded over the devices */
    if (argc >= 2 && (!strcmp(argv[argc - 1], "DB_OVER_RANKS") || !strcmp(argv[argc - 1], "PATTERNS_OVER_RANKS"))) {
        if (!strcmp(argv[argc - 1], "PATTERNS_OVER_RANKS")) partition = APM_PARTITION_PATTERNS; /* the pattern list is (with --gpus > 1) */
        argc -= 1;
    }

    if (argc < 4) {
        printf("Usage: %s approximation_factor dna_database pattern1 pattern2 ...\n", argv[0]);
        return 1;
    }

    const int approx_factor = atoi(argv[1]);
    const char *filename = argv[2];
    const int nb_patterns = argc - 3;

    int *len = (int *)malloc((size_t)nb_patterns * sizeof(int));
    uint64_t *n_matches = (uint64_t *)malloc((size_t)nb_patterns * sizeof(uint64_t));
    if (!len || !n_matches) {
        fprintf(stderr, "Unable to allocate array of pattern of size %d\n", nb_patterns);
        return 1;
    }
    for (int i = 0; i < nb_patterns; ++i) {
        len[i] = (int)strlen(argv[i + 3]);
        if (len[i] <= 0) {
            fprintf(stderr, "Error while parsing argument %d\n", i + 3);
            return 1;
        }
    }

    printf("Approximate Pattern Mathing: looking for %d pattern(s) in file %s w/ distance of %d\n",
           nb_patterns, filename, approx_factor);
    fflush(stdout);

    /* the reference opens the file before anything else can fail (src/sequential.c:84) */
    FILE *probe = fopen(filename, "rb");
    if (!probe) {
        fprintf(stderr, "Unable to open the text file <%s>\n", filename);
        return 1;
    }
    fclose(probe);

    apm_ctx *ctx = NULL;
    int rc = apm_create(&ctx, n_gpus);
    if (rc != APM_OK) {
        fprintf(stderr, "apm_parallel: %s\n", apm_last_error(NULL));
        return 1;
    }
    if ((rc = apm_set_partition(ctx, partition)) != APM_OK) {
        fprintf(stderr, "apm_parallel: %s\n", apm_last_error(ctx));
        apm_destroy(ctx);
        return 1;
    }
    if (kernel != APM_KERNEL_AUTO && (rc = apm_set_kernel(ctx, kernel)) != APM_OK) {
        fprintf(stderr, "apm_parallel: %s\n", apm_last_error(ctx));
        apm_destroy(ctx);
        return 1;
    }
    rc = apm_set_patterns(ctx, nb_patterns, (const char *const *)(argv + 3), len, approx_factor);
    if (rc != APM_OK) {
        fprintf(stderr, "apm_parallel: %s\n", apm_last_error(ctx));
        apm_destroy(ctx);
        return 1;
    }

    struct timeval t1, t2;
    gettimeofday(&t1, NULL);
    rc = apm_count_file(ctx, filename, n_matches);
    gettimeofday(&t2, NULL);
    if (rc != APM_OK) {
        fprintf(stderr, "%s\n", apm_last_error(ctx));
        apm_destroy(ctx);
        return 1;
    }
    const double duration = (double)(t2.tv_sec - t1.tv_sec) + (double)(t2.tv_usec - t1.tv_usec) / 1e6;
    printf("APM done in %lf s\n", duration);

    if (verbose) {
        apm_timing tm;
        if (apm_get_timing(ctx, &tm) == APM_OK)
            fprintf(stderr,
                    "[apm] devices=%d launches=%d h2d=%.3f ms kernels=%.3f ms reduce=%.3f ms total=%.3f ms "
                    "windows=%llu algorithmic_cells=%.4g (%.4g cells/s)\n",
                    tm.n_devices, tm.n_launches, tm.h2d_ms, tm.kernel_ms, tm.reduce_ms, tm.total_ms,
                    (unsigned long long)tm.windows, tm.cells_algorithmic,
                    tm.total_ms > 0 ? tm.cells_algorithmic / (tm.total_ms * 1e-3) : 0.0);
    }

    for (int i = 0; i < nb_patterns; ++i)
        printf("Number of matches for pattern <%s>: %llu\n", argv[i + 3], (unsigned long long)n_matches[i]);

    if (want_positions) { /* off by default: stdout stays identical to the reference */
        /* the text is mapped, not read a second time into a heap buffer */
        uint8_t *buf = NULL;
        uint64_t n = 0;
        const int fd = open(filename, O_RDONLY);
        struct stat st;
        if (fd >= 0 && fstat(fd, &st) == 0 && st.st_size > 0) {
            void *mp = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (mp != MAP_FAILED) {
                buf = (uint8_t *)mp;
                n = (uint64_t)st.st_size;
            }
        }
        if (fd >= 0) close(fd);
        if (!buf && fd >= 0 && st.st_size > 0) {
            fprintf(stderr, "Unable to map the text file for the positions pass\n");
            apm_destroy(ctx);
            return 1;
        }
        const uint64_t pcap = (uint64_t)1 << 20;
        uint64_t *pos = (uint64_t *)malloc((size_t)pcap * sizeof(uint64_t));
        for (int i = 0; pos && i < nb_patterns; ++i) {
            uint64_t found = 0;
            if (apm_find_buffer(ctx, buf, n, i, pos, pcap, &found) != APM_OK) {
                fprintf(stderr, "%s\n", apm_last_error(ctx));
                break;
            }
            printf("Positions for pattern <%s>:", argv[i + 3]);
            for (uint64_t q = 0; q < found && q < pcap; ++q) printf(" %llu", (unsigned long long)pos[q]);
            printf(found > pcap ? " ...\n" : "\n");
        }
        free(pos);
        if (buf) munmap(buf, (size_t)n);
    }

    apm_destroy(ctx);
    free(len);
    free(n_matches);
    return 0;
}
