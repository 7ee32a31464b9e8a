/*
 * Command-line front end: sift_cli <list file> <destination directory>
 * Counterpart of Sift/src/main.cc:12-44: the list file holds one image path
 * per line (binary PGM or PPM here; the reference reads whatever OpenCV
 * reads); every image gets <destination>/<file name>.sift.txt.
 * Options after the two arguments: --scales N, --octaves N, --cap N,
 * --profile ref|lowe, --upsample, --drop-last (the reference's count-1,
 * sift.cu:125), --gpus N (the list dealt over N GPUs, image i on GPU i mod N;
 * default 1), --pairwise (the reference's loop as it stands: two images at a
 * time, sift.cu:68-134), --timing (seconds the list took, on stderr).
 */
#include "../../include/sift_amd.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s <list file> <destination directory> [--scales N] [--octaves N] [--cap N] "
                        "[--profile ref|lowe] [--upsample] [--drop-last] [--gpus N] [--pairwise] [--timing]\n", argv[0]);
        return 2;
    }
    sift_params prm;
    sift_default_params(&prm);
    int gpus = 1, pairwise = 0, timing = 0, repeat = 1;
    for (int i = 3; i < argc; ++i) {
        if (!strcmp(argv[i], "--scales") && i + 1 < argc)
            prm.n_scales = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--octaves") && i + 1 < argc)
            prm.n_octaves = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--cap") && i + 1 < argc)
            prm.max_points = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--profile") && i + 1 < argc)
            prm.profile = !strcmp(argv[++i], "lowe") ? SIFT_PROFILE_LOWE : SIFT_PROFILE_REF;
        else if (!strcmp(argv[i], "--upsample"))
            prm.upsample = 1;
        else if (!strcmp(argv[i], "--drop-last"))
            sift_pipeline_drop_last_point(1);
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc)
            gpus = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--pairwise"))
            pairwise = 1;
        else if (!strcmp(argv[i], "--timing"))
            timing = 1;
        else if (!strcmp(argv[i], "--repeat") && i + 1 < argc) /* the list again (timing of a warm process) */
            repeat = atoi(argv[++i]);
        else {
            fprintf(stderr, "unknown option %s\n", argv[i]);
            return 2;
        }
    }
    if (sift_set_params(&prm)) {
        fprintf(stderr, "%s\n", sift_last_error());
        return 2;
    }
    FILE *f = fopen(argv[1], "r");
    if (!f) {
        fprintf(stderr, "Unable to open file %s\n", argv[1]); /* main.cc:17-21 */
        return 1;
    }
    char **paths = NULL;
    int n = 0, cap = 0;
    char line[4096];
    while (fgets(line, sizeof line, f)) {
        size_t len = strlen(line);
        while (len && (line[len - 1] == '\n' || line[len - 1] == '\r' || line[len - 1] == ' ')) line[--len] = 0;
        if (!len) continue;
        if (n == cap) {
            cap = cap ? 2 * cap : 16;
            paths = (char **)realloc(paths, sizeof(char *) * (size_t)cap);
        }
        paths[n++] = strdup(line);
    }
    fclose(f);
    int rc = 0;
    for (int rep = 0; rep < repeat && n > 0 && rc == 0; ++rep) {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        rc = pairwise ? sift(argv[2], (const char **)paths, n)
                      : sift_extract_list_multi(argv[2], (const char **)paths, n, gpus < 1 ? 1 : gpus, NULL);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (rc) fprintf(stderr, "sift: %s\n", sift_last_error());
        if (timing) {
            const double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
            fprintf(stderr, "pass %d: %d images in %.3f s = %.1f images/s\n", rep, n, dt, n / dt);
        }
    }
    for (int i = 0; i < n; ++i) free(paths[i]);
    free(paths);
    sift_shutdown();
    return rc ? 1 : 0;
}
