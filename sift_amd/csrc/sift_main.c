/*
 * Command-line front end: sift_cli <list file> <destination directory>
 * Counterpart of Sift/src/main.cc:12-44: the list file holds one image path
 * per line (binary PGM or PPM here; the reference reads whatever OpenCV
 * reads); every image gets <destination>/<file name>.sift.txt.
 * Options after the two arguments: --scales N, --octaves N, --cap N,
 * --profile ref|lowe, --upsample, --drop-last (the reference's count-1,
 * sift.cu:125).
 */
#include "../../include/sift_amd.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s <list file> <destination directory> [--scales N] [--octaves N] [--cap N] "
                        "[--profile ref|lowe] [--upsample] [--drop-last]\n", argv[0]);
        return 2;
    }
    sift_params prm;
    sift_default_params(&prm);
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
    if (n > 0) {
        rc = sift(argv[2], (const char **)paths, n);
        if (rc) fprintf(stderr, "sift: %s\n", sift_last_error());
    }
    for (int i = 0; i < n; ++i) free(paths[i]);
    free(paths);
    sift_shutdown();
    return rc ? 1 : 0;
}
