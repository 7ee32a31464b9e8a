/* Runs the CPU oracle over small inputs under AddressSanitizer / UBSan (host code only). */
#include "../../include/sift_amd.h"
#include "../../oracle/sift_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int run_case(int w, int h, int pitch, int S, int O, int profile, int upsample, int dense, unsigned long long seed)
{
    OracleParams pr;
    oracle_default_params(&pr);
    pr.n_scales = S;
    pr.n_octaves = O;
    pr.max_points = 3000;
    pr.profile = profile;
    pr.upsample = upsample;
    float *img = (float *)malloc(sizeof(float) * (size_t)pitch * h);
    for (long i = 0; i < (long)pitch * h; ++i) img[i] = NAN; /* pad columns are never to be read */
    sift_synth_image(img, w, pitch, h, seed, dense);
    OracleSiftPoint *pts = (OracleSiftPoint *)malloc(sizeof(OracleSiftPoint) * (size_t)pr.max_points);
    unsigned char *meta = (unsigned char *)malloc(2 * (size_t)pr.max_points);
    int total = 0;
    const int n = oracle_extract(&pr, img, w, pitch, h, 0, pts, meta, &total);
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < 128; ++k)
            if (!(pts[j].data[k] >= 0.f && pts[j].data[k] <= 1.f)) {
                fprintf(stderr, "descriptor value out of range\n");
                return 1;
            }
    unsigned char *q = (unsigned char *)malloc(128 * (size_t)(n + 1));
    oracle_quantize_descriptors(pts, n, q);
    const long cap = 2048L * (n + 1);
    char *text = (char *)malloc((size_t)cap);
    if (oracle_format_points(pts, n, text, cap) < 0) return 1;
    if (!profile) {
        long amb[7];
        oracle_ambiguity_report(&pr, img, w, pitch, h, 1e-3f, amb);
        if (amb[0] != total) {
            fprintf(stderr, "ambiguity report counts %ld accepted points, extract %d\n", amb[0], total);
            return 1;
        }
    }
    printf("%4dx%-4d pitch %4d S=%d O=%d profile %d up %d: %d points\n", w, h, pitch, S, O, profile, upsample, n);
    free(text);
    free(q);
    free(meta);
    free(pts);
    free(img);
    return 0;
}

int main(void)
{
    int bad = 0;
    bad |= run_case(160, 120, 160, 2, 4, 0, 0, 1, 1);
    bad |= run_case(97, 70, 101, 3, 3, 0, 0, 1, 2);   /* ragged, pitched */
    bad |= run_case(20, 18, 20, 2, 4, 0, 0, 1, 3);    /* smaller than the border: nothing */
    bad |= run_case(200, 33, 256, 2, 5, 0, 0, 0, 4);  /* octaves shrink to a few rows */
    bad |= run_case(128, 96, 128, 3, 4, 1, 0, 1, 5);  /* refinement */
    bad |= run_case(100, 80, 100, 2, 3, 1, 1, 1, 6);  /* doubled first octave */
    /* literal block simulation against the closed form on one scale */
    {
        OracleParams pr;
        oracle_default_params(&pr);
        const int w = 230, h = 150, p = 256;
        const long g = (long)p * h;
        float *img = (float *)calloc((size_t)g, sizeof(float));
        sift_synth_image(img, w, p, h, 7, 1);
        float *tmp = (float *)calloc((size_t)g * 5, sizeof(float)), *blur = (float *)calloc((size_t)g * 5, sizeof(float));
        float *dog = (float *)calloc((size_t)g * 4, sizeof(float)), *hess = (float *)calloc((size_t)g * 6, sizeof(float));
        oracle_xblur(&pr, img, tmp, w, p, h);
        oracle_yblur(&pr, tmp, blur, w, p, h);
        oracle_dog(&pr, blur, dog, w, p, h);
        oracle_hessian(&pr, blur, hess, w, p, h);
        int *a = (int *)malloc(sizeof(int) * 2 * 5000), *b = (int *)malloc(sizeof(int) * 2 * 5000);
        const int na = oracle_find_extrema_closed(&pr, dog, hess, w, p, h, a, 5000);
        const int nb = oracle_find_extrema_literal(&pr, dog, hess, w, p, h, b, 5000);
        if (na != nb) {
            fprintf(stderr, "closed form %d points, literal simulation %d\n", na, nb);
            bad = 1;
        }
        printf("closed form and literal simulation: %d and %d points\n", na, nb);
        free(a), free(b), free(tmp), free(blur), free(dog), free(hess), free(img);
    }
    unsigned char rgb[30], gray[10];
    for (int i = 0; i < 30; ++i) rgb[i] = (unsigned char)(i * 8);
    oracle_gray_q14(rgb, 10, gray);
    printf(bad ? "FAILED\n" : "oracle driver finished\n");
    return bad;
}
