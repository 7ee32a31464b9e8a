/* Runs the C host code (sift_host.c, pipeline.c) against the CPU stand-in of the HIP layer under
 * AddressSanitizer / UBSan: arena planning, argument checks, launch-set splitting, slots, the
 * no-allocation entry, the loaders, the formatter, both file pipelines and their threads. */
#include "../../include/sift_amd.h"

#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(c)                                                            \
    do {                                                                    \
        if (!(c)) {                                                         \
            fprintf(stderr, "%s:%d: check failed: %s (last error: %s)\n", __FILE__, __LINE__, #c, sift_last_error()); \
            exit(1);                                                        \
        }                                                                   \
    } while (0)

static void write_pnm(const char *path, int kind, int w, int h, unsigned seed, long truncate)
{
    FILE *f = fopen(path, "wb");
    fprintf(f, "P%d\n# a comment line\n%d %d\n255\n", kind, w, h);
    long n = (long)w * h * (kind == 6 ? 3 : 1);
    if (truncate >= 0 && truncate < n) n = truncate;
    for (long i = 0; i < n; ++i) fputc((int)((seed = seed * 1664525u + 1013904223u) >> 24), f);
    fclose(f);
}

extern int hl_stub_malloc_fail_at; /* hip_layer_stub.c */

static void *device_thread(void *arg)
{
    const int dev = (int)(intptr_t)arg;
    CHECK(sift_set_device(dev) == 0);
    CHECK(sift_get_device() == dev);
    float *img;
    SiftPoint *pts;
    CHECK(sift_dev_malloc((void **)&img, sizeof(float) * 300 * 200) == 0);
    CHECK(sift_dev_malloc((void **)&pts, sizeof(SiftPoint) * 2500) == 0);
    initDeviceVariables();
    for (int rep = 0; rep < 20; ++rep) {
        extractSift(pts, img, 300 - rep, 300, 200 - rep, NULL, rep % SIFT_MAX_SLOTS, 0);
        CHECK(sift_last_error()[0] == 0);
        CHECK(getPointCount(rep % SIFT_MAX_SLOTS) == 0);
    }
    {   /* the entry that allocates nothing, on this device's copy of the static buffer, and a batch
         * through this device's lane arenas */
        unsigned *cnt;
        int hc[3];
        CHECK(sift_dev_malloc((void **)&cnt, sizeof(unsigned)) == 0);
        CHECK(sift_extract_noalloc(pts, cnt, img, 300, 300, 200, NULL) == 0);
        CHECK(sift_extract_noalloc(pts, cnt, img, 150, 300, 100, NULL) == 0);
        SiftPoint *bp;
        CHECK(sift_dev_malloc((void **)&bp, sizeof(SiftPoint) * 2500 * 3) == 0);
        CHECK(sift_extract_batch(bp, NULL, hc, img, 100L * 200, 3, 100, 100, 200, NULL) == 0);
        CHECK(hc[0] == 0 && hc[2] == 0);
        sift_dev_free(bp);
        sift_dev_free(cnt);
    }
    sift_shutdown();
    sift_dev_free(img);
    sift_dev_free(pts);
    return NULL;
}

/* rewrites the (unchanged) parameters and reads them back while the device threads extract:
 * writers and readers of the parameters meet on the parameter lock */
static void *params_thread(void *arg)
{
    (void)arg;
    for (int rep = 0; rep < 200; ++rep) {
        sift_params p;
        sift_get_params(&p);
        CHECK(sift_set_params(&p) == 0);
    }
    return NULL;
}

int main(int argc, char **argv)
{
    const char *tmp = argc > 1 ? argv[1] : "/tmp";
    char path[8][512], dst[512];
    sift_params prm;
    sift_default_params(&prm);
    /* parameters */
    prm.n_scales = 5;
    CHECK(sift_set_params(&prm) != 0);
    sift_clear_error();
    sift_default_params(&prm);
    prm.upsample = 1;
    CHECK(sift_set_params(&prm) != 0); /* needs profile lowe */
    sift_clear_error();
    sift_default_params(&prm);
    CHECK(sift_set_params(&prm) == 0);
    float blur[6 * 19], wnd[3 * 16], dw[16];
    sift_get_tables(blur, wnd, dw);
    float sum = 0;
    for (int k = 0; k < 19; ++k) sum += blur[k];
    CHECK(fabsf(sum - 1.0f) < 1e-5f);

    /* reference entry points */
    float *d_img;
    int pitch;
    SiftPoint *d_pts;
    CHECK(sift_dev_malloc_pitch(&d_img, &pitch, 801, 603) == 0 && pitch >= 801);
    CHECK(sift_dev_malloc((void **)&d_pts, sizeof(SiftPoint) * 2500) == 0);
    initDeviceConstant();
    initDeviceVariables();
    void *stream;
    CHECK(sift_stream_create(&stream) == 0);
    extractSift(d_pts, d_img, 801, pitch, 603, stream, 0, 0);
    CHECK(sift_last_error()[0] == 0 && getPointCount(0) == 0);
    {   /* two calls in a row on one slot from two streams, no getPointCount in between: the second
         * is ordered behind the first through the slot's event */
        void *stream2;
        CHECK(sift_stream_create(&stream2) == 0);
        extractSift(d_pts, d_img, 801, pitch, 603, stream, 0, 0);
        extractSift(d_pts, d_img, 801, pitch, 603, stream2, 0, 0);
        CHECK(sift_last_error()[0] == 0 && getPointCount(0) == 0);
        initDeviceVariables();
        CHECK(getPointCount(0) == 0 && getPointCount(1) == 0);
        CHECK(sift_stream_destroy(stream2) == 0);
    }
    CHECK(sift_stream_destroy(stream) == 0);
    {   /* an allocation that fails half-way through a slot's arena leaves the slot empty (nothing
         * leaks, and the next call plans it afresh) */
        for (int k = 0; k < 5; ++k) {
            hl_stub_malloc_fail_at = k;
            extractSift(d_pts, d_img, 640 + k, pitch, 400, NULL, 3, 0);
            CHECK(sift_last_error()[0] != 0);
            sift_clear_error();
            hl_stub_malloc_fail_at = -1;
        }
        extractSift(d_pts, d_img, 640, pitch, 400, NULL, 3, 0);
        CHECK(sift_last_error()[0] == 0 && getPointCount(3) == 0);
    }
    extractSift(d_pts, d_img, 400, pitch, 300, NULL, 0, 1); /* other geometry, same slot, start octave 1 */
    CHECK(sift_last_error()[0] == 0 && getPointCount(0) == 0);
    extractSift(d_pts, d_img, 900, pitch, 300, NULL, 0, 0); /* pitch < width */
    CHECK(sift_last_error()[0] != 0);
    sift_clear_error();
    extractSift(d_pts, d_img, 400, pitch, 300, NULL, 99, 0);
    CHECK(sift_last_error()[0] != 0);
    sift_clear_error();
    unsigned acc[32], que[32];
    CHECK(sift_get_group_counts(0, acc, que, 32) == 6); /* octaves 1..3, two scales each */
    int gw, gp, gh;
    CHECK(sift_debug_read_planes(0, 0, 1, NULL, 0, &gw, &gp, &gh) == 0 && gw == 200 && gh == 150);
    float *planes = (float *)malloc(sizeof(float) * (size_t)gp * gh * 5);
    CHECK(sift_debug_read_planes(0, 0, 1, planes, (long)gp * gh * 5, &gw, &gp, &gh) == 0);
    CHECK(sift_debug_read_planes(0, 0, 1, planes, 10, &gw, &gp, &gh) != 0);
    sift_clear_error();
    free(planes);

    /* batch entries: one set, several sets, 8-bit, asynchronous, both profiles */
    const int n = 70, w = 200, h = 120;
    float *d_batch;
    SiftPoint *d_bpts;
    unsigned *d_counts;
    unsigned char *d_meta;
    int counts[70];
    CHECK(sift_dev_malloc((void **)&d_batch, sizeof(float) * (size_t)n * w * h) == 0);
    CHECK(sift_dev_malloc((void **)&d_bpts, sizeof(SiftPoint) * 2500 * (size_t)n) == 0);
    CHECK(sift_dev_malloc((void **)&d_counts, sizeof(unsigned) * (size_t)n) == 0);
    CHECK(sift_dev_malloc((void **)&d_meta, 2 * 2500 * (size_t)n) == 0);
    CHECK(sift_extract_batch(d_bpts, d_meta, counts, d_batch, (long)w * h, n, w, w, h, NULL) == 0);
    sift_set_batch_chunk(16);
    CHECK(sift_extract_batch(d_bpts, NULL, counts, d_batch, (long)w * h, n, w, w, h, NULL) == 0);
    int slot, image;
    CHECK(sift_debug_batch_slot(n, 69, &slot, &image) == 0 && image == 69 % 14);
    CHECK(sift_debug_batch_slot(n, 3, &slot, &image) != 0); /* that set's arena has been reused */
    sift_set_batch_lanes(1);
    CHECK(sift_extract_batch_u8(d_bpts, NULL, counts, (const unsigned char *)d_batch, (long)w * h, n, w, w, h, NULL) == 0);
    sift_set_batch_lanes(2);
    sift_set_batch_chunk(256);
    CHECK(sift_extract_batch_async(d_bpts, NULL, d_counts, d_batch, 0, (long)w * h, n, w, w, h, NULL) == 0);
    CHECK(sift_debug_batch_slot(n, 5, &slot, &image) == 0 && slot == SIFT_MAX_SLOTS && image == 5);
    /* the next non-blocking call starts on the other lane, and the slot reported follows it */
    CHECK(sift_extract_batch_async(d_bpts, NULL, d_counts, d_batch, 0, (long)w * h, n, w, w, h, NULL) == 0);
    CHECK(sift_debug_batch_slot(n, 5, &slot, &image) == 0 && slot == SIFT_MAX_SLOTS + 1 && image == 5);
    CHECK(sift_debug_batch_slot(n - 1, 5, &slot, &image) != 0); /* not the most recent batch's size */
    CHECK(sift_extract_batch_async(d_bpts, NULL, NULL, d_batch, 0, (long)w * h, n, w, w, h, NULL) != 0);
    CHECK(sift_extract_batch(d_bpts, NULL, counts, d_batch, 10, n, w, w, h, NULL) != 0); /* stride too small */
    sift_clear_error();
    prm.profile = SIFT_PROFILE_LOWE;
    prm.upsample = 1;
    prm.n_scales = 3;
    CHECK(sift_set_params(&prm) == 0);
    CHECK(sift_extract_batch(d_bpts, NULL, counts, d_batch, (long)w * h, 9, w, w, h, NULL) == 0);
    sift_set_queue_cap(0);
    CHECK(sift_extract_batch(d_bpts, NULL, counts, d_batch, (long)w * h, 9, w, w, h, NULL) == 0);
    sift_set_queue_cap(65536);
    unsigned char *d_q;
    CHECK(sift_dev_malloc((void **)&d_q, 128 * 10) == 0);
    CHECK(sift_quantize_descriptors(d_bpts, 10, d_q, NULL) == 0);
    sift_default_params(&prm);
    CHECK(sift_set_params(&prm) == 0);

    /* the entry that allocates nothing */
    CHECK(sift_extract_noalloc(d_bpts, d_counts, d_batch, w, w, h, NULL) == 0);
    CHECK(sift_extract_noalloc(d_bpts, d_counts, d_img, 801, pitch, 603, NULL) == 0);
    prm.n_octaves = 5;
    CHECK(sift_set_params(&prm) == 0);
    {
        float *big;
        CHECK(sift_dev_malloc((void **)&big, sizeof(float) * 1920 * 1080) == 0);
        CHECK(sift_extract_noalloc(d_bpts, d_counts, big, 1920, 1920, 1080, NULL) != 0);
        CHECK(strstr(sift_last_error(), "static buffer holds") != NULL);
        sift_clear_error();
        sift_dev_free(big);
    }
    sift_default_params(&prm);
    CHECK(sift_set_params(&prm) == 0);

    /* stage launchers */
    {
        float *stage;
        CHECK(sift_dev_malloc((void **)&stage, sizeof(float) * 256 * 100 * 6) == 0);
        blurOctave(stage, d_batch, 200, 256, 100, NULL);
        computeDiffOfGauss(stage, stage, 200, 256, 100, NULL);
        computeHessian(stage, stage, 200, 256, 100, NULL);
        computeGradient(stage, stage, 200, 256, 100, NULL);
        copyDeviceData(stage, stage, 200, 256, 100, NULL);
        CHECK(sift_last_error()[0] == 0);
        blurOctave(stage, d_batch, 200, 255, 100, NULL); /* pitch not a multiple of 4 */
        CHECK(sift_last_error()[0] != 0);
        sift_clear_error();
        sift_dev_free(stage);
    }

    /* loader: grey, colour, comments, truncated, wrong magic, missing */
    snprintf(path[0], sizeof path[0], "%s/a.pgm", tmp);
    snprintf(path[1], sizeof path[1], "%s/b.ppm", tmp);
    snprintf(path[2], sizeof path[2], "%s/c.pgm", tmp);
    snprintf(path[3], sizeof path[3], "%s/d.pgm", tmp);
    snprintf(path[4], sizeof path[4], "%s/short.pgm", tmp);
    snprintf(path[5], sizeof path[5], "%s/magic.pgm", tmp);
    snprintf(path[6], sizeof path[6], "%s/missing.pgm", tmp);
    write_pnm(path[0], 5, 123, 77, 1, -1);
    write_pnm(path[1], 6, 123, 77, 2, -1);
    write_pnm(path[2], 5, 64, 48, 3, -1);
    write_pnm(path[3], 5, 123, 77, 4, -1);
    write_pnm(path[4], 5, 50, 50, 5, 100);
    {
        FILE *f = fopen(path[5], "wb");
        fputs("P3\n2 2\n255\n0 0 0 0", f);
        fclose(f);
    }
    float *g;
    int lw, lh;
    CHECK(sift_load_pnm_gray(path[0], &g, &lw, &lh) == 0 && lw == 123 && lh == 77);
    sift_free_host(g);
    CHECK(sift_load_pnm_gray(path[1], &g, &lw, &lh) == 0 && g[0] >= 0.f && g[0] <= 255.f);
    SiftPoint two[2];
    memset(two, 0, sizeof two);
    two[0].xpos = 40, two[0].ypos = 30, two[0].scale = 3.2f, two[0].orientation = 45.f;
    two[1].xpos = 10, two[1].ypos = 70, two[1].scale = 1.6f, two[1].orientation = 348.75f;
    snprintf(dst, sizeof dst, "%s/drawn.ppm", tmp);
    CHECK(sift_draw_points(dst, g, lw, lw, lh, two, 2, NULL) == 0);
    sift_sort_points(two, 2);
    CHECK(two[0].scale == 1.6f);
    sift_free_host(g);
    CHECK(sift_load_pnm_gray(path[4], &g, &lw, &lh) != 0);
    CHECK(sift_load_pnm_gray(path[5], &g, &lw, &lh) != 0);
    CHECK(sift_load_pnm_gray(path[6], &g, &lw, &lh) != 0);

    /* formatter against printf, on awkward values */
    {
        SiftPoint p;
        memset(&p, 0, sizeof p);
        uint32_t st = 12345u;
        char buf[4096], want[64];
        for (int rep = 0; rep < 2000; ++rep) {
            for (int k = 0; k < 128; ++k) {
                st = st * 1664525u + 1013904223u;
                uint32_t bits = st;
                if (rep % 4 == 0) bits &= 0x3fffffffu;      /* below 2 */
                if (rep % 4 == 1) bits = (bits >> 9);         /* denormals */
                if (rep % 4 == 2) bits = 0x3a800000u + (bits & 0xffffu); /* near ties at the sixth digit */
                memcpy(&p.data[k], &bits, 4);
            }
            p.data[0] = 0.5f, p.data[1] = 999.9999f, p.data[2] = 1000.0f, p.data[3] = -0.25f, p.data[4] = INFINITY, p.data[5] = NAN;
            const long len = sift_format_points(&p, 1, buf, sizeof buf);
            CHECK(len > 0);
            buf[len] = 0;
            const char *q = buf;
            for (int t = 0; t < 4; ++t) q = strchr(q, '\t') + 1;
            for (int k = 0; k < 128; ++k) {
                snprintf(want, sizeof want, "%.6f ", (double)p.data[k]);
                CHECK(strncmp(q, want, strlen(want)) == 0);
                q += strlen(want);
            }
        }
        CHECK(sift_format_points(&p, 1, buf, 100) == -1);
    }

    /* the pairwise pipeline and the threaded one, on the pretend devices */
    const char *list[5] = {path[0], path[1], path[2], path[3], path[0]};
    CHECK(sift(tmp, list, 5) == 0);
    sift_pipeline_drop_last_point(1);
    CHECK(sift(tmp, list, 3) == 0);
    sift_pipeline_drop_last_point(0);
    int lc[5];
    for (int workers = 1; workers <= 4; ++workers) { /* 2 and 4 workers: both pretend devices, one and two workers each */
        CHECK(sift_extract_list_multi(tmp, list, 5, workers, lc) == 0);
        for (int i = 0; i < 5; ++i) CHECK(lc[i] == 0);
    }
    CHECK(sift_extract_list_multi(NULL, list, 5, 2, NULL) == 0);
    const char *bad_list[3] = {path[0], path[6], path[2]};
    sift_clear_error();
    CHECK(sift_extract_list_multi(tmp, bad_list, 3, 1, lc) != 0);
    CHECK(strstr(sift_last_error(), "missing.pgm") != NULL); /* the worker's message reaches this thread */
    const char *bad_list2[2] = {path[4], path[0]};
    sift_clear_error();
    CHECK(sift_extract_list_multi(tmp, bad_list2, 2, 2, lc) != 0);
    CHECK(sift_last_error()[0] != 0);
    sift_clear_error();
    CHECK(sift(tmp, bad_list, 3) != 0);
    CHECK(sift(tmp, bad_list2, 2) != 0);
    CHECK(sift_extract_list_multi(tmp, list, 0, 1, NULL) == 0);
    CHECK(sift_list_partition(13, 4) == 1);

    /* arenas given back and planned afresh: stream slots and batch lanes, blocking (the caller's
     * stream) and non-blocking (the private streams) calls in turn on one lane's arena, and the
     * two detection routes (a one-set job under the pixel limit skips the queues) */
    CHECK(sift_release_arenas() == 0);
    CHECK(sift_debug_batch_slot(n, 5, &slot, &image) != 0);
    sift_clear_error();
    extractSift(d_pts, d_img, 640, pitch, 400, NULL, 2, 0);
    CHECK(getPointCount(2) >= 0 && sift_last_error()[0] == 0);
    for (long lim = 0; lim <= 4000000; lim += 4000000) {
        sift_set_dense_route_pixels(lim);
        CHECK(sift_extract_batch(d_bpts, NULL, counts, d_batch, (long)w * h, n, w, w, h, NULL) == 0);
        CHECK(sift_extract_batch_async(d_bpts, NULL, d_counts, d_batch, 0, (long)w * h, n, w, w, h, NULL) == 0);
        CHECK(sift_extract_batch(d_bpts, NULL, counts, d_batch, (long)w * h, n, w, w, h, NULL) == 0);
        CHECK(sift_extract_batch_async(d_bpts, NULL, d_counts, d_batch, 0, (long)w * h, n, w, w, h, NULL) == 0);
        CHECK(sift_extract_batch_async(d_bpts, NULL, d_counts, d_batch, 0, (long)w * h, n, w, w, h, NULL) == 0);
        CHECK(sift_extract_batch(d_bpts, NULL, counts, d_batch, (long)w * h, n, w, w, h, NULL) == 0);
    }
    sift_set_dense_route_pixels(2200000);
    CHECK(sift_release_arenas() == 0);
    CHECK(sift_release_arenas() == 0); /* nothing left to free */

    /* two host threads, one per pretend device */
    pthread_t th[3];
    for (intptr_t d = 0; d < 2; ++d) pthread_create(&th[d], NULL, device_thread, (void *)d);
    pthread_create(&th[2], NULL, params_thread, NULL);
    for (int d = 0; d < 3; ++d) pthread_join(th[d], NULL);

    double gbps;
    CHECK(sift_measure_copy_ceiling(1 << 20, 1, &gbps) == 0);
    sift_dev_free(d_q);
    sift_dev_free(d_meta);
    sift_dev_free(d_counts);
    sift_dev_free(d_bpts);
    sift_dev_free(d_batch);
    sift_dev_free(d_pts);
    sift_dev_free(d_img);
    sift_set_device(0);
    sift_shutdown();
    printf("host driver finished\n");
    return 0;
}
