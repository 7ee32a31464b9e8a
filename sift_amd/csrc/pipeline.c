/*
 * Host pipeline around the extractor: image loading, the per-image-pair
 * upload / extract / read-back loop and the text writer.
 * Reference counterpart: Sift/src/sift.cu:64-163 and Sift/src/utils.cc:8-17.
 */
#include "../../include/sift_amd.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int g_drop_last = 0;

void sift_pipeline_drop_last_point(int on) { g_drop_last = on != 0; }

static int pnm_token(FILE *f, int *out)
{
    int c = fgetc(f);
    for (;;) {
        while (c != EOF && isspace(c)) c = fgetc(f);
        if (c == '#') {
            while (c != EOF && c != '\n') c = fgetc(f);
            continue;
        }
        break;
    }
    if (c == EOF || !isdigit(c)) return -1;
    int v = 0;
    while (c != EOF && isdigit(c)) {
        v = v * 10 + (c - '0');
        c = fgetc(f);
    }
    *out = v; /* the single whitespace after the token has been consumed */
    return 0;
}

/* Binary PGM (P5) or PPM (P6), 8 bit.  Colour is reduced to grey the way the
 * reference's imread(..., GRAYSCALE) does, as pinned by Sift/result/result.ppm:
 * Y = (1868*B + 9617*G + 4899*R + 8192) >> 14; then converted to float. */
int sift_load_pnm_gray(const char *path, float **out, int *w, int *h)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    int rc = -2;
    unsigned char *raw = NULL;
    do {
        if (fgetc(f) != 'P') break;
        int kind = fgetc(f);
        if (kind != '5' && kind != '6') break;
        int W, H, maxv;
        if (pnm_token(f, &W) || pnm_token(f, &H) || pnm_token(f, &maxv)) break;
        if (W < 1 || H < 1 || maxv < 1 || maxv > 255) break;
        const int ch = kind == '6' ? 3 : 1;
        const size_t n = (size_t)W * H;
        raw = (unsigned char *)malloc(n * ch);
        float *img = (float *)malloc(n * sizeof(float));
        if (!raw || !img || fread(raw, 1, n * ch, f) != n * ch) {
            free(img);
            break;
        }
        for (size_t i = 0; i < n; ++i) {
            if (ch == 1) {
                img[i] = (float)raw[i];
            } else {
                const int R = raw[3 * i], G = raw[3 * i + 1], B = raw[3 * i + 2];
                img[i] = (float)((1868 * B + 9617 * G + 4899 * R + 8192) >> 14);
            }
        }
        *out = img;
        *w = W;
        *h = H;
        rc = 0;
    } while (0);
    free(raw);
    fclose(f);
    return rc;
}

void sift_free_host(void *p) { free(p); }

/* sift.cu:151-160: x, y with one decimal, scale and orientation with three,
 * tab separated, then 128 values with six decimals each followed by a space */
long sift_format_points(const SiftPoint *pts, int n, char *buf, long cap)
{
    long off = 0;
    for (int j = 0; j < n; ++j) {
        if (cap - off < 2048) return -1;
        off += sprintf(buf + off, "%.1f\t%.1f\t%.3f\t%.3f\t", (double)pts[j].xpos, (double)pts[j].ypos,
                       (double)pts[j].scale, (double)pts[j].orientation);
        for (int k = 0; k < 128; ++k)
            off += sprintf(buf + off, "%.6f ", (double)pts[j].data[k]);
        buf[off++] = '\n';
    }
    return off;
}

/* sift.cu:144-163: <dst>/<basename of src>.sift.txt */
int sift_save_points(const char *dst_dir, const char *src_path, const SiftPoint *pts, int n)
{
    const char *base = src_path;
    for (const char *q = src_path; *q; ++q)
        if (*q == '/' || *q == '\\') base = q + 1;
    char name[4096];
    snprintf(name, sizeof name, "%s/%s.sift.txt", dst_dir, base);
    FILE *f = fopen(name, "w");
    if (!f) return -1;
    const long cap = 2048L * (n + 1);
    char *buf = (char *)malloc((size_t)cap);
    long len = buf ? sift_format_points(pts, n, buf, cap) : -1;
    int rc = -1;
    if (len >= 0 && fwrite(buf, 1, (size_t)len, f) == (size_t)len) rc = 0;
    free(buf);
    fclose(f);
    return rc;
}

/* sift.cu:64-143: images are taken two at a time, each on its own stream and
 * counter slot; upload, extract, read back, write one text file per image.
 * Unlike the reference the writer runs before the buffers are released and
 * every point is written (the reference passes count-1, sift.cu:125; call
 * sift_pipeline_drop_last_point(1) to reproduce that). */
int sift(const char *dst_dir, const char **src_paths, int n_imgs)
{
    sift_params prm;
    sift_get_params(&prm);
    int done = 0, rc = 0;
    while (done < n_imgs && rc == 0) {
        const int nq = (n_imgs - done > 1) ? 2 : 1;
        float *h_img[2] = {NULL, NULL}, *d_img[2] = {NULL, NULL};
        SiftPoint *d_pts[2] = {NULL, NULL}, *h_pts[2] = {NULL, NULL};
        void *stream[2] = {NULL, NULL};
        int w[2], h[2], pitch[2];
        for (int i = 0; i < nq && rc == 0; ++i) {
            if (sift_load_pnm_gray(src_paths[done + i], &h_img[i], &w[i], &h[i])) {
                fprintf(stderr, "Could not open or find the image %s\n", src_paths[done + i]);
                rc = -1;
                break;
            }
            rc |= sift_dev_malloc_pitch(&d_img[i], &pitch[i], w[i], h[i]);
            rc |= sift_dev_malloc((void **)&d_pts[i], sizeof(SiftPoint) * (size_t)prm.max_points);
            h_pts[i] = (SiftPoint *)malloc(sizeof(SiftPoint) * (size_t)prm.max_points);
            rc |= sift_stream_create(&stream[i]);
        }
        if (rc == 0) {
            initDeviceConstant();
            initDeviceVariables();
            for (int i = 0; i < nq; ++i)
                rc |= sift_memcpy2d_h2d_async(d_img[i], pitch[i], h_img[i], w[i], w[i], h[i], stream[i]);
            for (int i = 0; i < nq; ++i)
                extractSift(d_pts[i], d_img[i], w[i], pitch[i], h[i], stream[i], i, 0);
            for (int i = 0; i < nq && rc == 0; ++i) {
                int n = getPointCount(i);
                if (n < 0) {
                    rc = -1;
                    break;
                }
                rc |= sift_memcpy_d2h_async(h_pts[i], d_pts[i], sizeof(SiftPoint) * (size_t)n, stream[i]);
                rc |= sift_stream_sync(stream[i]);
                if (g_drop_last) n -= 1;
                if (n < 0) n = 0;
                if (rc == 0) rc = sift_save_points(dst_dir, src_paths[done + i], h_pts[i], n);
            }
        }
        for (int i = 0; i < nq; ++i) {
            if (stream[i]) sift_stream_destroy(stream[i]);
            if (d_img[i]) sift_dev_free(d_img[i]);
            if (d_pts[i]) sift_dev_free(d_pts[i]);
            free(h_pts[i]);
            free(h_img[i]);
        }
        done += nq;
    }
    return rc;
}

/* ---- visualisation ------------------------------------------------------------
 * The reference declares image::drawPoint (utils.h:32) and never defines it.
 * Keypoints are drawn into a colour copy of the grey image: a circle of radius
 * 3*scale around (xpos, ypos) and a spoke along the orientation, written as a
 * binary PPM.  pos_scale maps point coordinates to image pixels (1 for profile
 * lowe, whose positions are input pixels; for profile ref a point of octave o
 * lives on a grid 2^o times coarser, which the caller knows from the meta array
 * and passes per point through point_pos_scale when it is not NULL). */
static void put_px(unsigned char *rgb, int w, int h, int x, int y, const unsigned char c[3])
{
    if (x < 0 || x >= w || y < 0 || y >= h) return;
    unsigned char *p = rgb + 3L * (x + (long)w * y);
    p[0] = c[0];
    p[1] = c[1];
    p[2] = c[2];
}

int sift_draw_points(const char *path, const float *gray, int w, int pitch, int h,
                     const SiftPoint *pts, int n, const float *point_pos_scale)
{
    if (!path || !gray || w < 1 || h < 1 || pitch < w || n < 0 || (n && !pts)) return -1;
    unsigned char *rgb = (unsigned char *)malloc(3L * w * h);
    if (!rgb) return -1;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float v = gray[x + (long)pitch * y];
            unsigned char g = (unsigned char)(v < 0.f ? 0.f : (v > 255.f ? 255.f : v + 0.5f));
            rgb[3L * (x + (long)w * y)] = rgb[3L * (x + (long)w * y) + 1] = rgb[3L * (x + (long)w * y) + 2] = g;
        }
    static const unsigned char ring[3] = {255, 40, 40}, spoke[3] = {40, 255, 40};
    for (int j = 0; j < n; ++j) {
        const double ps = point_pos_scale ? point_pos_scale[j] : 1.0;
        const double cx = pts[j].xpos * ps, cy = pts[j].ypos * ps;
        double r = 3.0 * pts[j].scale;
        if (r < 2.0) r = 2.0;
        const int steps = (int)(8.0 * r) + 8;
        for (int k = 0; k < steps; ++k) {
            const double a = 6.283185307179586 * k / steps;
            put_px(rgb, w, h, (int)floor(cx + r * cos(a) + 0.5), (int)floor(cy + r * sin(a) + 0.5), ring);
        }
        const double th = pts[j].orientation * 3.141592653589793 / 180.0;
        for (int k = 0; k <= (int)r; ++k)
            put_px(rgb, w, h, (int)floor(cx + k * cos(th) + 0.5), (int)floor(cy + k * sin(th) + 0.5), spoke);
    }
    FILE *f = fopen(path, "wb");
    int rc = -1;
    if (f) {
        fprintf(f, "P6\n%d %d\n255\n", w, h);
        if (fwrite(rgb, 3, (size_t)w * h, f) == (size_t)w * h) rc = 0;
        fclose(f);
    }
    free(rgb);
    return rc;
}
