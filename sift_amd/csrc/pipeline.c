/*
 * Host pipeline around the extractor: image loading, the per-image-pair
 * upload / extract / read-back loop and the text writer.
 * Reference counterpart: Sift/src/sift.cu:64-163 and Sift/src/utils.cc:8-17.
 */
#include "../../include/sift_amd.h"

#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
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

/* a buffer that only grows: the pipeline keeps its buffers from image to image */
typedef struct {
    void *p;
    size_t cap;
    int pinned;
} grow_buf;

static int grow(grow_buf *b, size_t bytes)
{
    if (b->cap >= bytes) return 0;
    if (b->p) {
        if (b->pinned) sift_host_free_pinned(b->p);
        else free(b->p);
        b->p = NULL;
        b->cap = 0;
    }
    if (b->pinned) {
        if (sift_host_alloc_pinned(&b->p, bytes)) return -1;
    } else {
        b->p = malloc(bytes);
        if (!b->p) return -1;
    }
    b->cap = bytes;
    return 0;
}

static void release(grow_buf *b)
{
    if (b->p) {
        if (b->pinned) sift_host_free_pinned(b->p);
        else free(b->p);
    }
    b->p = NULL;
    b->cap = 0;
}

/* Binary PGM (P5) or PPM (P6), 8 bit, into img (grown as needed).  Colour is reduced to
 * grey the way the reference's imread(..., GRAYSCALE) does, as pinned by
 * Sift/result/result.ppm: Y = (1868*B + 9617*G + 4899*R + 8192) >> 14; then converted
 * to float (utils.cc:13). */
static int read_pnm_gray(const char *path, grow_buf *raw, grow_buf *img, int *w, int *h)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    int rc = -2;
    do {
        if (fgetc(f) != 'P') break;
        int kind = fgetc(f);
        if (kind != '5' && kind != '6') break;
        int W, H, maxv;
        if (pnm_token(f, &W) || pnm_token(f, &H) || pnm_token(f, &maxv)) break;
        if (W < 1 || H < 1 || maxv < 1 || maxv > 255) break;
        const int ch = kind == '6' ? 3 : 1;
        const size_t n = (size_t)W * H;
        if (grow(raw, n * ch) || grow(img, n * sizeof(float))) break;
        const unsigned char *r = (const unsigned char *)raw->p;
        float *o = (float *)img->p;
        if (fread(raw->p, 1, n * ch, f) != n * ch) break;
        if (ch == 1) {
            for (size_t i = 0; i < n; ++i) o[i] = (float)r[i];
        } else {
            for (size_t i = 0; i < n; ++i) {
                const int R = r[3 * i], G = r[3 * i + 1], B = r[3 * i + 2];
                o[i] = (float)((1868 * B + 9617 * G + 4899 * R + 8192) >> 14);
            }
        }
        *w = W;
        *h = H;
        rc = 0;
    } while (0);
    fclose(f);
    return rc;
}

int sift_load_pnm_gray(const char *path, float **out, int *w, int *h)
{
    grow_buf raw = {NULL, 0, 0}, img = {NULL, 0, 0};
    const int rc = read_pnm_gray(path, &raw, &img, w, h);
    release(&raw);
    if (rc) {
        release(&img);
        return rc;
    }
    *out = (float *)img.p;
    return 0;
}

void sift_free_host(void *p) { free(p); }

/* "%.6f" of v, the bytes printf would produce: the exact binary value times 10^6, rounded to
 * nearest with ties to even, by integer arithmetic.  Values outside [0, 1000) (never a
 * descriptor bin) go through printf itself. */
static int format_6f(char *out, float v)
{
    uint32_t bits;
    memcpy(&bits, &v, sizeof bits);
    const uint32_t ex = (bits >> 23) & 0xffu;
    if ((bits >> 31) || ex == 0xffu || !(v < 1000.0f)) return sprintf(out, "%.6f", (double)v);
    /* v = m * 2^e exactly */
    const uint64_t m = ex ? ((bits & 0x7fffffu) | 0x800000u) : (bits & 0x7fffffu);
    const int e = ex ? (int)ex - 150 : -149;
    const uint64_t num = m * 1000000ull; /* below 2^44 */
    uint64_t q;
    if (e >= 0) {
        q = num << e; /* not reached below 1000: kept for completeness */
    } else if (-e >= 64) {
        q = 0; /* num is below half of 2^-e */
    } else {
        const int sh = -e;
        const uint64_t rem = num & ((1ull << sh) - 1ull), half = 1ull << (sh - 1);
        q = num >> sh;
        if (rem > half || (rem == half && (q & 1ull))) ++q;
    }
    const uint32_t ip = (uint32_t)(q / 1000000ull);
    uint32_t fp = (uint32_t)(q % 1000000ull);
    int len = 0;
    if (ip >= 100) out[len++] = (char)('0' + ip / 100);
    if (ip >= 10) out[len++] = (char)('0' + ip / 10 % 10);
    out[len++] = (char)('0' + ip % 10);
    out[len++] = '.';
    for (int k = 5; k >= 0; --k) {
        out[len + k] = (char)('0' + fp % 10);
        fp /= 10;
    }
    return len + 6;
}

/* sift.cu:151-160: x, y with one decimal, scale and orientation with three,
 * tab separated, then 128 values with six decimals each followed by a space */
long sift_format_points(const SiftPoint *pts, int n, char *buf, long cap)
{
    long off = 0;
    for (int j = 0; j < n; ++j) {
        if (cap - off < 256) return -1; /* four values of at most 48 characters each */
        off += sprintf(buf + off, "%.1f\t%.1f\t%.3f\t%.3f\t", (double)pts[j].xpos, (double)pts[j].ypos,
                       (double)pts[j].scale, (double)pts[j].orientation);
        for (int k = 0; k < 128; ++k) {
            if (cap - off < 64) return -1;
            off += format_6f(buf + off, pts[j].data[k]);
            buf[off++] = ' ';
        }
        buf[off++] = '\n';
    }
    return off;
}

static int save_points_buf(const char *dst_dir, const char *src_path, const SiftPoint *pts, int n,
                           grow_buf *text)
{
    const char *base = src_path;
    for (const char *q = src_path; *q; ++q)
        if (*q == '/' || *q == '\\') base = q + 1;
    char name[4096];
    snprintf(name, sizeof name, "%s/%s.sift.txt", dst_dir, base);
    const long cap = 2048L * (n + 1);
    if (grow(text, (size_t)cap)) return -1;
    const long len = sift_format_points(pts, n, (char *)text->p, cap);
    if (len < 0) return -1;
    FILE *f = fopen(name, "w");
    if (!f) return -1;
    const int rc = fwrite(text->p, 1, (size_t)len, f) == (size_t)len ? 0 : -1;
    return fclose(f) ? -1 : rc;
}

/* sift.cu:144-163: <dst>/<basename of src>.sift.txt */
int sift_save_points(const char *dst_dir, const char *src_path, const SiftPoint *pts, int n)
{
    grow_buf text = {NULL, 0, 0};
    const int rc = save_points_buf(dst_dir, src_path, pts, n, &text);
    release(&text);
    return rc;
}

/* sift.cu:64-143: images are taken two at a time, each on its own stream and
 * counter slot; upload, extract, read back, write one text file per image.
 * Each image of a pair has its own host thread for the two host-side stages
 * (decode before, format and write after the extraction), and the buffers are kept
 * from pair to pair: the reference allocates and frees everything per pair, and its host
 * stages take a hundred times longer than the extraction here.  Unlike the reference
 * the writer runs before the buffers are released and every point is written (the
 * reference passes count-1, sift.cu:125; call sift_pipeline_drop_last_point(1) to
 * reproduce that). */
typedef struct {
    /* kept from pair to pair */
    grow_buf raw, img, text, h_pts; /* img and h_pts are pinned */
    float *d_img;
    size_t d_img_floats;
    SiftPoint *d_pts;
    void *stream;
    /* the current image */
    const char *path, *dst_dir;
    int w, h, pitch, n, rc;
} lane_t;

static void *lane_load(void *arg)
{
    lane_t *L = (lane_t *)arg;
    L->rc = read_pnm_gray(L->path, &L->raw, &L->img, &L->w, &L->h);
    return NULL;
}

static void *lane_save(void *arg)
{
    lane_t *L = (lane_t *)arg;
    L->rc = save_points_buf(L->dst_dir, L->path, (const SiftPoint *)L->h_pts.p, L->n, &L->text);
    return NULL;
}

/* runs fn for the nq lanes, the second on a thread of its own */
static void run_lanes(lane_t *L, int nq, void *(*fn)(void *))
{
    pthread_t th;
    int threaded = 0;
    if (nq > 1) threaded = pthread_create(&th, NULL, fn, &L[1]) == 0;
    fn(&L[0]);
    if (nq > 1) {
        if (threaded) pthread_join(th, NULL);
        else fn(&L[1]);
    }
}

int sift(const char *dst_dir, const char **src_paths, int n_imgs)
{
    sift_params prm;
    sift_get_params(&prm);
    lane_t L[2];
    memset(L, 0, sizeof L);
    for (int i = 0; i < 2; ++i) L[i].img.pinned = L[i].h_pts.pinned = 1;
    const size_t pts_bytes = sizeof(SiftPoint) * (size_t)prm.max_points;
    int done = 0, rc = 0;
    while (done < n_imgs && rc == 0) {
        const int nq = (n_imgs - done > 1) ? 2 : 1;
        for (int i = 0; i < nq; ++i) {
            L[i].path = src_paths[done + i];
            L[i].dst_dir = dst_dir;
        }
        run_lanes(L, nq, lane_load);
        for (int i = 0; i < nq && rc == 0; ++i) {
            if (L[i].rc) {
                fprintf(stderr, "Could not open or find the image %s\n", L[i].path);
                rc = -1;
                break;
            }
            const size_t need = (size_t)L[i].w * L[i].h;
            if (L[i].d_img_floats < need) {
                if (L[i].d_img) sift_dev_free(L[i].d_img);
                L[i].d_img = NULL;
                L[i].d_img_floats = 0;
                rc |= sift_dev_malloc((void **)&L[i].d_img, need * sizeof(float));
                if (!rc) L[i].d_img_floats = need;
            }
            L[i].pitch = L[i].w;
            if (!L[i].d_pts) rc |= sift_dev_malloc((void **)&L[i].d_pts, pts_bytes);
            if (grow(&L[i].h_pts, pts_bytes)) rc = -1;
            if (!L[i].stream) rc |= sift_stream_create(&L[i].stream);
        }
        if (rc == 0) {
            initDeviceConstant();
            initDeviceVariables();
            for (int i = 0; i < nq; ++i)
                rc |= sift_memcpy2d_h2d_async(L[i].d_img, L[i].pitch, (const float *)L[i].img.p, L[i].w,
                                              L[i].w, L[i].h, L[i].stream);
            for (int i = 0; i < nq; ++i)
                extractSift(L[i].d_pts, L[i].d_img, L[i].w, L[i].pitch, L[i].h, L[i].stream, i, 0);
            for (int i = 0; i < nq && rc == 0; ++i) {
                int n = getPointCount(i);
                if (n < 0) {
                    rc = -1;
                    break;
                }
                rc |= sift_memcpy_d2h_async(L[i].h_pts.p, L[i].d_pts, sizeof(SiftPoint) * (size_t)n, L[i].stream);
                rc |= sift_stream_sync(L[i].stream);
                if (g_drop_last) n -= 1;
                L[i].n = n < 0 ? 0 : n;
            }
            if (rc == 0) {
                run_lanes(L, nq, lane_save);
                for (int i = 0; i < nq; ++i) rc |= L[i].rc;
            }
        }
        done += nq;
    }
    for (int i = 0; i < 2; ++i) {
        if (L[i].stream) sift_stream_destroy(L[i].stream);
        if (L[i].d_img) sift_dev_free(L[i].d_img);
        if (L[i].d_pts) sift_dev_free(L[i].d_pts);
        release(&L[i].raw);
        release(&L[i].img);
        release(&L[i].text);
        release(&L[i].h_pts);
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
