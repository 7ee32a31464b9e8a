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
#include <unistd.h>

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

/* Inside an (octave, scale) group the reference's points stand in the order its atomic
 * counter handed out (cudaSiftD.cu:296), which is not a function of the input; two runs of it
 * write different files.  The writers here put the points in one order: scale (groups
 * ascend in scale, cudaSiftH.cu:94-95), then row, then column. */
static int point_order(const void *a, const void *b)
{
    const SiftPoint *p = (const SiftPoint *)a, *q = (const SiftPoint *)b;
    if (p->scale != q->scale) return p->scale < q->scale ? -1 : 1;
    if (p->ypos != q->ypos) return p->ypos < q->ypos ? -1 : 1;
    if (p->xpos != q->xpos) return p->xpos < q->xpos ? -1 : 1;
    if (p->orientation != q->orientation) return p->orientation < q->orientation ? -1 : 1;
    return 0;
}

void sift_sort_points(SiftPoint *pts, int n)
{
    if (pts && n > 1) qsort(pts, (size_t)n, sizeof(SiftPoint), point_order);
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

/* the pipelines' writer: the points in their one order first (the buffer is the pipeline's own) */
static int save_sorted(const char *dst_dir, const char *src_path, SiftPoint *pts, int n_all, int n_write,
                       grow_buf *text)
{
    sift_sort_points(pts, n_all);
    return save_points_buf(dst_dir, src_path, pts, n_write, text);
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

/* ---- the whole list on all GPUs -------------------------------------------------
 * sift.cu:68-134 treats its images as independent (two at a time, no state shared between
 * them but the counter slots).  Here the list is dealt over the GPUs, image i to worker
 * i mod n_workers, one host thread per GPU with its own device, stream and buffers; no data
 * moves between GPUs, and the only gather is the count array, filled in image order on the
 * host.  Inside a worker the stages overlap: a pool of host threads decodes the files of
 * batch j+1 into pinned memory and formats and writes the text of batch j-1 while the GPU
 * holds batch j (one upload of 8-bit pixels, sift_extract_batch_async, one read-back of the
 * counts, then only the points each image produced).  Output files are byte for byte those
 * of sift(). */
typedef struct {
    void (*fn)(void *);
    void *arg;
    int *group; /* jobs of one group still to finish */
} mt_job;

typedef struct {
    pthread_t *th;
    int n_th;
    mt_job *q;
    int cap, head, count, stop;
    pthread_mutex_t mu;
    pthread_cond_t cv_job, cv_done;
} mt_pool;

static void *pool_main(void *arg)
{
    mt_pool *P = (mt_pool *)arg;
    pthread_mutex_lock(&P->mu);
    for (;;) {
        while (!P->count && !P->stop) pthread_cond_wait(&P->cv_job, &P->mu);
        if (!P->count && P->stop) break;
        mt_job j = P->q[P->head];
        P->head = (P->head + 1) % P->cap;
        P->count--;
        pthread_mutex_unlock(&P->mu);
        j.fn(j.arg);
        pthread_mutex_lock(&P->mu);
        if (--*j.group == 0) pthread_cond_broadcast(&P->cv_done);
    }
    pthread_mutex_unlock(&P->mu);
    return NULL;
}

static int pool_start(mt_pool *P, int n_th, int cap)
{
    memset(P, 0, sizeof *P);
    pthread_mutex_init(&P->mu, NULL);
    pthread_cond_init(&P->cv_job, NULL);
    pthread_cond_init(&P->cv_done, NULL);
    P->q = (mt_job *)calloc((size_t)cap, sizeof(mt_job));
    P->th = (pthread_t *)calloc((size_t)n_th, sizeof(pthread_t));
    P->cap = cap;
    if (!P->q || !P->th) return -1;
    for (int i = 0; i < n_th; ++i) {
        if (pthread_create(&P->th[i], NULL, pool_main, P)) break;
        P->n_th++;
    }
    return P->n_th > 0 ? 0 : -1;
}

/* the queue holds every job of the sets in flight: it cannot fill up */
static void pool_submit(mt_pool *P, void (*fn)(void *), void *arg, int *group)
{
    pthread_mutex_lock(&P->mu);
    ++*group;
    P->q[(P->head + P->count) % P->cap] = (mt_job){fn, arg, group};
    P->count++;
    pthread_cond_signal(&P->cv_job);
    pthread_mutex_unlock(&P->mu);
}

static void pool_wait(mt_pool *P, int *group)
{
    pthread_mutex_lock(&P->mu);
    while (*group) pthread_cond_wait(&P->cv_done, &P->mu);
    pthread_mutex_unlock(&P->mu);
}

static void pool_stop(mt_pool *P)
{
    pthread_mutex_lock(&P->mu);
    P->stop = 1;
    pthread_cond_broadcast(&P->cv_job);
    pthread_mutex_unlock(&P->mu);
    for (int i = 0; i < P->n_th; ++i) pthread_join(P->th[i], NULL);
    free(P->th);
    free(P->q);
    pthread_mutex_destroy(&P->mu);
    pthread_cond_destroy(&P->cv_job);
    pthread_cond_destroy(&P->cv_done);
}

/* header of a binary PGM/PPM: 0 and the geometry, or -1 */
static int pnm_header(FILE *f, int *kind, int *w, int *h)
{
    if (fgetc(f) != 'P') return -1;
    *kind = fgetc(f);
    if (*kind != '5' && *kind != '6') return -1;
    int maxv;
    if (pnm_token(f, w) || pnm_token(f, h) || pnm_token(f, &maxv)) return -1;
    return (*w < 1 || *h < 1 || maxv < 1 || maxv > 255) ? -1 : 0;
}

/* the file's pixels as 8-bit grey into dst (n bytes); colour is reduced as read_pnm_gray does */
static int read_gray_u8(FILE *f, int kind, size_t n, unsigned char *dst, grow_buf *rgb)
{
    if (kind == '5') return fread(dst, 1, n, f) == n ? 0 : -1;
    if (grow(rgb, 3 * n)) return -1;
    const unsigned char *c = (const unsigned char *)rgb->p;
    if (fread(rgb->p, 1, 3 * n, f) != 3 * n) return -1;
    for (size_t i = 0; i < n; ++i)
        dst[i] = (unsigned char)((1868 * c[3 * i + 2] + 9617 * c[3 * i + 1] + 4899 * c[3 * i] + 8192) >> 14);
    return 0;
}

/* sift.cu:64-143: images are taken two at a time, each on its own stream and counter slot:
 * initDeviceConstant, initDeviceVariables, upload, extractSift, getPointCount, read back, one
 * text file per image, exactly the reference's sequence of calls per pair.  What differs is
 * where the host work runs.  The reference decodes on the calling thread, hands each file to a
 * detached writer thread (sift.cu:121-127) and allocates and frees every buffer per pair; its
 * host stages take a hundred times longer than the extraction here.  So the files of the next
 * pairs are decoded by pool threads while the GPU holds the current pair, the writers of the
 * last pairs run beside both (a ring of PW_RING pairs of host buffers), pixels cross the link
 * as the 8-bit values they are in the file and become floats on the device (convertTo(CV_32F),
 * utils.cc:13, after the upload), and buffers are kept from pair to pair.  Every point is
 * written (the reference passes count-1, sift.cu:125; sift_pipeline_drop_last_point(1)
 * reproduces that), and the writers are waited for before the call returns. */
#define PW_RING 4
typedef struct {
    grow_buf raw, rgb, h_pts, text; /* raw and h_pts are pinned */
    const char *path, *dst_dir;
    int device, cap;
    int w, h, n, n_all, rc;
} pw_img;

typedef struct {
    pw_img img[2];
    int nq, used;
    int load_left, save_left; /* job groups */
} pw_pair;

static void pw_load(void *arg)
{
    pw_img *I = (pw_img *)arg;
    I->rc = -1;
    sift_set_device(I->device); /* the pinned buffers belong to the caller's device */
    FILE *f = fopen(I->path, "rb");
    if (!f) return;
    int kind;
    if (pnm_header(f, &kind, &I->w, &I->h) == 0) {
        const size_t n = (size_t)I->w * I->h;
        if (grow(&I->raw, n) == 0 && grow(&I->h_pts, sizeof(SiftPoint) * (size_t)I->cap) == 0 &&
            read_gray_u8(f, kind, n, (unsigned char *)I->raw.p, &I->rgb) == 0)
            I->rc = 0;
    }
    fclose(f);
}

static void pw_save(void *arg)
{
    pw_img *I = (pw_img *)arg;
    I->rc = save_sorted(I->dst_dir, I->path, (SiftPoint *)I->h_pts.p, I->n_all, I->n, &I->text);
}

static int host_thread_budget(void);

int sift(const char *dst_dir, const char **src_paths, int n_imgs)
{
    if (!dst_dir || !src_paths || n_imgs < 0) return -1;
    sift_params prm;
    sift_get_params(&prm);
    const int device = sift_get_device();
    if (device < 0) return -1;
    mt_pool pool;
    int threads = host_thread_budget();
    threads = threads < 2 ? 2 : (threads > 8 ? 8 : threads);
    if (pool_start(&pool, threads, 4 * PW_RING + 8)) {
        pool_stop(&pool);
        return -1;
    }
    pw_pair *ring = (pw_pair *)calloc(PW_RING, sizeof(pw_pair));
    /* per lane of a pair: device buffers and the stream, kept from pair to pair */
    struct {
        unsigned char *d_u8;
        float *d_img;
        size_t px;
        SiftPoint *d_pts;
        void *stream;
    } D[2];
    memset(D, 0, sizeof D);
    int rc = ring ? 0 : -1;
    for (int s = 0; ring && s < PW_RING; ++s)
        for (int i = 0; i < 2; ++i) {
            pw_img *I = &ring[s].img[i];
            I->raw.pinned = I->h_pts.pinned = 1;
            I->dst_dir = dst_dir;
            I->device = device;
            I->cap = prm.max_points;
        }
    const int n_pairs = (n_imgs + 1) / 2;
    int submitted = 0; /* pairs whose decoding has been started */
    for (int j = 0; j < n_pairs && rc == 0; ++j) {
        /* decoding runs up to two pairs ahead; a ring slot is reused once its files are written */
        for (; submitted < n_pairs && submitted <= j + 2; ++submitted) {
            pw_pair *P = &ring[submitted % PW_RING];
            pool_wait(&pool, &P->save_left);
            for (int i = 0; i < P->nq && P->used; ++i)
                if (P->img[i].rc) rc = -1;
            P->nq = (n_imgs - 2 * submitted > 1) ? 2 : 1;
            P->used = 0;
            for (int i = 0; i < P->nq; ++i) {
                P->img[i].path = src_paths[2 * submitted + i];
                pool_submit(&pool, pw_load, &P->img[i], &P->load_left);
            }
        }
        if (rc) break;
        pw_pair *P = &ring[j % PW_RING];
        pool_wait(&pool, &P->load_left);
        for (int i = 0; i < P->nq && rc == 0; ++i) {
            pw_img *I = &P->img[i];
            if (I->rc) {
                fprintf(stderr, "Could not open or find the image %s\n", I->path); /* utils.cc:15 */
                rc = -1;
                break;
            }
            const size_t need = (size_t)I->w * I->h;
            if (D[i].px < need) {
                if (D[i].d_u8) sift_dev_free(D[i].d_u8);
                if (D[i].d_img) sift_dev_free(D[i].d_img);
                D[i].d_u8 = NULL;
                D[i].d_img = NULL;
                D[i].px = 0;
                rc |= sift_dev_malloc((void **)&D[i].d_u8, need);
                if (!rc) rc |= sift_dev_malloc((void **)&D[i].d_img, need * sizeof(float));
                if (!rc) D[i].px = need;
            }
            if (!rc && !D[i].d_pts) rc |= sift_dev_malloc((void **)&D[i].d_pts, sizeof(SiftPoint) * (size_t)prm.max_points);
            if (!rc && !D[i].stream) rc |= sift_stream_create(&D[i].stream);
        }
        if (rc) break;
        /* sift.cu:99-127 */
        initDeviceConstant();
        initDeviceVariables();
        for (int i = 0; i < P->nq; ++i) {
            pw_img *I = &P->img[i];
            rc |= sift_memcpy_h2d_async(D[i].d_u8, I->raw.p, (size_t)I->w * I->h, D[i].stream);
            rc |= sift_convert_u8_f32(D[i].d_img, I->w, D[i].d_u8, I->w, I->w, I->h, D[i].stream);
        }
        for (int i = 0; i < P->nq; ++i)
            extractSift(D[i].d_pts, D[i].d_img, P->img[i].w, P->img[i].w, P->img[i].h, D[i].stream, i, 0);
        for (int i = 0; i < P->nq && rc == 0; ++i) {
            int n = getPointCount(i);
            if (n < 0) {
                rc = -1;
                break;
            }
            pw_img *I = &P->img[i];
            if (n > 0) rc |= sift_memcpy_d2h_async(I->h_pts.p, D[i].d_pts, sizeof(SiftPoint) * (size_t)n, D[i].stream);
            I->n_all = n;
            if (g_drop_last) n -= 1;
            I->n = n < 0 ? 0 : n;
        }
        for (int i = 0; i < P->nq; ++i)
            if (D[i].stream) rc |= sift_stream_sync(D[i].stream);
        if (rc) break;
        P->used = 1;
        for (int i = 0; i < P->nq; ++i) pool_submit(&pool, pw_save, &P->img[i], &P->save_left);
    }
    for (int s = 0; ring && s < PW_RING; ++s) {
        pool_wait(&pool, &ring[s].load_left);
        pool_wait(&pool, &ring[s].save_left);
        for (int i = 0; i < ring[s].nq && ring[s].used; ++i)
            if (ring[s].img[i].rc) rc = -1;
    }
    pool_stop(&pool);
    for (int s = 0; ring && s < PW_RING; ++s)
        for (int i = 0; i < 2; ++i) {
            release(&ring[s].img[i].raw);
            release(&ring[s].img[i].rgb);
            release(&ring[s].img[i].h_pts);
            release(&ring[s].img[i].text);
        }
    for (int i = 0; i < 2; ++i) {
        if (D[i].stream) sift_stream_destroy(D[i].stream);
        if (D[i].d_u8) sift_dev_free(D[i].d_u8);
        if (D[i].d_img) sift_dev_free(D[i].d_img);
        if (D[i].d_pts) sift_dev_free(D[i].d_pts);
    }
    free(ring);
    return rc ? -1 : 0;
}

#define MT_SETS 3
#define MT_MAX_BATCH 32

struct mt_set;
typedef struct {
    struct mt_set *set;
    int k;          /* position in the set */
    int image;      /* index into the caller's list */
    int status;     /* 0 decoded; 1 another geometry than the set's (done later, alone); < 0 error */
    int n_points;
    grow_buf text, rgb;
} mt_item;

typedef struct mt_set {
    struct mt_worker *W;
    int n, w, h;
    grow_buf h_img, h_pts, h_counts; /* pinned */
    unsigned char *d_img;
    size_t d_img_cap;
    SiftPoint *d_pts;
    unsigned *d_counts;
    int decode_left, format_left; /* job groups */
    mt_item item[MT_MAX_BATCH];
} mt_set;

typedef struct mt_worker {
    int index, n_workers, device, pool_threads, batch, cap;
    const char *dst_dir;
    const char **paths;
    int n_imgs;
    int *counts;
    int rc;
    char err[512]; /* the worker thread's last library error, for the thread that started it */
    pthread_t thread;
    mt_pool pool;
    void *stream;
    mt_set set[MT_SETS];
} mt_worker;

static int peek_geometry(const char *path, int *w, int *h)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    int kind;
    const int rc = pnm_header(f, &kind, w, h);
    fclose(f);
    return rc;
}

/* one file into its place in the set's pinned 8-bit staging buffer; colour is reduced to grey as
 * read_pnm_gray does (the grey values are integers, so the 8-bit entry gives the float entry's result) */
static void decode_job(void *arg)
{
    mt_item *it = (mt_item *)arg;
    mt_set *S = it->set;
    const char *path = S->W->paths[it->image];
    it->status = -1;
    FILE *f = fopen(path, "rb");
    if (!f) return;
    int kind, w, h;
    if (pnm_header(f, &kind, &w, &h) == 0) {
        if (w != S->w || h != S->h) {
            it->status = 1;
        } else {
            const size_t n = (size_t)w * h;
            unsigned char *dst = (unsigned char *)S->h_img.p + n * (size_t)it->k;
            if (read_gray_u8(f, kind, n, dst, &it->rgb) == 0) it->status = 0;
        }
    }
    fclose(f);
}

static void format_job(void *arg)
{
    mt_item *it = (mt_item *)arg;
    mt_set *S = it->set;
    mt_worker *W = S->W;
    if (it->status) return;
    if (W->counts) W->counts[it->image] = it->n_points;
    if (!W->dst_dir) return;
    int n = it->n_points - (g_drop_last ? 1 : 0);
    if (n < 0) n = 0;
    if (save_sorted(W->dst_dir, W->paths[it->image], (SiftPoint *)S->h_pts.p + (size_t)it->k * W->cap, it->n_points, n,
                    &it->text))
        it->status = -2;
}

/* buffers of a set for n images of w x h */
static int set_prepare(mt_set *S, int n, int w, int h)
{
    mt_worker *W = S->W;
    const size_t img = (size_t)w * h, need = img * (size_t)W->batch;
    S->n = n;
    S->w = w;
    S->h = h;
    S->h_img.pinned = S->h_pts.pinned = S->h_counts.pinned = 1;
    if (grow(&S->h_img, need) || grow(&S->h_pts, sizeof(SiftPoint) * (size_t)W->cap * W->batch) ||
        grow(&S->h_counts, sizeof(unsigned) * (size_t)W->batch))
        return -1;
    if (S->d_img_cap < need) {
        if (S->d_img) sift_dev_free(S->d_img);
        S->d_img = NULL;
        S->d_img_cap = 0;
        if (sift_dev_malloc((void **)&S->d_img, need)) return -1;
        S->d_img_cap = need;
    }
    if (!S->d_pts && sift_dev_malloc((void **)&S->d_pts, sizeof(SiftPoint) * (size_t)W->cap * W->batch)) return -1;
    if (!S->d_counts && sift_dev_malloc((void **)&S->d_counts, sizeof(unsigned) * (size_t)W->batch)) return -1;
    return 0;
}

/* upload, extract, read back the counts, then the points each image produced */
static int set_gpu(mt_set *S)
{
    mt_worker *W = S->W;
    const size_t img = (size_t)S->w * S->h;
    int any = 0;
    for (int k = 0; k < S->n; ++k) any |= S->item[k].status == 0;
    if (!any) return 0;
    int rc = sift_memcpy_h2d_async(S->d_img, S->h_img.p, img * (size_t)S->n, W->stream);
    if (!rc)
        rc = sift_extract_batch_async(S->d_pts, NULL, S->d_counts, S->d_img, 1, (long)img, S->n, S->w, S->w, S->h, W->stream);
    if (!rc) rc = sift_memcpy_d2h_async(S->h_counts.p, S->d_counts, sizeof(unsigned) * (size_t)S->n, W->stream);
    if (!rc) rc = sift_stream_sync(W->stream);
    const unsigned *cnt = (const unsigned *)S->h_counts.p;
    for (int k = 0; k < S->n && !rc; ++k) {
        mt_item *it = &S->item[k];
        if (it->status) continue;
        it->n_points = (int)cnt[k];
        if (it->n_points > 0)
            rc = sift_memcpy_d2h_async((SiftPoint *)S->h_pts.p + (size_t)k * W->cap, S->d_pts + (size_t)k * W->cap,
                                       sizeof(SiftPoint) * (size_t)it->n_points, W->stream);
    }
    if (!rc) rc = sift_stream_sync(W->stream);
    return rc;
}

static void *worker_main(void *arg)
{
    mt_worker *W = (mt_worker *)arg;
    W->rc = -1;
    sift_clear_error();
    if (sift_set_device(W->device) || sift_stream_create(&W->stream)) {
        snprintf(W->err, sizeof W->err, "%s", sift_last_error());
        return NULL;
    }
    if (pool_start(&W->pool, W->pool_threads, 2 * MT_SETS * MT_MAX_BATCH + 8)) {
        pool_stop(&W->pool); /* (joins whatever threads did start) */
        sift_stream_destroy(W->stream);
        snprintf(W->err, sizeof W->err, "worker %d: could not start its host threads", W->index);
        return NULL;
    }
    /* this worker's images, in list order */
    int n_mine = 0;
    for (int i = W->index; i < W->n_imgs; i += W->n_workers) ++n_mine;
    int *mine = (int *)malloc(sizeof(int) * (size_t)(n_mine + 1));
    int *later = (int *)malloc(sizeof(int) * (size_t)(n_mine + 1)); /* images of another size than their batch */
    int n_later = 0, rc = (mine && later) ? 0 : -1;
    for (int i = W->index, k = 0; rc == 0 && i < W->n_imgs; i += W->n_workers) mine[k++] = i;
    for (int s = 0; s < MT_SETS; ++s) W->set[s].W = W;

    /* pass 0: the list in batches; pass 1: what pass 0 set aside, one image at a time */
    for (int pass = 0; pass < 2 && rc == 0; ++pass) {
        const int *list = pass ? later : mine;
        const int n_list = pass ? n_later : n_mine, B = pass ? 1 : W->batch;
        const int nb = (n_list + B - 1) / B;
        for (int j = 0; j <= nb && rc == 0; ++j) {
            /* decoding of batch j starts once the set's previous use (batch j - MT_SETS) is written out */
            if (j < nb) {
                mt_set *S = &W->set[j % MT_SETS];
                pool_wait(&W->pool, &S->format_left);
                for (int k = 0; k < S->n; ++k)
                    if (S->item[k].status < 0) rc = -1;
                const int first = j * B, n = (n_list - first) < B ? (n_list - first) : B;
                int w = 0, h = 0;
                if (peek_geometry(W->paths[list[first]], &w, &h)) {
                    fprintf(stderr, "Could not open or find the image %s\n", W->paths[list[first]]); /* utils.cc:15 */
                    char msg[400];
                    snprintf(msg, sizeof msg, "Could not open or find the image %.300s", W->paths[list[first]]);
                    sift_set_last_error(msg);
                    rc = -1;
                    break;
                }
                if (set_prepare(S, n, w, h)) {
                    rc = -1;
                    break;
                }
                for (int k = 0; k < n; ++k) {
                    mt_item *it = &S->item[k];
                    it->set = S;
                    it->k = k;
                    it->image = list[first + k];
                    it->n_points = 0;
                    pool_submit(&W->pool, decode_job, it, &S->decode_left);
                }
            }
            /* the GPU takes batch j - 1 while batch j is being decoded */
            if (j >= 1) {
                mt_set *S = &W->set[(j - 1) % MT_SETS];
                pool_wait(&W->pool, &S->decode_left);
                for (int k = 0; k < S->n; ++k) {
                    if (S->item[k].status < 0) {
                        fprintf(stderr, "Could not open or find the image %s\n", W->paths[S->item[k].image]);
                        char msg[400];
                        snprintf(msg, sizeof msg, "Could not open or find the image %.300s", W->paths[S->item[k].image]);
                        sift_set_last_error(msg);
                        rc = -1;
                    } else if (S->item[k].status == 1) {
                        if (pass == 0) later[n_later++] = S->item[k].image;
                        else rc = -1;
                    }
                }
                if (rc == 0 && set_gpu(S)) rc = -1;
                if (rc == 0)
                    for (int k = 0; k < S->n; ++k) pool_submit(&W->pool, format_job, &S->item[k], &S->format_left);
            }
        }
        for (int s = 0; s < MT_SETS; ++s) {
            pool_wait(&W->pool, &W->set[s].decode_left);
            pool_wait(&W->pool, &W->set[s].format_left);
            for (int k = 0; k < W->set[s].n; ++k)
                if (W->set[s].item[k].status < 0) rc = -1;
            W->set[s].n = 0;
        }
    }
    pool_stop(&W->pool);
    for (int s = 0; s < MT_SETS; ++s) {
        mt_set *S = &W->set[s];
        release(&S->h_img);
        release(&S->h_pts);
        release(&S->h_counts);
        if (S->d_img) sift_dev_free(S->d_img);
        if (S->d_pts) sift_dev_free(S->d_pts);
        if (S->d_counts) sift_dev_free(S->d_counts);
        for (int k = 0; k < MT_MAX_BATCH; ++k) {
            release(&S->item[k].text);
            release(&S->item[k].rgb);
        }
    }
    sift_stream_destroy(W->stream);
    free(mine);
    free(later);
    if (rc) snprintf(W->err, sizeof W->err, "%s", sift_last_error());
    W->rc = rc;
    return NULL;
}

/* worker w of n_workers takes images w, w + n_workers, ...: which GPU an image goes to */
int sift_list_partition(int image, int n_workers) { return n_workers > 0 ? image % n_workers : -1; }

static int host_thread_budget(void)
{
    const char *e = getenv("SIFT_HOST_THREADS");
    if (!e) e = getenv("OMP_NUM_THREADS");
    int n = e ? atoi(e) : 0;
    if (n < 1) {
        const long on = sysconf(_SC_NPROCESSORS_ONLN);
        n = on > 16 ? 16 : (int)on; /* a share of the machine, not all of it */
    }
    return n < 1 ? 1 : (n > 256 ? 256 : n);
}

int sift_extract_list_multi(const char *dst_dir, const char **src_paths, int n_imgs, int n_workers, int *counts)
{
    if (!src_paths || n_imgs < 0 || n_workers < 1 || n_workers > 64) return -1;
    const int n_dev = sift_device_count();
    if (n_dev < 1) return -1;
    sift_params prm;
    sift_get_params(&prm);
    initDeviceConstant();
    if (counts)
        for (int i = 0; i < n_imgs; ++i) counts[i] = -1;
    mt_worker *W = (mt_worker *)calloc((size_t)n_workers, sizeof(mt_worker));
    if (!W) return -1;
    int per = host_thread_budget() / n_workers;
    if (per < 2) per = 2;
    /* images per batch: the pinned read-back buffer of a set stays within 64 MB */
    long batch = (64L << 20) / (long)(sizeof(SiftPoint) * (size_t)prm.max_points);
    batch = batch < 1 ? 1 : (batch > MT_MAX_BATCH ? MT_MAX_BATCH : batch);
    int started = 0, rc = 0;
    for (int g = 0; g < n_workers; ++g) {
        W[g].index = g;
        W[g].n_workers = n_workers;
        W[g].device = g % n_dev; /* more workers than GPUs share them in turn (rehearsals on one GPU) */
        W[g].pool_threads = per;
        W[g].batch = (int)batch;
        W[g].cap = prm.max_points;
        W[g].dst_dir = dst_dir;
        W[g].paths = src_paths;
        W[g].n_imgs = n_imgs;
        W[g].counts = counts;
        if (pthread_create(&W[g].thread, NULL, worker_main, &W[g])) {
            rc = -1;
            break;
        }
        ++started;
    }
    const char *first_err = NULL;
    for (int g = 0; g < started; ++g) {
        pthread_join(W[g].thread, NULL);
        if (W[g].rc) {
            rc = -1;
            if (!first_err && W[g].err[0]) first_err = W[g].err;
        }
    }
    /* the workers' errors live in their own threads: the first one becomes the caller's */
    if (rc) sift_set_last_error(first_err ? first_err : "sift_extract_list_multi: a worker failed (see stderr)");
    free(W);
    return rc;
}
