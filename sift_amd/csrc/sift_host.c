/*
 * C host side of the MI355X SIFT extractor: configuration, Gaussian tables,
 * scratch arenas, the octave loop and the entry points of include/sift_amd.h.
 * Talks to the GPU only through the C-ABI layer in hip_layer.h.
 *
 * Reference counterpart: Sift/src/cudaSiftH.cu (host driver) and the table
 * construction in Sift/src/utils.cc:41-53.
 */
#include "../../include/sift_amd.h"
#include "hip_layer.h"

#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define REF_PI 3.14159265 /* utils.h:4 */
#define MAX_DEV 16
#define BATCH_SLOT SIFT_MAX_SLOTS /* two batch arenas follow the stream slots */
#define N_LANES 2
#define QUEUE_CAP 65536 /* first-octave contrast candidates per image and scale before the dense fall-back */
#define QUEUE_MIN 4096
#define COUNT_WORDS HL_COUNT_WORDS

/* ---- state ---------------------------------------------------------------- */
typedef struct {
    int valid;
    /* signature the arena was planned for */
    int w, h, S, O, oct0, cap, nimg, queue_cap, profile, upsample;
    int n_oct;         /* octaves processed: O - oct0, plus one in front when up-sampling */
    long up_off;       /* the up-sampled image (2w x 2h, pitch 2w), profile "lowe" with upsample */
    void *sub;         /* profile "lowe": nimg * groups * cap 16-byte records beside cand */
    float *arena;
    long arena_stride; /* floats per image */
    int ow[SIFT_MAX_OCTAVES], oh[SIFT_MAX_OCTAVES], op[SIFT_MAX_OCTAVES];
    long blur_off[SIFT_MAX_OCTAVES];
    long queue_off[SIFT_MAX_OCTAVES]; /* words, inside one image's queue block */
    int qcap[SIFT_MAX_OCTAVES];
    long queue_stride;                /* words per image */
    unsigned *counts; /* nimg * HL_COUNT_WORDS: accepted per group, queued per group, the full-group word */
    unsigned *cand;   /* nimg * groups * cap */
    long cand_stride;
    unsigned *queue;  /* nimg * queue_stride: contrast candidates per octave and scale */
    /* Library-owned event recorded behind the slot's latest work.  The caller's stream
     * handle is never kept: the reference's caller creates its streams per image pair and
     * destroys them afterwards (sift.cu:96,130), so a stored handle would dangle. */
    void *done;
    int busy; /* done has been recorded and not yet waited for */
    /* batch lanes: 1 + the index of the library's private stream that recorded `done` last,
     * 0 when it was a caller's stream (work on that private stream needs no wait) */
    int on_lane;
} slot_ctx;

/* g_mu guards the building of the tables; g_dev_mu[d] guards device d's slots, batch
 * streams and counters.  Calls on different devices run side by side (one host
 * thread per GPU), calls on one device take turns because they share its arenas.
 * Every entry that reads the parameters holds g_params_rw for reading while it plans and
 * enqueues; sift_set_params takes it for writing, so it waits for them and they for it. */
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_rwlock_t g_params_rw = PTHREAD_RWLOCK_INITIALIZER;
static pthread_mutex_t g_dev_mu[MAX_DEV] = {[0 ... MAX_DEV - 1] = PTHREAD_MUTEX_INITIALIZER};
static sift_params g_params = {4, 2, 2500, 1.6f, 10.0f, 10.0f, 0, 0};
static int g_tables_ready = 0;
static float g_blur[HL_MAX_PLANES * HL_TAPS];
static float g_wnd[4 * 16];
static float g_desc[16];
static float g_cos32[32], g_sin32[32];
static hl_taps g_taps;
static slot_ctx g_slots[MAX_DEV][SIFT_MAX_SLOTS + N_LANES];
/* the batch entry alternates launch sets between two private streams so the
 * short, latency-bound kernels of one set run beside the long ones of the other */
static void *g_lane_stream[MAX_DEV][N_LANES];
static void *g_lane_event[MAX_DEV][N_LANES + 1];
static unsigned *g_batch_count[MAX_DEV];
static int g_batch_count_n[MAX_DEV];
static unsigned *g_host_count[MAX_DEV]; /* pinned host memory the device writes the counts to */
static int g_host_count_n[MAX_DEV];
static int g_lanes = N_LANES;
static int g_next_lane[MAX_DEV]; /* the lane the next non-blocking batch call starts on */
static unsigned *g_slot_counter[MAX_DEV]; /* device, SIFT_MAX_SLOTS words */
/* Pinned host mirror of the slot counters: the last kernel of every extractSift writes the
 * slot's count there, so getPointCount is one event wait and a load, not a copy and a sync. */
static unsigned *g_slot_host[MAX_DEV];
/* initDeviceVariables has been called and the slot has not extracted since: its count is zero,
 * and the next extractSift zeroes the device word in its own first kernel, on its own stream. */
static int g_slot_reset[MAX_DEV][SIFT_MAX_SLOTS];
/* the most recent batch call on each device: where its images sit in the lane arenas */
static struct {
    int valid, n, chunk, n_sets, lane0, lanes;
} g_last_batch[MAX_DEV];
static int g_chunk = 256;
static int g_queue_cap = QUEUE_CAP; /* 0: no pre-filter, dense detector only */
/* A launch set of at most this many pixels (images x width x height of the first octave) is
 * detected by the dense detector alone: its kernels are one round of waves, so the pre-filter's
 * per-wave start (a counter read) and end (an atomic and a dependent store) sit on every
 * pyramid launch's critical path, and candidate verification plus the gated dense launch cost
 * what the dense detector costs by itself at this size.  Measured on one 1080p image: 141 us
 * against 155 us for the whole call sequence.  0: never. */
static long g_dense_route_px = 2200000;
static __thread char g_err[512] = ""; /* the calling thread's last error */

static void set_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

const char *sift_last_error(void) { return g_err; }
/* makes msg the calling thread's last error (a worker thread's message, handed to the thread
 * that started it) */
void sift_set_last_error(const char *msg)
{
    snprintf(g_err, sizeof g_err, "%s", msg ? msg : "");
}
void sift_clear_error(void) { g_err[0] = 0; }

static int fail_hl(int code, const char *what)
{
    if (code <= -1000)
        set_err("%s: invalid launch arguments (%d)", what, code);
    else
        set_err("%s: %s", what, hl_error_string(code));
    return code;
}

/* locks the calling thread's current device and returns its index, or -1 */
static int lock_device(const char *who)
{
    const int dev = hl_get_device();
    if (dev < 0 || dev >= MAX_DEV) {
        set_err("%s: no usable HIP device", who);
        return -1;
    }
    pthread_mutex_lock(&g_dev_mu[dev]);
    pthread_rwlock_rdlock(&g_params_rw);
    return dev;
}

static void unlock_device(int dev)
{
    pthread_rwlock_unlock(&g_params_rw);
    pthread_mutex_unlock(&g_dev_mu[dev]);
}

static int idivup(int a, int b) { return (a % b != 0) ? (a / b + 1) : (a / b); }      /* utils.cc:59 */
static int ialignup(int A, int a) { return (A % a != 0) ? (A + a - A % a) : A; }     /* utils.cc:60 */
static long lalign(long v, long a) { return (v + a - 1) / a * a; }

/* ---- tables ---------------------------------------------------------------
 * utils.cc:41-53: the density is evaluated in double, stored as float, summed
 * in float and normalised by a float division. */
static void gaussian1D(float *taps, float sigma, int n)
{
    float sum = 0;
    const float centre = (float)(n - 1) / 2;
    for (int i = 0; i < n; ++i) {
        const float d = (float)i - centre;
        const double amp = 1 / (pow(2.0f * REF_PI, (double)0.5f) * (double)sigma);
        const double arg = -pow((double)d, 2) / (2 * pow((double)sigma, 2));
        taps[i] = (float)(amp * exp(arg));
        sum += taps[i];
    }
    for (int i = 0; i < n; ++i)
        taps[i] /= sum;
}

static int build_tables(void)
{
    memset(&g_taps, 0, sizeof g_taps);
    const int S = g_params.n_scales;
    const float k = (float)pow(2, 0.5); /* cudaSiftH.cu:131 */
    for (int i = 0; i < S + 3; ++i) {   /* cudaSiftH.cu:135-139 */
        const float s = (float)(pow((double)k, (double)(i - 1)) * (double)g_params.sigma);
        gaussian1D(g_blur + i * HL_TAPS, s, HL_TAPS);
    }
    for (int i = 0; i < S; ++i) { /* cudaSiftH.cu:145-149 */
        const float s = (float)(pow((double)k, (double)i) * (double)g_params.sigma * 1.5);
        gaussian1D(g_wnd + i * 16, s, 16);
    }
    gaussian1D(g_desc, (float)((double)g_params.sigma * 1.5), 16); /* cudaSiftH.cu:153-155 */
    for (int i = 0; i < S + 3; ++i)
        for (int m = 0; m < HL_HALF_TAPS; ++m) {
            if (g_blur[i * HL_TAPS + m] != g_blur[i * HL_TAPS + 18 - m]) {
                set_err("blur taps of plane %d are not symmetric", i);
                return -1;
            }
            g_taps.t[m][i] = g_blur[i * HL_TAPS + m];
        }
    for (int q = 0; q < 32; ++q) { /* the 32 orientations the detector can assign */
        const double rad = (double)(11.25f * (float)q) * (3.14159265358979323846 / 180.0);
        g_cos32[q] = (float)cos(rad);
        g_sin32[q] = (float)sin(rad);
    }
    g_tables_ready = 1;
    return 0;
}

static int tables_ok(void)
{
    int r = 0;
    pthread_mutex_lock(&g_mu);
    if (!g_tables_ready) r = build_tables();
    pthread_mutex_unlock(&g_mu);
    return r;
}

static float scale_value(int i, int octave) /* cudaSiftH.cu:94-95 */
{
    const float k = (float)pow(2, 0.5);
    return (float)(pow((double)k, (double)(i + g_params.n_scales * octave)) * (double)g_params.sigma);
}

void sift_default_params(sift_params *p)
{
    p->n_octaves = 4;
    p->n_scales = 2;
    p->max_points = 2500;
    p->sigma = 1.6f;
    p->r_thresh = 10.0f;
    p->extrema_thresh = 10.0f;
    p->profile = SIFT_PROFILE_REF;
    p->upsample = 0;
}

int sift_set_params(const sift_params *p)
{
    if (p->profile != SIFT_PROFILE_REF && p->profile != SIFT_PROFILE_LOWE) {
        set_err("profile %d: 0 (ref) or 1 (lowe)", p->profile);
        return -1;
    }
    if (p->upsample && (p->profile != SIFT_PROFILE_LOWE || p->n_octaves + 1 > SIFT_MAX_OCTAVES)) {
        set_err("upsample needs profile lowe and n_octaves < %d", SIFT_MAX_OCTAVES);
        return -1;
    }
    if (p->n_octaves < 1 || p->n_octaves > SIFT_MAX_OCTAVES) {
        set_err("n_octaves %d outside 1..%d", p->n_octaves, SIFT_MAX_OCTAVES);
        return -1;
    }
    if (p->n_scales != 2 && p->n_scales != 3) {
        set_err("n_scales %d: kernels are built for 2 and 3", p->n_scales);
        return -1;
    }
    if (p->max_points < 1 || p->max_points > (1 << 22)) {
        set_err("max_points %d outside 1..2^22", p->max_points);
        return -1;
    }
    if (!(p->sigma > 0.f) || !(p->r_thresh > 0.f)) {
        set_err("sigma and r_thresh must be positive");
        return -1;
    }
    pthread_rwlock_wrlock(&g_params_rw);
    pthread_mutex_lock(&g_mu);
    g_params = *p;
    int r = build_tables();
    pthread_mutex_unlock(&g_mu);
    pthread_rwlock_unlock(&g_params_rw);
    return r;
}

void sift_get_params(sift_params *p)
{
    pthread_rwlock_rdlock(&g_params_rw);
    *p = g_params;
    pthread_rwlock_unlock(&g_params_rw);
}

void sift_get_tables(float *blur, float *wnd, float *desc)
{
    pthread_rwlock_rdlock(&g_params_rw);
    tables_ok();
    memcpy(blur, g_blur, sizeof(float) * (g_params.n_scales + 3) * HL_TAPS);
    memcpy(wnd, g_wnd, sizeof(float) * g_params.n_scales * 16);
    memcpy(desc, g_desc, sizeof(float) * 16);
    pthread_rwlock_unlock(&g_params_rw);
}

void sift_set_queue_cap(int n)
{
    if (n >= 0 && n <= (1 << 24))
        g_queue_cap = n;
}

void sift_set_dense_route_pixels(long px)
{
    if (px >= 0)
        g_dense_route_px = px;
}

void sift_set_batch_chunk(int n)
{
    if (n >= 1 && n <= 1024)
        g_chunk = n;
}

/* ---- arenas --------------------------------------------------------------- */
static void free_slot(slot_ctx *c)
{
    if (c->arena) hl_free(c->arena);
    if (c->counts) hl_free(c->counts);
    if (c->cand) hl_free(c->cand);
    if (c->queue) hl_free(c->queue);
    if (c->sub) hl_free(c->sub);
    if (c->done) hl_event_destroy(c->done);
    memset(c, 0, sizeof *c);
}

/* Waits for the slot's latest work.  A failure here means the device lost that work:
 * the slot is torn down so that the next call plans it afresh instead of meeting the
 * same error for ever. */
static int drain_slot(slot_ctx *c, const char *who)
{
    if (!c->valid || !c->busy)
        return 0;
    int r = hl_event_sync(c->done);
    c->busy = 0;
    if (r) {
        fail_hl(r, who);
        hl_device_sync();
        free_slot(c);
    }
    return r;
}

void sift_shutdown(void)
{
    const int dev = lock_device("sift_shutdown");
    if (dev >= 0) {
        hl_device_sync();
        for (int s = 0; s < SIFT_MAX_SLOTS + N_LANES; ++s)
            free_slot(&g_slots[dev][s]);
        for (int l = 0; l < N_LANES; ++l)
            if (g_lane_stream[dev][l]) {
                hl_stream_destroy(g_lane_stream[dev][l]);
                g_lane_stream[dev][l] = NULL;
            }
        for (int l = 0; l <= N_LANES; ++l)
            if (g_lane_event[dev][l]) {
                hl_event_destroy(g_lane_event[dev][l]);
                g_lane_event[dev][l] = NULL;
            }
        if (g_batch_count[dev]) {
            hl_free(g_batch_count[dev]);
            g_batch_count[dev] = NULL;
            g_batch_count_n[dev] = 0;
        }
        if (g_host_count[dev]) {
            hl_host_free(g_host_count[dev]);
            g_host_count[dev] = NULL;
            g_host_count_n[dev] = 0;
        }
        if (g_slot_counter[dev]) {
            hl_free(g_slot_counter[dev]);
            g_slot_counter[dev] = NULL;
        }
        if (g_slot_host[dev]) {
            hl_host_free(g_slot_host[dev]);
            g_slot_host[dev] = NULL;
        }
        memset(g_slot_reset[dev], 0, sizeof g_slot_reset[dev]);
        g_last_batch[dev].valid = 0;
        unlock_device(dev);
    }
}

/* Gives the planes, queues and lists of every slot and batch lane of the current device back
 * (14 GB after a batch of 256 1080p images); the next call plans and allocates afresh.  Counters,
 * streams and events stay: getPointCount keeps answering for finished slots. */
int sift_release_arenas(void)
{
    const int dev = lock_device("sift_release_arenas");
    if (dev < 0) return -1;
    int rc = 0;
    for (int s = 0; s < SIFT_MAX_SLOTS + N_LANES; ++s) {
        slot_ctx *c = &g_slots[dev][s];
        if (!c->valid) continue;
        if (drain_slot(c, "sift_release_arenas"))
            rc = -1; /* (the slot has been torn down already) */
        else
            free_slot(c);
    }
    g_last_batch[dev].valid = 0;
    unlock_device(dev);
    return rc;
}

static int ensure_slot_counter(int dev)
{
    if (g_slot_counter[dev])
        return 0;
    void *p = NULL;
    int r = hl_malloc(&p, sizeof(unsigned) * SIFT_MAX_SLOTS);
    if (r) return fail_hl(r, "allocating slot counters");
    r = hl_memset_async(p, 0, sizeof(unsigned) * SIFT_MAX_SLOTS, NULL);
    if (!r) r = hl_stream_sync(NULL);
    if (r) {
        hl_free(p);
        return fail_hl(r, "zeroing slot counters");
    }
    void *hp = NULL;
    r = hl_host_alloc(&hp, sizeof(unsigned) * SIFT_MAX_SLOTS);
    if (r) {
        hl_free(p);
        return fail_hl(r, "allocating the host copy of the slot counters");
    }
    memset(hp, 0, sizeof(unsigned) * SIFT_MAX_SLOTS);
    g_slot_host[dev] = (unsigned *)hp;
    g_slot_counter[dev] = (unsigned *)p;
    return 0;
}

/* Plans the per-image arena: per octave the S+3 blur planes.  A DoG value is formed
 * where it is read (one subtraction of two neighbouring planes, cudaSiftH.cu:190), so DoG
 * planes are never stored.  Every octave has its own planes so that all pyramid passes can
 * be queued back to back and detection can run once, over all octaves, afterwards.
 * Fills the geometry of *c (no pointers) and returns the bytes of each of the five regions
 * for nimg images: arena, counters, queues, accepted lists, refined-point records. */
static void plan_arena(slot_ctx *c, int w, int h, int oct0, int nimg, int queue_cap, size_t bytes[5])
{
    const int S = g_params.n_scales, O = g_params.n_octaves, cap = g_params.max_points;
    const int lowe = g_params.profile == SIFT_PROFILE_LOWE, up = lowe && g_params.upsample;
    memset(c, 0, sizeof *c);
    long off = 0, qoff = 0;
    int ww = up ? 2 * w : w, hh = up ? 2 * h : h, qc = queue_cap > 0 ? queue_cap : 1;
    const int n_oct = O - oct0 + (up ? 1 : 0);
    if (up) {
        c->up_off = 0;
        off = lalign(4L * w * h, 64);
    }
    for (int oi = 0; oi < n_oct; ++oi) {
        c->ow[oi] = ww;
        c->oh[oi] = hh;
        c->op[oi] = (oi == 0) ? ialignup(ww, 64) : ialignup(ww, 128); /* cudaSiftH.cu:46 */
        c->blur_off[oi] = off;
        off += lalign((long)c->op[oi] * hh * (S + 3), 64);
        c->qcap[oi] = qc;
        c->queue_off[oi] = qoff;
        qoff += (long)S * qc;
        if (qc > QUEUE_MIN) qc = qc / 4 < QUEUE_MIN ? QUEUE_MIN : qc / 4;
        ww = idivup(ww, 2); /* cudaSiftH.cu:44-45 */
        hh = idivup(hh, 2);
    }
    c->queue_stride = qoff;
    c->arena_stride = off;
    c->cand_stride = (long)n_oct * S * cap;
    c->n_oct = n_oct;
    c->profile = g_params.profile;
    c->upsample = up;
    c->w = w;
    c->h = h;
    c->S = S;
    c->O = O;
    c->oct0 = oct0;
    c->cap = cap;
    c->nimg = nimg;
    c->queue_cap = queue_cap;
    bytes[0] = sizeof(float) * (size_t)off * nimg;
    bytes[1] = sizeof(unsigned) * COUNT_WORDS * (size_t)nimg;
    bytes[2] = sizeof(unsigned) * (size_t)c->queue_stride * nimg;
    bytes[3] = sizeof(unsigned) * (size_t)c->cand_stride * nimg;
    bytes[4] = lowe ? 16 * (size_t)c->cand_stride * nimg : 0;
}

/* Plans (or re-plans) a slot and allocates its regions. */
static int ensure_arena(slot_ctx *c, int w, int h, int oct0, int nimg)
{
    const int S = g_params.n_scales, O = g_params.n_octaves, cap = g_params.max_points;
    const int lowe = g_params.profile == SIFT_PROFILE_LOWE, up = lowe && g_params.upsample;
    if (c->valid && c->w == w && c->h == h && c->S == S && c->O == O && c->oct0 == oct0 &&
        c->cap == cap && c->nimg >= nimg && c->queue_cap == g_queue_cap &&
        c->profile == g_params.profile && c->upsample == up)
        return 0;
    if (c->valid && drain_slot(c, "draining slot")) /* geometry changed: its memory goes away */
        return -1;
    free_slot(c);

    size_t bytes[5];
    plan_arena(c, w, h, oct0, nimg, g_queue_cap, bytes);
    static const char *const what[5] = {"allocating arena", "allocating group counters", "allocating candidate queue",
                                        "allocating candidate lists", "allocating refined-point records"};
    void *region[5] = {NULL, NULL, NULL, NULL, NULL};
    int r = 0;
    for (int k = 0; k < 5 && !r; ++k) {
        if (!bytes[k]) continue; /* (the refined-point records exist in profile "lowe" only) */
        r = hl_malloc(&region[k], bytes[k]);
        if (r) fail_hl(r, what[k]);
    }
    if (!r) {
        r = hl_event_create(&c->done);
        if (r) fail_hl(r, "creating slot event");
    }
    if (r) { /* nothing half-built stays behind: the slot is empty again */
        for (int k = 0; k < 5; ++k)
            if (region[k]) hl_free(region[k]);
        memset(c, 0, sizeof *c);
        return -1;
    }
    c->arena = (float *)region[0];
    c->counts = (unsigned *)region[1];
    c->queue = (unsigned *)region[2];
    c->cand = (unsigned *)region[3];
    c->sub = region[4];
    c->valid = 1;
    return 0;
}

/* Enqueues the whole path for nimg images of one geometry on `stream`.
 * cudaSiftH.cu:9-117 does this per image with a recursion, 12 allocations and
 * a dozen blocking copies per octave; here the octaves are a loop over a
 * preplanned arena and nothing blocks. */
static int enqueue_extract(slot_ctx *c, SiftPoint *pts, unsigned char *meta, const void *imgs,
                           int imgs_u8, long img_stride, int nimg, int pitch, int oct0,
                           unsigned *base_count, int zero_base, unsigned *host_counts, void *stream)
{
    const int S = g_params.n_scales, cap = g_params.max_points;
    const int lowe = c->profile == SIFT_PROFILE_LOWE, up = c->upsample, n_oct = c->n_oct;
    /* the group counters start at zero; so do the point counts unless they carry on from an
     * earlier call (extractSift, one call per octave) */
    int r = hl_zero_words(c->counts, (long)COUNT_WORDS * nimg, zero_base ? base_count : NULL,
                          zero_base ? nimg : 0, stream);
    if (r) return fail_hl(r, "resetting group counters");
    float *A = c->arena;
    const long AS = c->arena_stride;
    unsigned *queued = c->counts + HL_MAX_GROUPS; /* second half of each image's counter block */
    const long set_px = (long)nimg * c->ow[0] * c->oh[0];
    const int prefilter = c->queue_cap > 0 && set_px > g_dense_route_px;
    hl_detect_args da;
    memset(&da, 0, sizeof da);
    da.n_scales = S;
    da.n_oct = n_oct;
    da.cap = cap;
    da.edge_limit = (g_params.r_thresh * g_params.r_thresh) / g_params.r_thresh; /* cudaSiftD.cu:292 */
    da.contrast = g_params.extrema_thresh;
    da.pre_thresh = 0.5f * g_params.extrema_thresh / (float)S;
    da.r_thresh = g_params.r_thresh;
    da.sigma = g_params.sigma;
    /* the test the pyramid pass applies before it queues a sample */
    const float queue_thresh = lowe ? da.pre_thresh : da.contrast;
    float pix = up ? 0.5f : 1.0f;
    for (int k = 0; k < oct0; ++k)
        pix *= 2.0f;
    if (up) {
        r = hl_upsample2x(imgs, imgs_u8, pitch, img_stride, A + c->up_off, 2 * c->w, AS, c->w, c->h, nimg,
                          stream);
        if (r) return fail_hl(r, "up-sampling");
    }
    for (int oi = 0; oi < n_oct; ++oi, pix *= 2.0f) {
        const int ww = c->ow[oi], hh = c->oh[oi], pp = c->op[oi];
        da.oct[oi].w = ww;
        da.oct[oi].h = hh;
        da.oct[oi].p = pp;
        da.oct[oi].qcap = c->qcap[oi];
        da.oct[oi].blur_off = c->blur_off[oi];
        da.oct[oi].queue_off = c->queue_off[oi];
        da.pix[oi] = pix;
        da.dense0[oi + 1] = da.dense0[oi] + idivup(ww, 96) * idivup(hh, 32);
        unsigned *q = prefilter ? c->queue + c->queue_off[oi] : NULL;
        if (oi == 0 && up)
            r = hl_pyramid(A + c->up_off, ww, hh, ww, AS, 0, A + c->blur_off[oi], AS, ww, pp, hh, S, nimg,
                           &g_taps, queue_thresh, q, c->queue_stride, queued, COUNT_WORDS, c->qcap[oi],
                           oi * S, lowe, stream);
        else if (oi == 0)
            r = hl_pyramid(imgs, ww, hh, pitch, img_stride, imgs_u8 ? 1 : 0, A + c->blur_off[oi], AS, ww,
                           pp, hh, S, nimg, &g_taps, queue_thresh, q, c->queue_stride, queued,
                           COUNT_WORDS, c->qcap[oi], oi * S, lowe, stream);
        else /* cudaSiftH.cu:50-52: the next octave is the even pixels of blur plane S */
            r = hl_pyramid(A + c->blur_off[oi - 1] + (long)S * c->op[oi - 1] * c->oh[oi - 1],
                           c->ow[oi - 1], c->oh[oi - 1], c->op[oi - 1], AS, 2, A + c->blur_off[oi],
                           AS, ww, pp, hh, S, nimg, &g_taps, queue_thresh, q, c->queue_stride, queued,
                           COUNT_WORDS, c->qcap[oi], oi * S, lowe, stream);
        if (r) return fail_hl(r, "pyramid pass");
    }
    if (lowe) {
        /* strict extrema, refinement and rejection over the queues; a dense pass
         * redoes the groups whose queue overflowed */
        if (prefilter) {
            r = hl_verify_lowe(A, AS, c->queue, c->queue_stride, queued, COUNT_WORDS, c->cand, c->sub,
                               c->cand_stride, c->counts, COUNT_WORDS, &da, nimg, stream);
            if (r) return fail_hl(r, "candidate refinement");
        }
        r = hl_detect_lowe(A, AS, c->cand, c->sub, c->cand_stride, c->counts, COUNT_WORDS,
                           prefilter ? queued : NULL, COUNT_WORDS, &da, nimg, stream);
        if (r) return fail_hl(r, "dense detection");
    } else {
        /* detection, once for all octaves: the queues are verified, and (only where a
         * queue overflowed) a dense pass redoes the group */
        if (prefilter) {
            r = hl_verify(A, AS, c->queue, c->queue_stride, queued, COUNT_WORDS, c->cand, c->cand_stride,
                          c->counts, COUNT_WORDS, &da, nimg, stream);
            if (r) return fail_hl(r, "candidate verification");
        }
        r = hl_detect(A, AS, c->cand, c->cand_stride, c->counts, COUNT_WORDS, prefilter ? queued : NULL,
                      COUNT_WORDS, &da, nimg, stream);
        if (r) return fail_hl(r, "dense detection");
    }
    hl_keypoint_args ka;
    memset(&ka, 0, sizeof ka);
    ka.n_scales = S;
    ka.n_groups = n_oct * S;
    ka.first_octave = oct0;
    ka.cap = cap;
    for (int oi = 0; oi < n_oct; ++oi) {
        ka.oct[oi].w = c->ow[oi];
        ka.oct[oi].h = c->oh[oi];
        ka.oct[oi].p = c->op[oi];
        ka.oct[oi].blur_off = c->blur_off[oi];
        for (int i = 0; i < S; ++i)
            ka.scale_value[oi * S + i] = scale_value(i, oct0 + oi);
    }
    memcpy(ka.wnd, g_wnd, sizeof(float) * S * 16);
    memcpy(ka.desc_wnd, g_desc, sizeof g_desc);
    memcpy(ka.cos32, g_cos32, sizeof g_cos32);
    memcpy(ka.sin32, g_sin32, sizeof g_sin32);
    r = hl_keypoints(pts, cap, meta, A, AS, c->cand, lowe ? c->sub : NULL, c->cand_stride, c->counts,
                     COUNT_WORDS, base_count, &ka, nimg, stream);
    if (r) return fail_hl(r, "keypoint kernel");
    r = hl_finalize_counts(base_count, host_counts, c->counts, COUNT_WORDS, ka.n_groups, cap, nimg, stream);
    if (r) return fail_hl(r, "count update");
    if (c->done) { /* (the entry that allocates nothing has no event: its caller orders by stream) */
        r = hl_event_record(c->done, stream);
        if (r) return fail_hl(r, "recording slot event");
        c->busy = 1;
        c->on_lane = 0; /* (the batch entry sets it when the stream was one of its own) */
    }
    return 0;
}

/* ---- reference entry points ------------------------------------------------ */
void initDeviceConstant(void)
{
    pthread_mutex_lock(&g_mu);
    build_tables();
    pthread_mutex_unlock(&g_mu);
}

void initDeviceVariables(void)
{
    const int dev = lock_device("initDeviceVariables");
    if (dev < 0) return;
    {
        /* set-up work that may allocate happens here, not in the entry that must not:
         * the device's address of the static scratch buffer is looked up once */
        void *base = NULL;
        size_t have = 0;
        int r = hl_scratch(&base, &have);
        if (r) fail_hl(r, "initDeviceVariables: static scratch buffer");
    }
    if (ensure_slot_counter(dev) == 0) {
        /* The reference zeroes its two counters with a blocking copy that follows all earlier
         * work (cudaSiftH.cu:120-127).  Here the slots' own work is waited for through their
         * events (normally long finished: getPointCount has been called), the host copies of the
         * counts are zeroed now, and each device word is zeroed by the first kernel of the slot's
         * next extractSift, on that call's stream: no device-wide wait, no copy. */
        for (int sl = 0; sl < SIFT_MAX_SLOTS; ++sl) {
            drain_slot(&g_slots[dev][sl], "initDeviceVariables");
            g_slot_host[dev][sl] = 0u;
            g_slot_reset[dev][sl] = 1;
        }
    }
    unlock_device(dev);
}

void extractSift(SiftPoint *siftPt, float *d_src, int width, int pitch, int height, void *stream,
                 int streamIdx, int octaveIdx)
{
    const int dev = lock_device("extractSift");
    if (dev < 0) return;
    do {
        if (tables_ok())
            break;
        if (!siftPt || !d_src || width < 1 || height < 1 || pitch < width) {
            set_err("extractSift: bad image arguments (w=%d pitch=%d h=%d)", width, pitch, height);
            break;
        }
        if (width > 65535 || height > 65535) {
            set_err("extractSift: image larger than 65535 pixels on a side");
            break;
        }
        if (streamIdx < 0 || streamIdx >= SIFT_MAX_SLOTS) {
            set_err("extractSift: streamIdx %d outside 0..%d", streamIdx, SIFT_MAX_SLOTS - 1);
            break;
        }
        if (octaveIdx < 0 || octaveIdx >= g_params.n_octaves) /* cudaSiftH.cu:41: nothing to do */
            break;
        if (octaveIdx > 0 && g_params.upsample) {
            set_err("extractSift: octaveIdx must be 0 when the first octave is up-sampled");
            break;
        }
        if (ensure_slot_counter(dev))
            break;
        slot_ctx *c = &g_slots[dev][streamIdx];
        if (ensure_arena(c, width, height, octaveIdx, 1))
            break;
        /* The slot's planes, queues and counters may still be in use by its previous call,
         * which may have been given another stream (the reference's extractSift blocks, so
         * there any two calls on one streamIdx are ordered, cudaSiftH.cu:22-24,158-177). */
        if (c->busy) {
            int r = hl_stream_wait_event(stream, c->done);
            if (r) {
                fail_hl(r, "extractSift: ordering behind the slot's previous call");
                break;
            }
        }
        const int reset = g_slot_reset[dev][streamIdx];
        if (enqueue_extract(c, siftPt, NULL, d_src, 0, 0, 1, pitch, octaveIdx, g_slot_counter[dev] + streamIdx,
                            reset, g_slot_host[dev] + streamIdx, stream) == 0)
            g_slot_reset[dev][streamIdx] = 0;
    } while (0);
    unlock_device(dev);
}

int getPointCount(int streamIdx)
{
    int result = -1;
    const int dev = lock_device("getPointCount");
    if (dev < 0) return -1;
    do {
        if (streamIdx < 0 || streamIdx >= SIFT_MAX_SLOTS) {
            set_err("getPointCount: streamIdx %d outside 0..%d", streamIdx, SIFT_MAX_SLOTS - 1);
            break;
        }
        if (ensure_slot_counter(dev))
            break;
        slot_ctx *c = &g_slots[dev][streamIdx];
        if (drain_slot(c, "getPointCount"))
            break;
        /* the slot's last kernel wrote the count to pinned host memory (cudaSiftH.cu:323-328
         * copies it from the device symbol with a blocking call) */
        result = (int)*(volatile unsigned *)(g_slot_host[dev] + streamIdx);
    } while (0);
    unlock_device(dev);
    return result;
}

int sift_get_group_counts(int streamIdx, unsigned *accepted, unsigned *queued, int n)
{
    if (streamIdx < 0 || streamIdx >= SIFT_MAX_SLOTS || !accepted || !queued || n < 1) {
        set_err("sift_get_group_counts: bad arguments");
        return -1;
    }
    const int dev = lock_device("sift_get_group_counts");
    if (dev < 0) return -1;
    int result = -1;
    slot_ctx *c = &g_slots[dev][streamIdx];
    if (!c->valid) {
        set_err("sift_get_group_counts: slot %d has not been used", streamIdx);
    } else {
        unsigned words[COUNT_WORDS];
        int r = drain_slot(c, "sift_get_group_counts");
        if (r) {
            unlock_device(dev);
            return -1;
        }
        r = hl_memcpy_async(words, c->counts, sizeof words, 2, NULL);
        if (!r) r = hl_stream_sync(NULL);
        if (r) {
            fail_hl(r, "sift_get_group_counts");
        } else {
            const int groups = c->n_oct * c->S;
            result = groups < n ? groups : n;
            for (int g = 0; g < result; ++g) {
                accepted[g] = words[g];
                queued[g] = words[HL_MAX_GROUPS + g];
            }
        }
    }
    unlock_device(dev);
    return result;
}

int sift_debug_read_planes(int slot, int image, int octave, float *h_dst, long dst_floats, int *width,
                           int *pitch, int *height)
{
    if (slot < 0 || slot >= SIFT_MAX_SLOTS + N_LANES || image < 0 || octave < 0) {
        set_err("sift_debug_read_planes: bad arguments");
        return -1;
    }
    const int dev = lock_device("sift_debug_read_planes");
    if (dev < 0) return -1;
    int rc = -1;
    slot_ctx *c = &g_slots[dev][slot];
    do {
        if (!c->valid || image >= c->nimg || octave >= c->n_oct) {
            set_err("sift_debug_read_planes: slot %d holds no image %d / octave %d", slot, image, octave);
            break;
        }
        if (drain_slot(c, "sift_debug_read_planes"))
            break;
        const long n = (long)c->op[octave] * c->oh[octave] * (c->S + 3);
        if (width) *width = c->ow[octave];
        if (pitch) *pitch = c->op[octave];
        if (height) *height = c->oh[octave];
        if (!h_dst) { /* geometry query */
            rc = 0;
            break;
        }
        if (dst_floats < n) {
            set_err("sift_debug_read_planes: destination holds %ld floats, %ld needed", dst_floats, n);
            break;
        }
        int r = hl_memcpy_async(h_dst, c->arena + (long)image * c->arena_stride + c->blur_off[octave],
                                sizeof(float) * (size_t)n, 2, NULL);
        if (!r) r = hl_stream_sync(NULL);
        if (r) {
            fail_hl(r, "sift_debug_read_planes");
            break;
        }
        rc = 0;
    } while (0);
    unlock_device(dev);
    return rc;
}

/* ---- one image, no allocation ------------------------------------------------------ */
#define NOALLOC_QUEUE_CAP 16384 /* first-octave candidates per scale before the dense detector takes over */

int sift_extract_noalloc(SiftPoint *d_pts, unsigned *d_count, const float *d_img, int width, int pitch,
                         int height, void *stream)
{
    int rc = -1;
    const int dev = lock_device("sift_extract_noalloc");
    if (dev < 0) return -1;
    do {
        if (tables_ok())
            break;
        if (!d_pts || !d_count || !d_img || width < 1 || height < 1 || pitch < width) {
            set_err("sift_extract_noalloc: bad arguments (w=%d pitch=%d h=%d)", width, pitch, height);
            break;
        }
        if (width > 65535 || height > 65535) {
            set_err("sift_extract_noalloc: image larger than 65535 pixels on a side");
            break;
        }
        void *base = NULL;
        size_t have = 0;
        int r = hl_scratch(&base, &have); /* the device's copy of the static buffer (looked up once per device) */
        if (r) {
            fail_hl(r, "sift_extract_noalloc: static scratch buffer");
            break;
        }
        slot_ctx c;
        size_t bytes[5], off[5], need = 0;
        plan_arena(&c, width, height, 0, 1, NOALLOC_QUEUE_CAP, bytes);
        for (int k = 0; k < 5; ++k) {
            off[k] = need;
            need += (bytes[k] + 255) & ~(size_t)255;
        }
        if (need > have) {
            set_err("sift_extract_noalloc: a %dx%d image with %d scales, %d octaves and max_points %d needs %zu "
                    "bytes of scratch, the static buffer holds %zu",
                    width, height, c.S, c.O, c.cap, need, have);
            break;
        }
        char *b = (char *)base;
        c.arena = (float *)(b + off[0]);
        c.counts = (unsigned *)(b + off[1]);
        c.queue = (unsigned *)(b + off[2]);
        c.cand = (unsigned *)(b + off[3]);
        c.sub = bytes[4] ? (void *)(b + off[4]) : NULL;
        /* the list starts empty: the count is an output, not an accumulator as in extractSift;
         * the launches are never timed here (per-kernel timing creates events, which allocates) */
        hl_timing_hold(1);
        const int bad = enqueue_extract(&c, d_pts, NULL, d_img, 0, 0, 1, pitch, 0, d_count, 1, NULL, stream);
        hl_timing_hold(0);
        if (bad)
            break;
        rc = 0;
    } while (0);
    unlock_device(dev);
    return rc;
}

size_t sift_noalloc_scratch_bytes(void) { return hl_scratch_bytes(); }

/* ---- batch ------------------------------------------------------------------ */
static int ensure_lanes(int dev)
{
    for (int l = 0; l < N_LANES; ++l)
        if (!g_lane_stream[dev][l]) {
            int r = hl_stream_create(&g_lane_stream[dev][l]);
            if (r) return fail_hl(r, "creating batch streams");
        }
    for (int l = 0; l <= N_LANES; ++l)
        if (!g_lane_event[dev][l]) {
            int r = hl_event_create(&g_lane_event[dev][l]);
            if (r) return fail_hl(r, "creating batch events");
        }
    return 0;
}

void sift_set_batch_lanes(int n)
{
    if (n >= 1 && n <= N_LANES)
        g_lanes = n;
}

/* How a batch of n images is cut into launch sets: the fewest sets of at most g_chunk
 * images, of equal size.  Set k holds images [k*chunk, min(n, (k+1)*chunk)) and runs on
 * lane k % lanes.  (Cutting a batch that fits one set in two, one per lane, was measured
 * slower: 48.4 k against 49.8 k images/s at 256 images of 1080p.) */
static void batch_sets(int n, int *chunk, int *n_sets)
{
    const int sets = (n + g_chunk - 1) / g_chunk;
    *chunk = (n + sets - 1) / sets;
    *n_sets = (n + *chunk - 1) / *chunk;
}

int sift_debug_batch_slot(int n, int i, int *slot, int *image)
{
    if (n < 1 || i < 0 || i >= n || !slot || !image) return -1;
    const int dev = hl_get_device();
    if (dev < 0 || dev >= MAX_DEV) return -1;
    int rc = -1;
    pthread_mutex_lock(&g_dev_mu[dev]);
    /* the lanes the most recent batch call (blocking or not) really used */
    if (g_last_batch[dev].valid && g_last_batch[dev].n == n) {
        const int chunk = g_last_batch[dev].chunk, n_sets = g_last_batch[dev].n_sets;
        const int lanes = g_last_batch[dev].lanes, k = i / chunk;
        if (k + lanes >= n_sets) { /* (a later set has not reused that lane's arena) */
            *slot = BATCH_SLOT + (g_last_batch[dev].lane0 + k) % g_lanes;
            *image = i % chunk;
            rc = 0;
        }
    }
    pthread_mutex_unlock(&g_dev_mu[dev]);
    return rc;
}

void sift_set_strip_rows(int rows) { hl_set_pyramid_strip_rows(rows > 0 ? rows : 0); }
void sift_set_pyramid_columns(int cols, int min_width) { hl_set_pyramid_columns(cols, min_width); }
void sift_set_pyramid_split(int on) { hl_set_pyramid_split(on); }
int sift_debug_strip_rows(int width, int height, int images, int n_scales, int kernel)
{
    return hl_debug_strip_rows(width, height, images, n_scales, kernel);
}

/* h_counts != NULL: the counts go to the host and the call blocks until they are there.
 * Otherwise they go to d_counts (device, n words) and the call returns after enqueueing. */
static int extract_batch(SiftPoint *d_pts, unsigned char *d_meta, int *h_counts, unsigned *d_counts,
                         const void *d_imgs, int imgs_u8, long img_stride, int n, int width, int pitch,
                         int height, void *stream)
{
    const long esz = imgs_u8 ? 1 : (long)sizeof(float);
    int rc = -1;
    const int dev = lock_device("sift_extract_batch");
    if (dev < 0) return -1;
    do {
        if (tables_ok())
            break;
        if (!d_pts || !d_imgs || (!h_counts && !d_counts) || n < 1 || width < 1 || height < 1 || pitch < width ||
            img_stride < (long)pitch * (height - 1) + width) {
            set_err("sift_extract_batch: bad arguments");
            break;
        }
        if (width > 65535 || height > 65535) {
            set_err("sift_extract_batch: image larger than 65535 pixels on a side");
            break;
        }
        int chunk, n_sets;
        batch_sets(n, &chunk, &n_sets);
        const int lanes = n_sets < g_lanes ? n_sets : g_lanes;
        /* A blocking call starts on the first lane.  Non-blocking calls take the lanes in turn,
         * so that the short kernels at the end of one call run beside the long ones of the next
         * (given that the caller queues them on different streams): each lane has its own
         * arena, and calls that share a lane follow each other on its stream. */
        const int lane0 = h_counts ? 0 : g_next_lane[dev] % g_lanes;
        if (!h_counts) g_next_lane[dev] = (lane0 + n_sets) % g_lanes;
        g_last_batch[dev].valid = 0;
        if (ensure_lanes(dev))
            break;
        int bad = 0;
        for (int l = 0; l < lanes && !bad; ++l)
            bad = ensure_arena(&g_slots[dev][BATCH_SLOT + (lane0 + l) % g_lanes], width, height, 0, chunk);
        if (bad)
            break;
        if (!d_counts && g_batch_count_n[dev] < n) {
            if (g_batch_count[dev]) {
                hl_device_sync();
                hl_free(g_batch_count[dev]);
                g_batch_count[dev] = NULL;
                g_batch_count_n[dev] = 0;
            }
            void *p = NULL;
            int r = hl_malloc(&p, sizeof(unsigned) * (size_t)n);
            if (r) {
                fail_hl(r, "allocating batch counters");
                break;
            }
            g_batch_count[dev] = (unsigned *)p;
            g_batch_count_n[dev] = n;
        }
        if (h_counts && g_host_count_n[dev] < n) { /* pinned mirror of the counts, written by the device */
            if (g_host_count[dev]) {
                hl_device_sync();
                hl_host_free(g_host_count[dev]);
                g_host_count[dev] = NULL;
                g_host_count_n[dev] = 0;
            }
            void *p = NULL;
            int r = hl_host_alloc(&p, sizeof(unsigned) * (size_t)n);
            if (r) {
                fail_hl(r, "allocating the host copy of the batch counters");
                break;
            }
            g_host_count[dev] = (unsigned *)p;
            g_host_count_n[dev] = n;
        }
        unsigned *bc = d_counts ? d_counts : g_batch_count[dev];
        unsigned *hc = h_counts ? g_host_count[dev] : NULL;
        /* A blocking call of one launch set has nothing to run beside: it goes on the caller's
         * stream as it is, without the fork to a private stream and the join back (two
         * cross-stream waits: 60 us of a one-image call, 1 % of a step of 256 1080p images). */
        const int direct = h_counts && n_sets == 1;
        int r = 0;
        if (!direct) {
            /* fork: the private streams start after everything queued on the caller's stream */
            r = hl_event_record(g_lane_event[dev][N_LANES], stream);
            for (int l = 0; l < lanes && !r; ++l)
                r = hl_stream_wait_event(g_lane_stream[dev][(lane0 + l) % g_lanes], g_lane_event[dev][N_LANES]);
            if (r) {
                fail_hl(r, "starting batch streams");
                break;
            }
        }
        const int cap = g_params.max_points;
        for (int k = 0; k < n_sets && !bad; ++k) {
            const int i0 = k * chunk;
            const int m = (n - i0) < chunk ? (n - i0) : chunk;
            const int l = (lane0 + k) % g_lanes;
            slot_ctx *c = &g_slots[dev][BATCH_SLOT + l];
            void *st = direct ? stream : g_lane_stream[dev][l];
            /* the arena's previous launch set ran on this lane's private stream (then stream
             * order is enough) or on some caller's stream (then its event is waited for) */
            if (c->busy && (direct || c->on_lane != 1 + l)) {
                r = hl_stream_wait_event(st, c->done);
                if (r) {
                    fail_hl(r, "ordering behind the arena's previous launch set");
                    bad = 1;
                    break;
                }
            }
            bad = enqueue_extract(c, d_pts + (long)i0 * cap, d_meta ? d_meta + 2L * i0 * cap : NULL,
                                  (const char *)d_imgs + (long)i0 * img_stride * esz, imgs_u8,
                                  img_stride, m, pitch, 0, bc + i0, 1, hc ? hc + i0 : NULL, st);
            if (!bad && !direct) c->on_lane = 1 + l;
        }
        /* join: the caller's stream continues after both private streams */
        for (int q = 0; q < lanes && !direct; ++q) {
            const int l = (lane0 + q) % g_lanes;
            int r2 = hl_event_record(g_lane_event[dev][l], g_lane_stream[dev][l]);
            if (!r2) r2 = hl_stream_wait_event(stream, g_lane_event[dev][l]);
            if (r2 && !r) r = r2;
        }
        if (bad)
            break;
        if (!r && h_counts) { /* the last kernel of every set wrote its counts to the pinned mirror */
            r = hl_stream_sync(stream);
            if (!r) memcpy(h_counts, hc, sizeof(unsigned) * (size_t)n);
        }
        if (r) {
            fail_hl(r, "finishing batch");
            break;
        }
        g_last_batch[dev].n = n;
        g_last_batch[dev].chunk = chunk;
        g_last_batch[dev].n_sets = n_sets;
        g_last_batch[dev].lane0 = lane0;
        g_last_batch[dev].lanes = lanes;
        g_last_batch[dev].valid = 1;
        rc = 0;
    } while (0);
    unlock_device(dev);
    return rc;
}

int sift_quantize_descriptors(const SiftPoint *d_pts, int n, unsigned char *d_out, void *stream)
{
    int r = hl_quantize(d_pts, n, d_out, stream);
    return r ? fail_hl(r, "sift_quantize_descriptors") : 0;
}

int sift_extract_batch(SiftPoint *d_pts, unsigned char *d_meta, int *h_counts, const float *d_imgs,
                       long img_stride, int n, int width, int pitch, int height, void *stream)
{
    return extract_batch(d_pts, d_meta, h_counts, NULL, d_imgs, 0, img_stride, n, width, pitch, height, stream);
}

int sift_extract_batch_async(SiftPoint *d_pts, unsigned char *d_meta, unsigned *d_counts, const void *d_imgs,
                             int imgs_u8, long img_stride, int n, int width, int pitch, int height,
                             void *stream)
{
    if (!d_counts) {
        set_err("sift_extract_batch_async: d_counts is NULL");
        return -1;
    }
    return extract_batch(d_pts, d_meta, NULL, d_counts, d_imgs, imgs_u8 != 0, img_stride, n, width, pitch,
                         height, stream);
}

int sift_extract_batch_u8(SiftPoint *d_pts, unsigned char *d_meta, int *h_counts,
                          const unsigned char *d_imgs, long img_stride, int n, int width, int pitch,
                          int height, void *stream)
{
    return extract_batch(d_pts, d_meta, h_counts, NULL, d_imgs, 1, img_stride, n, width, pitch, height, stream);
}

/* ---- stage launchers (dense planes, reference layout) ------------------------ */
static int stage_ok(const char *name, const void *a, const void *b, int w, int p, int h)
{
    if (tables_ok())
        return 0;
    if (!a || !b || w < 1 || h < 1 || p < w) {
        set_err("%s: bad arguments (w=%d pitch=%d h=%d)", name, w, p, h);
        return 0;
    }
    return 1;
}

static void blurOctave_held(float *dst, float *src, int width, int pitch, int height, void *stream)
{
    if (!stage_ok("blurOctave", dst, src, width, pitch, height))
        return;
    if (pitch & 3) {
        set_err("blurOctave: pitch must be a multiple of 4 floats");
        return;
    }
    const int S = g_params.n_scales;
    const long plane = (long)pitch * height;
    if ((pitch & 63) == 0 && plane * (S + 3) * 4 < 0xfffffff0L) {
        /* the fused pass of the extraction path, without its candidate queues: the same planes
         * bit for bit, nothing allocated and no wait */
        int r = hl_pyramid(src, width, height, pitch, 0, 0, dst, 0, width, pitch, height, S, 1, &g_taps, 0.f,
                           NULL, 0, NULL, 0, 0, 0, 0, stream);
        if (r) fail_hl(r, "blurOctave");
        return;
    }
    /* other pitches: row pass into scratch, column pass out of it (cudaSiftH.cu:218-258) */
    void *tmp = NULL;
    int r = hl_malloc(&tmp, sizeof(float) * (size_t)plane * (S + 3));
    if (r) {
        fail_hl(r, "blurOctave scratch");
        return;
    }
    r = hl_row_blur(src, pitch, 0, (float *)tmp, 0, width, pitch, height, S, 1, &g_taps, stream);
    if (r) fail_hl(r, "blurOctave row pass");
    if (!r) {
        r = hl_col_blur_dog((const float *)tmp, 0, dst, 0, 1, NULL, 0, NULL, 0, 0, width, pitch,
                            height, S, 1, &g_taps, stream);
        if (r) fail_hl(r, "blurOctave column pass");
    }
    hl_stream_sync(stream); /* the scratch is freed here, as cudaSiftH.cu:258 does */
    hl_free(tmp);
}

static void computeDiffOfGauss_held(float *dst, float *src, int width, int pitch, int height, void *stream)
{
    if (!stage_ok("computeDiffOfGauss", dst, src, width, pitch, height))
        return;
    const long plane = (long)pitch * height;
    for (int i = 0; i < g_params.n_scales + 2; ++i) { /* cudaSiftH.cu:314-315 */
        int r = hl_subtract(dst + i * plane, src + i * plane, src + (i + 1) * plane, width, pitch,
                            height, stream);
        if (r) {
            fail_hl(r, "computeDiffOfGauss");
            return;
        }
    }
}

static void computeHessian_held(float *dst, float *src, int width, int pitch, int height, void *stream)
{
    if (!stage_ok("computeHessian", dst, src, width, pitch, height))
        return;
    const long plane = (long)pitch * height;
    for (int i = 0; i < g_params.n_scales; ++i) { /* cudaSiftH.cu:277-282 */
        int r = hl_hessian(dst + 3L * i * plane, src + (i + 1) * plane, width, pitch, height, stream);
        if (r) {
            fail_hl(r, "computeHessian");
            return;
        }
    }
}

static void computeGradient_held(float *dst, float *src, int width, int pitch, int height, void *stream)
{
    if (!stage_ok("computeGradient", dst, src, width, pitch, height))
        return;
    const long plane = (long)pitch * height;
    for (int i = 0; i < g_params.n_scales; ++i) { /* cudaSiftH.cu:299-304 */
        int r = hl_gradient(dst + 2L * i * plane, src + (i + 1) * plane, width, pitch, height, stream);
        if (r) {
            fail_hl(r, "computeGradient");
            return;
        }
    }
}

/* the stage entries read the parameters and tables: held for reading while they enqueue */
#define STAGE_ENTRY(name)                                                                      \
    void name(float *dst, float *src, int width, int pitch, int height, void *stream)         \
    {                                                                                          \
        pthread_rwlock_rdlock(&g_params_rw);                                                   \
        name##_held(dst, src, width, pitch, height, stream);                                   \
        pthread_rwlock_unlock(&g_params_rw);                                                   \
    }
STAGE_ENTRY(blurOctave)
STAGE_ENTRY(computeDiffOfGauss)
STAGE_ENTRY(computeHessian)
STAGE_ENTRY(computeGradient)
#undef STAGE_ENTRY

void copyDeviceData(float *dst, float *src, int width, int pitch, int height, void *stream)
{
    if (!stage_ok("copyDeviceData", dst, src, width, pitch, height))
        return;
    int r = hl_copy(dst, src, width, pitch, height, stream);
    if (r) fail_hl(r, "copyDeviceData");
}

/* ---- thin device layer ---------------------------------------------------------- */
#define WRAP(call, what)            \
    do {                            \
        int r_ = (call);            \
        if (r_) fail_hl(r_, what);  \
        return r_;                  \
    } while (0)

int sift_device_count(void) { return hl_device_count(); }
int sift_set_device(int dev) { WRAP(hl_set_device(dev), "sift_set_device"); }
int sift_get_device(void) { return hl_get_device(); }
int sift_dev_malloc(void **p, size_t bytes) { WRAP(hl_malloc(p, bytes), "sift_dev_malloc"); }
int sift_dev_malloc_pitch(float **p, int *pitch_floats, int width, int height)
{
    size_t pb = 0;
    void *q = NULL;
    int r = hl_malloc_pitch(&q, &pb, sizeof(float) * (size_t)width, (size_t)height);
    if (r) return fail_hl(r, "sift_dev_malloc_pitch");
    *p = (float *)q;
    *pitch_floats = (int)(pb / sizeof(float)); /* cudaImage.cu:32 */
    return 0;
}
int sift_dev_free(void *p) { WRAP(hl_free(p), "sift_dev_free"); }
int sift_memcpy2d_h2d_async(float *d, int dp, const float *h, int hp, int w, int ht, void *s)
{
    WRAP(hl_memcpy2d_async(d, sizeof(float) * (size_t)dp, h, sizeof(float) * (size_t)hp,
                           sizeof(float) * (size_t)w, (size_t)ht, 1, s),
         "sift_memcpy2d_h2d_async");
}
int sift_memcpy2d_d2h_async(float *h, int hp, const float *d, int dp, int w, int ht, void *s)
{
    WRAP(hl_memcpy2d_async(h, sizeof(float) * (size_t)hp, d, sizeof(float) * (size_t)dp,
                           sizeof(float) * (size_t)w, (size_t)ht, 2, s),
         "sift_memcpy2d_d2h_async");
}
int sift_memcpy_h2d_async(void *d, const void *h, size_t n, void *s)
{
    WRAP(hl_memcpy_async(d, h, n, 1, s), "sift_memcpy_h2d_async");
}
int sift_memcpy_d2h_async(void *h, const void *d, size_t n, void *s)
{
    WRAP(hl_memcpy_async(h, d, n, 2, s), "sift_memcpy_d2h_async");
}
int sift_convert_u8_f32(float *d, int dp, const unsigned char *src, int sp, int w, int h, void *s)
{
    WRAP(hl_u8_to_f32(d, dp, src, sp, w, h, s), "sift_convert_u8_f32");
}
int sift_stream_create(void **s) { WRAP(hl_stream_create(s), "sift_stream_create"); }
int sift_stream_destroy(void *s) { WRAP(hl_stream_destroy(s), "sift_stream_destroy"); }
int sift_stream_sync(void *s) { WRAP(hl_stream_sync(s), "sift_stream_sync"); }
int sift_host_alloc_pinned(void **p, size_t n) { WRAP(hl_host_alloc(p, n), "sift_host_alloc_pinned"); }
int sift_host_free_pinned(void *p) { WRAP(hl_host_free(p), "sift_host_free_pinned"); }

int sift_measure_copy_ceiling(size_t bytes, int repeats, double *gbytes_per_s)
{
    WRAP(hl_copy_ceiling(bytes, repeats, gbytes_per_s), "sift_measure_copy_ceiling");
}

void sift_timing_enable(int on) { hl_timing_enable(on); }
int sift_timing_read(int kernel, double *ms, long *launches)
{
    return hl_timing_read(kernel, ms, launches);
}
void sift_timing_reset(void) { hl_timing_reset(); }
