/*
 * Thin C-ABI layer over the HIP runtime and the gfx950 kernels.  The C host
 * code (sift_host.c) sees only this header: plain pointers, ints and structs
 * passed by value.  Implemented in sift_kernels.hip.
 */
#ifndef SIFT_HIP_LAYER_H
#define SIFT_HIP_LAYER_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HL_TAPS 19
#define HL_HALF_TAPS 10 /* taps are symmetric: c[k] == c[18-k] */
#define HL_MAX_PLANES 7 /* n_scales + 3, n_scales <= 4 */
#define HL_MAX_OCT 8
#define HL_MAX_GROUPS 32 /* (octave, scale) pairs */
/* Per-image counter block, HL_COUNT_WORDS words in three cache lines: accepted
 * points per group, queued candidates per group, and at HL_FULL_WORD the
 * announcement HL_MAX_GROUPS - g of the first group g that reached the cap
 * (0 = none), kept away from the counters the atomics hammer. */
#define HL_FULL_WORD (2 * HL_MAX_GROUPS)
#define HL_COUNT_WORDS (3 * HL_MAX_GROUPS)

/* blur taps folded by symmetry and interleaved by plane:
 * t[m][i] = c[i*19 + m], m = 0..9, i = plane (padded to 8) */
typedef struct {
    float t[HL_HALF_TAPS][8];
} hl_taps;

/* geometry of one octave inside a per-image arena (offsets in floats) */
typedef struct {
    int w, h, p;        /* internal pitch p */
    int qcap;           /* capacity of each scale's contrast-candidate queue */
    long blur_off;      /* the S+3 blur planes, plane 0 first */
    long queue_off;     /* first of the octave's S queues, in words, inside an image's queue block */
} hl_octave;

/* everything the detection kernels need to work on all octaves in one launch */
typedef struct {
    int n_scales, n_oct, cap;
    float edge_limit;   /* (r_thresh^2)/r_thresh, cudaSiftD.cu:292 */
    float contrast;     /* EXTREMA_THRESH, cudaSiftD.cu:293 */
    /* profile "lowe" only */
    float pre_thresh;   /* 0.5 * contrast / n_scales: the test a sample must pass to be examined */
    float r_thresh;     /* edge ratio r */
    float sigma;        /* base sigma */
    float pix[HL_MAX_OCT]; /* input pixels per pixel of each octave (a power of two) */
    hl_octave oct[HL_MAX_OCT];
    int dense0[HL_MAX_OCT + 1]; /* first 96x32-pixel block of each octave (dense detector units) */
} hl_detect_args;

typedef struct {
    int n_scales;
    int n_groups;            /* octaves processed * n_scales */
    int first_octave;        /* octave index of group 0 */
    int cap;                 /* max_points */
    hl_octave oct[HL_MAX_OCT]; /* indexed by group / n_scales */
    float scale_value[HL_MAX_GROUPS];
    float wnd[4 * 16];       /* orientation window per scale index */
    float desc_wnd[16];
    float cos32[32], sin32[32];
} hl_keypoint_args;

/* runtime */
int hl_device_count(void);
int hl_set_device(int dev);
int hl_get_device(void);
int hl_malloc(void **p, size_t bytes);
int hl_malloc_pitch(void **p, size_t *pitch_bytes, size_t width_bytes, size_t height);
int hl_free(void *p);
int hl_memset_async(void *p, int v, size_t bytes, void *stream);
int hl_memcpy_async(void *dst, const void *src, size_t bytes, int kind, void *stream); /* 1 h2d 2 d2h 3 d2d */
int hl_memcpy2d_async(void *dst, size_t dpitch, const void *src, size_t spitch, size_t wbytes,
                      size_t h, int kind, void *stream);
int hl_stream_create(void **s);
int hl_stream_destroy(void *s);
int hl_stream_sync(void *s);
int hl_device_sync(void);
int hl_event_create(void **e);
int hl_event_destroy(void *e);
int hl_event_record(void *e, void *stream);
int hl_event_sync(void *e);
int hl_stream_wait_event(void *stream, void *e);
int hl_host_alloc(void **p, size_t bytes);
int hl_host_free(void *p);
const char *hl_error_string(int code);

/* rows per strip of the pyramid pass (0 = chosen per launch); for experiments */
void hl_set_pyramid_strip_rows(int rows);
/* 0: single-round launches without a candidate queue keep pyramid_kernel (default 1: the planes
 * are dealt over waves, pyramid_plane_kernel) */
void hl_set_pyramid_split(int on);
/* rows per strip the launchers would choose (kind 0: pyramid_kernel, 1: pyramid_plane_kernel,
 * 2: pyramid2_kernel), ignoring any override; -1 for bad arguments */
int hl_debug_strip_rows(int w, int h, int nimg, int n_scales, int kind);
/* columns per lane of the pyramid pass: 1, or 2 for octaves at least min_width wide (0 = keep) */
void hl_set_pyramid_columns(int cols, int min_width);

/* The static device buffer behind the entry point that allocates nothing: the current
 * device's address of it (looked up on the first call per device) and its size. */
int hl_scratch(void **ptr, size_t *bytes);
size_t hl_scratch_bytes(void);

/* device-to-device copy of `bytes` bytes, best of `repeats` timed passes: read + write GB/s */
int hl_copy_ceiling(size_t bytes, int repeats, double *gbps);

/* timing */
void hl_timing_enable(int on);
/* on: launches of the calling thread are not timed even when timing is enabled (timing creates events) */
void hl_timing_hold(int on);
int hl_timing_read(int kernel, double *ms, long *launches);
void hl_timing_reset(void);

/* kernels.  nimg images per launch; image z of a buffer at base + z*stride. */
int hl_row_blur(const float *src, int src_pitch, long src_stride, float *tmp, long tmp_stride,
                int w, int p, int h, int n_scales, int nimg, const hl_taps *taps, void *stream);
/* all_planes != 0: blur receives all S+3 planes and nothing else is stored;
 * otherwise blur receives planes 1..S (S planes), dog the S+2 DoG planes and
 * next_base (may be NULL) the even pixels of plane S with pitch next_p */
int hl_col_blur_dog(const float *tmp, long tmp_stride, float *blur, long blur_stride,
                    int all_planes, float *dog, long dog_stride, float *next_base,
                    long next_stride, int next_p, int w, int p, int h, int n_scales, int nimg,
                    const hl_taps *taps, void *stream);
/* fused row pass + column pass: source -> the S+3 blur planes, no intermediate
 * in HBM (DoG plane i is blur plane i minus blur plane i+1, formed by its readers).
 * src_mode 0: float image.  1: 8-bit image (src points at bytes, src_pitch and
 * src_stride count bytes).  2: the source is the previous octave's blur plane S
 * (src_w x src_h) and is read at its even rows and columns, so w = ceil(src_w/2),
 * h = ceil(src_h/2).  p must be a multiple of 64: columns [w, p) of the
 * destination planes are written with don't-care values.
 * queue != NULL: pixels passing the detector's contrast and border tests (where the
 * reference reads the centre value through an aliased shared-memory cell, y mod 32 in
 * {30, 31}: the value it reads) are queued per scale: entry (x | y<<16) at
 * queue[i*qcap + k], count in qcount[group0 + i]; a count above qcap means the
 * queue is incomplete and the dense detector must redo that scale.
 * plain != 0: every sample inside the border qualifies (profile "lowe"); 0: the
 * reference's tile rules apply (middle tiles never fire, aliased centre cells). */
int hl_pyramid(const void *src, int src_w, int src_h, int src_pitch, long src_stride, int src_mode,
               float *blur, long blur_stride, int w, int p, int h,
               int n_scales, int nimg, const hl_taps *taps, float contrast, unsigned *queue,
               long queue_stride, unsigned *qcount, long qcount_stride, int qcap, int group0,
               int plain, void *stream);
/* The detection launches cover every octave of the launch set.
 * hl_verify: rank + edge tests over the queues; accepted pixels are appended as
 *   (x | y<<16) to cand + group*cap, counted in counts[group].
 * hl_detect: the dense detector; with gate set (the queue counters) it only
 *   redoes the groups whose queue overflowed, with gate NULL it does everything. */
int hl_verify(const float *arena, long arena_stride, const unsigned *queue, long queue_stride,
              const unsigned *qcount, long qcount_stride, unsigned *cand, long cand_stride,
              unsigned *counts, long counts_stride, const hl_detect_args *a, int nimg, void *stream);
int hl_detect(const float *arena, long arena_stride, unsigned *cand, long cand_stride,
              unsigned *counts, long counts_stride, const unsigned *gate, long gate_stride,
              const hl_detect_args *a, int nimg, void *stream);
/* orientation + descriptor for every candidate; writes SiftPoint records
 * behind base_count[z] and meta (may be NULL) */
int hl_keypoints(void *pts, long pts_stride_points, unsigned char *meta, const float *arena,
                 long arena_stride, const unsigned *cand, const void *sub, long cand_stride,
                 const unsigned *counts, long counts_stride, const unsigned *base_count,
                 const hl_keypoint_args *args, int nimg, void *stream);
/* Profile "lowe" (not in the reference).  sub: one 16-byte record per candidate
 * slot beside cand (x, y in input pixels, scale, DoG plane index); NULL in
 * hl_keypoints means profile "ref".
 * hl_verify_lowe: strict extremum + refinement + rejection over the queues.
 * hl_detect_lowe: the same over every pixel of the groups whose queue overflowed
 *   (gate = the queue counters) or of all groups (gate NULL).
 * hl_upsample2x: float or 8-bit image -> float image of twice the size. */
int hl_verify_lowe(const float *arena, long arena_stride, const unsigned *queue, long queue_stride,
                   const unsigned *qcount, long qcount_stride, unsigned *cand, void *sub,
                   long cand_stride, unsigned *counts, long counts_stride, const hl_detect_args *a,
                   int nimg, void *stream);
int hl_detect_lowe(const float *arena, long arena_stride, unsigned *cand, void *sub, long cand_stride,
                   unsigned *counts, long counts_stride, const unsigned *gate, long gate_stride,
                   const hl_detect_args *a, int nimg, void *stream);
int hl_upsample2x(const void *src, int src_u8, int src_pitch, long src_stride, float *dst, int dst_pitch,
                  long dst_stride, int w, int h, int nimg, void *stream);
/* out[128*j + k] = min(255, round(512 * pts[j].data[k])) */
int hl_quantize(const void *pts, int n, unsigned char *out, void *stream);
/* base_count[z] = min(base_count[z] + sum(min(counts, cap)), cap); host_copy (pinned host memory
 * or NULL) receives the same values */
int hl_finalize_counts(unsigned *base_count, unsigned *host_copy, const unsigned *counts, long counts_stride,
                       int n_groups, int cap, int nimg, void *stream);
/* a[0..na) = 0 and b[0..nb) = 0 in one launch (either range may be empty) */
int hl_zero_words(unsigned *a, long na, unsigned *b, long nb, void *stream);

/* dense stage kernels */
int hl_subtract(float *dst, const float *a, const float *b, int w, int p, int h, void *stream);
int hl_hessian(float *dst, const float *src, int w, int p, int h, void *stream);
int hl_gradient(float *dst, const float *src, int w, int p, int h, void *stream);
int hl_copy(float *dst, const float *src, int w, int p, int h, void *stream);
/* 8-bit pixels to floats on the device (pitches in elements) */
int hl_u8_to_f32(float *dst, int dp, const unsigned char *src, int sp, int w, int h, void *stream);

#ifdef __cplusplus
}
#endif
#endif
