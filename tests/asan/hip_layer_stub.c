/*
 * CPU stand-in for the HIP layer (hip_layer.h), for sanitizer builds of the C host code only:
 * "device" memory is host memory, streams and events are tokens, copies happen at once, and
 * the kernels are replaced by argument checks that write nothing (every image yields zero
 * keypoints).  Test infrastructure: never linked into libsift_amd.so.
 */
#include "../../sift_amd/csrc/hip_layer.h"

#include <stdlib.h>
#include <string.h>

static __thread int g_dev = 0; /* the current device belongs to the calling thread, as in HIP */
int hl_device_count(void) { return 2; } /* two pretend devices: the per-device paths get exercised */
int hl_set_device(int dev)
{
    if (dev < 0 || dev >= 2) return -101;
    g_dev = dev;
    return 0;
}
int hl_get_device(void) { return g_dev; }
/* test hook: the n-th device allocation from now on fails (0 = the next one; -1 = none) */
int hl_stub_malloc_fail_at = -1;
int hl_malloc(void **p, size_t bytes)
{
    if (hl_stub_malloc_fail_at >= 0 && hl_stub_malloc_fail_at-- == 0) {
        *p = NULL;
        return -2;
    }
    *p = malloc(bytes ? bytes : 1);
    return *p ? 0 : -2;
}
int hl_malloc_pitch(void **p, size_t *pitch_bytes, size_t width_bytes, size_t height)
{
    *pitch_bytes = (width_bytes + 255) / 256 * 256;
    return hl_malloc(p, *pitch_bytes * height);
}
int hl_free(void *p)
{
    free(p);
    return 0;
}
int hl_memset_async(void *p, int v, size_t bytes, void *stream)
{
    (void)stream;
    memset(p, v, bytes);
    return 0;
}
int hl_memcpy_async(void *dst, const void *src, size_t bytes, int kind, void *stream)
{
    (void)kind;
    (void)stream;
    memmove(dst, src, bytes);
    return 0;
}
int hl_memcpy2d_async(void *dst, size_t dpitch, const void *src, size_t spitch, size_t wbytes, size_t h, int kind,
                      void *stream)
{
    (void)kind;
    (void)stream;
    for (size_t y = 0; y < h; ++y) memmove((char *)dst + y * dpitch, (const char *)src + y * spitch, wbytes);
    return 0;
}
int hl_stream_create(void **s) { return hl_malloc(s, 1); }
int hl_stream_destroy(void *s) { return hl_free(s); }
int hl_stream_sync(void *s)
{
    (void)s;
    return 0;
}
int hl_device_sync(void) { return 0; }
int hl_event_create(void **e) { return hl_malloc(e, 1); }
int hl_event_destroy(void *e) { return hl_free(e); }
int hl_event_record(void *e, void *stream)
{
    (void)stream;
    return e ? 0 : -1;
}
int hl_event_sync(void *e) { return e ? 0 : -1; }
int hl_stream_wait_event(void *stream, void *e)
{
    (void)stream;
    return e ? 0 : -1;
}
int hl_host_alloc(void **p, size_t bytes)
{
    *p = malloc(bytes ? bytes : 1);
    return *p ? 0 : -2;
}
int hl_host_free(void *p) { return hl_free(p); }
const char *hl_error_string(int code)
{
    (void)code;
    return "stub error";
}
void hl_set_pyramid_strip_rows(int rows) { (void)rows; }
void hl_set_pyramid_columns(int cols, int min_width) { (void)cols, (void)min_width; }
void hl_set_pyramid_split(int on) { (void)on; }
int hl_debug_strip_rows(int w, int h, int nimg, int n_scales, int kind) { (void)w, (void)nimg, (void)n_scales, (void)kind; return h; }
static unsigned char g_scratch[2][32u << 20];
int hl_scratch(void **ptr, size_t *bytes)
{
    *ptr = g_scratch[g_dev];
    *bytes = sizeof g_scratch[0];
    return 0;
}
size_t hl_scratch_bytes(void) { return sizeof g_scratch[0]; }
int hl_copy_ceiling(size_t bytes, int repeats, double *gbps)
{
    (void)bytes;
    (void)repeats;
    *gbps = 0.0;
    return 0;
}
void hl_timing_enable(int on) { (void)on; }
void hl_timing_hold(int on) { (void)on; }
int hl_timing_read(int kernel, double *ms, long *launches)
{
    (void)kernel;
    *ms = 0.0;
    *launches = 0;
    return 0;
}
void hl_timing_reset(void) {}

/* the kernels: the argument checks of the real launchers, nothing computed */
int hl_row_blur(const float *src, int src_pitch, long src_stride, float *tmp, long tmp_stride, int w, int p, int h,
                int n_scales, int nimg, const hl_taps *taps, void *stream)
{
    (void)src_stride, (void)tmp_stride, (void)taps, (void)stream;
    if (!src || !tmp || w <= 0 || h <= 0 || nimg <= 0 || p < w || (p & 3) || src_pitch < w) return -1001;
    return (n_scales == 2 || n_scales == 3) ? 0 : -1002;
}
int hl_col_blur_dog(const float *tmp, long tmp_stride, float *blur, long blur_stride, int all_planes, float *dog,
                    long dog_stride, float *next_base, long next_stride, int next_p, int w, int p, int h, int n_scales,
                    int nimg, const hl_taps *taps, void *stream)
{
    (void)tmp_stride, (void)blur_stride, (void)dog_stride, (void)next_base, (void)next_stride, (void)next_p;
    (void)taps, (void)stream, (void)n_scales;
    if (w <= 0 || h <= 0 || nimg <= 0 || p < w || !tmp || !blur) return -1001;
    return (!all_planes && !dog) ? -1001 : 0;
}
int hl_pyramid(const void *src, int src_w, int src_h, int src_pitch, long src_stride, int src_mode, float *blur,
               long blur_stride, int w, int p, int h, int n_scales, int nimg, const hl_taps *taps, float contrast,
               unsigned *queue, long queue_stride, unsigned *qcount, long qcount_stride, int qcap, int group0, int plain,
               void *stream)
{
    (void)src_stride, (void)blur_stride, (void)taps, (void)contrast, (void)queue_stride, (void)qcount_stride;
    (void)group0, (void)plain, (void)stream, (void)n_scales;
    if (w <= 0 || h <= 0 || nimg <= 0 || !src || !blur || src_pitch < src_w) return -1001;
    if (p < w || (p & 63)) return -1005;
    if (src_mode < 0 || src_mode > 2) return -1002;
    if (src_mode == 2 ? (w != (src_w + 1) / 2 || h != (src_h + 1) / 2) : (w != src_w || h != src_h)) return -1006;
    if (queue && (!qcount || qcap < 1)) return -1008;
    /* touch the first and last byte of what the kernel would write: the arena must hold it */
    ((volatile float *)blur)[0] = 0.f;
    ((volatile float *)blur)[(long)(nimg - 1) * blur_stride + (long)p * h * (n_scales + 3) - 1] = 0.f;
    return 0;
}
static int args_ok(const hl_detect_args *a, int nimg)
{
    return a && nimg > 0 && a->n_oct >= 1 && a->n_oct <= HL_MAX_OCT && a->cap >= 1;
}
int hl_verify(const float *arena, long arena_stride, const unsigned *queue, long queue_stride, const unsigned *qcount,
              long qcount_stride, unsigned *cand, long cand_stride, unsigned *counts, long counts_stride,
              const hl_detect_args *a, int nimg, void *stream)
{
    (void)arena_stride, (void)queue, (void)queue_stride, (void)qcount, (void)qcount_stride, (void)cand;
    (void)cand_stride, (void)counts, (void)counts_stride, (void)stream;
    return (arena && args_ok(a, nimg)) ? 0 : -1001;
}
int hl_detect(const float *arena, long arena_stride, unsigned *cand, long cand_stride, unsigned *counts,
              long counts_stride, const unsigned *gate, long gate_stride, const hl_detect_args *a, int nimg, void *stream)
{
    (void)arena_stride, (void)cand, (void)cand_stride, (void)counts, (void)counts_stride, (void)gate, (void)gate_stride;
    (void)stream;
    return (arena && args_ok(a, nimg)) ? 0 : -1001;
}
int hl_keypoints(void *pts, long pts_stride_points, unsigned char *meta, const float *arena, long arena_stride,
                 const unsigned *cand, const void *sub, long cand_stride, const unsigned *counts, long counts_stride,
                 const unsigned *base_count, const hl_keypoint_args *args, int nimg, void *stream)
{
    (void)pts_stride_points, (void)meta, (void)arena, (void)arena_stride, (void)cand, (void)sub, (void)cand_stride;
    (void)counts, (void)counts_stride, (void)base_count, (void)stream;
    return (pts && nimg > 0 && args->n_groups > 0 && args->n_groups <= HL_MAX_GROUPS) ? 0 : -1001;
}
int hl_verify_lowe(const float *arena, long arena_stride, const unsigned *queue, long queue_stride,
                   const unsigned *qcount, long qcount_stride, unsigned *cand, void *sub, long cand_stride,
                   unsigned *counts, long counts_stride, const hl_detect_args *a, int nimg, void *stream)
{
    (void)arena_stride, (void)queue, (void)queue_stride, (void)qcount, (void)qcount_stride, (void)cand;
    (void)cand_stride, (void)counts, (void)counts_stride, (void)stream;
    return (arena && sub && args_ok(a, nimg)) ? 0 : -1001;
}
int hl_detect_lowe(const float *arena, long arena_stride, unsigned *cand, void *sub, long cand_stride, unsigned *counts,
                   long counts_stride, const unsigned *gate, long gate_stride, const hl_detect_args *a, int nimg,
                   void *stream)
{
    (void)arena_stride, (void)cand, (void)cand_stride, (void)counts, (void)counts_stride, (void)gate, (void)gate_stride;
    (void)stream;
    return (arena && sub && args_ok(a, nimg)) ? 0 : -1001;
}
int hl_upsample2x(const void *src, int src_u8, int src_pitch, long src_stride, float *dst, int dst_pitch,
                  long dst_stride, int w, int h, int nimg, void *stream)
{
    (void)src_u8, (void)src_stride, (void)dst_stride, (void)stream;
    return (src && dst && w > 0 && h > 0 && nimg > 0 && src_pitch >= w && dst_pitch >= 2 * w) ? 0 : -1001;
}
int hl_quantize(const void *pts, int n, unsigned char *out, void *stream)
{
    (void)stream;
    return (pts && out && n >= 0) ? 0 : -1001;
}
int hl_zero_words(unsigned *a, long na, unsigned *b, long nb, void *stream)
{
    (void)stream;
    if (na < 0 || nb < 0 || (na > 0 && !a) || (nb > 0 && !b)) return -1001;
    if (na) memset(a, 0, sizeof(unsigned) * (size_t)na);
    if (nb) memset(b, 0, sizeof(unsigned) * (size_t)nb);
    return 0;
}
int hl_finalize_counts(unsigned *base_count, unsigned *host_copy, const unsigned *counts, long counts_stride, int n_groups,
                       int cap, int nimg, void *stream)
{
    (void)stream;
    for (int z = 0; z < nimg; ++z) {
        unsigned long long tot = base_count[z];
        for (int g = 0; g < n_groups; ++g) {
            const unsigned c = counts[z * counts_stride + g];
            tot += c < (unsigned)cap ? c : (unsigned)cap;
        }
        base_count[z] = (unsigned)(tot < (unsigned long long)cap ? tot : (unsigned long long)cap);
        if (host_copy) host_copy[z] = base_count[z];
    }
    return 0;
}
int hl_subtract(float *dst, const float *a, const float *b, int w, int p, int h, void *stream)
{
    (void)stream;
    return (dst && a && b && w > 0 && p >= w && h > 0) ? 0 : -1001;
}
int hl_hessian(float *dst, const float *src, int w, int p, int h, void *stream)
{
    (void)stream;
    return (dst && src && w > 0 && p >= w && h > 0) ? 0 : -1001;
}
int hl_gradient(float *dst, const float *src, int w, int p, int h, void *stream)
{
    (void)stream;
    return (dst && src && w > 0 && p >= w && h > 0) ? 0 : -1001;
}
int hl_copy(float *dst, const float *src, int w, int p, int h, void *stream)
{
    (void)stream;
    return (dst && src && w > 0 && p >= w && h > 0) ? 0 : -1001;
}
int hl_u8_to_f32(float *dst, int dp, const unsigned char *src, int sp, int w, int h, void *stream)
{
    (void)stream;
    if (!dst || !src || w < 1 || h < 1 || dp < w || sp < w) return -1001;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) dst[(long)dp * y + x] = (float)src[(long)sp * y + x];
    return 0;
}
