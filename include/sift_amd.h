/*
 * sift_amd -- MI355X (gfx950) SIFT extractor, C ABI.
 *
 * Drop-in boundary for the reference pipeline IjlalBaig/SIFT (Sift/src): the
 * four functions sift.cu calls (Sift/src/sift.cu:99-125) keep their names and
 * argument order; the CUDA stream reference becomes an opaque handle by value.
 * Everything is plain C: pointers, ints, floats.  No HIP type appears here.
 *
 * All device pointers are owned by the caller.  The library owns one scratch
 * arena per stream slot, sized on first use and reused afterwards.
 */
#ifndef SIFT_AMD_H
#define SIFT_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Sift/src/sift.h:7-13 -- 528-byte POD, identical layout. */
typedef struct {
    float xpos;        /* integer pixel column in the octave's own grid */
    float ypos;        /* integer pixel row in the octave's own grid */
    float scale;       /* 1.6 * sqrt(2)^(scaleIdx + nScales*octave), cudaSiftH.cu:94-95 */
    float orientation; /* degrees, multiple of 11.25 */
    float data[128];   /* L1-normalised 4x4x8 histogram */
} SiftPoint;

/* Runtime mirror of the compile-time configuration in Sift/src/utils.h:4-19.
 * A non-default value behaves as the reference would if recompiled with it
 * (the scale ratio stays sqrt(2) and the blur stays 19 taps). */
typedef struct {
    int n_octaves;        /* N_OCTAVES      4     utils.h:7  (1..8)  */
    int n_scales;         /* N_SCALES       2     utils.h:8  (2 or 3) */
    int max_points;       /* MAX_POINTCOUNT 2500  utils.h:10 */
    float sigma;          /* SIGMA          1.6f  utils.h:14 */
    float r_thresh;       /* R_THRESH       10.0f utils.h:16 */
    float extrema_thresh; /* EXTREMA_THRESH 10.0f utils.h:18 */
    int profile;          /* SIFT_PROFILE_REF (default) or SIFT_PROFILE_LOWE */
    int upsample;         /* profile lowe: put an octave built from the 2x up-sampled image in front */
} sift_params;

/* Semantics profiles.
 * REF: the reference's behaviour, quirks of its detector included; positions are
 *   integers in the grid of their octave (cudaSiftD.cu:299-300).
 * LOWE: what the reference lacks (no counterpart in Sift/src): strict 26-neighbour
 *   extrema of the same DoG pyramid, 3-D quadratic refinement of position and
 *   scale, contrast test |D(refined)|*S >= extrema_thresh, edge test on the DoG
 *   Hessian with det = Dxx*Dyy - Dxy^2; xpos, ypos in pixels of the input image,
 *   scale = sigma * 2^(octave + (scale index + offset)/S).  Orientation and
 *   descriptor are taken as in REF, at the sample the refinement ended on. */
#define SIFT_PROFILE_REF 0
#define SIFT_PROFILE_LOWE 1

#define SIFT_MAX_OCTAVES 8
#define SIFT_MAX_SLOTS 8 /* the reference has 2: d_PointCount[2], cudaSiftD.h:24-25 */

/* ---- the reference's entry points (Sift/src/cudaSiftH.h:6-30) ------------- */

/* cudaSiftH.cu:128-157 -- builds the Gaussian tap tables (host side; they
 * travel to the kernels as launch arguments, so there is nothing to upload). */
void initDeviceConstant(void);
/* cudaSiftH.cu:120-127 -- zeroes the point counter of every stream slot on
 * the current device.  Waits for work the slots still have in flight (the
 * reference's symbol copy follows all earlier work); getPointCount reads zero
 * from then on, and the slot's next extractSift starts its list at zero. */
void initDeviceVariables(void);
/* cudaSiftH.cu:9-60 -- one image: octaves octaveIdx..n_octaves-1, appending
 * keypoints with orientation and descriptor to siftPt (device array of
 * max_points entries) behind the slot's current count.  d_src: device image,
 * pitch in floats (cudaImage.cu:31-32).  stream: hipStream_t passed as void*
 * (NULL = default stream).  streamIdx selects the slot (0..SIFT_MAX_SLOTS-1).
 * Asynchronous: returns after enqueueing; errors via sift_last_error(). */
void extractSift(SiftPoint *siftPt, float *d_src, int width, int pitch, int height,
                 void *stream, int streamIdx, int octaveIdx);
/* cudaSiftH.cu:323-328 -- waits for the slot's work, returns its point count
 * (clamped to max_points), or -1 on error.  The count arrives in pinned host
 * memory with the slot's last kernel: one event wait, no copy. */
int getPointCount(int streamIdx);

/* ---- stage launchers (Sift/src/cudaSiftH.h:17-20), dense planes -----------
 * Plane i of a multi-plane buffer starts at base + i*pitch*height floats, as
 * in the reference.  All asynchronous on `stream`. */
/* cudaSiftH.cu:210-259: src -> dst[(S+3) planes], both passes */
void blurOctave(float *dst, float *src, int width, int pitch, int height, void *stream);
/* cudaSiftH.cu:306-316: src[(S+3)] -> dst[(S+2)] */
void computeDiffOfGauss(float *dst, float *src, int width, int pitch, int height, void *stream);
/* cudaSiftH.cu:260-283: src[(S+3)] blur planes -> dst[3S] (Dxx,Dyy,Dxy per scale) */
void computeHessian(float *dst, float *src, int width, int pitch, int height, void *stream);
/* cudaSiftH.cu:284-305: src[(S+3)] blur planes -> dst[2S] (magnitude, direction per scale) */
void computeGradient(float *dst, float *src, int width, int pitch, int height, void *stream);
/* cudaSiftH.cu:317-322 */
void copyDeviceData(float *dst, float *src, int width, int pitch, int height, void *stream);

/* ---- configuration and errors (not in the reference: it uses macros and
 * exit(1), utils.h:4-19, cudaUtils.h:8-15) ---------------------------------- */
void sift_default_params(sift_params *p);
int sift_set_params(const sift_params *p); /* 0 = ok; rebuilds the tables */
void sift_get_params(sift_params *p);
/* message of the calling thread's most recent failure, "" if none.
 * Threading: calls on different devices (one host thread per GPU, each with its
 * own current device) run side by side; calls on one device take turns, since
 * they share that device's scratch arenas.  sift_set_params must not run while
 * another thread extracts (the reference's tables are process-wide too,
 * cudaSiftD.h:16-25). */
const char *sift_last_error(void);
void sift_clear_error(void);
/* Makes msg the calling thread's last error: how a thread that ran library calls on
 * behalf of another hands its message over (sift_extract_list_multi does so for its
 * worker threads). */
void sift_set_last_error(const char *msg);
/* tap tables as the kernels receive them: blur[(S+3)*19], wnd[S*16], desc[16] */
void sift_get_tables(float *blur, float *wnd, float *desc);
/* frees every arena on the current device */
void sift_shutdown(void);

/* ---- batch entry (not in the reference: it loops over image pairs,
 * sift.cu:68-134).  n images of identical geometry, image i at
 * d_imgs + i*img_stride floats; points of image i at d_pts + i*max_points;
 * h_counts[i] (host) receives the clamped count.  d_meta (optional, device,
 * 2 bytes per point slot) receives {octave, scale index}.  Blocks until done.
 * Returns 0 or a negative error. */
int sift_extract_batch(SiftPoint *d_pts, unsigned char *d_meta, int *h_counts,
                       const float *d_imgs, long img_stride, int n,
                       int width, int pitch, int height, void *stream);
/* The same for 8-bit grey images, the form images have on disk and on the wire:
 * a pixel is converted to float (image::imload's convertTo(CV_32F), utils.cc:13)
 * as the pyramid kernel fetches it, so results equal those of the float entry on
 * the converted image.  img_stride and pitch count bytes. */
int sift_extract_batch_u8(SiftPoint *d_pts, unsigned char *d_meta, int *h_counts,
                          const unsigned char *d_imgs, long img_stride, int n,
                          int width, int pitch, int height, void *stream);
/* The same without the wait: the counts go to d_counts (device, n words), the
 * call returns as soon as the work is queued, and everything is ordered on
 * `stream` (work queued on it before the call is waited for; work queued on it
 * after the call sees the results).  imgs_u8 != 0: 8-bit images as above.  For
 * callers that keep the link and the GPU busy at the same time: upload batch
 * k+1 on a copy stream while batch k runs, read counts and points back later.
 * Consecutive calls take the library's two private streams and arenas in turn:
 * calls queued on different caller streams (with their own d_pts and d_counts)
 * run side by side, calls on one stream follow each other. */
int sift_extract_batch_async(SiftPoint *d_pts, unsigned char *d_meta, unsigned *d_counts,
                             const void *d_imgs, int imgs_u8, long img_stride, int n,
                             int width, int pitch, int height, void *stream);
/* One image without any allocation.  No counterpart in the reference (its extractSift
 * allocates and frees a dozen buffers per octave, cudaSiftH.cu:22-24,58,158-177); for callers
 * that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  A variant of sift_extract_batch_async for n = 1 float image: the same
 * kernels, with every scratch buffer (blur planes, counters, candidate queues and lists)
 * taken from one static 32 MiB device array instead of the arenas the other entries
 * allocate on first use.  d_count (device, one word) receives the clamped point count,
 * d_pts (device, max_points entries) the points; the call clears d_count itself, queues its
 * work on `stream` and returns.  It calls no hipMalloc, hipFree or other allocating routine
 * (the address of the static array is looked up on the first call per device, or by
 * initDeviceVariables; call either once before capturing).
 * Limit: the image's scratch must fit the static array, else -1 with a message that names
 * both sizes; 1024x575 at 3 scales and 4 octaves fits (19 MB), 1920x1080 does not.
 * Every call on a GPU uses that GPU's one array: calls that overlap on one GPU race (queue
 * them on one stream, or order them with events).  Returns 0 or -1. */
int sift_extract_noalloc(SiftPoint *d_pts, unsigned *d_count, const float *d_img, int width,
                         int pitch, int height, void *stream);
/* size of that static array in bytes */
size_t sift_noalloc_scratch_bytes(void);
/* 8-bit descriptors (no counterpart in the reference, whose descriptors are
 * float and L1-normalised, cudaSiftD.cu:186-200): d_out[128*j + k] =
 * min(255, round(512 * d_pts[j].data[k])) for the first n points.  Asynchronous
 * on `stream`.  Returns 0 or a negative error. */
int sift_quantize_descriptors(const SiftPoint *d_pts, int n, unsigned char *d_out, void *stream);
/* Frees the scratch arenas (blur planes, queues, lists) of all stream slots and batch lanes of
 * the calling thread's GPU after waiting for their work: 55 MB per 1080p image of the largest
 * launch set seen, 14 GB after a batch of 256.  The reference frees its planes inside every call
 * (cudaSiftH.cu:58, 172-177); here they are kept from call to call, and this is the way to give
 * them back short of sift_shutdown.  The next call allocates again.  Returns 0 or -1. */
int sift_release_arenas(void);
/* most images worked on per launch set (default 256); arenas grow to match */
void sift_set_batch_chunk(int images_per_launch);
/* Detection is sparse: the pyramid pass queues the pixels that pass the contrast
 * test and only those are examined further.  n = first-octave queue capacity per
 * image and scale (default 65536; later octaves get a quarter each, at least
 * 4096).  A queue that overflows is redone by the dense detector, so results do
 * not depend on n; 0 disables the queues (dense detector only). */
void sift_set_queue_cap(int n);
/* Launch sets of at most this many first-octave pixels (images x width x height; default
 * 2,200,000: one 1080p image) skip the queues and run the dense detector alone, which is
 * the faster route when every kernel of the set is a single round of waves (one image at a
 * time through extractSift).  0: never.  Results do not depend on it.  No counterpart in
 * the reference. */
void sift_set_dense_route_pixels(long pixels);
/* Diagnostics: after the work of slot streamIdx has completed (it is waited
 * for), copies per (octave, scale) group, octave-major, the number of keypoints
 * the group produced (before the cap) into accepted[] and the number of
 * candidates the pyramid pass queued for it into queued[] (a value above the
 * queue's capacity marks a group the dense detector redid).  Returns the number
 * of groups written (at most n), or -1.  No counterpart in the reference. */
int sift_get_group_counts(int streamIdx, unsigned *accepted, unsigned *queued, int n);
/* Diagnostics: the S+3 blur planes the pyramid pass left in the scratch arena for
 * one image and octave (what blurOctave's x and y passes produce, cudaSiftH.cu:210-259;
 * for octave o > 0 the source was resizeKernel's even pixels of plane S of octave o-1,
 * cudaSiftD.cu:544-559), copied to host memory after the slot's work has completed.
 * slot: 0..SIFT_MAX_SLOTS-1 = the stream slots of extractSift, SIFT_MAX_SLOTS + l = launch
 * lane l of the batch entries (a batch of at most one launch set leaves image i of the
 * batch at image i of lane 0).  octave counts from the first octave processed.  Plane k
 * starts at h_dst + k*pitch*height; columns [width, pitch) hold don't-care values.
 * h_dst NULL: only the geometry is returned.  Returns 0 or -1. */
int sift_debug_read_planes(int slot, int image, int octave, float *h_dst, long dst_floats,
                           int *width, int *pitch, int *height);
/* Where image i of the most recent batch of n images sits for sift_debug_read_planes
 * (a batch is cut into launch sets that take turns on the lanes).  Returns 0, or -1 if
 * a later launch set has reused that arena. */
int sift_debug_batch_slot(int n, int i, int *slot, int *image);
/* rows per strip of the pyramid pass; 0 (default) = chosen per launch from the image
 * size and the number of images.  Results do not depend on it. */
void sift_set_strip_rows(int rows);
/* Columns each lane of the pyramid pass owns: 1 (64-column tiles, three waves per SIMD) or 2
 * (128-column tiles, 512-byte plane stores, two waves per SIMD) for octaves at least min_width
 * pixels wide (0 keeps the current width limit); the reference profile only.  Results do not
 * depend on it. */
void sift_set_pyramid_columns(int cols, int min_width);
/* Pyramid launches the chip holds in a single round of waves and that feed no candidate queue
 * (one image at a time on the dense route, blurOctave) deal the S+3 planes over waves, one plane
 * per wave (pyramid_plane_kernel): a fifth of the arithmetic per row step and five times the
 * waves, for calls whose time is the sum of their launches' longest waves.  on = 0 keeps the
 * one-wave-per-tile kernel for them too (default 1).  Results do not depend on it. */
void sift_set_pyramid_split(int on);
/* Diagnostics: the strip height (rows) a pyramid launch of `images` images of width x height would
 * be cut into, whatever sift_set_strip_rows says.  kernel 0: one wave per tile (pyramid_kernel),
 * 1: one plane per wave, 2: two columns per lane.  Needs no GPU.  -1 for bad arguments. */
int sift_debug_strip_rows(int width, int height, int images, int n_scales, int kernel);
/* launch sets in flight at once, each on a private stream and arena: 1 or 2
 * (default 2: the short kernels of the small octaves of one set overlap the
 * long kernels of the other) */
void sift_set_batch_lanes(int n);

/* ---- thin device layer (replaces CudaImage / SiftData helpers,
 * cudaImage.cu:23-68, sift.cu:30-61, and the stream calls in sift.cu:94-130) */
int sift_device_count(void);
int sift_set_device(int dev);
int sift_get_device(void); /* the calling thread's current device, or -1 */
int sift_dev_malloc(void **p, size_t bytes);
int sift_dev_malloc_pitch(float **p, int *pitch_floats, int width, int height);
int sift_dev_free(void *p);
int sift_memcpy2d_h2d_async(float *d_dst, int d_pitch, const float *h_src, int h_pitch,
                            int width, int height, void *stream);
int sift_memcpy2d_d2h_async(float *h_dst, int h_pitch, const float *d_src, int d_pitch,
                            int width, int height, void *stream);
int sift_memcpy_h2d_async(void *d_dst, const void *h_src, size_t bytes, void *stream);
int sift_memcpy_d2h_async(void *h_dst, const void *d_src, size_t bytes, void *stream);
int sift_stream_create(void **stream);
int sift_stream_destroy(void *stream);
int sift_stream_sync(void *stream);
int sift_host_alloc_pinned(void **p, size_t bytes);
int sift_host_free_pinned(void *p);
/* 8-bit grey pixels on the device to the float image extractSift takes (the convertTo(CV_32F)
 * of image::imload, utils.cc:13, done after the upload instead of before it: a quarter of the
 * bytes cross the link).  Pitches in elements.  Asynchronous on `stream`. */
int sift_convert_u8_f32(float *d_dst, int dst_pitch, const unsigned char *d_src, int src_pitch,
                        int width, int height, void *stream);

/* ---- host pipeline (counterpart of Sift/src/sift.cu:64-163, utils.cc:8-17) ---
 * Binary PGM/PPM -> float grey image (malloc'ed; release with sift_free_host).
 * Colour is reduced as the reference's OpenCV grayscale load does:
 * Y = (1868*B + 9617*G + 4899*R + 8192) >> 14. */
int sift_load_pnm_gray(const char *path, float **out, int *width, int *height);
void sift_free_host(void *p);
/* sift.cu:151-160 text format; returns bytes written into buf or -1 */
long sift_format_points(const SiftPoint *pts, int n, char *buf, long cap);
/* sift.cu:144-163: writes <dst_dir>/<basename(src_path)>.sift.txt */
int sift_save_points(const char *dst_dir, const char *src_path, const SiftPoint *pts, int n);
/* The order the pipelines' writers put points in: scale, then row, then column (inside a group
 * the reference's order is its atomic counter's, cudaSiftD.cu:296, and differs from run to run) */
void sift_sort_points(SiftPoint *pts, int n);
/* sift.cu:64-143: images two at a time on two streams and two counter slots */
int sift(const char *dst_dir, const char **src_paths, int n_imgs);
/* The same work for a whole list on several GPUs (sift.cu:68-134 treats its images as
 * independent; nothing in the reference spans GPUs).  Image i goes to worker
 * sift_list_partition(i, n_workers) = i mod n_workers; worker g is one host thread on device
 * g mod sift_device_count() with its own stream and buffers, no data moves between GPUs, and
 * the only gather is counts[] (n_imgs entries, optional), filled on the host in image order.
 * Inside a worker file decoding, the GPU and the text writer overlap (batches of up to 32
 * same-size images through sift_extract_batch_async on 8-bit pixels; an image of another size
 * than its batch is done afterwards, alone).  dst_dir NULL: no files are written.  The files
 * are byte for byte those of sift().  Host threads: SIFT_HOST_THREADS, else OMP_NUM_THREADS,
 * else at most 16, shared out over the workers.  Returns 0 or -1. */
int sift_extract_list_multi(const char *dst_dir, const char **src_paths, int n_imgs, int n_workers,
                            int *counts);
int sift_list_partition(int image, int n_workers);
/* the reference hands count-1 to its writer (sift.cu:125); 1 reproduces that */
void sift_pipeline_drop_last_point(int on);
/* Visual check (image::drawPoint is declared at utils.h:32 and never defined):
 * writes a binary PPM of the grey image with a circle of radius 3*scale and an
 * orientation spoke per point.  point_pos_scale (optional, one per point) maps
 * point coordinates to image pixels: 2^octave for profile ref, NULL (= 1) for
 * profile lowe.  Host memory throughout.  Returns 0 or -1. */
int sift_draw_points(const char *path, const float *gray, int w, int pitch, int h,
                     const SiftPoint *pts, int n, const float *point_pos_scale);

/* ---- measurement hooks ---------------------------------------------------- */
enum {
    SIFT_K_ROW_BLUR = 0,
    SIFT_K_COL_BLUR_DOG = 1,
    SIFT_K_DETECT = 2,
    SIFT_K_KEYPOINT = 3,
    SIFT_K_PYRAMID = 4, /* fused row + column pass of the hot path */
    SIFT_K_VERIFY = 5,  /* sparse detector over queued contrast candidates */
    SIFT_K_ALIAS = 6,   /* unused (that pre-filter is part of the pyramid kernel now); keeps the numbering */
    SIFT_K_COUNT = 7
};
/* When enabled, every launch of the pyramid kernels is bracketed by events on
 * its own stream; sift_timing_read drains them (synchronises). */
void sift_timing_enable(int on);
/* total milliseconds and number of launches per kernel since the last reset */
int sift_timing_read(int kernel, double *ms_total, long *launches);
void sift_timing_reset(void);

/* The rate of a plain device-to-device copy on the current GPU (a buffer of `bytes` bytes
 * copied to another, 16 bytes per lane; best of `repeats` passes; bytes read plus bytes
 * written per second, in GB/s): what the memory system gives a kernel that only moves
 * bytes, to price bandwidth-bound kernels against beside the data-sheet figure.
 * Allocates and frees its two buffers.  Returns 0 or a negative error. */
int sift_measure_copy_ceiling(size_t bytes, int repeats, double *gbytes_per_s);

/* ---- synthetic input (deterministic, no dependencies) ---------------------
 * 128 + value noise + Gaussian blobs + pixel noise, clamped to [0,255] and
 * rounded to integers; dense != 0 selects the high-keypoint-density variant. */
void sift_synth_image(float *dst, int width, int pitch, int height, unsigned long long seed,
                      int dense);

#ifdef __cplusplus
}
#endif
#endif
