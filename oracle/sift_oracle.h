/*
 * CPU restatement of the reference SIFT path (IjlalBaig/SIFT, Sift/src).
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker and the CPU
 * baseline.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py may load it; nothing in the shipped GPU path links or calls it.
 *
 * Pinning: the reference has no tests, golden vectors or CPU path, and cannot
 * be built here (CUDA + OpenCV).  For the path itself parity is UNPINNED by
 * the reference; it is pinned by (i) a second, literal thread-block simulation
 * of findExtremaKernel that must agree with the closed form, (ii) closed-form
 * known answers (constant image, impulse response, border fall-off), and
 * (iii) Sift/result/result.ppm == Q14 gray of Sift/img/img3.ppm for the loader.
 *
 * Semantics profile "ref": every deterministic behaviour of the reference is
 * reproduced (including its detector quirks); behaviour that is not a function
 * of the input (data races, uninitialised reads, atomic ordering) is replaced
 * by the evident intent with the reference's parameters, and every float sum
 * whose order the reference leaves to atomics is given one fixed order here.
 *
 * Semantics profile "lowe" (params.profile = 1) is NOT the reference's
 * behaviour: it is the feature set the reference lacks -- strict 26-neighbour
 * extrema on the same pyramid, a 3-D quadratic refinement of position and
 * scale, contrast and edge rejection on the DoG Hessian with the usual sign,
 * coordinates in pixels of the input image, optionally an up-sampled first
 * octave.  Nothing in the reference pins it; it is defined here and by the
 * comments at oracle_find_extrema_lowe.
 */
#ifndef SIFT_ORACLE_H
#define SIFT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Sift/src/sift.h:7-13 */
typedef struct {
    float xpos;
    float ypos;
    float scale;
    float orientation;
    float data[128];
} OracleSiftPoint;

/* Runtime mirror of the compile-time macros in Sift/src/utils.h:4-19. */
typedef struct {
    int n_octaves;        /* N_OCTAVES 4        utils.h:7  */
    int n_scales;         /* N_SCALES 2         utils.h:8  */
    int max_points;       /* MAX_POINTCOUNT 2500 utils.h:10 */
    float sigma;          /* SIGMA 1.6f         utils.h:14 */
    float r_thresh;       /* R_THRESH 10.0f     utils.h:16 */
    float extrema_thresh; /* EXTREMA_THRESH 10.0f utils.h:18 */
    int profile;          /* 0 = ref, 1 = lowe */
    int upsample;         /* lowe only: first octave is the image up-sampled by 2 */
} OracleParams;

#define ORACLE_BLUR_TAPS 19 /* 2*B_KERNEL_RADIUS+1, utils.h:5, cudaSiftD.h:6 */
#define ORACLE_WND_TAPS 16  /* WND_KERNEL_SIZE, utils.h:6 */

void oracle_default_params(OracleParams *p);

/* utils.cc:41-53 */
void oracle_gaussian1d(float *taps, float sigma, int n);
/* cudaSiftH.cu:128-157: blur[(S+3)*19], wnd[S*16], desc[16] */
void oracle_tables(const OracleParams *p, float *blur, float *wnd, float *desc);
/* cudaSiftH.cu:94-95 */
float oracle_scale_value(const OracleParams *p, int scale_idx, int octave_idx);
/* cos/sin of k*11.25 degrees, k = 0..31, as float */
void oracle_rotation_table(float *cos32, float *sin32);

/* utils.cc:59-60 */
int oracle_idivup(int a, int b);
int oracle_ialignup(int a, int b);

/* deterministic atan2f: double-precision evaluation rounded once to float */
float oracle_atan2f(float y, float x);
/* cudaSiftD.h:35-41 */
int oracle_assign_bin(float theta, int nbins);

/* Stage functions.  All planes are pitched (pitch p in floats), plane i at
 * base + i*p*h, exactly as the reference lays them out. */
/* xblurMultiKernel cudaSiftD.cu:430-486 : src -> tmp[(S+3) planes] */
void oracle_xblur(const OracleParams *pr, const float *src, float *tmp, int w, int p, int h);
/* yblurKernel cudaSiftD.cu:488-534 : tmp -> blur[(S+3) planes] */
void oracle_yblur(const OracleParams *pr, const float *tmp, float *blur, int w, int p, int h);
/* subtractKernel cudaSiftD.cu:5-15 : blur -> dog[(S+2) planes] */
void oracle_dog(const OracleParams *pr, const float *blur, float *dog, int w, int p, int h);
/* hessianKernel cudaSiftD.cu:366-428 on blur plane i+1 -> hess[3S planes] */
void oracle_hessian(const OracleParams *pr, const float *blur, float *hess, int w, int p, int h);
/* gradientKernel cudaSiftD.cu:306-364 on blur plane i+1 -> grad[2S planes] */
void oracle_gradient(const OracleParams *pr, const float *blur, float *grad, int w, int p, int h);
/* resizeKernel cudaSiftD.cu:544-559 : blur plane S -> next base */
void oracle_downsample(const float *src, float *dst, int w, int p, int h);

/* findExtremaKernel cudaSiftD.cu:216-305 for one scale.
 * dog3: 3 consecutive DoG planes, hess3: Dxx,Dyy,Dxy planes.  Writes up to
 * max_out (x,y) pairs in raster order (closed) or block/thread order
 * (literal) and returns the number found (may exceed max_out). */
int oracle_find_extrema_closed(const OracleParams *pr, const float *dog3, const float *hess3,
                               int w, int p, int h, int *xy, int max_out);
int oracle_find_extrema_literal(const OracleParams *pr, const float *dog3, const float *hess3,
                                int w, int p, int h, int *xy, int max_out);
/* the rank value a reference thread computes for pixel (x,y) of its own tile */
int oracle_pixel_rank(const float *dog3, int w, int p, int h, int x, int y, float *center);

/* OrientationKernel cudaSiftD.cu:17-94 (intent), one keypoint.
 * grad2: magnitude plane then direction plane of this scale. */
float oracle_orientation(const float *grad2, int w, int p, int h, const float *wnd16,
                         int x, int y, float *hist32);
/* DescriptorKernel cudaSiftD.cu:96-214 (intent), one keypoint. */
void oracle_descriptor(const float *grad2, int w, int p, int h, const float *desc16,
                       int x, int y, float orientation_deg, float *out128);

/* ---- profile "lowe" -------------------------------------------------------
 * One keypoint of the lowe detector: integer sample it was refined at (octave
 * grid), DoG plane index c (1..S), and the refined offsets. */
typedef struct {
    int x, y, c;
    float dx, dy, ds; /* each within (-0.5, 0.5) */
} OracleLowePoint;
/* 2^f for f in [-1, 2], evaluated in double with a fixed operation sequence */
double oracle_det_exp2(double f);
/* Detector of one scale index i (DoG plane i+1 is the starting plane) over the
 * S+3 blur planes of one octave, raster order of the starting sample.  Returns
 * the number found (may exceed max_out). */
int oracle_find_extrema_lowe(const OracleParams *pr, const float *blur, int w, int p, int h, int i,
                             OracleLowePoint *out, int max_out);
/* dst (2w x 2h, pitch 2w): dst[2y][2x] = src[y][x], odd samples are the mean of
 * their two (or four) even neighbours, the last odd row/column repeats the edge */
void oracle_upsample2x(const float *src, int w, int pitch, int h, float *dst);
/* 8-bit descriptors: min(255, round-half-up(512 * v)) */
void oracle_quantize_descriptors(const OracleSiftPoint *pts, int n, unsigned char *out);

/* extractSift cudaSiftH.cu:9-60 for octaves octave0..n_octaves-1.
 * pts: capacity max_points.  meta (optional, 2 bytes per point): octave, scale
 * index.  Returns the clamped point count.  total_found (optional) receives
 * the unclamped number of detections. */
int oracle_extract(const OracleParams *pr, const float *img, int w, int pitch, int h,
                   int octave0, OracleSiftPoint *pts, unsigned char *meta, int *total_found);

/* n images of identical geometry, image i at imgs + i*img_stride floats,
 * points at pts + i*max_points, worked on by n_threads host threads. */
int oracle_extract_batch(const OracleParams *pr, const float *imgs, long img_stride, int n,
                         int w, int pitch, int h, OracleSiftPoint *pts, int *counts,
                         int n_threads);

/* How many threshold decisions of cudaSiftD.cu:288-293 sit within `tol` of their limit, and how
 * many edge decisions change under roundings a CUDA compiler may have chosen (FMA contraction,
 * double-width pow, reciprocal division); see sift_oracle.c for the seven counters of out[]. */
void oracle_extrema_ambiguity(const OracleParams *pr, const float *dog3, const float *hess3, int w,
                              int p, int h, float tol, long *out);
void oracle_ambiguity_report(const OracleParams *pr, const float *img, int w, int pitch, int h,
                             float tol, long *out);

/* OpenCV grayscale load as pinned by Sift/result/result.ppm (utils.cc:13):
 * Y = (1868*B + 9617*G + 4899*R + 8192) >> 14 */
void oracle_gray_q14(const unsigned char *rgb, int npix, unsigned char *gray);

/* saveSift sift.cu:144-163: one text line per point into buf; returns bytes */
long oracle_format_points(const OracleSiftPoint *pts, int n, char *buf, long cap);

#ifdef __cplusplus
}
#endif
#endif
