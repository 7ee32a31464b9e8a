/*
 * CPU restatement of the reference SIFT path -- TEST INFRASTRUCTURE ONLY
 * (see sift_oracle.h for the pinning statement and the semantics profile).
 *
 * Build: gcc -O2 -ftree-vectorize -ffp-contract=off -mfma -mavx2 (Makefile).
 * Every fused multiply-add the reference performs is written as fmaf();
 * nothing else may be contracted.
 */
#include "sift_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define REF_PI 3.14159265 /* utils.h:4 */
#define TAPS ORACLE_BLUR_TAPS
#define RADIUS 9 /* B_KERNEL_RADIUS utils.h:5 */

void oracle_default_params(OracleParams *p)
{
    p->n_octaves = 4;
    p->n_scales = 2;
    p->max_points = 2500;
    p->sigma = 1.6f;
    p->r_thresh = 10.0f;
    p->extrema_thresh = 10.0f;
    p->profile = 0;
    p->upsample = 0;
}

int oracle_idivup(int num, int den) { return (num % den != 0) ? (num / den + 1) : (num / den); }
int oracle_ialignup(int A, int a) { return (A % a != 0) ? (A + a - A % a) : A; }

/* utils.cc:41-53.  C++ promotion rules of the original expression: 2.0f*PI is
 * double, pow(float,int) and pow(double,float) evaluate in double, the product
 * is rounded to float on store, the running sum and the division are float. */
void oracle_gaussian1d(float *kp, float sigma, int n)
{
    float sum = 0;
    for (int i = 0; i < n; ++i) {
        float d = (float)i - ((float)(n - 1) / 2);
        double norm = 1 / (pow(2.0f * REF_PI, (double)0.5f) * (double)sigma);
        double e = exp(-pow((double)d, 2) / (2 * pow((double)sigma, 2)));
        kp[i] = (float)(norm * e);
        sum += kp[i];
    }
    for (int i = 0; i < n; ++i)
        kp[i] /= sum;
}

/* cudaSiftH.cu:128-157 */
void oracle_tables(const OracleParams *p, float *blur, float *wnd, float *desc)
{
    float k = (float)pow(2, 0.5);
    int S = p->n_scales;
    for (int i = 0; i < S + 3; ++i) {
        float sigmaNew = (float)(pow((double)k, (double)(i - 1)) * (double)p->sigma);
        oracle_gaussian1d(blur + i * TAPS, sigmaNew, TAPS);
    }
    for (int i = 0; i < S; ++i) {
        float sigmaNew = (float)(pow((double)k, (double)i) * (double)p->sigma * 1.5);
        oracle_gaussian1d(wnd + i * ORACLE_WND_TAPS, sigmaNew, ORACLE_WND_TAPS);
    }
    float sigmaNew = (float)((double)p->sigma * 1.5);
    oracle_gaussian1d(desc, sigmaNew, 16);
}

/* cudaSiftH.cu:94-95 */
float oracle_scale_value(const OracleParams *p, int i, int octaveIdx)
{
    float k = (float)pow(2, 0.5);
    return (float)(pow((double)k, (double)(i + p->n_scales * octaveIdx)) * (double)p->sigma);
}

void oracle_rotation_table(float *c32, float *s32)
{
    for (int k = 0; k < 32; ++k) {
        double rad = (double)(11.25f * (float)k) * (3.14159265358979323846 / 180.0);
        c32[k] = (float)cos(rad);
        s32[k] = (float)sin(rad);
    }
}

/* ---- deterministic atan2f ------------------------------------------------
 * The reference calls the CUDA device atan2f (cudaSiftD.cu:356), which is not
 * available.  This evaluation uses only IEEE double +,-,*,/ and fma in a fixed
 * order, so the same source gives the same bits on any IEEE machine; the
 * double result (rel. error < 1e-15) is rounded once to float.
 * With a = min/max of |x|,|y| and c the nearest of 0, 1/4, 1/2, 3/4, 1:
 * atan(a) = atan(c) + atan(t), t = (a - c)/(1 + a c) = (min - c max)/(max + c min),
 * |t| < 0.1245, so one division and a series up to t^17 suffice. */
static double atan_small(double t) /* |t| < 1/8 */
{
    static const double C[8] = {-1.0 / 3.0, 1.0 / 5.0,  -1.0 / 7.0,  1.0 / 9.0,
                                -1.0 / 11.0, 1.0 / 13.0, -1.0 / 15.0, 1.0 / 17.0};
    double s = t * t;
    double q = C[7];
    for (int k = 6; k >= 0; --k)
        q = fma(q, s, C[k]);
    return fma(t * s, q, t);
}

float oracle_atan2f(float yf, float xf)
{
    static const double ATAN_C[5] = {0.0, 0.24497866312686414, 0.46364760900080609,
                                     0.64350110879328437, 0.78539816339744831};
    double x = (double)xf, y = (double)yf;
    double ax = fabs(x), ay = fabs(y);
    double mx = ax > ay ? ax : ay;
    double mn = ax > ay ? ay : ax;
    double r = 0.0;
    if (mx != 0.0) {
        const double m8 = 8.0 * mn;
        const int j = (m8 >= mx) + (m8 >= 3.0 * mx) + (m8 >= 5.0 * mx) + (m8 >= 7.0 * mx);
        const double c = 0.25 * (double)j;
        const double t = fma(-c, mx, mn) / fma(c, mn, mx);
        r = ATAN_C[j] + atan_small(t);
    }
    if (ay > ax)
        r = 1.57079632679489662 - r;
    if (x < 0.0)
        r = 3.14159265358979323846 - r;
    if (y < 0.0)
        r = -r;
    return (float)r;
}

/* cudaSiftD.h:35-41 */
int oracle_assign_bin(float theta, int nBins)
{
    float binRange = 360.0f / (float)nBins;
    theta = (theta < 0) ? (theta + 360) : theta;
    theta = fmodf(theta, 360.0f);
    return (int)floorf(theta / binRange);
}

/* ---- pyramid --------------------------------------------------------------*/

/* xblurMultiKernel cudaSiftD.cu:430-486 with the zero-filling tile loader
 * cudaSiftD.h:43-78: out_i[x,y] = chain_k fmaf(c[i*19+k], src[x-9+k,y], sum),
 * sum0 = 0, k ascending; src is 0 outside [0,w). */
void oracle_xblur(const OracleParams *pr, const float *src, float *tmp, int w, int p, int h)
{
    float taps[8 * TAPS], wnd[8 * 16], desc[16];
    oracle_tables(pr, taps, wnd, desc);
    int nP = pr->n_scales + 3;
    long gDim = (long)p * h;
    float *row = (float *)malloc(sizeof(float) * (w + 2 * RADIUS));
    float *acc = (float *)malloc(sizeof(float) * w);
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < RADIUS; ++x) {
            row[x] = 0.0f;
            row[w + RADIUS + x] = 0.0f;
        }
        memcpy(row + RADIUS, src + (long)y * p, sizeof(float) * w);
        for (int i = 0; i < nP; ++i) {
            const float *c = taps + i * TAPS;
            for (int x = 0; x < w; ++x)
                acc[x] = 0.0f;
            for (int k = 0; k < TAPS; ++k) {
                float ck = c[k];
                const float *s = row + k;
                for (int x = 0; x < w; ++x)
                    acc[x] = fmaf(ck, s[x], acc[x]);
            }
            memcpy(tmp + i * gDim + (long)y * p, acc, sizeof(float) * w);
        }
    }
    free(row);
    free(acc);
}

/* yblurKernel cudaSiftD.cu:488-534: column chain, zero outside [0,h). */
void oracle_yblur(const OracleParams *pr, const float *tmp, float *blur, int w, int p, int h)
{
    float taps[8 * TAPS], wnd[8 * 16], desc[16];
    oracle_tables(pr, taps, wnd, desc);
    int nP = pr->n_scales + 3;
    long gDim = (long)p * h;
    float *acc = (float *)malloc(sizeof(float) * w);
    for (int i = 0; i < nP; ++i) {
        const float *c = taps + i * TAPS;
        const float *src = tmp + i * gDim;
        for (int y = 0; y < h; ++y) {
            for (int x = 0; x < w; ++x)
                acc[x] = 0.0f;
            for (int j = 0; j < TAPS; ++j) {
                int yy = y - RADIUS + j;
                float cj = c[j];
                if (yy >= 0 && yy < h) {
                    const float *s = src + (long)yy * p;
                    for (int x = 0; x < w; ++x)
                        acc[x] = fmaf(cj, s[x], acc[x]);
                } else {
                    for (int x = 0; x < w; ++x)
                        acc[x] = fmaf(cj, 0.0f, acc[x]);
                }
            }
            memcpy(blur + i * gDim + (long)y * p, acc, sizeof(float) * w);
        }
    }
    free(acc);
}

/* subtractKernel cudaSiftD.cu:5-15, launched for i = 0..S+1 (cudaSiftH.cu:314-315) */
void oracle_dog(const OracleParams *pr, const float *blur, float *dog, int w, int p, int h)
{
    long gDim = (long)p * h;
    for (int i = 0; i < pr->n_scales + 2; ++i)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                long g = x + (long)p * y;
                dog[i * gDim + g] = blur[i * gDim + g] - blur[(i + 1) * gDim + g];
            }
}

static inline float zpad(const float *pl, int w, int p, int h, int x, int y)
{
    return (x >= 0 && x < w && y >= 0 && y < h) ? pl[x + (long)p * y] : 0.0f;
}

/* hessianKernel cudaSiftD.cu:366-428; source plane i+1 (cudaSiftH.cu:278) */
void oracle_hessian(const OracleParams *pr, const float *blur, float *hess, int w, int p, int h)
{
    long gDim = (long)p * h;
    for (int i = 0; i < pr->n_scales; ++i) {
        const float *src = blur + (i + 1) * gDim;
        float *dst = hess + (long)i * 3 * gDim;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float c = src[x + (long)p * y];
                float Dxx = -2 * c;
                Dxx += zpad(src, w, p, h, x - 1, y);
                Dxx += zpad(src, w, p, h, x + 1, y);
                float Dyy = -2 * c;
                Dyy += zpad(src, w, p, h, x, y - 1);
                Dyy += zpad(src, w, p, h, x, y + 1);
                float Dxy = c;
                Dxy += zpad(src, w, p, h, x - 1, y - 1);
                Dxy -= zpad(src, w, p, h, x - 1, y);
                Dxy -= zpad(src, w, p, h, x, y - 1);
                long g = x + (long)p * y;
                dst[g] = Dxx;
                dst[g + gDim] = Dyy;
                dst[g + 2 * gDim] = Dxy;
            }
    }
}

/* one gradient sample as gradientKernel computes it (cudaSiftD.cu:349-356) */
static inline void grad_at(const float *src, int w, int p, int h, int x, int y, float *mag,
                           float *dir)
{
    float lUp = zpad(src, w, p, h, x, y - 1);
    float lDown = zpad(src, w, p, h, x, y + 1);
    float lLeft = zpad(src, w, p, h, x - 1, y);
    float lRight = zpad(src, w, p, h, x + 1, y);
    float dx = lRight - lLeft;
    float dy = lDown - lUp;
    float dx2 = dx * dx; /* powf(v, 2) is v*v correctly rounded */
    float dy2 = dy * dy;
    *mag = sqrtf(dx2 + dy2);
    *dir = (float)((180 / REF_PI) * (double)oracle_atan2f(dy, dx));
}

/* gradientKernel cudaSiftD.cu:306-364; source plane i+1 (cudaSiftH.cu:300) */
void oracle_gradient(const OracleParams *pr, const float *blur, float *grad, int w, int p, int h)
{
    long gDim = (long)p * h;
    for (int i = 0; i < pr->n_scales; ++i) {
        const float *src = blur + (i + 1) * gDim;
        float *dst = grad + (long)i * 2 * gDim;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                long g = x + (long)p * y;
                grad_at(src, w, p, h, x, y, &dst[g], &dst[g + gDim]);
            }
    }
}

/* resizeKernel cudaSiftD.cu:544-559: even pixels, dst pitch alignUp(ceil(w/2),128) */
void oracle_downsample(const float *src, float *dst, int w, int p, int h)
{
    int p_ = oracle_ialignup(oracle_idivup(w, 2), 128);
    for (int y = 0; y < h; y += 2)
        for (int x = 0; x < w; x += 2)
            dst[x / 2 + (long)(y / 2) * p_] = src[x + (long)p * y];
}

/* ---- detection ------------------------------------------------------------
 * Launch geometry of findExtremaKernel (cudaSiftH.cu:64-89): 32x32 threads,
 * nTilesX = 3, apron 1 on every side, bankOffset 3. */
#define XT_BDIM 32
#define XT_NTILES 3
#define XT_DATAX (XT_NTILES * XT_BDIM)          /* 96 */
#define XT_SDIMX (1 + XT_DATAX + 1 + 3)         /* 101 */
#define XT_SDIMY (1 + XT_BDIM + 1)              /* 34 */
#define XT_WRITTEN (XT_SDIMX - 3)               /* loader writes sx < 98 */
#define XT_PLANE_OFF ((XT_DATAX + 3) * XT_BDIM) /* 3168, cudaSiftD.cu:240-241 */
#define XT_SMEM ((3 * 3 * 32 + 2 + 3) * 34)     /* 9962 floats, cudaSiftH.cu:73-74 */

/* Value a thread of block (bx,by) reads from tile `s` at (sx,sy) after the
 * three sequential tile loads (cudaSiftD.cu:250-261).  The tiles are 101x34
 * but spaced only 3168 floats apart, so the head of tile s+1 overwrites the
 * tail of tile s, except in the columns the loader never writes. */
static float tile_read(const float *dog3, int w, int p, int h, int bx, int by, int s, int sx,
                       int sy)
{
    long gDim = (long)p * h;
    if (s < 2) {
        int lin = sx + XT_SDIMX * sy;
        if (lin >= XT_PLANE_OFF) {
            int j = lin - XT_PLANE_OFF;
            int sx2 = j % XT_SDIMX, sy2 = j / XT_SDIMX;
            if (sx2 < XT_WRITTEN) {
                s = s + 1;
                sx = sx2;
                sy = sy2;
            }
        }
    }
    return zpad(dog3 + s * gDim, w, p, h, XT_DATAX * bx - 1 + sx, XT_BDIM * by - 1 + sy);
}

/* contribution of one (thread, tile) visit to pxRank (cudaSiftD.cu:271-284) */
int oracle_pixel_rank(const float *dog3, int w, int p, int h, int x, int y, float *center)
{
    int bx = x / XT_DATAX, by = y / XT_BDIM;
    int sx = 1 + x % XT_DATAX, sy = 1 + y % XT_BDIM;
    float pxVal = tile_read(dog3, w, p, h, bx, by, 1, sx, sy);
    int r = 0;
    for (int l = -1; l < 2; ++l)
        for (int m = -1; m < 2; ++m)
            for (int s = 0; s < 3; ++s)
                r += (pxVal >= tile_read(dog3, w, p, h, bx, by, s, sx + l, sy + m)) ? 1 : -1;
    if (center)
        *center = pxVal;
    return r;
}

/* acceptance test of cudaSiftD.cu:286-294 without the counter clause */
static int accept_point(const OracleParams *pr, const float *hess3, int w, int p, int h, int x,
                        int y, int pxRank, float pxVal)
{
    long gDim = (long)p * h;
    if (!(x > 8 && x < w - 8 && y > 8 && y < h - 8))
        return 0;
    if (pxRank != 9)
        return 0;
    long g = x + (long)p * y;
    float Dxx = hess3[g], Dyy = hess3[g + gDim], Dxy = hess3[g + 2 * gDim];
    float trace = Dxx + Dyy;
    float prod = Dxx * Dyy;
    float sq = Dxy * Dxy;
    float det = prod + sq; /* PLUS, cudaSiftD.cu:288 */
    float lhs = trace * trace;
    lhs = lhs / det;
    float rhs = (pr->r_thresh * pr->r_thresh) / pr->r_thresh;
    if (!(lhs < rhs))
        return 0;
    if (!(fabsf(pxVal) > pr->extrema_thresh))
        return 0;
    return 1;
}

int oracle_find_extrema_closed(const OracleParams *pr, const float *dog3, const float *hess3,
                               int w, int p, int h, int *xy, int max_out)
{
    int n = 0;
    for (int y = 9; y < h - 8; ++y)
        for (int x = 9; x < w - 8; ++x) {
            int tile = (x % XT_DATAX) / XT_BDIM;
            if (tile == 1)
                continue; /* r0 + r1 is even, never 9 */
            float pv;
            int r = oracle_pixel_rank(dog3, w, p, h, x, y, &pv);
            if (tile == 2) { /* pxRank is never reset between tiles, cudaSiftD.cu:243,263 */
                r += oracle_pixel_rank(dog3, w, p, h, x - 64, y, NULL);
                r += oracle_pixel_rank(dog3, w, p, h, x - 32, y, NULL);
            }
            if (accept_point(pr, hess3, w, p, h, x, y, r, pv)) {
                if (n < max_out) {
                    xy[2 * n] = x;
                    xy[2 * n + 1] = y;
                }
                ++n;
            }
        }
    return n;
}

/* How firm are the two float threshold decisions of cudaSiftD.cu:288-293?  The reference was
 * built by a compiler this restatement cannot see: whether nvcc contracted Dxx*Dyy + Dxy^2
 * into an FMA, whether pow(float, 2) ran in double, and whether the division was the IEEE
 * one are all unknowable here.  For every pixel that passes the rank and border tests this
 * counts, into out[0..6]:
 *   0 accepted points                         1 accepted, |trace^2/det - limit| < tol
 *   2 rejected by the edge test alone, within tol of the limit
 *   3 accepted, ||D| - contrast| < tol        4 rejected by the contrast test alone, within tol
 *   5 pixels whose edge decision differs between this restatement's arithmetic (separate
 *     float multiply, float add, IEEE divide) and any of: det by fmaf(Dxx, Dyy, Dxy*Dxy),
 *     det by fmaf(Dxy, Dxy, Dxx*Dyy), det summed in double and rounded once, quotient by a
 *     rounded reciprocal
 *   6 of those, the ones that also pass the contrast test (their presence in the output
 *     depends on the compiler) */
void oracle_extrema_ambiguity(const OracleParams *pr, const float *dog3, const float *hess3, int w,
                              int p, int h, float tol, long *out)
{
    long gDim = (long)p * h;
    const float rhs = (pr->r_thresh * pr->r_thresh) / pr->r_thresh;
    for (int y = 9; y < h - 8; ++y)
        for (int x = 9; x < w - 8; ++x) {
            int tile = (x % XT_DATAX) / XT_BDIM;
            if (tile == 1)
                continue;
            float pv;
            int r = oracle_pixel_rank(dog3, w, p, h, x, y, &pv);
            if (tile == 2) {
                r += oracle_pixel_rank(dog3, w, p, h, x - 64, y, NULL);
                r += oracle_pixel_rank(dog3, w, p, h, x - 32, y, NULL);
            }
            if (r != 9)
                continue;
            long g = x + (long)p * y;
            float Dxx = hess3[g], Dyy = hess3[g + gDim], Dxy = hess3[g + 2 * gDim];
            float trace = Dxx + Dyy;
            float tt = trace * trace;
            float prod = Dxx * Dyy, sq = Dxy * Dxy;
            float det[4];
            det[0] = prod + sq;
            det[1] = fmaf(Dxx, Dyy, sq);
            det[2] = fmaf(Dxy, Dxy, prod);
            det[3] = (float)((double)prod + (double)Dxy * (double)Dxy);
            int edge_ok = (tt / det[0]) < rhs;
            int differs = 0;
            for (int k = 1; k < 4; ++k)
                differs |= ((tt / det[k]) < rhs) != edge_ok;
            float rcp = 1.0f / det[0];
            differs |= ((tt * rcp) < rhs) != edge_ok;
            int con_ok = fabsf(pv) > pr->extrema_thresh;
            int near_edge = fabsf(tt / det[0] - rhs) < tol;
            int near_con = fabsf(fabsf(pv) - pr->extrema_thresh) < tol;
            if (edge_ok && con_ok) {
                out[0]++;
                out[1] += near_edge;
                out[3] += near_con;
            }
            if (!edge_ok && con_ok)
                out[2] += near_edge;
            if (edge_ok && !con_ok)
                out[4] += near_con;
            out[5] += differs;
            out[6] += differs && con_ok;
        }
}

/* The same over every octave and scale of one image (profile ref), out[0..6] summed. */
void oracle_ambiguity_report(const OracleParams *pr, const float *img, int w, int pitch, int h,
                             float tol, long *out)
{
    int S = pr->n_scales;
    const float *src = img;
    float *owned = NULL;
    int p = pitch;
    for (int k = 0; k < 7; ++k)
        out[k] = 0;
    for (int o = 0; o < pr->n_octaves; ++o) {
        long gDim = (long)p * h;
        float *tmp = (float *)calloc((size_t)gDim * (S + 3), sizeof(float));
        float *blur = (float *)calloc((size_t)gDim * (S + 3), sizeof(float));
        float *dog = (float *)calloc((size_t)gDim * (S + 2), sizeof(float));
        float *hess = (float *)calloc((size_t)gDim * 3 * S, sizeof(float));
        oracle_xblur(pr, src, tmp, w, p, h);
        oracle_yblur(pr, tmp, blur, w, p, h);
        oracle_dog(pr, blur, dog, w, p, h);
        oracle_hessian(pr, blur, hess, w, p, h);
        for (int i = 0; i < S; ++i)
            oracle_extrema_ambiguity(pr, dog + i * gDim, hess + (long)i * 3 * gDim, w, p, h, tol, out);
        float *next = NULL;
        int w_ = oracle_idivup(w, 2), h_ = oracle_idivup(h, 2), p_ = oracle_ialignup(w_, 128);
        if (o + 1 < pr->n_octaves) {
            next = (float *)calloc((size_t)p_ * h_, sizeof(float));
            oracle_downsample(blur + (long)S * gDim, next, w, p, h);
        }
        free(tmp);
        free(blur);
        free(dog);
        free(hess);
        free(owned);
        owned = next;
        src = next;
        w = w_;
        h = h_;
        p = p_;
    }
    free(owned);
}

/* Literal simulation: one shared array per block, the three cooperative tile
 * loads of cudaSiftD.h:43-78 run to completion one after another (each ends in
 * a barrier), then every thread runs the loop of cudaSiftD.cu:263-304. */
static void literal_load(float *s, const float *g, int tx, int ty, int gx, int gy, int w, int p,
                         int h)
{
    const int bDimX = 32, bDimY = 32, nTilesX = 3, nTilesY = 1;
    const int apronLeft = 1, apronRight = 1, apronUp = 1, apronDown = 1, bankOffset = 3;
    int sDimX = apronLeft + bDimX * nTilesX + apronRight + bankOffset;
    int sDimY = apronUp + bDimY * nTilesY + apronDown;
    int apronBlocksX = oracle_idivup(apronLeft, bDimX) + oracle_idivup(apronRight, bDimX);
    int apronBlocksY = oracle_idivup(apronUp, bDimY) + oracle_idivup(apronDown, bDimY);
    int jLoop = nTilesY + apronBlocksY;
    int iLoop = nTilesX + apronBlocksX;
    for (int j = 0; j < jLoop; ++j) {
        int sy = ty + j * bDimY;
        for (int i = 0; i < iLoop; ++i) {
            int sx = tx + i * bDimX;
            if (sx >= 0 && sx < sDimX - bankOffset && sy >= 0 && sy < sDimY) {
                int gx_ = gx - apronLeft + i * bDimX;
                int gy_ = gy - apronUp + j * bDimY;
                if (gx_ >= 0 && gx_ < w && gy_ >= 0 && gy_ < h)
                    s[sx + sy * sDimX] = g[gx_ + (long)gy_ * p];
                else
                    s[sx + sy * sDimX] = 0;
            }
        }
    }
}

int oracle_find_extrema_literal(const OracleParams *pr, const float *dog3, const float *hess3,
                                int w, int p, int h, int *xy, int max_out)
{
    long gDim = (long)p * h;
    int n = 0;
    float *shared = (float *)malloc(sizeof(float) * XT_SMEM);
    int gridX = oracle_idivup(w, XT_DATAX), gridY = oracle_idivup(h, XT_BDIM);
    const int sDimX = XT_SDIMX;
    for (int bIdxY = 0; bIdxY < gridY; ++bIdxY)
        for (int bIdxX = 0; bIdxX < gridX; ++bIdxX) {
            for (int i = 0; i < XT_SMEM; ++i)
                shared[i] = NAN; /* uninitialised shared memory */
            float *sharedScale0 = shared;
            float *sharedScale1 = shared + (XT_DATAX + 3) * 32;
            float *sharedScale2 = shared + 2 * (XT_DATAX + 3) * 32;
            float *tiles[3] = {sharedScale0, sharedScale1, sharedScale2};
            for (int s = 0; s < 3; ++s)
                for (int ty = 0; ty < 32; ++ty)
                    for (int tx = 0; tx < 32; ++tx)
                        literal_load(tiles[s], dog3 + s * gDim, tx, ty, tx + 32 * bIdxX * 3,
                                     ty + 32 * bIdxY, w, p, h);
            for (int ty = 0; ty < 32; ++ty)
                for (int tx = 0; tx < 32; ++tx) {
                    int gx = tx + 32 * bIdxX * 3;
                    int gy = ty + 32 * bIdxY;
                    float pxRank = 0.0f;
                    for (int i = 0; i < 3; ++i) {
                        int gx_ = gx + i * 32;
                        if (gx_ < w || gy < h) {
                            int sx = 1 + tx + i * 32;
                            int sy = 1 + ty;
                            float pxVal = sharedScale1[sx + sy * sDimX];
                            for (int l = -1; l < 2; ++l)
                                for (int m = -1; m < 2; ++m) {
                                    pxRank += (pxVal >= sharedScale0[sx + l + (sy + m) * sDimX]) ? 1.0f : -1.0f;
                                    pxRank += (pxVal >= sharedScale1[sx + l + (sy + m) * sDimX]) ? 1.0f : -1.0f;
                                    pxRank += (pxVal >= sharedScale2[sx + l + (sy + m) * sDimX]) ? 1.0f : -1.0f;
                                }
                            if (gx_ > 8 && gx_ < w - 8 && gy > 8 && gy < h - 8 &&
                                accept_point(pr, hess3, w, p, h, gx_, gy, (int)pxRank, pxVal) &&
                                pxRank == 9.0f) {
                                if (n < max_out) {
                                    xy[2 * n] = gx_;
                                    xy[2 * n + 1] = gy;
                                }
                                ++n;
                            }
                        }
                    }
                }
        }
    free(shared);
    return n;
}

/* ---- orientation (intent of OrientationKernel, cudaSiftD.cu:17-94) --------
 * 16x16 patch [x-7,x+8] x [y-7,y+8]; weight mag*g[tx]*g[ty]; 32 bins; argmax,
 * first maximum wins; orientation = 11.25*bin degrees.
 * Summation order (the reference leaves it to shared-memory atomics): the 256
 * samples (index s = ty*16+tx) form 64 groups of four consecutive samples;
 * within a group, samples of one bin are added in ascending order; a bin is
 * then (partials of groups 0..31 added in ascending order) + (the same for
 * groups 32..63). */
float oracle_orientation(const float *grad2, int w, int p, int h, const float *g16, int x, int y,
                         float *hist32)
{
    long gDim = (long)p * h;
    static __thread float part[64][32];
    memset(part, 0, sizeof(part));
    for (int ty = 0; ty < 16; ++ty)
        for (int tx = 0; tx < 16; ++tx) {
            long g = (x - 7 + tx) + (long)p * (y - 7 + ty);
            float m = grad2[g];
            float d = grad2[g + gDim];
            float wgt = m * g16[tx];
            wgt = wgt * g16[ty];
            int b = oracle_assign_bin(d, 32);
            int s = ty * 16 + tx;
            part[s >> 2][b] += wgt;
        }
    float hist[32];
    for (int b = 0; b < 32; ++b) {
        float half[2];
        for (int hf = 0; hf < 2; ++hf) {
            float a = 0.0f;
            for (int L = 32 * hf; L < 32 * hf + 32; ++L)
                a = a + part[L][b];
            half[hf] = a;
        }
        hist[b] = half[0] + half[1];
    }
    float best = hist[0];
    for (int b = 1; b < 32; ++b)
        best = fmaxf(best, hist[b]);
    int idx = 0;
    for (int b = 0; b < 32; ++b)
        if (hist[b] == best) {
            idx = b;
            break;
        }
    if (hist32)
        memcpy(hist32, hist, sizeof(hist));
    (void)w;
    return 360.0f / 32.0f * (float)idx;
}

/* ---- descriptor (intent of DescriptorKernel, cudaSiftD.cu:96-214) ---------
 * sample (tx,ty): offset (dx,dy) = (tx-7.5, ty-7.5) rotated by
 * u = dx*cos + dy*sin, v = dy*cos - dx*sin (cudaSiftD.cu:141-142), bilinear
 * fetch of magnitude and direction with clamp-to-edge, window c_DescWnd, hard
 * binning floor(dir/45), cell (tx/4, ty/4), index bin + 8*cx + 32*cy, then
 * L1 normalise, clamp 0.2, L1 normalise (cudaSiftD.cu:186-200). */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float lerpf(float a, float b, float t) { return fmaf(t, b - a, a); }

static float warp32_sum(const float *h) /* shflsum cudaSiftD.h:96-106, lane 0 */
{
    float a[32];
    memcpy(a, h, sizeof(a));
    for (int off = 16; off >= 1; off >>= 1)
        for (int j = 0; j < off; ++j)
            a[j] = a[j] + a[j + off];
    return a[0];
}

static float hist_sum128(const float *h)
{
    float s = warp32_sum(h);
    s = s + warp32_sum(h + 32);
    s = s + warp32_sum(h + 64);
    s = s + warp32_sum(h + 96);
    return s;
}

void oracle_descriptor(const float *grad2, int w, int p, int h, const float *dw, int x, int y,
                       float orientation_deg, float *out)
{
    long gDim = (long)p * h;
    float c32[32], s32[32];
    oracle_rotation_table(c32, s32);
    int k = (int)(orientation_deg / 11.25f);
    float cs = c32[k & 31], sn = s32[k & 31];
    float wgt[256];
    int bin[256];
    for (int ty = 0; ty < 16; ++ty)
        for (int tx = 0; tx < 16; ++tx) {
            float dx = (float)tx - 7.5f, dy = (float)ty - 7.5f;
            float t1 = dx * cs, t2 = dy * sn;
            float u = t1 + t2;
            float t3 = dy * cs, t4 = dx * sn;
            float v = t3 - t4;
            float px = (float)x + u, py = (float)y + v;
            float fx = floorf(px), fy = floorf(py);
            float ax = px - fx, ay = py - fy;
            int x0 = clampi((int)fx, 0, w - 1), x1 = clampi((int)fx + 1, 0, w - 1);
            int y0 = clampi((int)fy, 0, h - 1), y1 = clampi((int)fy + 1, 0, h - 1);
            float val[2];
            for (int c = 0; c < 2; ++c) {
                const float *pl = grad2 + c * gDim;
                float v00 = pl[x0 + (long)p * y0], v10 = pl[x1 + (long)p * y0];
                float v01 = pl[x0 + (long)p * y1], v11 = pl[x1 + (long)p * y1];
                float top = lerpf(v00, v10, ax);
                float bot = lerpf(v01, v11, ax);
                val[c] = lerpf(top, bot, ay);
            }
            float m = val[0] * dw[tx];
            m = m * dw[ty];
            wgt[ty * 16 + tx] = m;
            bin[ty * 16 + tx] = oracle_assign_bin(val[1], 8);
        }
    float hist[128];
    for (int cy = 0; cy < 4; ++cy)
        for (int cx = 0; cx < 4; ++cx)
            for (int b = 0; b < 8; ++b) {
                /* rows of the cell are summed separately (ascending tx), then
                 * combined as ((P0 + P1) + P2) + P3 */
                float s = 0.0f;
                for (int ty = 4 * cy; ty < 4 * cy + 4; ++ty) {
                    float row = 0.0f;
                    for (int tx = 4 * cx; tx < 4 * cx + 4; ++tx)
                        if (bin[ty * 16 + tx] == b)
                            row += wgt[ty * 16 + tx];
                    s = (ty == 4 * cy) ? row : s + row;
                }
                hist[b + 8 * cx + 32 * cy] = s;
            }
    float S1 = hist_sum128(hist);
    for (int i = 0; i < 128; ++i) {
        hist[i] /= S1;
        if ((double)hist[i] > 0.2)
            hist[i] = (float)0.2;
    }
    float S2 = hist_sum128(hist);
    for (int i = 0; i < 128; ++i)
        out[i] = hist[i] / S2;
}

/* ---- profile "lowe" ---------------------------------------------------------
 * Not in the reference.  The pyramid is the reference's (same planes, same
 * DoG); detection follows D. Lowe's method in the form OpenCV made common:
 *   start   8 < x < w-8, 8 < y < h-8, |D| > 0.5*T/S, D strictly above or
 *           strictly below all 26 neighbours in planes c-1, c, c+1
 *   refine  up to 5 times: gradient g and Hessian H of D by central
 *           differences, X = -H^-1 g by the adjugate; all |X_k| < 0.5 ends the
 *           loop; otherwise the sample moves by floor(X_k + 0.5) and the point
 *           is dropped if it leaves planes 1..S or the border
 *   reject  |D + 0.5 g.X| * S < T;  det(Hxy) <= 0;  tr^2 * r >= (r+1)^2 * det
 * T = extrema_thresh, r = r_thresh, S = n_scales.  Single precision, one
 * operation at a time, in the order written. */
double oracle_det_exp2(double f)
{
    /* exp(f ln 2) by its series in Horner form: |f ln 2| <= 1.39, 22 terms leave < 1e-19 */
    const double t = f * 0.69314718055994530942;
    double q = 1.0;
    for (int k = 22; k >= 1; --k)
        q = fma(q, t / (double)k, 1.0);
    return q;
}

static inline float lowe_dog(const float *blur, long gDim, int p, int s, int x, int y)
{
    const float *c = blur + (long)s * gDim + x + (long)p * y;
    return c[0] - c[gDim]; /* cudaSiftH.cu:190: DoG_s = blur_s - blur_{s+1} */
}

int oracle_find_extrema_lowe(const OracleParams *pr, const float *blur, int w, int p, int h, int i,
                             OracleLowePoint *out, int max_out)
{
    const int S = pr->n_scales;
    const long gDim = (long)p * h;
    const float T = pr->extrema_thresh, r = pr->r_thresh;
    const float pre = 0.5f * T / (float)S;
    int n = 0;
    for (int y0 = 9; y0 < h - 8; ++y0)
        for (int x0 = 9; x0 < w - 8; ++x0) {
            int x = x0, y = y0, c = i + 1;
            float v = lowe_dog(blur, gDim, p, c, x, y);
            if (!(fabsf(v) > pre))
                continue;
            int above = 1, below = 1;
            for (int s = -1; s < 2; ++s)
                for (int m = -1; m < 2; ++m)
                    for (int l = -1; l < 2; ++l) {
                        if (!s && !m && !l)
                            continue;
                        float q = lowe_dog(blur, gDim, p, c + s, x + l, y + m);
                        above &= v > q;
                        below &= v < q;
                    }
            if (!above && !below)
                continue;
            float Xx = 0, Xy = 0, Xs = 0, gx = 0, gy = 0, gs = 0, dxx = 0, dyy = 0, dxy = 0;
            int ok = 0;
            for (int it = 0; it < 5; ++it) {
                float d[3][3][3]; /* [plane][row][column] */
                for (int s = 0; s < 3; ++s)
                    for (int m = 0; m < 3; ++m)
                        for (int l = 0; l < 3; ++l)
                            d[s][m][l] = lowe_dog(blur, gDim, p, c + s - 1, x + l - 1, y + m - 1);
                v = d[1][1][1];
                gx = (d[1][1][2] - d[1][1][0]) * 0.5f;
                gy = (d[1][2][1] - d[1][0][1]) * 0.5f;
                gs = (d[2][1][1] - d[0][1][1]) * 0.5f;
                const float v2 = v * 2.0f;
                dxx = d[1][1][2] + d[1][1][0] - v2;
                dyy = d[1][2][1] + d[1][0][1] - v2;
                const float dss = d[2][1][1] + d[0][1][1] - v2;
                dxy = (d[1][2][2] - d[1][2][0] - d[1][0][2] + d[1][0][0]) * 0.25f;
                const float dxs = (d[2][1][2] - d[2][1][0] - d[0][1][2] + d[0][1][0]) * 0.25f;
                const float dys = (d[2][2][1] - d[2][0][1] - d[0][2][1] + d[0][0][1]) * 0.25f;
                const float c00 = dyy * dss - dys * dys;
                const float c01 = dxs * dys - dxy * dss;
                const float c02 = dxy * dys - dxs * dyy;
                const float c11 = dxx * dss - dxs * dxs;
                const float c12 = dxs * dxy - dxx * dys;
                const float c22 = dxx * dyy - dxy * dxy;
                const float det = dxx * c00 + dxy * c01 + dxs * c02;
                if (det == 0.0f)
                    break;
                Xx = -(c00 * gx + c01 * gy + c02 * gs) / det;
                Xy = -(c01 * gx + c11 * gy + c12 * gs) / det;
                Xs = -(c02 * gx + c12 * gy + c22 * gs) / det;
                if (fabsf(Xx) < 0.5f && fabsf(Xy) < 0.5f && fabsf(Xs) < 0.5f) {
                    ok = 1;
                    break;
                }
                if (!(fabsf(Xx) < 1e6f && fabsf(Xy) < 1e6f && fabsf(Xs) < 1e6f))
                    break; /* also catches NaN */
                x += (int)floorf(Xx + 0.5f);
                y += (int)floorf(Xy + 0.5f);
                c += (int)floorf(Xs + 0.5f);
                if (c < 1 || c > S || x <= 8 || x >= w - 8 || y <= 8 || y >= h - 8)
                    break;
            }
            if (!ok)
                continue;
            const float dhat = v + 0.5f * (gx * Xx + gy * Xy + gs * Xs);
            if (fabsf(dhat) * (float)S < T)
                continue;
            const float tr = dxx + dyy, det2 = dxx * dyy - dxy * dxy;
            if (det2 <= 0.0f || tr * tr * r >= (r + 1.0f) * (r + 1.0f) * det2)
                continue;
            if (n < max_out) {
                out[n].x = x;
                out[n].y = y;
                out[n].c = c;
                out[n].dx = Xx;
                out[n].dy = Xy;
                out[n].ds = Xs;
            }
            ++n;
        }
    return n;
}

void oracle_upsample2x(const float *src, int w, int pitch, int h, float *dst)
{
    const int W = 2 * w;
    for (int Y = 0; Y < 2 * h; ++Y)
        for (int X = 0; X < W; ++X) {
            const int x0 = X >> 1, y0 = Y >> 1;
            const int x1 = (X & 1) ? (x0 + 1 < w ? x0 + 1 : x0) : x0;
            const int y1 = (Y & 1) ? (y0 + 1 < h ? y0 + 1 : y0) : y0;
            const float a = src[x0 + (long)pitch * y0], b = src[x1 + (long)pitch * y0];
            const float c = src[x0 + (long)pitch * y1], d = src[x1 + (long)pitch * y1];
            dst[X + (long)W * Y] = ((a + b) + (c + d)) * 0.25f;
        }
}

void oracle_quantize_descriptors(const OracleSiftPoint *pts, int n, unsigned char *out)
{
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < 128; ++k) {
            float q = floorf(pts[j].data[k] * 512.0f + 0.5f);
            out[128 * j + k] = (unsigned char)(q > 255.0f ? 255.0f : q);
        }
}

/* ---- extractSift cudaSiftH.cu:9-117 --------------------------------------- */
int oracle_extract(const OracleParams *pr, const float *img, int w, int pitch, int h, int octave0,
                   OracleSiftPoint *pts, unsigned char *meta, int *total_found)
{
    int S = pr->n_scales;
    float taps[8 * TAPS], wnd[8 * 16], dwnd[16];
    oracle_tables(pr, taps, wnd, dwnd);
    int count = 0, total = 0;
    const float *src = img;
    float *owned_src = NULL;
    int p = pitch;
    int *xy = (int *)malloc(sizeof(int) * 2 * pr->max_points);
    const int lowe = pr->profile == 1;
    OracleLowePoint *lp = lowe ? (OracleLowePoint *)malloc(sizeof(OracleLowePoint) * pr->max_points) : NULL;
    int n_oct = pr->n_octaves;
    double pix = 1.0; /* lowe: input pixels per pixel of the current octave */
    if (lowe && pr->upsample) { /* one more octave in front, built from the image at twice its size */
        float *up = (float *)malloc(sizeof(float) * 4L * w * h);
        oracle_upsample2x(img, w, pitch, h, up);
        owned_src = up;
        src = up;
        w *= 2;
        h *= 2;
        p = w;
        n_oct += 1;
        pix = 0.5;
    }
    for (int k = 0; k < octave0; ++k)
        pix *= 2.0;
    for (int o = octave0; o < n_oct; ++o, pix *= 2.0) {
        if (count >= pr->max_points) /* cudaSiftH.cu:12-13 */
            break;
        long gDim = (long)p * h;
        float *tmp = (float *)malloc(sizeof(float) * gDim * (S + 3));
        float *blur = (float *)malloc(sizeof(float) * gDim * (S + 3));
        float *dog = (float *)malloc(sizeof(float) * gDim * (S + 2));
        float *hess = (float *)malloc(sizeof(float) * gDim * 3 * S);
        float *grad = (float *)malloc(sizeof(float) * gDim * 2 * S);
        /* pad columns [w,p) are never written by the reference: poison them */
        for (long i = 0; i < gDim * (S + 3); ++i)
            tmp[i] = blur[i] = NAN;
        for (long i = 0; i < gDim * (S + 2); ++i)
            dog[i] = NAN;
        for (long i = 0; i < gDim * 3 * S; ++i)
            hess[i] = NAN;
        for (long i = 0; i < gDim * 2 * S; ++i)
            grad[i] = NAN;
        oracle_xblur(pr, src, tmp, w, p, h);
        oracle_yblur(pr, tmp, blur, w, p, h);
        oracle_dog(pr, blur, dog, w, p, h);
        oracle_hessian(pr, blur, hess, w, p, h);
        oracle_gradient(pr, blur, grad, w, p, h);
        for (int i = 0; i < S && lowe; ++i) {
            if (count >= pr->max_points)
                break;
            int room = pr->max_points - count;
            int n = oracle_find_extrema_lowe(pr, blur, w, p, h, i, lp, room);
            total += n;
            if (n > room)
                n = room;
            for (int j = 0; j < n; ++j) {
                OracleSiftPoint *q = &pts[count + j];
                const int sc = lp[j].c - 1; /* scale index the point ended on */
                const float *g2 = grad + (long)sc * 2 * gDim;
                q->xpos = (float)(((double)((float)lp[j].x + lp[j].dx)) * pix);
                q->ypos = (float)(((double)((float)lp[j].y + lp[j].dy)) * pix);
                q->scale = (float)((double)pr->sigma * pix *
                                   oracle_det_exp2((double)((float)sc + lp[j].ds) / (double)S));
                /* orientation and descriptor at the integer sample, as the reference takes them */
                q->orientation = oracle_orientation(g2, w, p, h, wnd + sc * 16, lp[j].x, lp[j].y, NULL);
                oracle_descriptor(g2, w, p, h, dwnd, lp[j].x, lp[j].y, q->orientation, q->data);
                if (meta) {
                    meta[2 * (count + j)] = (unsigned char)o;
                    meta[2 * (count + j) + 1] = (unsigned char)sc;
                }
            }
            count += n;
        }
        for (int i = 0; i < S && !lowe; ++i) { /* computeOctaveSift cudaSiftH.cu:80-116 */
            if (count >= pr->max_points)
                break;
            int room = pr->max_points - count;
            int n = oracle_find_extrema_closed(pr, dog + i * gDim, hess + (long)i * 3 * gDim, w,
                                               p, h, xy, room);
            total += n;
            if (n > room)
                n = room; /* clampPtCount cudaSiftH.cu:340-352 */
            float scale = oracle_scale_value(pr, i, o);
            const float *g2 = grad + (long)i * 2 * gDim;
            for (int j = 0; j < n; ++j) {
                OracleSiftPoint *q = &pts[count + j];
                q->xpos = (float)xy[2 * j];
                q->ypos = (float)xy[2 * j + 1];
                q->scale = scale;
                q->orientation = oracle_orientation(g2, w, p, h, wnd + i * 16, xy[2 * j],
                                                    xy[2 * j + 1], NULL);
                oracle_descriptor(g2, w, p, h, dwnd, xy[2 * j], xy[2 * j + 1], q->orientation,
                                  q->data);
                if (meta) {
                    meta[2 * (count + j)] = (unsigned char)o;
                    meta[2 * (count + j) + 1] = (unsigned char)i;
                }
            }
            count += n;
        }
        float *next = NULL;
        int w_ = oracle_idivup(w, 2), h_ = oracle_idivup(h, 2), p_ = oracle_ialignup(w_, 128);
        if (o + 1 < n_oct) { /* cudaSiftH.cu:40-55 */
            next = (float *)malloc(sizeof(float) * (long)p_ * h_);
            for (long i = 0; i < (long)p_ * h_; ++i)
                next[i] = NAN;
            oracle_downsample(blur + (long)S * gDim, next, w, p, h);
        }
        free(tmp);
        free(blur);
        free(dog);
        free(hess);
        free(grad);
        free(owned_src);
        owned_src = next;
        src = next;
        w = w_;
        h = h_;
        p = p_;
    }
    free(owned_src);
    free(xy);
    free(lp);
    if (total_found)
        *total_found = total;
    return count;
}

typedef struct {
    const OracleParams *pr;
    const float *imgs;
    long img_stride;
    int n, w, pitch, h;
    OracleSiftPoint *pts;
    int *counts;
    int next;
    pthread_mutex_t mu;
} BatchJob;

static void *batch_worker(void *arg)
{
    BatchJob *jb = (BatchJob *)arg;
    for (;;) {
        pthread_mutex_lock(&jb->mu);
        int i = jb->next++;
        pthread_mutex_unlock(&jb->mu);
        if (i >= jb->n)
            break;
        jb->counts[i] = oracle_extract(jb->pr, jb->imgs + (long)i * jb->img_stride, jb->w,
                                       jb->pitch, jb->h, 0,
                                       jb->pts + (long)i * jb->pr->max_points, NULL, NULL);
    }
    return NULL;
}

int oracle_extract_batch(const OracleParams *pr, const float *imgs, long img_stride, int n, int w,
                         int pitch, int h, OracleSiftPoint *pts, int *counts, int n_threads)
{
    BatchJob jb = {pr, imgs, img_stride, n, w, pitch, h, pts, counts, 0, PTHREAD_MUTEX_INITIALIZER};
    if (n_threads < 1)
        n_threads = 1;
    if (n_threads > 256)
        n_threads = 256;
    pthread_t th[256];
    for (int t = 0; t < n_threads; ++t)
        pthread_create(&th[t], NULL, batch_worker, &jb);
    for (int t = 0; t < n_threads; ++t)
        pthread_join(th[t], NULL);
    return 0;
}

void oracle_gray_q14(const unsigned char *rgb, int npix, unsigned char *gray)
{
    for (int i = 0; i < npix; ++i) {
        int R = rgb[3 * i], G = rgb[3 * i + 1], B = rgb[3 * i + 2];
        gray[i] = (unsigned char)((1868 * B + 9617 * G + 4899 * R + 8192) >> 14);
    }
}

/* saveSift sift.cu:151-160 */
long oracle_format_points(const OracleSiftPoint *pts, int n, char *buf, long cap)
{
    long off = 0;
    for (int j = 0; j < n; ++j) {
        if (cap - off < 4096)
            return -1;
        off += sprintf(buf + off, "%.1f\t%.1f\t%.3f\t%.3f\t", pts[j].xpos, pts[j].ypos,
                       pts[j].scale, pts[j].orientation);
        for (int k = 0; k < 128; ++k)
            off += sprintf(buf + off, "%.6f ", pts[j].data[k]);
        buf[off++] = '\n';
    }
    return off;
}
