/*
 * Deterministic synthetic test images (no dependencies): mid-grey + smooth
 * value noise + Gaussian blobs + pixel noise, clamped to [0,255] and rounded
 * to integers so the values lie in the domain an 8-bit image load produces
 * (the reference feeds imread(...GRAYSCALE) converted to float, utils.cc:13).
 */
#include "../../include/sift_amd.h"

#include <math.h>
#include <stdlib.h>

static unsigned long long splitmix64(unsigned long long *s)
{
    unsigned long long z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double unit(unsigned long long *s) { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); }
static unsigned long long hash2(unsigned long long seed, unsigned long long i)
{
    unsigned long long s = seed ^ (i * 0xD6E8FEB86659FD93ull);
    return splitmix64(&s);
}

void sift_synth_image(float *dst, int w, int pitch, int h, unsigned long long seed, int dense)
{
    unsigned long long st = seed;
    const int L = 64; /* value-noise lattice pitch */
    const int gw = w / L + 2, gh = h / L + 2;
    float *lat = (float *)malloc(sizeof(float) * gw * gh);
    for (int i = 0; i < gw * gh; ++i)
        lat[i] = (float)((unit(&st) * 2.0 - 1.0) * 32.0);
    for (int y = 0; y < h; ++y) {
        const int gy = y / L;
        const float fy = (float)(y % L) / L;
        for (int x = 0; x < w; ++x) {
            const int gx = x / L;
            const float fx = (float)(x % L) / L;
            const float a = lat[gy * gw + gx], b = lat[gy * gw + gx + 1];
            const float c = lat[(gy + 1) * gw + gx], d = lat[(gy + 1) * gw + gx + 1];
            const float top = a + (b - a) * fx, bot = c + (d - c) * fx;
            dst[(long)y * pitch + x] = 128.0f + top + (bot - top) * fy;
        }
    }
    free(lat);
    const long K = dense ? ((long)w * h / 150) : ((long)w * h / 1500);
    const double smin = dense ? 1.2 : 1.5, smax = dense ? 4.0 : 12.0;
    for (long k = 0; k < K; ++k) {
        const double cx = unit(&st) * w, cy = unit(&st) * h;
        const double sg = smin * exp(unit(&st) * log(smax / smin));
        double amp = 20.0 + unit(&st) * 70.0;
        if (splitmix64(&st) & 1)
            amp = -amp;
        const int R = (int)ceil(4.0 * sg);
        const int x0 = (int)floor(cx) - R, x1 = (int)floor(cx) + R;
        const int y0 = (int)floor(cy) - R, y1 = (int)floor(cy) + R;
        const float inv = (float)(1.0 / (2.0 * sg * sg));
        for (int y = y0 < 0 ? 0 : y0; y <= y1 && y < h; ++y)
            for (int x = x0 < 0 ? 0 : x0; x <= x1 && x < w; ++x) {
                const float dx = (float)(x - cx), dy = (float)(y - cy);
                dst[(long)y * pitch + x] += (float)amp * expf(-(dx * dx + dy * dy) * inv);
            }
    }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const unsigned long long r = hash2(seed, (unsigned long long)y * (unsigned)w + (unsigned)x);
            float v = dst[(long)y * pitch + x] + (float)(((double)(r >> 11) * (1.0 / 9007199254740992.0)) * 4.0 - 2.0);
            v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
            dst[(long)y * pitch + x] = rintf(v);
        }
}
