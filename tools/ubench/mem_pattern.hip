// Micro-benchmark: which access pattern of "read one plane, write NPL planes, walking down a
// strip of rows" the MI355X memory system sustains best.  The fused pyramid kernel moves exactly
// these bytes; this program moves them with a chosen amount of arithmetic per element, so the
// memory ceiling of each pattern and how well arithmetic hides behind it can be read off.
//
//   VW     floats per lane and row (1, 2 or 4): a wave covers 64*VW columns
//   WAVES  waves per workgroup, side by side in x
//   LAYOUT 0 planar (plane i at i*p*h), 1 pixel-interleaved (pixel x holds its NPL values together)
//   FMAS   dependent-free FMAs per element and plane-row (stands in for the blur arithmetic)
//   PF     rows fetched ahead
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int VW> struct Vec;
template <> struct Vec<1> { typedef float T; };
template <> struct Vec<2> { typedef float2 T; };
template <> struct Vec<4> { typedef float4 T; };

template <int VW> __device__ __forceinline__ void fma_all(typename Vec<VW>::T &v, float a, float b);
template <> __device__ __forceinline__ void fma_all<1>(float &v, float a, float b) { v = __builtin_fmaf(v, a, b); }
template <> __device__ __forceinline__ void fma_all<2>(float2 &v, float a, float b)
{
    v.x = __builtin_fmaf(v.x, a, b);
    v.y = __builtin_fmaf(v.y, a, b);
}
template <> __device__ __forceinline__ void fma_all<4>(float4 &v, float a, float b)
{
    v.x = __builtin_fmaf(v.x, a, b);
    v.y = __builtin_fmaf(v.y, a, b);
    v.z = __builtin_fmaf(v.z, a, b);
    v.w = __builtin_fmaf(v.w, a, b);
}

//   SYNC   1: the waves of a workgroup meet at a barrier after every row (they stay in lockstep, so a
//          workgroup's stores of one row form one contiguous 256*WAVES-byte run in time as well as in space)
//   PERSIST 1: the grid is the resident wave count; each workgroup walks the tile list with stride gridDim
template <int NPL, int VW, int WAVES, int LAYOUT, int FMAS, int PF, int LDSB, int REMAP, int SYNC = 0, int PERSIST = 0>
__global__ __launch_bounds__(64 * WAVES) void strips(float *__restrict__ out, const float *__restrict__ in, int w, int p,
                                                     int h, int rows, long out_stride, float a, float b, int vgx = 0,
                                                     int vgy = 0, int vgz = 0)
{
    typedef typename Vec<VW>::T V;
    __shared__ float occ[LDSB / 4 + 1]; // caps the waves per CU
    if (a == 12345.f) occ[threadIdx.x] = b;
    if (a == 54321.f) out[0] = occ[threadIdx.x + 1];
    // XCD-aware remap: each XCD gets a contiguous run of tiles
    const unsigned gx = PERSIST ? vgx : gridDim.x, gy = PERSIST ? vgy : gridDim.y;
    const unsigned nwg = gx * gy * (PERSIST ? vgz : gridDim.z);
    for (unsigned orig = PERSIST ? blockIdx.x : blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z); orig < nwg;
         orig += PERSIST ? gridDim.x : nwg) {
    int tx, ty, tz;
    {
        const unsigned q = nwg / 8, r = nwg % 8, xcd = orig % 8;
        const unsigned t = REMAP ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8 : orig;
        tx = (int)(t % gx);
        ty = (int)((t / gx) % gy);
        tz = (int)(t / (gx * gy));
    }
    const int x = (tx * 64 * WAVES + threadIdx.x) * VW;
    if (x >= w) { if (SYNC || PERSIST) continue; else return; }
    const int y0 = ty * rows, y1 = min(y0 + rows, h);
    const float *s = in + tz * (long)p * h + x;
    float *o = out + tz * out_stride;
    const long plane = (long)p * h;
    V g[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) g[q] = *reinterpret_cast<const V *>(s + (long)min(y0 + q, h - 1) * p);
    for (int y = y0; y < y1; y += PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            V v = g[q];
            g[q] = *reinterpret_cast<const V *>(s + (long)min(y + q + PF, h - 1) * p);
            if (y + q < y1) {
                V pv[NPL];
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
                    pv[i] = v;
#pragma unroll
                    for (int k = 0; k < FMAS; ++k) fma_all<VW>(pv[i], a, b + (float)i);
                }
                if (LAYOUT == 0) {
#pragma unroll
                    for (int i = 0; i < NPL; ++i) *reinterpret_cast<V *>(o + i * plane + (long)(y + q) * p + x) = pv[i];
                } else if (LAYOUT == 3) { // tile-major: a 64*VW*WAVES-column tile's rows are contiguous, plane by plane
                    const long tw = 64 * VW * WAVES;
#pragma unroll
                    for (int i = 0; i < NPL; ++i)
                        *reinterpret_cast<V *>(o + i * plane + tx * tw * h + (long)(y + q) * tw + threadIdx.x * VW) = pv[i];
                } else if (LAYOUT == 4) { // tile-major with the planes interleaved row by row: one stream per wave
                    const long tw = 64 * VW * WAVES;
#pragma unroll
                    for (int i = 0; i < NPL; ++i)
                        *reinterpret_cast<V *>(o + tx * tw * h * NPL + ((long)(y + q) * NPL + i) * tw + threadIdx.x * VW) = pv[i];
                } else if (LAYOUT == 2) { // planes interleaved row by row
#pragma unroll
                    for (int i = 0; i < NPL; ++i) *reinterpret_cast<V *>(o + ((long)(y + q) * NPL + i) * p + x) = pv[i];
                } else {
                    // pixel-interleaved: pixel x of row y at ((y*p + x) * NPL); VW pixels per lane are contiguous
                    float *d = o + ((long)(y + q) * p + x) * NPL;
                    float t[NPL * VW];
#pragma unroll
                    for (int i = 0; i < NPL; ++i)
#pragma unroll
                        for (int c = 0; c < VW; ++c) t[c * NPL + i] = reinterpret_cast<const float *>(&pv[i])[c];
                    if ((NPL * VW) % 4 == 0) {
#pragma unroll
                        for (int j = 0; j < NPL * VW / 4; ++j)
                            reinterpret_cast<float4 *>(d)[j] = float4{t[4 * j], t[4 * j + 1], t[4 * j + 2], t[4 * j + 3]};
                    } else {
#pragma unroll
                        for (int j = 0; j < NPL * VW; ++j) d[j] = t[j];
                    }
                }
            }
            if (SYNC) __syncthreads();
        }
    }
    }
}

// 64 columns per wave as the pyramid kernel has them, but the stores of four rows go out together:
// lane l writes 16 bytes, columns 4*(l%16).., of row y + l/16, so one instruction covers four
// 256-byte row segments (what a transposition of four finished rows through LDS would allow).
template <int NPL, int PF, int LDSB>
__global__ __launch_bounds__(64) void strips_rows4(float *__restrict__ out, const float *__restrict__ in, int w, int p, int h,
                                                   int rows, long out_stride, float a, float b)
{
    __shared__ float occ[LDSB / 4 + 1];
    if (a == 12345.f) occ[threadIdx.x] = b;
    if (a == 54321.f) out[0] = occ[threadIdx.x + 1];
    int tx, ty, tz;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y;
        const unsigned nwg = gx * gy * gridDim.z;
        const unsigned orig = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const unsigned q = nwg / 8, r = nwg % 8, xcd = orig % 8;
        const unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
        tx = (int)(t % gx);
        ty = (int)((t / gx) % gy);
        tz = (int)(t / (gx * gy));
    }
    const int lane = threadIdx.x;
    const int x = tx * 64 + lane;
    const int y0 = ty * rows, y1 = min(y0 + rows, h);
    const float *s = in + tz * (long)p * h + x;
    float *o = out + tz * out_stride + tx * 64 + (lane & 15) * 4;
    const long plane = (long)p * h;
    float g[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) g[q] = s[(long)min(y0 + q, h - 1) * p];
    for (int y = y0; y < y1; y += PF) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            v += g[q];
            g[q] = s[(long)min(y + q + PF, h - 1) * p];
        }
#pragma unroll
        for (int q4 = 0; q4 < PF; q4 += 4) {
            const int yr = y + q4 + (lane >> 4);
            if (yr < y1) {
#pragma unroll
                for (int i = 0; i < NPL; ++i)
                    *reinterpret_cast<float4 *>(o + i * plane + (long)yr * p) = float4{v, v + a, v + b, v + (float)i};
            }
        }
    }
}

__global__ __launch_bounds__(256) void flat_copy(float4 *out, const float4 *in, long n)
{
    long i = blockIdx.x * 256L + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = in[i];
}
__global__ __launch_bounds__(256) void flat_store(float4 *out, long n)
{
    long i = blockIdx.x * 256L + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = float4{1.f, 2.f, 3.f, (float)i};
}

__global__ __launch_bounds__(256) void flat_store1(float *out, long n)
{
    long i = blockIdx.x * 256L + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = (float)i;
}
__global__ __launch_bounds__(256) void flat_read(const float4 *in, long n, float *sink)
{
    long i = blockIdx.x * 256L + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    float acc = 0.f;
    for (; i < n; i += stride) {
        const float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) *sink = acc;
}

static hipEvent_t e0, e1;
template <class F> static float best_ms(F f)
{
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0, 0);
        f();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    return best;
}

constexpr int NPL = 5;
static const int W = 1920, H = 1080, P = 1920, NIMG = 128;
static float *g_out, *g_in;

template <int VW, int WAVES, int LAYOUT, int FMAS, int PF, int LDSB = 0, int REMAP = 1, int SYNC = 0, int PERSIST = 0>
static void run(int rows, int resident_blocks = 0)
{
    const long plane = (long)P * H;
    const dim3 vgrid((W + 64 * WAVES * VW - 1) / (64 * WAVES * VW), (H + rows - 1) / rows, NIMG);
    const dim3 grid = PERSIST ? dim3(resident_blocks) : vgrid;
    const float ms = best_ms([&] {
        hipLaunchKernelGGL((strips<NPL, VW, WAVES, LAYOUT, FMAS, PF, LDSB, REMAP, SYNC, PERSIST>), grid, dim3(64 * WAVES), 0, 0, g_out,
                           g_in, W, P, H, rows, plane * NPL, 0.999f, 0.001f, (int)vgrid.x, (int)vgrid.y, (int)vgrid.z);
    });
    const double bytes = (double)plane * NIMG * 4 * (NPL + 1);
    printf("VW=%d waves/WG=%d layout=%s FMA/elem/plane=%3d PF=%d rows=%4d lds=%5d remap=%d sync=%d persist=%d(%d): %7.3f ms  %5.2f TB/s  (%.1f us/image)\n", VW, WAVES,
           LAYOUT == 0 ? "planar" : (LAYOUT == 1 ? "pixel " : (LAYOUT == 2 ? "rowint" : (LAYOUT == 3 ? "tiled " : "tilint"))), FMAS, PF, rows, LDSB, REMAP, SYNC, PERSIST, resident_blocks, ms, bytes / ms / 1e9, ms * 1e3 / NIMG);
    fflush(stdout);
}

template <int PF, int LDSB> static void run_rows4(int rows)
{
    const long plane = (long)P * H;
    dim3 grid(W / 64, (H + rows - 1) / rows, NIMG);
    const float ms = best_ms([&] {
        hipLaunchKernelGGL((strips_rows4<NPL, PF, LDSB>), grid, dim3(64), 0, 0, g_out, g_in, W, P, H, rows, plane * NPL, 0.999f, 0.001f);
    });
    const double bytes = (double)plane * NIMG * 4 * (NPL + 1);
    printf("64 columns per wave, stores of 4 rows per instruction, PF=%d rows=%4d lds=%5d: %7.3f ms  %5.2f TB/s  (%.1f us/image)\n", PF,
           rows, LDSB, ms, bytes / ms / 1e9, ms * 1e3 / NIMG);
    fflush(stdout);
}

// The kernel's shape (64 columns per wave, one wave per workgroup, strips of rows, planar) with the mix
// taken apart: NPLX planes written per row (0..5), with (READ = 1) or without the source row read.
// SPLIT = 1: twice the waves, every second one only reads its tile's source rows and the others only
// write the planes: the same bytes in the same launch, but no wave waits for a load behind its stores.
template <int NPLX, int READ, int LDSB, int SPLIT, int PF>
__global__ __launch_bounds__(64) void strips_mix(float *__restrict__ out, const float *__restrict__ in, int w, int p, int h,
                                                 int rows, long out_stride, float a, float b)
{
    __shared__ float occ[LDSB / 4 + 1];
    if (a == 12345.f) occ[threadIdx.x] = b;
    if (a == 54321.f) out[0] = occ[threadIdx.x + 1];
    const unsigned gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
    const unsigned orig = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned q = nwg / 8, r = nwg % 8, xcd = orig % 8;
    unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    // SPLIT = 2: every image reads the source rows of image 0, so the reads hit in the caches and HBM sees writes only
    const bool reads = SPLIT == 1 ? (t & 1u) == 0 : READ != 0, writes = SPLIT == 1 ? (t & 1u) != 0 : NPLX > 0;
    if (SPLIT == 1) t >>= 1;
    const int tx = (int)(t % gx), ty = (int)((t / gx) % gy), tz = (int)(t / (gx * gy));
    const int x = tx * 64 + threadIdx.x;
    if (x >= w) return;
    const int y0 = ty * rows, y1 = min(y0 + rows, h);
    const long plane = (long)p * h;
    const float *s = in + (SPLIT == 2 ? 0 : tz) * plane + x;
    float *o = out + tz * out_stride + x;
    float g[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) g[k] = reads ? s[(long)min(y0 + k, h - 1) * p] : a;
    float sink = 0.f;
    for (int y = y0; y < y1; y += PF) {
        float cur[PF];
        if (SPLIT == 3) { // the PF loads of the next chunk go out together, then PF rows of stores
#pragma unroll
            for (int k = 0; k < PF; ++k) cur[k] = g[k];
#pragma unroll
            for (int k = 0; k < PF; ++k) g[k] = s[(long)min(y + k + PF, h - 1) * p];
        }
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const float v = SPLIT == 3 ? cur[k] : g[k];
            if (SPLIT != 3 && reads) g[k] = s[(long)min(y + k + PF, h - 1) * p];
            if (y + k < y1) {
                if (!writes) sink += v;
                else {
#pragma unroll
                    for (int i = 0; i < NPLX; ++i) o[i * plane + (long)(y + k) * p] = v + (float)i;
                }
            }
        }
    }
    if (!writes && sink == 123.456f) out[0] = sink;
}
template <int NPLX, int READ, int LDSB, int SPLIT = 0, int PF = 4> static void run_mix(int rows)
{
    const long plane = (long)P * H;
    dim3 grid(W / 64, (H + rows - 1) / rows, NIMG * (SPLIT == 1 ? 2 : 1));
    const float ms = best_ms([&] {
        hipLaunchKernelGGL((strips_mix<NPLX, READ, LDSB, SPLIT, PF>), grid, dim3(64), 0, 0, g_out, g_in, W, P, H, rows, plane * NPL, 0.999f,
                           0.001f);
    });
    const double bytes = (double)plane * NIMG * 4 * (NPLX + (READ ? 1 : 0));
    printf("kernel's shape, %d plane(s) written, %2d rows ahead, source row %s, lds=%5d: %7.3f ms  %5.2f TB/s\n", NPLX, PF,
           SPLIT == 3 ? "read in bursts" : SPLIT == 1 ? "read by other waves" : (SPLIT == 2 ? "read from the caches" : (READ ? "read    " : "not read")), LDSB, ms, bytes / ms / 1e9);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int set = argc > 1 ? atoi(argv[1]) : 0;
    const long plane = (long)P * H;
    hipMalloc(&g_out, plane * NPL * NIMG * 4);
    hipMalloc(&g_in, plane * NIMG * 4);
    hipMemset(g_in, 0, plane * NIMG * 4);
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    if (set == 0 || set == 1) { // flat copies, stores and reads over grid sizes (256-thread blocks)
        const long n = plane * NIMG / 4, m = plane * NPL * NIMG / 4;
        for (int blocks : {256, 512, 768, 1024, 1280, 1536, 2048, 4096, 16384}) {
            float ms = best_ms([&] { hipLaunchKernelGGL(flat_copy, dim3(blocks), dim3(256), 0, 0, (float4 *)g_out, (const float4 *)g_in, n); });
            printf("%5d blocks: flat float4 copy %.3f ms %.2f TB/s (read + write)", blocks, ms, 2.0 * n * 16 / ms / 1e9);
            ms = best_ms([&] { hipLaunchKernelGGL(flat_store, dim3(blocks), dim3(256), 0, 0, (float4 *)g_out, m); });
            printf(" | float4 store %.3f ms %.2f TB/s", ms, (double)m * 16 / ms / 1e9);
            ms = best_ms([&] { hipLaunchKernelGGL(flat_store1, dim3(blocks), dim3(256), 0, 0, g_out, m * 4); });
            printf(" | dword store %.3f ms %.2f TB/s", ms, (double)m * 16 / ms / 1e9);
            ms = best_ms([&] { hipLaunchKernelGGL(flat_read, dim3(blocks), dim3(256), 0, 0, (const float4 *)g_out, m, g_in); });
            printf(" | float4 read %.3f ms %.2f TB/s\n", ms, (double)m * 16 / ms / 1e9);
            fflush(stdout);
        }
    }
    if (set == 0 || set == 2) { // the pass's own pattern: launch shape, lockstep, resident grid
        run<1, 1, 0, 0, 4, 13000>(540);
        run<1, 1, 0, 0, 4, 13000, 1, 0, 1>(540, 3072);
        run<1, 1, 0, 0, 4, 13000, 1, 0, 1>(540, 2048);
        run<1, 1, 0, 0, 4, 13000, 1, 0, 1>(540, 1536);
        run<1, 1, 0, 0, 4, 13000, 1, 0, 1>(540, 1024);
        run<1, 1, 0, 0, 4, 13000, 1, 0, 1>(1080, 3072);
        run<1, 1, 0, 0, 4, 13000, 1, 0, 1>(180, 3072);
        run<2, 1, 0, 0, 4, 13000>(540);
        run<2, 1, 0, 0, 4, 13000, 1, 0, 1>(540, 3072);
        run<2, 1, 0, 0, 4, 13000, 1, 0, 1>(540, 2048);
        run<2, 1, 0, 0, 4, 20000, 1, 0, 1>(540, 2048);
        run<2, 1, 0, 0, 8, 20000, 1, 0, 1>(540, 2048);
        // lockstep workgroups: 5, 6, 10 or 15 waves side by side, a barrier per row
        run<1, 5, 0, 0, 4, 0, 1, 1>(540);
        run<1, 6, 0, 0, 4, 0, 1, 1>(540);
        run<1, 6, 0, 0, 4, 0, 1, 0>(540);
        run<1, 10, 0, 0, 4, 0, 1, 1>(540);
        run<1, 15, 0, 0, 4, 0, 1, 1>(540);
        run<1, 15, 0, 0, 4, 0, 1, 0>(540);
        run<1, 6, 0, 0, 4, 78000, 1, 1>(540);  // two 6-wave workgroups per CU = 3 waves per SIMD
        run<1, 6, 0, 0, 4, 78000, 1, 1, 1>(540, 512);
        run<1, 15, 0, 0, 4, 0, 1, 1, 1>(540, 512);
        run<2, 5, 0, 0, 4, 0, 1, 1>(540);
        run<2, 15, 0, 0, 4, 0, 1, 1>(540);
        run<1, 6, 0, 38, 4, 78000, 1, 1>(540);
        run<1, 1, 0, 38, 4, 13000>(540);
        run<1, 1, 0, 38, 4, 13000, 1, 0, 1>(540, 3072);
        run<2, 1, 0, 38, 4, 20000>(540);
        run<2, 1, 0, 38, 4, 20000, 1, 0, 1>(540, 2048);
    }
    if (set == 3) { // the mix taken apart, at the kernel's occupancy (3 waves per SIMD) and unthrottled
        run_mix<0, 1, 13000>(540);
        run_mix<1, 1, 13000>(540);
        run_mix<2, 1, 13000>(540);
        run_mix<3, 1, 13000>(540);
        run_mix<5, 1, 13000>(540);
        run_mix<1, 0, 13000>(540);
        run_mix<2, 0, 13000>(540);
        run_mix<3, 0, 13000>(540);
        run_mix<5, 0, 13000>(540);
        run_mix<0, 1, 0>(540);
        run_mix<1, 1, 0>(540);
        run_mix<5, 1, 0>(540);
        run_mix<1, 0, 0>(540);
        run_mix<5, 0, 0>(540);
        run_mix<5, 1, 13000, 1>(540);
        run_mix<5, 1, 6500, 1>(540);
        run_mix<5, 1, 0, 1>(540);
        run_mix<1, 1, 13000, 1>(540);
        run_mix<5, 1, 13000, 2>(540);
        run_mix<5, 1, 0, 2>(540);
        run_mix<1, 1, 13000, 2>(540);
        run_mix<5, 1, 13000, 0, 2>(540);
        run_mix<5, 1, 13000, 0, 6>(540);
        run_mix<5, 1, 13000, 0, 8>(540);
        run_mix<5, 1, 13000, 0, 10>(540);
        run_mix<5, 1, 13000, 0, 20>(540);
        run_mix<5, 1, 13000, 3, 4>(540);
        run_mix<5, 1, 13000, 3, 8>(540);
        run_mix<5, 1, 13000, 3, 16>(540);
        run_mix<5, 1, 0, 3, 16>(540);
        run_mix<5, 1, 13000, 2, 8>(540);
        run_mix<5, 1, 13000, 2, 10>(540);
        run_mix<5, 1, 13000, 2, 20>(540);
    }
    return 0;
}
