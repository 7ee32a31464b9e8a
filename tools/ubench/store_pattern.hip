// Micro-benchmark: HBM write rate of the fused pyramid kernel's store pattern
// (every wave writes one 256-byte row segment into each of NPL planes per row,
// walking down a strip) against a flat 16-byte-per-lane streaming store, and
// the same with a concurrent read stream of one plane.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NPL, bool READ, int NT>
__global__ __launch_bounds__(64) void strips(float *out, const float *in, int p, int h, int rows, long img_stride)
{
    const int lane = threadIdx.x;
    const int x = blockIdx.x * 64 + lane;
    const int y0 = blockIdx.y * rows;
    const int y1 = min(y0 + rows, h);
    float *o = out + blockIdx.z * img_stride;
    const float *s = in + blockIdx.z * (long)p * h;
    const long plane = (long)p * h;
    float v = (float)lane;
    if (NT & 16) { // two planes per store instruction: lanes 0..31 write plane 2j, lanes 32..63 plane 2j+1, 8 bytes each
        float g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = s[(long)min(y0 + q, h - 1) * p + x];
        const int half = lane >> 5, xc = blockIdx.x * 64 + 2 * (lane & 31);
        for (int y = y0; y < y1; y += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v += g[q];
                g[q] = s[(long)min(y + q + 4, h - 1) * p + x];
                if (y + q < y1) {
#pragma unroll
                    for (int i = 0; i + 1 < NPL; i += 2)
                        *reinterpret_cast<float2 *>(&o[(i + half) * plane + (long)(y + q) * p + xc]) = float2{v + i, v - i};
                    if (NPL & 1) o[(NPL - 1) * plane + (long)(y + q) * p + x] = v;
                }
            }
        }
        return;
    }
    if (NT & 4) { // source rows fetched four rows ahead, as the pyramid kernel does
        float g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = s[(long)min(y0 + q, h - 1) * p + x];
        for (int y = y0; y < y1; y += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v += g[q];
                g[q] = s[(long)min(y + q + 4, h - 1) * p + x];
                if (y + q < y1) {
#pragma unroll
                    for (int i = 0; i < NPL; ++i) {
                        if (NT & 8) o[((long)(y + q) * NPL + i) * p + x] = v + i; // planes interleaved row by row
                        else o[i * plane + (long)(y + q) * p + x] = v + i;
                    }
                }
            }
        }
        return;
    }
    for (int y = y0; y < y1; ++y) {
        if (READ) v += (NT & 2) ? __builtin_nontemporal_load(&s[(long)y * p + x]) : s[(long)y * p + x];
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            if (NT & 1) __builtin_nontemporal_store(v + i, &o[i * plane + (long)y * p + x]);
            else o[i * plane + (long)y * p + x] = v + i;
        }
    }
}

__global__ __launch_bounds__(256) void flat(float4 *out, long n)
{
    long i = blockIdx.x * 256L + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = float4{1.f, 2.f, 3.f, (float)i};
}

int main()
{
    const int w = 1920, h = 1080, p = 1920, nimg = 64, NPL = 6;
    const long plane = (long)p * h, img_stride = plane * NPL;
    float *out, *in;
    hipMalloc(&out, img_stride * nimg * 4);
    hipMalloc(&in, plane * nimg * 4);
    hipMemset(in, 0, plane * nimg * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double bytes = (double)img_stride * nimg * 4;
    for (int rows : {182}) {
        for (int rd = 0; rd < 8; ++rd) {
            dim3 grid(w / 64, (h + rows - 1) / rows, nimg);
            float best = 1e9;
            for (int it = 0; it < 4; ++it) {
                hipEventRecord(e0, 0);
                switch (rd) {
                case 0: hipLaunchKernelGGL((strips<NPL, false, 0>), grid, dim3(64), 0, 0, out, in, p, h, rows, img_stride); break;
                case 1: hipLaunchKernelGGL((strips<NPL, true, 0>), grid, dim3(64), 0, 0, out, in, p, h, rows, img_stride); break;
                case 2: hipLaunchKernelGGL((strips<NPL, false, 1>), grid, dim3(64), 0, 0, out, in, p, h, rows, img_stride); break;
                case 3: hipLaunchKernelGGL((strips<5, true, 16>), grid, dim3(64), 0, 0, out, in, p, h, rows, img_stride); break;
                case 4: hipLaunchKernelGGL((strips<5, true, 4>), grid, dim3(64), 0, 0, out, in, p, h, rows, img_stride); break;
                case 5: hipLaunchKernelGGL((strips<NPL, true, 4>), grid, dim3(64), 0, 0, out, in, p, h, rows, img_stride); break;
                case 6: hipLaunchKernelGGL((strips<5, true, 0>), grid, dim3(64), 0, 0, out, in, p, h, rows, img_stride); break;
                case 7: hipLaunchKernelGGL((strips<5, true, 12>), grid, dim3(64), 0, 0, out, in, p, h, rows, img_stride); break;
                }
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (it && ms < best) best = ms;
            }
            static const char *nm[] = {"6pl store", "6pl store+load", "6pl nt-store", "5pl prefetched, paired-plane 8B stores", "5pl store+prefetched load", "6pl store+prefetched load", "5pl store+load", "5pl prefetched, rows interleaved"};
            const double by = bytes * ((rd == 6 || rd == 4 || rd == 7 || rd == 3) ? 5.0 / 6 : 1.0);
            printf("%-22s rows=%d: %.3f ms  write %.2f TB/s\n", nm[rd], rows, best, by / best / 1e9);
        }
    }
    {
        float best = 1e9;
        for (int it = 0; it < 4; ++it) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(flat, dim3(256 * 16), dim3(256), 0, 0, (float4 *)out, (long)(bytes / 16));
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (it && ms < best) best = ms;
        }
        printf("flat 16B/lane stores: %.3f ms  write %.2f TB/s\n", best, bytes / best / 1e9);
    }
    return 0;
}
