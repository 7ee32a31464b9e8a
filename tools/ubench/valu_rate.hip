// Micro-benchmark: issue cost of v_fma_f32 vs v_pk_fma_f32 (VGPR and SGPR tap
// operand) on gfx950, one to four waves per SIMD.  Prints cycles per
// wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, const float *tapsg, unsigned long long *cyc, int iters)
{
    float tapv = tapsg[threadIdx.x & 7];
    float ts0 = tapsg[0], ts1 = tapsg[1]; // uniform -> SGPRs
    f2 a[8];
    float s[16];
    for (int i = 0; i < 8; ++i) a[i] = f2{(float)i, (float)(threadIdx.x + i)};
    for (int i = 0; i < 16; ++i) s[i] = (float)(threadIdx.x + i);
    f2 v = f2{tapv, tapv + 1.f};
    double d[8];
    for (int i = 0; i < 8; ++i) d[i] = (double)(threadIdx.x + i) + 0.5;
    const double dv = (double)tapv;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { // 16 independent scalar fma, SGPR tap
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = __builtin_fmaf(ts0, tapv, s[i]);
        } else if (MODE == 1) { // 8 independent packed fma, SGPR pair tap
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(f2{ts0, ts1}, v, a[i]);
        } else if (MODE == 2) { // 8 packed fma, all VGPR
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(v, v, a[i]);
        } else if (MODE == 3) { // 8 packed fma, SGPR pair tap, splat of one scalar VGPR (op_sel)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(f2{ts0, ts1}, f2{tapv, tapv}, a[i]);
        } else if (MODE == 4) { // 8 independent double-precision fma
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = __builtin_fma((double)ts0, dv, d[i]);
        } else { // 4 independent double-precision divisions (correctly rounded: the compiler's expansion)
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = d[i + 4] / (d[i] + dv);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    for (int i = 0; i < 16; ++i) r += s[i];
    for (int i = 0; i < 8; ++i) r += (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int waves_per_simd, int per_iter)
{
    const int iters = 20000;
    const int blocks = 256 * waves_per_simd; // 256-thread blocks: one wave per SIMD each
    float *out, *taps;
    unsigned long long *cyc;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&taps, 64);
    hipMalloc(&cyc, 8 * blocks);
    float h[8] = {0.5f, 0.25f, 1, 2, 3, 4, 5, 6};
    hipMemcpy(taps, h, 32, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, taps, cyc, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, taps, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : c) avg += v;
    avg /= blocks;
    const double instr_per_simd = (double)iters * per_iter * waves_per_simd;
    printf("%-26s waves/SIMD %d: %8.3f ms  %.2f ns per wave-instr per SIMD (= %.2f cyc at 2.4 GHz); "
           "memtime ticks per instr per wave %.2f\n",
           name, waves_per_simd, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4,
           avg / ((double)iters * per_iter));
    hipFree(out); hipFree(taps); hipFree(cyc);
}
int main()
{
    for (int w : {1, 2, 3, 4, 6, 8}) {
        run<0>("v_fma_f32 sgpr tap", w, 16);
        run<1>("v_pk_fma_f32 sgpr pair", w, 8);
        run<2>("v_pk_fma_f32 vgpr", w, 8);
        run<3>("v_pk_fma_f32 sgpr x splat", w, 8);
        run<4>("v_fma_f64", w, 8);
        run<5>("f64 division (whole sequence)", w, 4);
    }
    return 0;
}
