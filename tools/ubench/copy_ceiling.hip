// Micro-benchmark: the device-to-device copy rate of this MI355X, over launch shapes, to find
// the ceiling a bandwidth-bound kernel can be priced against (read + write bytes per second).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy_k(f4 *__restrict__ out, const f4 *__restrict__ in, long n)
{
    const long stride = (long)gridDim.x * 256;
    long i = blockIdx.x * 256L + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            v[u] = NT ? __builtin_nontemporal_load(&in[i + u * stride]) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], &out[i + u * stride]);
            else out[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) out[i] = in[i];
}

// each block copies one contiguous chunk (no grid-stride): consecutive blocks, consecutive memory
template <int PER_THREAD>
__global__ __launch_bounds__(256) void copy_chunk(f4 *__restrict__ out, const f4 *__restrict__ in, long n)
{
    const long base = (long)blockIdx.x * 256 * PER_THREAD + threadIdx.x;
    f4 v[PER_THREAD];
#pragma unroll
    for (int u = 0; u < PER_THREAD; ++u)
        if (base + u * 256 < n) v[u] = in[base + u * 256];
#pragma unroll
    for (int u = 0; u < PER_THREAD; ++u)
        if (base + u * 256 < n) out[base + u * 256] = v[u];
}

static hipEvent_t e0, e1;
template <class F> static float best_ms(F f)
{
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
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

int main()
{
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (long gb : {1L, 4L}) {
        const long bytes = gb << 30, n = bytes / 16;
        f4 *a, *b;
        hipMalloc(&a, bytes);
        hipMalloc(&b, bytes);
        hipMemset(a, 1, bytes);
        hipMemset(b, 0, bytes);
        auto rep = [&](const char *name, float ms) {
            printf("%ld GiB  %-44s %8.3f ms  %5.2f TB/s (read + write)\n", gb, name, ms, 2.0 * bytes / ms / 1e9);
            fflush(stdout);
        };
        for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
            char nm[96];
            snprintf(nm, sizeof nm, "grid-stride x1, %d blocks", blocks);
            rep(nm, best_ms([&] { hipLaunchKernelGGL((copy_k<1, false>), dim3(blocks), dim3(256), 0, 0, b, a, n); }));
            snprintf(nm, sizeof nm, "grid-stride x4, %d blocks", blocks);
            rep(nm, best_ms([&] { hipLaunchKernelGGL((copy_k<4, false>), dim3(blocks), dim3(256), 0, 0, b, a, n); }));
            snprintf(nm, sizeof nm, "grid-stride x4 nontemporal, %d blocks", blocks);
            rep(nm, best_ms([&] { hipLaunchKernelGGL((copy_k<4, true>), dim3(blocks), dim3(256), 0, 0, b, a, n); }));
        }
        rep("block chunks, 4 f4 per thread", best_ms([&] {
                hipLaunchKernelGGL((copy_chunk<4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, b, a, n);
            }));
        rep("block chunks, 8 f4 per thread", best_ms([&] {
                hipLaunchKernelGGL((copy_chunk<8>), dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, 0, b, a, n);
            }));
        rep("hipMemcpyAsync device to device", best_ms([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }));
        hipFree(a);
        hipFree(b);
    }
    return 0;
}
