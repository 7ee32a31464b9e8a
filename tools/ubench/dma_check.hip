#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((address_space(3))) float lds_float;
__global__ __launch_bounds__(64) void k(const float *src, const unsigned char *srcb, unsigned *dst, int n)
{
    __shared__ float lds[8][128];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, n * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(srcb), 0, n, 0x00020000);
    const int lane = threadIdx.x;
    for (int r = 0; r < 8; ++r) lds[r][lane] = -1.f, lds[r][lane + 64] = -1.f;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(lds_float *)&lds[0][0];
    // row 0: plain. row 1: lanes >= 40 out of range by voffset. row 2: soffset beyond range. row 3: soffset "negative".
    // row 4: stride-2 gather. row 5: bytes. row 6: bytes, lanes >= 50 out of range. row 7: second half cells (offset 256)
    int voff[8]; int soff[8];
    voff[0] = lane * 4; soff[0] = 0;
    voff[1] = lane < 40 ? lane * 4 : (int)0xfffffff0u; soff[1] = 256;
    voff[2] = lane * 4; soff[2] = n * 4;
    voff[3] = lane * 4; soff[3] = -256;
    voff[4] = lane * 8; soff[4] = 512;
    voff[5] = lane; soff[5] = 3;
    voff[6] = lane < 50 ? lane : (int)0xfffffff0u; soff[6] = 64;
    voff[7] = lane * 4; soff[7] = 1024;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const unsigned dstaddr = base + r * 512 + (r == 7 ? 256 : 0);
        if (r == 5 || r == 6)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_ubyte %1, %2, %3 offen lds"
                         :: "s"(dstaddr), "v"(voff[r]), "s"(rb), "s"(soff[r]) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                         :: "s"(dstaddr), "v"(voff[r]), "s"(rs), "s"(soff[r]) : "memory", "m0");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int r = 0; r < 8; ++r) {
        dst[r * 128 + lane] = __float_as_uint(lds[r][lane]);
        dst[r * 128 + 64 + lane] = __float_as_uint(lds[r][lane + 64]);
    }
}
int main()
{
    const int n = 4096;
    std::vector<float> h(n);
    std::vector<unsigned char> hb(n);
    for (int i = 0; i < n; ++i) h[i] = 1000.f + i, hb[i] = (unsigned char)(i * 7 + 1);
    float *d; unsigned char *db; unsigned *o;
    hipMalloc(&d, n * 4); hipMalloc(&db, n); hipMalloc(&o, 8 * 128 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, db, o, n);
    std::vector<unsigned> r(8 * 128);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    auto f = [&](int row, int i) { float v; unsigned u = r[row * 128 + i]; memcpy(&v, &u, 4); return v; };
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        if (f(0, l) != h[l]) bad++, printf("row0 lane %d: %g\n", l, f(0, l));
        if (f(1, l) != (l < 40 ? h[64 + l] : 0.f)) bad++, printf("row1 lane %d: %g\n", l, f(1, l));
        if (f(2, l) != 0.f) bad++, printf("row2 lane %d: %g\n", l, f(2, l));
        if (f(3, l) != 0.f) bad++, printf("row3 lane %d: %g\n", l, f(3, l));
        if (f(4, l) != h[128 + 2 * l]) bad++, printf("row4 lane %d: %g\n", l, f(4, l));
        if (r[5 * 128 + l] != hb[3 + l]) bad++, printf("row5 lane %d: %08x want %02x\n", l, r[5 * 128 + l], hb[3 + l]);
        if (r[6 * 128 + l] != (l < 50 ? hb[64 + l] : 0u)) bad++, printf("row6 lane %d: %08x\n", l, r[6 * 128 + l]);
        if (f(7, 64 + l) != h[256 + l]) bad++, printf("row7 lane %d: %g\n", l, f(7, 64 + l));
        if (f(7, l) != -1.f || f(0, 64 + l) != -1.f) bad++, printf("untouched cells changed, lane %d\n", l);
        if (bad > 20) break;
    }
    printf(bad ? "FAILED %d\n" : "all LDS-DMA checks passed\n", bad);
    return bad != 0;
}
