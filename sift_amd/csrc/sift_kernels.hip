// gfx950 kernels of the SIFT path and the thin C-ABI layer over the HIP runtime.
//
// Build with -ffp-contract=off: results must be bit-identical to a CPU
// evaluation of the same arithmetic, so the only fused operations are the
// explicit fmaf()/fma() calls.
//
// Kernels (reference counterpart in Sift/src/excludes/cudaSiftD.cu):
//   pyramid_kernel          xblurMultiKernel :430-486 + yblurKernel :488-534 + resizeKernel
//                           :544-559 in one pass, plus the contrast clause of
//                           findExtremaKernel :293 as a pre-filter (the pixels whose centre
//                           value the reference reads through an aliased shared-memory
//                           cell :239-261 included)                          [the hot path]
//   pyramid_plane_kernel    the same pass with one blur plane per wave, for launches small enough
//                           that the longest wave decides (one image at a time)
//   pyramid2_kernel         the same pass with two columns per lane (opt-in)
//   verify_kernel           findExtremaKernel :216-305 on the queued pixels
//                           (+ hessianKernel :366-428 at candidates)
//   detect_kernel           findExtremaKernel over every pixel (fall-back, and the whole
//                           detector when queues are switched off)
//   keypoint_kernel         gradientKernel :306-364 (in a patch) + OrientationKernel :17-94
//                           + DescriptorKernel :96-214
//   verify_lowe_kernel, detect_lowe_kernel, upsample2x_kernel, quantize_kernel
//                           profile "lowe": no counterpart in the reference
//   row_blur_kernel, col_blur_dog_kernel, subtract/hessian/gradient/copy kernels
//                           the reference's stages one by one (blurOctave, computeDiffOfGauss,
//                           computeHessian, computeGradient, copyDeviceData cudaSiftH.cu:210-316)
#include <hip/hip_runtime.h>

#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "hip_layer.h"

namespace {

constexpr int TAPS = HL_TAPS;
constexpr int RADIUS = 9;

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>).
// Used where a register ring must be indexed statically (a "#pragma unroll"
// that the optimiser declines turns the ring into scratch memory).
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// blur tap k (0..18) of plane `plane`; the table stores the 10 distinct values
// of each symmetric kernel with the planes interleaved, so the taps of planes
// (2j, 2j+1) sit in one aligned scalar register pair
__device__ __forceinline__ float tap(const hl_taps &t, int plane, int k)
{
    return t.t[k <= 9 ? k : 18 - k][plane];
}
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float tap1(const hl_taps &t, int plane, int k) { return tap(t, plane, k); }
__device__ __forceinline__ float2v tap2(const hl_taps &t, int pair, int k)
{
    const int m = k <= 9 ? k : 18 - k;
    return float2v{t.t[m][2 * pair], t.t[m][2 * pair + 1]};
}
// d = tap * b + c on both halves of b and c, the tap being ONE half of the aligned scalar pair that
// holds planes (2j, 2j+1): the instruction's half selects broadcast it, so a tap costs no scalar
// register beyond its place in the table (the compiler, left to itself, copies every tap into a
// scalar pair of its own and runs out of scalar registers).  ZERO: c is 0.
template <int HI, bool ZERO>
__device__ __forceinline__ float2v pk_fma_tap(float2v tap_pair, float2v b, float2v c)
{
    float2v d;
    if (ZERO) {
        if (HI) asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(d) : "s"(tap_pair), "v"(b));
        else asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(d) : "s"(tap_pair), "v"(b));
    } else {
        if (HI) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(d) : "s"(tap_pair), "v"(b), "v"(c));
        else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "s"(tap_pair), "v"(b), "v"(c));
    }
    return d;
}
// two IEEE fused multiply-adds in one instruction (v_pk_fma_f32)
__device__ __forceinline__ float2v pk_fma(float2v a, float2v b, float2v c)
{
    return __builtin_elementwise_fma(a, b, c);
}

// ---------------------------------------------------------------------------
// Row pass.  One wave per image row segment of 256 pixels: the segment plus a
// 12-pixel apron on each side is staged once in LDS as 70 float4 (zero outside
// the image), each lane then holds 28 consecutive inputs in registers and
// produces 4 consecutive pixels of every plane: out_i[x] = chain over k
// ascending of fmaf(c_i[k], src[x-9+k], sum), sum0 = 0.
// ---------------------------------------------------------------------------
template <int NP, bool ALIGNED>
__global__ __launch_bounds__(256) void row_blur_kernel(const float *__restrict__ src, int src_pitch,
                                                       long src_stride, float *__restrict__ tmp,
                                                       long tmp_stride, int w, int p, int h,
                                                       hl_taps taps)
{
    __shared__ float4 lds[4][72];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int y = blockIdx.y * 4 + wave;
    const int x0 = blockIdx.x * 256;
    const bool row_ok = y < h;
    const float *srow = src + (long)blockIdx.z * src_stride + (long)(row_ok ? y : 0) * src_pitch;

    for (int j = lane; j < 70; j += 64) {
        const int xs = x0 - 12 + 4 * j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row_ok && xs >= 0 && xs < w) {
            if (ALIGNED) {
                v = *reinterpret_cast<const float4 *>(srow + xs);
                if (xs + 1 >= w) v.y = 0.f;
                if (xs + 2 >= w) v.z = 0.f;
                if (xs + 3 >= w) v.w = 0.f;
            } else {
                v.x = srow[xs];
                if (xs + 1 < w) v.y = srow[xs + 1];
                if (xs + 2 < w) v.z = srow[xs + 2];
                if (xs + 3 < w) v.w = srow[xs + 3];
            }
        }
        lds[wave][j] = v;
    }
    __syncthreads();

    const int x = x0 + 4 * lane;
    if (!row_ok || x >= w) return;

    float in[28];
#pragma unroll
    for (int m = 0; m < 7; ++m) {
        const float4 v = lds[wave][lane + m];
        in[4 * m + 0] = v.x;
        in[4 * m + 1] = v.y;
        in[4 * m + 2] = v.z;
        in[4 * m + 3] = v.w;
    }
    float *orow = tmp + (long)blockIdx.z * tmp_stride + (long)y * p + x;
    const long plane = (long)p * h;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < TAPS; ++k) sum = __builtin_fmaf(tap(taps, i, k), in[q + 3 + k], sum);
            o[q] = sum;
        }
        *reinterpret_cast<float4 *>(orow + i * plane) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---------------------------------------------------------------------------
// Column pass with fused DoG and down-sampling.  One lane per column, a wave
// covers 64 consecutive columns (256-byte coalesced rows) and walks down a
// strip of rows.  Each input row r is loaded once and scattered into the 19
// partial sums it belongs to: out[yo] accumulates fmaf(c[k], in[yo-9+k], .)
// with k ascending as the input row ascends, which is exactly the reference's
// chain order.  The live sums of a plane sit in a register ring of 20 slots
// (19 in use, one being recycled); the row loop is unrolled by 20 so every
// slot index is static.  Input rows are fetched COL_PF rows ahead of their use
// to keep enough bytes in flight per wave.
// STAGE = true: store all S+3 blur planes and nothing else (blurOctave entry);
// STAGE = false: store blur planes 1..S, the S+2 DoG planes and the even
// pixels of plane S as the next octave's base.
// ---------------------------------------------------------------------------
constexpr int COL_RING = 20;
// waves per SIMD the fused pyramid kernel is compiled for (2 and 3 scales).  Measured on MI355X at
// 2 scales: 2 waves 19.3 us per 1080p image, 3 waves 16.6, 4 waves (128 registers) 17.3.
#ifndef PYR_WAVES_S2
#define PYR_WAVES_S2 3
#endif
#ifndef PYR_WAVES_S3
#define PYR_WAVES_S3 3
#endif
constexpr int COL_PF = 4;

template <int S, bool STAGE>
__global__ __launch_bounds__(64) void col_blur_dog_kernel(
    const float *__restrict__ tmp, long tmp_stride, float *__restrict__ blur, long blur_stride,
    float *__restrict__ dog, long dog_stride, float *__restrict__ next_base, long next_stride,
    int next_p, int w, int p, int h, int rows, hl_taps taps)
{
    constexpr int NP = S + 3;
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= w) return;
    const int y0 = blockIdx.y * rows;
    const int y1 = min(y0 + rows, h);
    const int nin = (y1 - y0) + 2 * RADIUS;
    const long plane = (long)p * h;
    const int z = blockIdx.z;
    const float *tin = tmp + z * tmp_stride + x;
    float *bl = blur + z * blur_stride + x;
    float *dg = STAGE ? nullptr : dog + z * dog_stride + x;
    float *nb = (!STAGE && next_base && !(x & 1)) ? next_base + z * next_stride + (x >> 1) : nullptr;

    float acc[NP][COL_RING];
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int k = 0; k < COL_RING; ++k) acc[i][k] = 0.f;

    float pre[COL_PF][NP];
#pragma unroll
    for (int q = 0; q < COL_PF; ++q) {
        const int yi = y0 - RADIUS + q;
        const bool ok = yi >= 0 && yi < h; // q < nin always: nin >= 19
#pragma unroll
        for (int i = 0; i < NP; ++i) pre[q][i] = ok ? tin[i * plane + (long)yi * p] : 0.f;
    }

    for (int rb = 0; rb < nin; rb += COL_RING) {
        static_for<COL_RING>([&](auto rr_c) __attribute__((always_inline)) {
            constexpr int rr = decltype(rr_c)::value;
            const int r = rb + rr;
            float v[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) v[i] = pre[rr % COL_PF][i];
            {
                const int rn = r + COL_PF;
                const int yn = y0 - RADIUS + rn;
                const bool ok = rn < nin && yn >= 0 && yn < h;
#pragma unroll
                for (int i = 0; i < NP; ++i) pre[rr % COL_PF][i] = ok ? tin[i * plane + (long)yn * p] : 0.f;
            }
            if (r < nin) {
#pragma unroll
                for (int i = 0; i < NP; ++i) {
#pragma unroll
                    for (int k = 0; k < TAPS; ++k) {
                        const int slot = (rr - k + COL_RING) % COL_RING;
                        acc[i][slot] = __builtin_fmaf(tap(taps, i, k), v[i], k == 0 ? 0.f : acc[i][slot]);
                    }
                }
                if (r >= 2 * RADIUS) { // output row yo = y0 + r - 18 is complete; yo < y1 because r < nin
                    const int yo = y0 + r - 2 * RADIUS;
                    const int e = (rr - 2 * RADIUS + COL_RING) % COL_RING;
                    float b[NP];
#pragma unroll
                    for (int i = 0; i < NP; ++i) b[i] = acc[i][e];
                    const long off = (long)yo * p;
                    if (STAGE) {
#pragma unroll
                        for (int i = 0; i < NP; ++i) bl[i * plane + off] = b[i];
                    } else {
#pragma unroll
                        for (int i = 1; i <= S; ++i) bl[(i - 1) * plane + off] = b[i];
#pragma unroll
                        for (int i = 0; i < NP - 1; ++i) dg[i * plane + off] = b[i] - b[i + 1];
                        if (nb && !(yo & 1)) nb[(long)(yo >> 1) * next_p] = b[S];
                    }
                }
            }
        });
    }
}

// ---------------------------------------------------------------------------
// Fused pyramid pass: row blur, column blur and DoG in one kernel, so the S+3
// row-blurred planes never travel to HBM and back.
// One wave per block owns 64 columns and walks down a strip of rows.  Per
// input row it stages the 88 source pixels [x0-12, x0+76) in LDS (zero outside
// the image), every lane reads its own 19-pixel window and forms the S+3
// row-blurred values of its column (xblurMultiKernel's chain), which are then
// scattered into the column ring exactly as col_blur_dog_kernel does
// (yblurKernel's chain).
//
// Memory pipeline.  On gfx9-family hardware loads and stores retire through
// one in-order counter.  The source rows travel from HBM straight into LDS
// (buffer_load ... lds: no register is held while a row is in flight), FZ_PF
// rows ahead of their use, and every row step issues the same vector-memory
// operations unconditionally: two source fetches and S+3 stores.  The wait in
// front of a row's first LDS read is therefore a constant: the number of
// operations issued since that row's fetch.  Pixels outside the image come
// back as zeros from the buffer's range check: a column outside [0, w) has an
// out-of-range lane offset, a row outside [0, h) an out-of-range row offset.
// The stores of rows that complete no output row of this strip (the first 18
// input rows) carry an out-of-range offset too and are dropped by the hardware;
// a wave leaves the row loop at its strip's last row.  All 64 lanes
// store: columns [w, pitch) of the destination planes are scratch.
//
// SRC_MODE 0: source is a float image (any pitch or alignment)
//          1: source is an 8-bit image (src points at bytes; pitch and stride in
//             bytes); a pixel becomes the float of the same value when staged
//          2: source is the previous octave's blur plane S, read at even rows
//             and even columns (resizeKernel cudaSiftD.cu:544-559 folded into
//             the fetch; the down-sampled base is never materialised)
// ---------------------------------------------------------------------------
#ifndef PYR_FZ_PF
#define PYR_FZ_PF 3
#endif
constexpr int FZ_PF = PYR_FZ_PF;    // source rows in flight per wave (measured: 3 and 4 within 0.5 %, 9 is 5 % slower)
constexpr int FZ_SLOTS = FZ_PF == 2 ? 4 : FZ_PF + 1; // LDS row buffers: the rows in flight and the row being read (a divisor of 20)
static_assert(20 % FZ_SLOTS == 0, "the row buffers must be used in whole turns per row chunk");
constexpr int FZ_STAGE = 88; // floats staged per row
// Cache policy of the plane stores: non-temporal, device scope (the "nt" and "sc1" bits).  The planes
// stream out and are read again only by later kernels; kept out of the way, they leave L2 to the
// source lines neighbouring column tiles share.  Measured: the pass 1.3 % faster, and the verifier,
// which runs after it, 20 % faster.
#ifndef FZ_STORE_POLICY
#define FZ_STORE_POLICY 18
#endif
#ifndef PYR_PQ_CAP
#define PYR_PQ_CAP 1024
#endif
constexpr int PQ_CAP = PYR_PQ_CAP; // contrast candidates one wave can hold per scale between flushes
constexpr unsigned FZ_DROP = 0xfffffff0u; // per-lane buffer offset past every plane: the store is dropped

// launch geometry of the reference's detector (cudaSiftH.cu:64-89): blocks of
// 96x32 pixels in three 32-wide tiles, shared-memory tiles of 101x34 floats
// (98 columns written) spaced 3168 floats apart
constexpr int XT_DATAX = 96;
constexpr int XT_BDIM = 32;
constexpr int XT_SDIMX = 101;
constexpr int XT_WRITTEN = 98;
constexpr int XT_PLANE_OFF = 3168;

template <int S, int SRC_MODE, bool PLAIN>
__global__ __launch_bounds__(64, (S == 2 ? PYR_WAVES_S2 : PYR_WAVES_S3)) void pyramid_kernel(
    const float *__restrict__ src, int src_w, int src_h, int src_pitch, long src_stride,
    float *__restrict__ blur, long blur_stride, int w, int p, int h, int rows, hl_taps taps, float contrast, unsigned *__restrict__ queue,
    long queue_stride, unsigned *__restrict__ qcount, long qcount_stride, int qcap, int group0, int vgx, int vgy, int vgz)
{
    constexpr int NP = S + 3;
    constexpr bool plain = PLAIN; // the detector without tiles ("lowe" profile)
    __shared__ float lds[FZ_SLOTS][128];
    // contrast candidates of this strip: pixels whose centre-plane DoG exceeds
    // the detector's threshold (cudaSiftD.cu:293), kept per scale in LDS and
    // flushed to the global queue once, after the last row
    __shared__ unsigned lq[S][PQ_CAP];
    // plain detector only: the centre DoG values of the previous output row
    __shared__ float d_up[PLAIN ? S : 1][64];
    const int lane = threadIdx.x;
    // Workgroups are handed to the 8 XCDs round-robin, each XCD with its own
    // L2.  The remap gives every XCD one contiguous run of (column, strip,
    // image) tiles, so the waves that share source lines (neighbouring columns)
    // fetch them through the same L2.  Bijective for any grid size.
    // (vgx, vgy, vgz) is the grid of tiles.  Normally it is the launch grid.  Experiment
    // PYR_EXP_PERSIST: the launch is only as many blocks as the chip holds at once (a multiple
    // of 8, so a block stays on its XCD's run of tiles) and each block walks the tile list.
    const unsigned nwg = (unsigned)vgx * (unsigned)vgy * (unsigned)vgz;
#ifdef PYR_EXP_PERSIST
    for (unsigned orig = blockIdx.x; orig < nwg; orig += gridDim.x) {
#else
    {
        const unsigned orig = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
#endif
    int tile_x, tile_y, tile_z;
    {
        const unsigned gx = (unsigned)vgx, gy = (unsigned)vgy;
        const unsigned q = nwg / 8, r = nwg % 8, xcd = orig % 8;
        const unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
        tile_x = (int)(t % gx);
        tile_y = (int)((t / gx) % gy);
        tile_z = (int)(t / (gx * gy));
    }
    const int x0 = tile_x * 64;
    const int y0 = tile_y * rows;
    const int y1 = min(y0 + rows, h);
    const int nin = (y1 - y0) + 2 * RADIUS;
    const long plane = (long)p * h;
    const int z = tile_z;
    constexpr int ES = SRC_MODE == 1 ? 1 : 4; // bytes per source element
    // buffer descriptors (wave-uniform): 32-bit per-lane offsets in one VGPR and
    // per-row / per-plane offsets in scalar registers replace 64-bit per-lane
    // address arithmetic
#ifdef PYR_EXP_SRC0 // experiment (results are wrong): every image reads the first image's source, so the reads hit in the caches
    const int zs = 0;
#else
    const int zs = z;
#endif
    const __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(src) + zs * src_stride * ES), 0,
        (int)(unsigned)min((long)src_pitch * src_h * ES, 0xffffffffL), 0x00020000);
    const __amdgpu_buffer_rsrc_t bl_rs = __builtin_amdgcn_make_buffer_rsrc(
        blur + z * blur_stride, 0, (int)(unsigned)min(plane * NP * 4, (long)FZ_DROP), 0x00020000);
    // destination of the next completed output row; advanced row by row (a
    // loop-carried pointer: offsets formed from the unrolled row index would be
    // hoisted out of the loop as dozens of live scalars)
    // byte offsets are unsigned 32-bit (the launcher checks the planes fit)
    const int out_lane = (x0 + lane) * 4;         // this lane's column
    unsigned out_row = (unsigned)y0 * p * 4u;     // the next completed output row
    const unsigned plane_b = (unsigned)plane * 4u;
    // detector pre-filter: columns the reference can accept at all
    // (cudaSiftD.cu:289: 8 < x < w-8; the middle tile of a 96-wide block never fires)
    const bool pre = queue != nullptr;
    const int xq = x0 + lane;
    const int xq_tile = (xq % XT_DATAX) / XT_BDIM;
    // plain: every column and row inside the border qualifies (the detector of the
    // "lowe" profile has no tiles)
    const bool pre_lane = xq > 8 && xq < w - 8 && (plain || xq_tile != 1);
    const bool pre_lane0 = pre_lane && xq_tile == 0; // rows with y mod 32 = 30: the right tile reads an aliased centre cell
    // the block is one wave: the fill of each list is a wave-uniform register, no
    // LDS atomic is needed to reserve slots
    unsigned lq_n[S];
#pragma unroll
    for (int i = 0; i < S; ++i) lq_n[i] = 0;
    // A queue that has already overflowed is lost (the dense detector will redo its
    // group): strips that start later stop feeding it.
    float thr[S];
#pragma unroll
    for (int i = 0; i < S; ++i)
        thr[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(
            (pre && qcount[z * qcount_stride + group0 + i] > (unsigned)qcap) ? __builtin_inff() : contrast))); // wave-uniform: a scalar register

    // ---- source fetch: one row from HBM into an LDS row buffer ----
    // lane j fetches staged float j and, for j < 24, staged float 64+j
    constexpr int SX = SRC_MODE == 2 ? 2 : 1; // column and row step in the source
    int col_off[2];
    {
        const int xa = x0 - 12 + lane, xb = xa + 64;
        const bool oka = xa >= 0 && xa < w, okb = lane < FZ_STAGE - 64 && xb >= 0 && xb < w;
        col_off[0] = oka ? SX * xa * ES : (int)FZ_DROP;
        col_off[1] = okb ? SX * xb * ES : (int)FZ_DROP; // (lanes >= 24 fill unused cells with zeros)
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float *)&lds[0][0];
    int yi_next = y0 - RADIUS; // rows are fetched strictly in order
    // The fetches are inline assembly so that the compiler neither tracks them (it would
    // wait for the youngest of them in front of every LDS read) nor counts them; extra
    // operations the compiler adds only make the waits below more conservative.
    // (m0 is named as clobbered so that the compiler keeps no value of its own in it across the
    // statements; it answers with a warning about reserved registers, which is switched off here)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    auto fetch = [&](const int slot) __attribute__((always_inline)) {
        const int yi = yi_next++;
        const unsigned row = (yi >= 0 && yi < h) ? (unsigned)(SX * yi) * (unsigned)src_pitch * (unsigned)ES : FZ_DROP;
        const unsigned dst = lds_base + (unsigned)slot * 512u;
        if (SRC_MODE == 1) { // a byte arrives as the integer of its value
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_ubyte %1, %2, %3 offen lds" ::"s"(dst), "v"(col_off[0]), "s"(src_rs), "s"(row) : "memory", "m0");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_ubyte %1, %2, %3 offen lds" ::"s"(dst + 256u), "v"(col_off[1]), "s"(src_rs), "s"(row) : "memory", "m0");
        } else {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(dst), "v"(col_off[0]), "s"(src_rs), "s"(row) : "memory", "m0");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(dst + 256u), "v"(col_off[1]), "s"(src_rs), "s"(row) : "memory", "m0");
        }
    };
#pragma clang diagnostic pop
    // vector-memory operations issued between a row's fetch and its first read: the NP
    // stores of the step that fetched it, FZ_PF - 1 whole steps, and this step's fetch
    constexpr int FZ_WAIT = NP + (NP + 2) * (FZ_PF - 1) + 2;
    static_assert(FZ_WAIT < 64, "the counter holds 63");

    constexpr int NPAIR = NP / 2;
    constexpr bool ODD = (NP & 1) != 0;
    float2v accp[NPAIR][COL_RING];
    float accs[COL_RING];
#pragma unroll
    for (int k = 0; k < COL_RING; ++k) {
#pragma unroll
        for (int j = 0; j < NPAIR; ++j) accp[j][k] = float2v{0.f, 0.f};
        accs[k] = 0.f;
    }

    // lanes the detector pre-filter may queue, as wave masks
    const unsigned long long pre_mask = __ballot(pre_lane), pre_mask0 = __ballot(pre_lane0);

    // appends the lanes of `hits` to the wave's list of scale i as pixels of row y
    // (dx: the queued pixel lies dx columns from the lane's own, see the aliased rows below)
    auto push = [&](const int i, const unsigned long long hits, const int y, const int dx = 0) __attribute__((always_inline)) {
        if (hits) {
            // lanes below this one among the hits: the mask-bit-count instruction needs no per-lane mask register
            const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(hits >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hits, 0u));
            const unsigned slot = lq_n[i] + below;
            if (((hits >> lane) & 1ull) && slot < PQ_CAP) lq[i][slot] = (unsigned)(x0 + lane + dx) | ((unsigned)y << 16);
            lq_n[i] += (unsigned)__popcll(hits);
        }
    };
    // Pixels whose centre value the reference reads through an aliased shared-memory cell.  For
    // y mod 32 = 30 or 31 the centre cell of a pixel has been overwritten by the next DoG plane's
    // tile (cudaSiftD.cu:239-261), i.e. it is DoG_{i+2} at a shifted position:
    //   y mod 32 = 30, x mod 96 >= 64:  DoG_{i+2}[x-37, y-31]
    //   y mod 32 = 31, x mod 96 <  32:  DoG_{i+2}[x+64, y-32]
    //   y mod 32 = 31, x mod 96 >= 64:  DoG_{i+2}[x-37, y-31]
    // Seen from the source side: rows 32b-1 and 32b of plane i+2 decide the contrast test of
    // pixels 31 or 32 rows below and 37 columns to the right or 64 to the left.  The wave that
    // completes such a row queues those targets.  alias_a / alias_b: lanes whose right / left
    // target is a column the detector can accept.
    const int xta = xq + 37, xtb = xq - 64;
    const unsigned long long alias_a = __ballot(!plain && xq < w && (xta % XT_DATAX) >= 2 * XT_BDIM && xta > 8 && xta < w - 8);
    const unsigned long long alias_b = __ballot(!plain && xq < w && xtb > 8 && (xtb % XT_DATAX) < XT_BDIM && xtb < w - 8);
    unsigned pend = 0u; // plain detector: this lane's candidates of the row above (see below)

    // One chunk of 20 input rows, straight line.  Input row r of the strip
    // completes output row y0 + r - 18.
    auto chunk = [&](const int rb) __attribute__((always_inline)) {
        static_for<COL_RING>([&](auto rr_c) __attribute__((always_inline)) {
            constexpr int rr = decltype(rr_c)::value;
            const int r = rb + rr;
            // keep each row's instructions together
            __builtin_amdgcn_sched_barrier(0);
            if (r >= nin) return; // wave-uniform: the strip ended inside this chunk
            // row r + FZ_PF goes to the buffer row r - 1 was read from
            fetch((rr + FZ_PF) % FZ_SLOTS);
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FZ_WAIT) : "memory"); // row r has landed
            constexpr int slot = rr % FZ_SLOTS;
            if (SRC_MODE == 1) { // integers to floats, in place (LDS operations of a wave stay in order)
                const unsigned a0 = __float_as_uint(lds[slot][lane]), a1 = __float_as_uint(lds[slot][lane + 64]);
                lds[slot][lane] = (float)a0;
                lds[slot][lane + 64] = (float)a1;
                // other lanes read these cells: keep the compiler from moving their reads up
                // (the hardware runs a wave's LDS operations in order)
                asm volatile("" ::: "memory");
            }
            float win[TAPS + 1]; // the 20th value only completes the last register pair
            // the whole window at once: ten two-float reads in flight, the chains start as the first arrive
#pragma unroll
            for (int k = 0; k < TAPS + 1; ++k) win[k] = lds[slot][lane + 3 + k];
            float2v vp[NPAIR];
            float vs = 0.f;
#pragma unroll
            for (int j = 0; j < NPAIR; ++j) vp[j] = float2v{0.f, 0.f};
#pragma unroll
            for (int k = 0; k < TAPS; ++k) {
                const float2v wk = float2v{win[k], win[k]};
#pragma unroll
                for (int j = 0; j < NPAIR; ++j) vp[j] = pk_fma(tap2(taps, j, k), wk, vp[j]);
                if (ODD) vs = __builtin_fmaf(tap1(taps, NP - 1, k), win[k], vs);
            }
#ifndef PYR_EXP_NO_COLUMN_FMA
#pragma unroll
            for (int k = 0; k < TAPS; ++k) {
                const int slot = (rr - k + COL_RING) % COL_RING;
#pragma unroll
                for (int j = 0; j < NPAIR; ++j)
                    accp[j][slot] = pk_fma(tap2(taps, j, k), vp[j], k == 0 ? float2v{0.f, 0.f} : accp[j][slot]);
                if (ODD) accs[slot] = __builtin_fmaf(tap1(taps, NP - 1, k), vs, k == 0 ? 0.f : accs[slot]);
            }
#else // experiment: what the kernel costs without its column arithmetic
            {
                constexpr int slot = (rr - 18 + COL_RING) % COL_RING;
#pragma unroll
                for (int j = 0; j < NPAIR; ++j) accp[j][slot] = vp[j];
                if (ODD) accs[slot] = vs;
            }
#endif
            // output row y0 + r - 18 is complete if it belongs to this strip
            const bool valid = r >= 2 * RADIUS && r < nin; // wave-uniform
#ifdef PYR_EXP_NO_STORES // experiment: what the kernel costs without its write traffic
            const int vo = out_lane | (int)FZ_DROP;
#else
            const int vo = out_lane | (valid ? 0 : (int)FZ_DROP);
#endif
            constexpr int e = (rr - 2 * RADIUS + COL_RING) % COL_RING;
            float b[NP];
#pragma unroll
            for (int j = 0; j < NPAIR; ++j) {
                b[2 * j] = accp[j][e].x;
                b[2 * j + 1] = accp[j][e].y;
            }
            if (ODD) b[NP - 1] = accs[e];
            // a wave that has a row to send goes first until its next fetch is out as well
            // (measured: 0.8 % on the whole pass)
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < NP; ++i)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(b[i]), bl_rs, vo,
                                                      (int)(out_row + i * plane_b), FZ_STORE_POLICY);
            out_row += valid ? (unsigned)p * 4u : 0u;
            // detector pre-filter: queue the pixels that pass the contrast test
            // (rare).  Pixels whose centre cell the reference reads through an
            // aliased shared-memory cell (y mod 32 = 31, and the right tile at
            // y mod 32 = 30) are queued from their source rows, below.  No global memory
            // is touched here, so the counted waits of the row loop stay exact.
            const int yo = y0 + r - 2 * RADIUS;
            const bool row_ok = pre && valid && yo > 8 && yo < h - 8 && (plain || (yo & 31) < 31); // wave-uniform
            if constexpr (!PLAIN) {
                if (row_ok) {
                    const unsigned long long lanes = (yo & 31) == 30 ? pre_mask0 : pre_mask;
#pragma unroll
                    for (int i = 0; i < S; ++i)
                        push(i, __ballot(fabsf(__fsub_rn(b[i + 1], b[i + 2])) > thr[i]) & lanes, yo);
                }
                const int ym = yo & 31;
                if (pre && valid && (ym == 31 || ym == 0)) { // wave-uniform: a source row of the aliased cells
                    const int yt = yo + 31;                 // target row with y mod 32 = 30 (from row 32b-1) or 31 (from row 32b)
                    const bool yt_ok = yt > 8 && yt < h - 8, yt1_ok = yt + 1 > 8 && yt + 1 < h - 8;
#pragma unroll
                    for (int i = 0; i < S; ++i) {
                        const unsigned long long big = __ballot(fabsf(__fsub_rn(b[i + 2], b[i + 3])) > thr[i]);
                        if (yt_ok) push(i, big & alias_a, yt, 37);
                        if (ym == 31 && yt1_ok) push(i, big & alias_b, yt + 1, -64);
                    }
                }
            } else if (pre) { // wave-uniform
                // The plain detector wants a strict extremum over 26 neighbours.  Being one
                // among the four in-plane neighbours and the two at the same pixel of the
                // neighbouring DoG planes is necessary and leaves a few percent of the
                // contrast candidates.  Left and right come from the lanes beside (lanes
                // 0 and 63 have only one in this wave and pass), the row above from LDS, and
                // the row below decides one row later: `pend` holds, per lane, which scales
                // wait as a maximum (bit i) or a minimum (bit S + i).
                float dn[S];
#pragma unroll
                for (int i = 0; i < S; ++i) dn[i] = __fsub_rn(b[i + 1], b[i + 2]);
                if (__ballot(pend != 0u)) { // rare
#pragma unroll
                    for (int i = 0; i < S; ++i) {
                        const float dp = d_up[i][lane];
                        const bool keep = (((pend >> i) & 1u) && (!valid || dp > dn[i])) ||
                                          (((pend >> (S + i)) & 1u) && (!valid || dp < dn[i]));
                        push(i, __ballot(keep), yo - 1);
                    }
                }
                pend = 0u;
                if (row_ok) {
#pragma unroll
                    for (int i = 0; i < S; ++i) {
                        const float d = dn[i];
                        const bool big = pre_lane && fabsf(d) > thr[i];
                        if (__ballot(big)) { // wave-uniform: some lane passes the contrast test
                            const float dl = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x138, 0xf, 0xf, false)); // lane - 1
                            const float dr = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x130, 0xf, 0xf, false)); // lane + 1
                            const bool edge = lane == 0 || lane == 63;
                            bool is_max = big && ((d > dl && d > dr) || edge);
                            bool is_min = big && ((d < dl && d < dr) || edge);
                            // the same pixel one DoG plane down and up
                            const float d_lo = __fsub_rn(b[i], b[i + 1]), d_hi = __fsub_rn(b[i + 2], b[i + 3]);
                            is_max = is_max && d > d_lo && d > d_hi;
                            is_min = is_min && d < d_lo && d < d_hi;
                            if (r > 2 * RADIUS) { // the row above was completed by this wave
                                const float du = d_up[i][lane];
                                is_max = is_max && d > du;
                                is_min = is_min && d < du;
                            }
                            pend |= (is_max ? 1u << i : 0u) | (is_min ? 1u << (S + i) : 0u);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < S; ++i) d_up[i][lane] = dn[i];
            }
        });
    };
    // Moves the strip's candidates so far to the global queue: one global atomic
    // per scale.  A list that overflowed poisons the count: the dense kernel then
    // redoes this (image, scale) from the DoG planes.
    auto flush = [&]() __attribute__((always_inline)) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const unsigned n = lq_n[i];
            if (n == 0) continue;
            unsigned base = 0;
            // a list that overflowed marks the count with its top bit: however many waves do
            // so, the count stays above every capacity (sums of poison values could wrap)
            if (lane == 0)
                base = n > PQ_CAP ? (atomicOr(&qcount[z * qcount_stride + group0 + i], 0x80000000u) | 0x80000000u)
                                  : atomicAdd(&qcount[z * qcount_stride + group0 + i], n);
            base = __builtin_amdgcn_readfirstlane(__shfl(base, 0));
            if (n <= PQ_CAP)
                for (unsigned e = lane; e < n; e += 64)
                    if (base + e < (unsigned)qcap) queue[z * queue_stride + (long)i * qcap + base + e] = lq[i][e];
            // the queue is full: its group goes to the dense detector, stop feeding it
            if (base + n > (unsigned)qcap) thr[i] = __builtin_inff();
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < S; ++i) lq_n[i] = 0;
    };
    // a row adds at most 64 entries per scale, a chunk of 20 rows rarely more
    // than a few: the lists are emptied between chunks once half full
    auto flush_if_filling = [&]() __attribute__((always_inline)) {
        if (!pre) return;
        unsigned fill = 0;
#pragma unroll
        for (int i = 0; i < S; ++i) fill = max(fill, lq_n[i]);
        if (fill > PQ_CAP / 2) flush();
    };
    // prologue: the first FZ_PF rows, waited for in full, so that the constant wait of the
    // row loop holds from its first step on (it assumes FZ_PF steps of operations behind a fetch)
#pragma unroll
    for (int q = 0; q < FZ_PF; ++q) fetch(q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int rb = 0;
    for (; rb < nin; rb += COL_RING) {
        chunk(rb);
        flush_if_filling();
    }
    // rows fetched beyond the strip's last are still on their way into this block's LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (PLAIN && pre && __ballot(pend != 0u)) { // the strip's last row has no row below in this wave
#pragma unroll
        for (int i = 0; i < S; ++i) push(i, __ballot(((pend >> i) | (pend >> (S + i))) & 1u), y1 - 1);
    }
    if (pre) flush();
    } // tile
}


// ---------------------------------------------------------------------------
// The same pass with the planes dealt over waves: a wave owns 64 columns of ONE blur plane.
// For launches the chip holds in a single round (one image at a time through extractSift,
// the small octaves of a few images), where a call's time is the sum of its launches'
// longest waves: a wave of pyramid_kernel needs 0.45 us per row step when it has its SIMD to
// itself (190 FMAs in chains of 19, its window's LDS reads, the counted wait), this one a
// fraction of that (38 FMAs), and S+3 times as many waves spread over the SIMDs.  The source
// row is staged once per wave (S+3 times per tile: the copies after the first come from L2),
// so this is not the kernel for a batch.  The window of the next row is read from LDS before
// the chains of the current one run, which hides the LDS latency a lone wave cannot hide.
// No detector pre-filter: launch sets this small are detected by the dense detector alone.
// Results are bit-identical to pyramid_kernel's: the same chains in the same order.
// ---------------------------------------------------------------------------
#ifndef PYRP_FZ_PF
#define PYRP_FZ_PF 4
#endif
constexpr int FZP_PF = PYRP_FZ_PF;       // source rows in flight per wave
constexpr int FZP_SLOTS = FZP_PF + 1;    // LDS row buffers (a divisor of 20)
static_assert(20 % FZP_SLOTS == 0 && FZP_PF >= 2, "the row buffers must be used in whole turns per row chunk");
#ifndef PYRP_STORE_POLICY
#define PYRP_STORE_POLICY 0 // the planes of a small launch set are read again at once: let the caches keep them
#endif

template <int SRC_MODE>
__global__ __launch_bounds__(64) void pyramid_plane_kernel(
    const float *__restrict__ src, int src_w, int src_h, int src_pitch, long src_stride,
    float *__restrict__ blur, long blur_stride, int w, int p, int h, int rows, int np, hl_taps taps,
    int vgx, int vgy, int vgz)
{
    __shared__ float lds[FZP_SLOTS][128];
    const int lane = threadIdx.x;
    // XCD-aware remap as in pyramid_kernel; the plane is the fastest index, so the np waves that
    // stage the same source rows run on one XCD (one L2) side by side
    const unsigned nwg = (unsigned)vgx * (unsigned)vgy * (unsigned)vgz * (unsigned)np;
    const unsigned orig = blockIdx.x;
    int tile_x, tile_y, tile_z, pl;
    {
        const unsigned gx = (unsigned)vgx, gy = (unsigned)vgy, n = (unsigned)np;
        const unsigned q = nwg / 8, r = nwg % 8, xcd = orig % 8;
        const unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
        pl = (int)(t % n);
        tile_x = (int)((t / n) % gx);
        tile_y = (int)((t / (n * gx)) % gy);
        tile_z = (int)(t / (n * gx * gy));
    }
    const int x0 = tile_x * 64;
    const int y0 = tile_y * rows;
    const int y1 = min(y0 + rows, h);
    const int nin = (y1 - y0) + 2 * RADIUS;
    const long plane = (long)p * h;
    const int z = tile_z;
    constexpr int ES = SRC_MODE == 1 ? 1 : 4;
    const __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(src) + z * src_stride * ES), 0,
        (int)(unsigned)min((long)src_pitch * src_h * ES, 0xffffffffL), 0x00020000);
    const __amdgpu_buffer_rsrc_t bl_rs = __builtin_amdgcn_make_buffer_rsrc(
        blur + z * blur_stride, 0, (int)(unsigned)min(plane * np * 4, (long)FZ_DROP), 0x00020000);
    const int out_lane = (x0 + lane) * 4;
    unsigned out_row = (unsigned)y0 * p * 4u + (unsigned)pl * (unsigned)plane * 4u; // the next completed row of this wave's plane
    // this plane's ten distinct taps (wave-uniform: scalar registers)
    float c[HL_HALF_TAPS];
#pragma unroll
    for (int m = 0; m < HL_HALF_TAPS; ++m)
        c[m] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(taps.t[m][pl])));
    auto tapk = [&](const int k) __attribute__((always_inline)) { return c[k <= 9 ? k : 18 - k]; };

    constexpr int SX = SRC_MODE == 2 ? 2 : 1;
    int col_off[2];
    {
        const int xa = x0 - 12 + lane, xb = xa + 64;
        const bool oka = xa >= 0 && xa < w, okb = lane < FZ_STAGE - 64 && xb >= 0 && xb < w;
        col_off[0] = oka ? SX * xa * ES : (int)FZ_DROP;
        col_off[1] = okb ? SX * xb * ES : (int)FZ_DROP;
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float *)&lds[0][0];
    int yi_next = y0 - RADIUS;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    auto fetch = [&](const int slot) __attribute__((always_inline)) {
        const int yi = yi_next++;
        const unsigned row = (yi >= 0 && yi < h) ? (unsigned)(SX * yi) * (unsigned)src_pitch * (unsigned)ES : FZ_DROP;
        const unsigned dst = lds_base + (unsigned)slot * 512u;
        if (SRC_MODE == 1) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_ubyte %1, %2, %3 offen lds" ::"s"(dst), "v"(col_off[0]), "s"(src_rs), "s"(row) : "memory", "m0");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_ubyte %1, %2, %3 offen lds" ::"s"(dst + 256u), "v"(col_off[1]), "s"(src_rs), "s"(row) : "memory", "m0");
        } else {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(dst), "v"(col_off[0]), "s"(src_rs), "s"(row) : "memory", "m0");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(dst + 256u), "v"(col_off[1]), "s"(src_rs), "s"(row) : "memory", "m0");
        }
    };
#pragma clang diagnostic pop
    // Row r + 1 must have landed when step r reads its window.  Vector-memory operations issued
    // after that row's fetch (which step r + 1 - FZP_PF issued): that step's store, FZP_PF - 2 whole
    // steps of two fetches and a store, and this step's two fetches.
    constexpr int FZP_WAIT = 1 + 3 * (FZP_PF - 2) + 2;
    // the 19-pixel window of the row in LDS buffer `slot` (8-bit sources: integers to floats first,
    // in place; a wave's LDS operations stay in order)
    auto window = [&](const int slot, float (&win)[TAPS]) __attribute__((always_inline)) {
        if (SRC_MODE == 1) {
            const unsigned a0 = __float_as_uint(lds[slot][lane]), a1 = __float_as_uint(lds[slot][lane + 64]);
            lds[slot][lane] = (float)a0;
            lds[slot][lane + 64] = (float)a1;
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int k = 0; k < TAPS; ++k) win[k] = lds[slot][lane + 3 + k];
    };

    float acc[COL_RING];
#pragma unroll
    for (int k = 0; k < COL_RING; ++k) acc[k] = 0.f;
    float win_a[TAPS], win_b[TAPS]; // the windows of even and odd rows of a chunk

    auto chunk = [&](const int rb) __attribute__((always_inline)) {
        static_for<COL_RING>([&](auto rr_c) __attribute__((always_inline)) {
            constexpr int rr = decltype(rr_c)::value;
            const int r = rb + rr;
            __builtin_amdgcn_sched_barrier(0);
            if (r >= nin) return; // wave-uniform: the strip ended inside this chunk
            fetch((rr + FZP_PF) % FZP_SLOTS); // row r + FZP_PF goes to the buffer row r - 1 was read from
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FZP_WAIT) : "memory"); // row r + 1 has landed
            float(&win)[TAPS] = (rr & 1) ? win_b : win_a;
            float(&nxt)[TAPS] = (rr & 1) ? win_a : win_b;
            window((rr + 1) % FZP_SLOTS, nxt); // in flight while this row's chains run
            float vs = 0.f;
#pragma unroll
            for (int k = 0; k < TAPS; ++k) vs = __builtin_fmaf(tapk(k), win[k], vs);
#pragma unroll
            for (int k = 0; k < TAPS; ++k) {
                const int slot = (rr - k + COL_RING) % COL_RING;
                acc[slot] = __builtin_fmaf(tapk(k), vs, k == 0 ? 0.f : acc[slot]);
            }
            const bool valid = r >= 2 * RADIUS; // (r < nin holds here)
            const int vo = out_lane | (valid ? 0 : (int)FZ_DROP);
            constexpr int e = (rr - 2 * RADIUS + COL_RING) % COL_RING;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[e]), bl_rs, vo, (int)out_row, PYRP_STORE_POLICY);
            out_row += valid ? (unsigned)p * 4u : 0u;
        });
    };
    // prologue: the first FZP_PF rows, waited for in full; the first row's window
#pragma unroll
    for (int q = 0; q < FZP_PF; ++q) fetch(q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    window(0, win_a);
    for (int rb = 0; rb < nin; rb += COL_RING) chunk(rb);
    // rows fetched beyond the strip's last are still on their way into this block's LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------
// The same pass with TWO columns per lane: a wave owns 128 columns, lane l the columns
// x0 + 2l and x0 + 2l + 1.  Every packed FMA then works on the two columns of one plane
// (the tap is one scalar for both halves), each plane row leaves the wave as one 512-byte
// store (8 bytes per lane), the apron columns are 18 in 146 instead of 24 in 88, and the
// per-row overhead (fetch issue, waits, address updates, branches) is paid once per 128
// pixels.  The price is the ring: 20 slots x (S+3) planes x 2 columns = 200 registers at two
// scales, so two waves per SIMD instead of three.
// Results are bit-identical to pyramid_kernel's: the same chains in the same order.
// Reference profile only (the plain pre-filter needs its neighbours across the two columns).
// ---------------------------------------------------------------------------
#ifndef PYR2_FZ_PF
#define PYR2_FZ_PF 3
#endif
constexpr int FZ2_PF = PYR2_FZ_PF;
constexpr int FZ2_SLOTS = FZ2_PF == 2 ? 4 : FZ2_PF + 1;
static_assert(20 % FZ2_SLOTS == 0, "the row buffers must be used in whole turns per row chunk");
constexpr int FZ2_STAGE = 192; // floats per LDS row buffer: three fetch instructions of 64
constexpr int FZ2_USED = 146;  // staged float j is pixel x0 - 9 + j; the last one a lane reads is 2*63 + 19

template <int S, int SRC_MODE>
__global__ __launch_bounds__(64, 2) void pyramid2_kernel(
    const float *__restrict__ src, int src_w, int src_h, int src_pitch, long src_stride,
    float *__restrict__ blur, long blur_stride, int w, int p, int h, int rows, hl_taps taps, float contrast, unsigned *__restrict__ queue,
    long queue_stride, unsigned *__restrict__ qcount, long qcount_stride, int qcap, int group0, int vgx, int vgy, int vgz)
{
    constexpr int NP = S + 3;
    __shared__ __attribute__((aligned(16))) float lds[FZ2_SLOTS][FZ2_STAGE];
    __shared__ unsigned lq[S][PQ_CAP];
    const int lane = threadIdx.x;
    const unsigned nwg = (unsigned)vgx * (unsigned)vgy * (unsigned)vgz;
    const unsigned orig = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    int tile_x, tile_y, tile_z;
    { // XCD-aware remap, as in pyramid_kernel
        const unsigned gx = (unsigned)vgx, gy = (unsigned)vgy;
        const unsigned q = nwg / 8, r = nwg % 8, xcd = orig % 8;
        const unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
        tile_x = (int)(t % gx);
        tile_y = (int)((t / gx) % gy);
        tile_z = (int)(t / (gx * gy));
    }
    const int x0 = tile_x * 128;
    const int y0 = tile_y * rows;
    const int y1 = min(y0 + rows, h);
    const int nin = (y1 - y0) + 2 * RADIUS;
    const long plane = (long)p * h;
    const int z = tile_z;
    constexpr int ES = SRC_MODE == 1 ? 1 : 4;
    const __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(src) + z * src_stride * ES), 0,
        (int)(unsigned)min((long)src_pitch * src_h * ES, 0xffffffffL), 0x00020000);
    const __amdgpu_buffer_rsrc_t bl_rs = __builtin_amdgcn_make_buffer_rsrc(
        blur + z * blur_stride, 0, (int)(unsigned)min(plane * NP * 4, (long)FZ_DROP), 0x00020000);
    const int xl = x0 + 2 * lane;                                    // this lane's first column
    const int out_lane = xl < p ? xl * 4 : (int)FZ_DROP;             // (the pitch is even: both columns or none)
    unsigned out_row = (unsigned)y0 * p * 4u;
    const unsigned plane_b = (unsigned)plane * 4u;
    const bool pre = queue != nullptr;
    // lanes the detector pre-filter may queue, per column of the lane (cudaSiftD.cu:289; the
    // middle tile of a 96-wide block never fires)
    unsigned long long pre_mask[2], pre_mask0[2], alias_a[2], alias_b[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int xq = xl + c;
        const int xq_tile = (xq % XT_DATAX) / XT_BDIM;
        const bool pre_lane = xq > 8 && xq < w - 8 && xq_tile != 1;
        pre_mask[c] = __ballot(pre_lane);
        pre_mask0[c] = __ballot(pre_lane && xq_tile == 0);
        const int xta = xq + 37, xtb = xq - 64; // targets of the aliased centre cells, see pyramid_kernel
        alias_a[c] = __ballot(xq < w && (xta % XT_DATAX) >= 2 * XT_BDIM && xta > 8 && xta < w - 8);
        alias_b[c] = __ballot(xq < w && xtb > 8 && (xtb % XT_DATAX) < XT_BDIM && xtb < w - 8);
    }
    unsigned lq_n[S];
#pragma unroll
    for (int i = 0; i < S; ++i) lq_n[i] = 0;
    float thr[S];
#pragma unroll
    for (int i = 0; i < S; ++i)
        thr[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(
            (pre && qcount[z * qcount_stride + group0 + i] > (unsigned)qcap) ? __builtin_inff() : contrast)));

    // ---- source fetch: lane j fetches staged floats j, 64 + j and (j < 18) 128 + j ----
    constexpr int SX = SRC_MODE == 2 ? 2 : 1;
    int col_off[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int xa = x0 - RADIUS + 64 * q + lane;
        const bool ok = 64 * q + lane < FZ2_USED && xa >= 0 && xa < w;
        col_off[q] = ok ? SX * xa * ES : (int)FZ_DROP;
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float *)&lds[0][0];
    int yi_next = y0 - RADIUS;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    auto fetch = [&](const int slot) __attribute__((always_inline)) {
        const int yi = yi_next++;
        const unsigned row = (yi >= 0 && yi < h) ? (unsigned)(SX * yi) * (unsigned)src_pitch * (unsigned)ES : FZ_DROP;
        const unsigned dst = lds_base + (unsigned)slot * (unsigned)(FZ2_STAGE * 4);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (SRC_MODE == 1)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_ubyte %1, %2, %3 offen lds" ::"s"(dst + 256u * q), "v"(col_off[q]), "s"(src_rs), "s"(row) : "memory", "m0");
            else
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(dst + 256u * q), "v"(col_off[q]), "s"(src_rs), "s"(row) : "memory", "m0");
        }
    };
#pragma clang diagnostic pop
    // vector-memory operations issued between a row's fetch and its first read
    constexpr int FZ2_WAIT = NP + (NP + 3) * (FZ2_PF - 1) + 3;
    static_assert(FZ2_WAIT < 64, "the counter holds 63");

    float2v acc[NP][COL_RING]; // [plane][ring slot] = the partial column sums of the lane's two columns
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int k = 0; k < COL_RING; ++k) acc[i][k] = float2v{0.f, 0.f};

    auto push = [&](const int i, const unsigned long long hits, const int y, const int dx) __attribute__((always_inline)) {
        if (hits) {
            const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(hits >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hits, 0u));
            const unsigned slot = lq_n[i] + below;
            if (((hits >> lane) & 1ull) && slot < PQ_CAP) lq[i][slot] = (unsigned)(xl + dx) | ((unsigned)y << 16);
            lq_n[i] += (unsigned)__popcll(hits);
        }
    };

    auto chunk = [&](const int rb) __attribute__((always_inline)) {
        static_for<COL_RING>([&](auto rr_c) __attribute__((always_inline)) {
            constexpr int rr = decltype(rr_c)::value;
            const int r = rb + rr;
            __builtin_amdgcn_sched_barrier(0);
            if (r >= nin) return;
            fetch((rr + FZ2_PF) % FZ2_SLOTS);
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FZ2_WAIT) : "memory");
            constexpr int slot = rr % FZ2_SLOTS;
            if (SRC_MODE == 1) {
                unsigned a[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) a[q] = __float_as_uint(lds[slot][lane + 64 * q]);
#pragma unroll
                for (int q = 0; q < 3; ++q) lds[slot][lane + 64 * q] = (float)a[q];
                asm volatile("" ::: "memory");
            }
            // window value j of the lane is staged float 2*lane + j.  Tap k multiplies (w[k], w[k+1]) for
            // the lane's two columns: an aligned 8-byte read for even k, two 4-byte reads for odd k.
            float2v vrow[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) vrow[i] = float2v{0.f, 0.f};
            const float *wl = &lds[slot][2 * lane];
#pragma unroll
            for (int k = 0; k < TAPS; ++k) {
                float2v wk;
                if (k & 1) wk = float2v{wl[k], wl[k + 1]};
                else wk = *reinterpret_cast<const float2v *>(wl + k);
                static_for<NP>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    if (k == 0) vrow[i] = pk_fma_tap<i & 1, true>(tap2(taps, i / 2, k), wk, vrow[i]);
                    else vrow[i] = pk_fma_tap<i & 1, false>(tap2(taps, i / 2, k), wk, vrow[i]);
                });
            }
#pragma unroll
            for (int k = 0; k < TAPS; ++k) {
                const int slot = (rr - k + COL_RING) % COL_RING;
                static_for<NP>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    if (k == 0) acc[i][slot] = pk_fma_tap<i & 1, true>(tap2(taps, i / 2, k), vrow[i], acc[i][slot]);
                    else acc[i][slot] = pk_fma_tap<i & 1, false>(tap2(taps, i / 2, k), vrow[i], acc[i][slot]);
                });
            }
            const bool valid = r >= 2 * RADIUS && r < nin; // wave-uniform
            const int vo = out_lane | (valid ? 0 : (int)FZ_DROP);
            constexpr int e = (rr - 2 * RADIUS + COL_RING) % COL_RING;
            float2v b[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) b[i] = acc[i][e];
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                typedef unsigned u2v __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(u2v{__float_as_uint(b[i].x), __float_as_uint(b[i].y)}, bl_rs, vo,
                                                      (int)(out_row + i * plane_b), FZ_STORE_POLICY);
            }
            out_row += valid ? (unsigned)p * 4u : 0u;
            // detector pre-filter, as in pyramid_kernel, once per column of the lane
            const int yo = y0 + r - 2 * RADIUS;
            const bool row_ok = pre && valid && yo > 8 && yo < h - 8 && (yo & 31) < 31; // wave-uniform
            if (row_ok) {
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    const float2v d = float2v{__fsub_rn(b[i + 1].x, b[i + 2].x), __fsub_rn(b[i + 1].y, b[i + 2].y)};
                    push(i, __ballot(fabsf(d.x) > thr[i]) & ((yo & 31) == 30 ? pre_mask0[0] : pre_mask[0]), yo, 0);
                    push(i, __ballot(fabsf(d.y) > thr[i]) & ((yo & 31) == 30 ? pre_mask0[1] : pre_mask[1]), yo, 1);
                }
            }
            const int ym = yo & 31;
            if (pre && valid && (ym == 31 || ym == 0)) { // a source row of the aliased centre cells
                const int yt = yo + 31;
                const bool yt_ok = yt > 8 && yt < h - 8, yt1_ok = yt + 1 > 8 && yt + 1 < h - 8;
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    const unsigned long long big0 = __ballot(fabsf(__fsub_rn(b[i + 2].x, b[i + 3].x)) > thr[i]);
                    const unsigned long long big1 = __ballot(fabsf(__fsub_rn(b[i + 2].y, b[i + 3].y)) > thr[i]);
                    if (yt_ok) {
                        push(i, big0 & alias_a[0], yt, 37);
                        push(i, big1 & alias_a[1], yt, 38);
                    }
                    if (ym == 31 && yt1_ok) {
                        push(i, big0 & alias_b[0], yt + 1, -64);
                        push(i, big1 & alias_b[1], yt + 1, -63);
                    }
                }
            }
        });
    };
    auto flush = [&]() __attribute__((always_inline)) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const unsigned n = lq_n[i];
            if (n == 0) continue;
            unsigned base = 0;
            if (lane == 0)
                base = n > PQ_CAP ? (atomicOr(&qcount[z * qcount_stride + group0 + i], 0x80000000u) | 0x80000000u)
                                  : atomicAdd(&qcount[z * qcount_stride + group0 + i], n);
            base = __builtin_amdgcn_readfirstlane(__shfl(base, 0));
            if (n <= PQ_CAP)
                for (unsigned e = lane; e < n; e += 64)
                    if (base + e < (unsigned)qcap) queue[z * queue_stride + (long)i * qcap + base + e] = lq[i][e];
            if (base + n > (unsigned)qcap) thr[i] = __builtin_inff();
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < S; ++i) lq_n[i] = 0;
    };
#pragma unroll
    for (int q = 0; q < FZ2_PF; ++q) fetch(q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int rb = 0; rb < nin; rb += COL_RING) {
        chunk(rb);
        if (pre) { // a chunk of 20 rows adds at most 2560 entries per scale, normally a few
            unsigned fill = 0;
#pragma unroll
            for (int i = 0; i < S; ++i) fill = max(fill, lq_n[i]);
            if (fill > PQ_CAP / 2) flush();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (pre) flush();
}

// ---------------------------------------------------------------------------
// Detection.  The reference's acceptance rule depends on where a pixel sits in
// ITS launch geometry (blocks of 96x32 pixels, three 32-wide tiles per block,
// tiles of 101x34 floats spaced 3168 floats apart in shared memory).  These
// constants restate that geometry so the rule can be evaluated per pixel from
// global memory under any tiling.
// ---------------------------------------------------------------------------

__device__ __forceinline__ float zpad(const float *pl, int w, int p, int h, int x, int y)
{
    return (x >= 0 && x < w && y >= 0 && y < h) ? pl[x + (long)p * y] : 0.f;
}

// value found at (sx,sy) of tile s by a thread of block (bx,by): the head of
// tile s+1 has overwritten the tail of tile s (cudaSiftD.cu:239-261).
// Branch-free, so independent reads issue back to back.
__device__ __forceinline__ float tile_read(const float *b3, long plane, int w, int p, int h, int bx,
                                           int by, int s, int sx, int sy)
{
    const int lin = sx + XT_SDIMX * sy;
    const int j = lin - XT_PLANE_OFF;
    const int sy2 = j / XT_SDIMX;
    const int sx2 = j - sy2 * XT_SDIMX;
    const bool moved = s < 2 && j >= 0 && sx2 < XT_WRITTEN;
    s = moved ? s + 1 : s;
    sx = moved ? sx2 : sx;
    sy = moved ? sy2 : sy;
    const int x = XT_DATAX * bx - 1 + sx, y = XT_BDIM * by - 1 + sy;
    const bool in = x >= 0 && x < w && y >= 0 && y < h;
    const float *c = b3 + s * plane + (in ? x + (long)p * y : 0);
    const float v = __fsub_rn(c[0], c[plane]); // DoG plane s (cudaSiftH.cu:190)
    return in ? v : 0.f;
}

struct __attribute__((packed, aligned(4))) float3u {
    float a, b, c;
};

// 27-cell rank of cudaSiftD.cu:271-284: sum of +1 where centre >= cell, else -1.
// rank_plain: no aliased cell in reach and no image border, i.e. the plain
// neighbourhood in three DoG planes, from four blur planes.  Twelve 12-byte
// loads (three neighbours of one row of one plane each) instead of 36 single ones.
__device__ __forceinline__ bool rank_is_plain(int w, int h, int x, int y)
{
    return (y & 31) < 29 && x > 0 && x < w - 1 && y > 0 && y < h - 1;
}

__device__ __forceinline__ int rank_plain(const float *__restrict__ b3, long plane, int p, int x, int y)
{
    const float *c = b3 + x + (long)p * y;
    float v[4][3][3];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int m = -1; m < 2; ++m) {
            const float3u t = *reinterpret_cast<const float3u *>(c + s * plane + m * p - 1);
            v[s][m + 1][0] = t.a, v[s][m + 1][1] = t.b, v[s][m + 1][2] = t.c;
        }
    const float pv = __fsub_rn(v[1][1][1], v[2][1][1]);
    int r = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int l = 0; l < 3; ++l) r += (pv >= __fsub_rn(v[s][m][l], v[s + 1][m][l])) ? 1 : -1;
    return r;
}

// the same through the reference's tile addressing, aliased cells and zero padding included
__device__ __forceinline__ int rank_tiled(const float *__restrict__ b3, long plane, int w, int p, int h, int x, int y)
{
    const int bx = x / XT_DATAX, by = y >> 5;
    const int sx = 1 + x % XT_DATAX, sy = 1 + (y & 31);
    const float pv = tile_read(b3, plane, w, p, h, bx, by, 1, sx, sy);
    int r = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int m = -1; m < 2; ++m)
#pragma unroll
            for (int l = -1; l < 2; ++l)
                r += (pv >= tile_read(b3, plane, w, p, h, bx, by, s, sx + l, sy + m)) ? 1 : -1;
    return r;
}

// in the right tile of a 96-wide block the rank is carried over from the two
// tiles to the left: ranks at x, x-32 and x-64 are added
__device__ __forceinline__ int rank_count(int x) { return (x % XT_DATAX) >= 2 * XT_BDIM ? 3 : 1; }

// true if every rank of the candidate can be taken on the plain neighbourhood
__device__ __forceinline__ bool candidate_is_plain(int w, int h, int x, int y)
{
    return rank_is_plain(w, h, x, y) && (rank_count(x) == 1 || x - 64 > 0);
}

// Full acceptance test of one contrast candidate (cudaSiftD.cu:271-293 after the
// contrast clause): rank 9 with the carry of the reference's tile loop, then
// the edge test on a Hessian of blur plane i+1 (hessianKernel, cudaSiftD.cu:405-419).
// PLAIN: the caller has checked candidate_is_plain.
template <bool PLAIN>
__device__ __forceinline__ bool accept_candidate(const float *__restrict__ b3, long plane, int w, int p,
                                                 int h, float edge_limit, int x, int y)
{
    const int n_ranks = rank_count(x);
    int r = 0;
    for (int t = 0; t < n_ranks; ++t) {
        const int xt = x - 32 * t;
        if (PLAIN)
            r += rank_plain(b3, plane, p, xt, y);
        else
            r += rank_is_plain(w, h, xt, y) ? rank_plain(b3, plane, p, xt, y) : rank_tiled(b3, plane, w, p, h, xt, y);
    }
    if (r != 9) return false;
    const float *b = b3 + plane + x + (long)p * y; // blur plane i+1
    const float c = b[0];
    float Dxx = -2.f * c;
    Dxx += b[-1];
    Dxx += b[1];
    float Dyy = -2.f * c;
    Dyy += b[-p];
    Dyy += b[p];
    float Dxy = c;
    Dxy += b[-p - 1];
    Dxy -= b[-1];
    Dxy -= b[-p];
    const float trace = Dxx + Dyy;
    const float det = Dxx * Dyy + Dxy * Dxy; // PLUS, and no contraction (cudaSiftD.cu:288)
    return (trace * trace) / det < edge_limit;
}

// wave-aggregated append: one global atomic per wave (ballot + prefix count)
__device__ __forceinline__ void append_hits(bool hit, unsigned xy, int lane, unsigned *count,
                                            unsigned *list, int cap)
{
    const unsigned long long m = __ballot(hit);
    if (m) {
        const int leader = __ffsll((long long)m) - 1;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(count, (unsigned)__popcll(m));
        base = __shfl(base, leader);
        const unsigned idx = base + __popcll(m & ((1ull << lane) - 1ull));
        if (hit && idx < (unsigned)cap) list[idx] = xy;
    }
}

// The three detection kernels work on all octaves of a launch set at once:
// blockIdx.y enumerates work units, the octave table says where each octave's
// units start and where its planes, queues and sizes are.
__device__ __forceinline__ int find_octave(const int *unit0, int n_oct, int unit)
{
    int o = 0;
    while (o + 1 < n_oct && unit >= unit0[o + 1]) ++o;
    return o;
}

// Sparse detector: consumes the contrast candidates queued by pyramid_kernel.  The queues of all (octave, scale) groups of an
// image form one list of 256-candidate items which the image's blocks
// (gridDim.x of them) take in turn, so the launch has the same number of blocks
// whatever the images hold and each block's share is even.  (A block per group
// and queue segment is mostly blocks that find nothing to do, and their
// dispatch alone outlasts the useful work.)
template <int S>
__global__ __launch_bounds__(256) void verify_kernel(const float *__restrict__ arena, long arena_stride,
                                                     const unsigned *queue, long queue_stride,
                                                     const unsigned *qcount, long qcount_stride,
                                                     unsigned *cand, long cand_stride,
                                                     unsigned *counts, long counts_stride,
                                                     hl_detect_args a)
{
    const int lane = threadIdx.x & 63;
    const int z = blockIdx.y;
    qcount += z * qcount_stride;
    const int n_groups = a.n_oct * S;
    // first[g]: index of group g's first item (an overflowed queue has none: the dense
    // kernel takes that group).  Lane g of the first wave counts group g, a wave scan adds up.
    __shared__ unsigned first[HL_MAX_GROUPS + 1];
    // Candidates that need the tile addressing (a tenth of them, spread over
    // every wave) are set aside in LDS and handled in full batches, so that the
    // long path runs once per 256 of them instead of once per wave.
    __shared__ unsigned slow_xy[512];
    __shared__ unsigned slow_n;
    if (threadIdx.x < 64) {
        unsigned items = 0;
        if (lane < n_groups) {
            const unsigned n = qcount[lane];
            items = n > (unsigned)a.oct[lane / S].qcap ? 0u : (n + 255u) / 256u;
        }
#pragma unroll
        for (int off = 1; off < HL_MAX_GROUPS; off <<= 1) {
            const unsigned up = __shfl_up(items, off);
            if (lane >= off) items += up;
        }
        if (lane == 0) first[0] = 0, slow_n = 0;
        if (lane < HL_MAX_GROUPS) first[lane + 1] = items;
    }
    __syncthreads();
    const unsigned total = first[n_groups];
    int g = 0, g_slow = -1; // g_slow: the group the set-aside candidates belong to
    // the set-aside candidates of group gs, `count` of them from slot `from` (block-uniform arguments)
    auto run_slow = [&](const int gs, const unsigned from, const unsigned count) __attribute__((always_inline)) {
        const int oi = gs / S, i = gs - oi * S;
        const hl_octave oc = a.oct[oi];
        const long plane = (long)oc.p * oc.h;
        const float *b3 = arena + z * arena_stride + oc.blur_off + i * plane;
        bool hit = false;
        unsigned xy = 0;
        if (threadIdx.x < count) {
            xy = slow_xy[from + threadIdx.x];
            hit = accept_candidate<false>(b3, plane, oc.w, oc.p, oc.h, a.edge_limit, (int)(xy & 0xffffu),
                                          (int)(xy >> 16));
        }
        append_hits(hit, xy, lane, counts + z * counts_stride + gs, cand + z * cand_stride + (long)gs * a.cap, a.cap);
    };
    for (unsigned item = blockIdx.x; item < total; item += gridDim.x) { // block-uniform
        while (item >= first[g + 1]) ++g;
        if (g != g_slow) { // a new group: finish what the last one set aside
            const unsigned pending = slow_n;
            __syncthreads();
            if (pending) {
                run_slow(g_slow, 0, pending);
                if (threadIdx.x == 0) slow_n = 0;
                __syncthreads();
            }
            g_slow = g;
        }
        const int oi = g / S, i = g - oi * S;
        const hl_octave oc = a.oct[oi];
        const unsigned n = qcount[g];
        const long plane = (long)oc.p * oc.h;
        const float *b3 = arena + z * arena_stride + oc.blur_off + i * plane; // blur planes i..i+3
        const unsigned e = (item - first[g]) * 256u + threadIdx.x;
        bool hit = false, slow = false;
        unsigned xy = 0;
        if (e < n) {
            xy = queue[z * queue_stride + oc.queue_off + (long)i * oc.qcap + e];
            const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
            if (candidate_is_plain(oc.w, oc.h, x, y))
                hit = accept_candidate<true>(b3, plane, oc.w, oc.p, oc.h, a.edge_limit, x, y);
            else
                slow = true;
        }
        append_hits(hit, xy, lane, counts + z * counts_stride + g, cand + z * cand_stride + (long)g * a.cap, a.cap);
        const unsigned long long sm = __ballot(slow);
        if (sm) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&slow_n, (unsigned)__popcll(sm));
            base = __shfl(base, 0);
            if (slow) slow_xy[base + __popcll(sm & ((1ull << lane) - 1ull))] = xy;
        }
        __syncthreads();
        const unsigned pending = slow_n; // at most 255 kept + 256 new
        // every thread has read the word before any wave can add to it again in the next turn:
        // the branch below contains barriers, so the whole block must agree on it
        __syncthreads();
        if (pending >= 256) {
            run_slow(g, pending - 256, 256);
            __syncthreads();
            if (threadIdx.x == 0) slow_n = pending - 256;
            __syncthreads();
        }
    }
    const unsigned pending = slow_n;
    if (pending) run_slow(g_slow, 0, pending);
}

// Dense detector: the reference's kernel as it stands.  One block per 96 x 32
// pixels loads the three DoG tiles (101 x 34 floats each, 98 columns written,
// zero outside the image) into LDS at the reference's spacing of 3168 floats,
// in its order, so a later tile overwrites the tail of the one before exactly
// as it does there (cudaSiftD.cu:239-261).  The rank is then read straight out
// of LDS (cudaSiftD.cu:271-284), with the carry over the three 32-wide column
// tiles.  The contrast test runs first and only the left and right column tiles
// are looked at (the middle one can never reach rank 9).  Cost is a streaming
// read of the planes whatever the number of candidates, which is what makes it
// the right tool when a queue overflows.  With `gate` set a group is processed
// only if its queue counter overflowed (gate[group] > qcap); with gate NULL it
// is the whole detector.
constexpr int XT_TILE = XT_SDIMX * 34; // cells of one tile

__device__ __forceinline__ bool edge_ok(const float *__restrict__ b, int p, float edge_limit)
{
    // hessianKernel cudaSiftD.cu:405-419 on blur plane i+1, then cudaSiftD.cu:287-292
    const float c = b[0];
    float Dxx = -2.f * c;
    Dxx += b[-1];
    Dxx += b[1];
    float Dyy = -2.f * c;
    Dyy += b[-p];
    Dyy += b[p];
    float Dxy = c;
    Dxy += b[-p - 1];
    Dxy -= b[-1];
    Dxy -= b[-p];
    const float trace = Dxx + Dyy;
    const float det = Dxx * Dyy + Dxy * Dxy; // PLUS, and no contraction (cudaSiftD.cu:288)
    return (trace * trace) / det < edge_limit;
}

constexpr int DET_THREADS = 512; // threads of a dense-detector block (a multiple of 32 dividing 1024)

template <int S, bool SCALE_MAJOR>
__global__ __launch_bounds__(DET_THREADS, 4) void detect_kernel(const float *__restrict__ arena, long arena_stride,
                                                     unsigned *cand, long cand_stride,
                                                     unsigned *counts, long counts_stride,
                                                     const unsigned *gate, long gate_stride,
                                                     hl_detect_args a)
{
    __shared__ float sh[2 * XT_PLANE_OFF + XT_TILE + 2];
    // pixels of the block that passed the contrast test: (tile << 16 | ty << 8 | tx)
    __shared__ unsigned hot[2 * XT_BDIM * XT_BDIM];
    __shared__ unsigned hot_n;
    // The announcement word as thread 0 read it for the current block of pixels.  The
    // word changes while the kernel runs; threads that read it themselves would see
    // different values, skip different scales, and meet at different barriers.  Two
    // slots, used in turn: a slot is rewritten only after every wave has passed the
    // barrier that follows the read of the other one.
    __shared__ unsigned s_announced[2];
    unsigned turn = 0;
    const int lane = threadIdx.x & 63;
    const int z = blockIdx.z;
    cand += z * cand_stride;
    counts += z * counts_stride;
    unsigned *full = counts + HL_FULL_WORD;
    // the groups to work on, as a bit mask read once (block-uniform)
    unsigned todo = 0xffffffffu;
    if (gate) { // nothing to do for this image unless one of its queues overflowed
        gate += z * gate_stride;
        todo = 0;
        for (int g = 0; g < a.n_oct * S; ++g)
            if (gate[g] > (unsigned)a.oct[g / S].qcap) todo |= 1u << g;
        if (!todo) return;
    }
    constexpr int CELLS = XT_WRITTEN * 34;          // cells of a tile that are written
    constexpr int ROUNDS = (CELLS + DET_THREADS - 1) / DET_THREADS; // per thread
    // the tile cells of this thread do not depend on the block being worked on:
    // coordinates, LDS address and whether the next tile overwrites the cell
    int cell_sx[ROUNDS], cell_sy[ROUNDS];
    unsigned own = 0, taken = 0; // bit it: the cell exists / is overwritten by the next tile
#pragma unroll
    for (int it = 0; it < ROUNDS; ++it) {
        const int c = threadIdx.x + DET_THREADS * it;
        const int cc = min(c, CELLS - 1);
        cell_sy[it] = cc / XT_WRITTEN;
        cell_sx[it] = cc - cell_sy[it] * XT_WRITTEN;
        const int j = cell_sx[it] + XT_SDIMX * cell_sy[it] - XT_PLANE_OFF; // the address seen from the next tile
        if (c < CELLS) own |= 1u << it;
        if (j >= 0 && (j % XT_SDIMX) < XT_WRITTEN) taken |= 1u << it;
    }
    // Scale i of the 96 x 32 block (bx, by) of octave oc, group g: val(it, t) is the value of
    // blur plane i + t (t = 0..3) at this thread's tile cell `it`; b3 points at blur plane i.
    auto eval = [&](const int g, const hl_octave &oc, const float *b3, const int bx, const int by,
                    const unsigned inside, auto val) __attribute__((always_inline)) {
        const int w = oc.w, h = oc.h, p = oc.p;
        const long plane = (long)p * h;
        __syncthreads(); // readers of the previous tiles are done
        if (threadIdx.x == 0) hot_n = 0;
        // The reference writes tile 0, then 1, then 2, and a later tile overwrites the
        // tail of the one before.  Which write a cell keeps is known from its address:
        // a cell of tile t survives unless tile t+1 writes the same address, so the
        // three tiles go out together and only the surviving writes are made.
#pragma unroll
        for (int it = 0; it < ROUNDS; ++it) {
            const int lin = cell_sx[it] + XT_SDIMX * cell_sy[it];
            const bool in = (inside >> it) & 1u;
            if ((own >> it) & 1u) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    if (t == 2 || !((taken >> it) & 1u))
                        sh[t * XT_PLANE_OFF + lin] = in ? __fsub_rn(val(it, t), val(it, t + 1)) : 0.f;
            }
        }
        __syncthreads();
        // contrast test first: the survivors are listed, then shared out over the block
        const int tx = threadIdx.x & 31;
#pragma unroll
        for (int pass = 0; pass < XT_BDIM * XT_BDIM / DET_THREADS; ++pass) {
            const int ty = pass * (DET_THREADS / XT_BDIM) + (threadIdx.x >> 5);
            const int y = XT_BDIM * by + ty;
#pragma unroll
            for (int tile = 0; tile < 3; tile += 2) {
                const int x = XT_DATAX * bx + tx + XT_BDIM * tile;
                const float pv = sh[XT_PLANE_OFF + 1 + tx + XT_BDIM * tile + XT_SDIMX * (1 + ty)];
                const bool pass = x > 8 && x < w - 8 && y > 8 && y < h - 8 && fabsf(pv) > a.contrast;
                const unsigned long long pm = __ballot(pass); // one LDS atomic per wave, not per pixel
                if (pm) {
                    unsigned base = 0;
                    if (lane == 0) base = atomicAdd(&hot_n, (unsigned)__popcll(pm));
                    base = __shfl(base, 0);
                    if (pass)
                        hot[base + __popcll(pm & ((1ull << lane) - 1ull))] =
                            (unsigned)tx | ((unsigned)ty << 8) | ((unsigned)tile << 16);
                }
            }
        }
        __syncthreads();
        const unsigned n_hot = hot_n;
        for (unsigned e0 = 0; e0 < n_hot; e0 += DET_THREADS) { // block-uniform trip count
            const unsigned e = e0 + threadIdx.x;
            bool hit = false;
            unsigned xy = 0;
            if (e < n_hot) {
                const unsigned hv = hot[e];
                const int htx = hv & 0xff, hty = (hv >> 8) & 0xff, tile = hv >> 16;
                const int sy = 1 + hty;
                int r = 0;
                for (int tt = 0; tt <= tile; ++tt) { // the reference never resets its rank between tiles
                    const int sx = 1 + htx + XT_BDIM * tt;
                    const float cv = sh[XT_PLANE_OFF + sx + XT_SDIMX * sy];
#pragma unroll
                    for (int l = -1; l < 2; ++l)
#pragma unroll
                        for (int m = -1; m < 2; ++m)
#pragma unroll
                            for (int k = 0; k < 3; ++k)
                                r += (cv >= sh[k * XT_PLANE_OFF + sx + l + XT_SDIMX * (sy + m)]) ? 1 : -1;
                }
                const int x = XT_DATAX * bx + htx + XT_BDIM * tile, y = XT_BDIM * by + hty;
                xy = (unsigned)x | ((unsigned)y << 16);
                hit = r == 9 && edge_ok(b3 + plane + x + (long)p * y, p, a.edge_limit);
            }
            // append, and announce the group once it holds `cap` points
            const unsigned long long hm = __ballot(hit);
            if (hm) {
                const int leader = __ffsll((long long)hm) - 1;
                const unsigned n_new = (unsigned)__popcll(hm);
                unsigned base = 0;
                if (lane == leader) {
                    base = atomicAdd(&counts[g], n_new);
                    if (base < (unsigned)a.cap && base + n_new >= (unsigned)a.cap)
                        atomicMax(full, (unsigned)(HL_MAX_GROUPS - g));
                }
                base = __shfl(base, leader);
                const unsigned idx = base + __popcll(hm & ((1ull << lane) - 1ull));
                if (hit && idx < (unsigned)a.cap) cand[(long)g * a.cap + idx] = xy;
            }
        }
    };
    // which tile cells of the block (bx, by) are pixels of the image (bit it)
    auto cells_inside = [&](const hl_octave &oc, const int bx, const int by) __attribute__((always_inline)) {
        const int xb = XT_DATAX * bx - 1, yb = XT_BDIM * by - 1; // image position of tile cell (0, 0)
        // a block whose tile lies inside the image needs no bounds tests (block-uniform)
        if (xb >= 0 && xb + XT_WRITTEN <= oc.w && yb >= 0 && yb + 34 <= oc.h) return 0xffffffffu;
        unsigned inside = 0;
#pragma unroll
        for (int it = 0; it < ROUNDS; ++it) {
            const int x = xb + cell_sx[it], y = yb + cell_sy[it];
            if (x >= 0 && x < oc.w && y >= 0 && y < oc.h) inside |= 1u << it;
        }
        return inside;
    };
    // The output keeps groups in order up to `cap` points and drops what follows
    // (cudaSiftH.cu:12-13, 82).  A block that sees a group reach the cap announces
    // it in a word of its own cache line; from then on that group and all later
    // ones cannot change the output.  The word is read past the vector L1 together
    // with the plane loads of a block.
    // thread 0 reads the word (before the plane loads are issued, so its latency passes
    // with theirs) ...
    auto read_announcement = [&]() __attribute__((always_inline)) {
        return threadIdx.x == 0 ? __hip_atomic_load(full, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    };
    // ... and the block agrees on it: first group that is full, if any (block-uniform)
    auto first_full_group = [&](const unsigned mine) __attribute__((always_inline)) {
        if (threadIdx.x == 0) s_announced[turn] = mine;
        __syncthreads();
        const unsigned announced = s_announced[turn];
        turn ^= 1u;
        return announced ? HL_MAX_GROUPS - (int)announced : HL_MAX_GROUPS;
    };
    if constexpr (SCALE_MAJOR) {
        // Group after group: all blocks work on one (octave, scale) at a time and load only
        // its four planes.  With a small cap the first group usually fills it, and all the
        // later ones (the other scales of the octave included) are never looked at; the price
        // when no group fills the cap is that the planes shared by two scales are read twice.
        for (int oi = 0; oi < a.n_oct; ++oi) {
            const hl_octave oc = a.oct[oi];
            const int p = oc.p;
            const int xblocks = (oc.w + XT_DATAX - 1) / XT_DATAX;
            const int n_units = a.dense0[oi + 1] - a.dense0[oi];
            const long plane = (long)p * oc.h;
            const float *blur = arena + z * arena_stride + oc.blur_off;
#pragma unroll
            for (int i = 0; i < S; ++i) {
                const int g = oi * S + i;
                if (!((todo >> g) & 1u)) continue; // block-uniform
                for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
                    const int bx = u % xblocks, by = u / xblocks;
                    const unsigned mine = read_announcement();
                    const unsigned inside = cells_inside(oc, bx, by);
                    const float *corner = blur + i * plane + (XT_DATAX * bx - 1) + (long)p * (XT_BDIM * by - 1);
                    float v[ROUNDS][4];
#pragma unroll
                    for (int it = 0; it < ROUNDS; ++it) {
                        const float *q = ((inside >> it) & 1u) ? corner + cell_sx[it] + p * cell_sy[it] : blur;
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[it][k] = q[k * plane];
                    }
                    const int full_from = first_full_group(mine);
                    if (g >= full_from) return; // this group and all later ones are beyond the cap
                    eval(g, oc, blur + i * plane, bx, by, inside, [&](int it, int t) { return v[it][t]; });
                }
            }
        }
    } else {
    // Block after block, all scales of an octave at a time.  (The same steps as `eval`,
    // written out: sharing the code costs this path 19 registers and a block per CU.)
    // dense0[] counts blocks of 96 x 32 pixels, octave after octave, row-major
    for (int unit = blockIdx.x; unit < a.dense0[a.n_oct]; unit += gridDim.x) {
        const int oi = find_octave(a.dense0, a.n_oct, unit);
        const hl_octave oc = a.oct[oi];
        const int w = oc.w, h = oc.h, p = oc.p;
        const int xblocks = (w + XT_DATAX - 1) / XT_DATAX;
        const int u = unit - a.dense0[oi];
        const int bx = u % xblocks, by = u / xblocks;
        const long plane = (long)p * h;
        const float *blur = arena + z * arena_stride + oc.blur_off;
        if (!((todo >> (oi * S)) & ((1u << S) - 1u))) continue; // block-uniform
        // The output keeps groups in order up to `cap` points and drops what follows
        // (cudaSiftH.cu:12-13, 82).  A block that sees a group reach the cap announces
        // it in a word of its own cache line; from then on that group and all later
        // ones cannot change the output.  Thread 0 reads the word past the vector L1
        // together with the plane loads below; the block agrees on it afterwards.
        const unsigned mine = read_announcement();
        // Every cell's S+3 blur values are requested before the first is used: one
        // memory round trip serves the tiles of all scales of the octave (the scales
        // share all but two of their planes).
        float v[ROUNDS][S + 3];
        const int xb = XT_DATAX * bx - 1, yb = XT_BDIM * by - 1; // image position of tile cell (0, 0)
        // a block whose tile lies inside the image needs no bounds tests (block-uniform)
        const bool interior = xb >= 0 && xb + XT_WRITTEN <= w && yb >= 0 && yb + 34 <= h;
        unsigned inside = 0xffffffffu; // bit it: the cell is a pixel of the image
        if (!interior) {
            inside = 0;
#pragma unroll
            for (int it = 0; it < ROUNDS; ++it) {
                const int x = xb + cell_sx[it], y = yb + cell_sy[it];
                if (x >= 0 && x < w && y >= 0 && y < h) inside |= 1u << it;
            }
        }
        const float *corner = blur + xb + (long)p * yb;
#pragma unroll
        for (int it = 0; it < ROUNDS; ++it) {
            const float *q = ((inside >> it) & 1u) ? corner + cell_sx[it] + p * cell_sy[it] : blur;
#pragma unroll
            for (int k = 0; k < S + 3; ++k) v[it][k] = q[k * plane];
        }
        // first group that is full, if any, as the whole block sees it
        const int full_from = first_full_group(mine);
        if (oi * S >= full_from) return; // this octave and all later ones are beyond the cap
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const int g = oi * S + i;
            if (!((todo >> g) & 1u) || g >= full_from) continue; // block-uniform
            const float *b3 = blur + i * plane; // blur planes i..i+3: DoG planes i..i+2
            __syncthreads(); // readers of the previous tiles are done
            if (threadIdx.x == 0) hot_n = 0;
            // The reference writes tile 0, then 1, then 2, and a later tile overwrites the
            // tail of the one before.  Which write a cell keeps is known from its address:
            // a cell of tile t survives unless tile t+1 writes the same address, so the
            // three tiles go out together and only the surviving writes are made.
#pragma unroll
            for (int it = 0; it < ROUNDS; ++it) {
                const int lin = cell_sx[it] + XT_SDIMX * cell_sy[it];
                const bool in = (inside >> it) & 1u;
                if ((own >> it) & 1u) {
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        if (t == 2 || !((taken >> it) & 1u))
                            sh[t * XT_PLANE_OFF + lin] = in ? __fsub_rn(v[it][i + t], v[it][i + t + 1]) : 0.f;
                }
            }
            __syncthreads();
            // contrast test first: the survivors are listed, then shared out over the block
            const int tx = threadIdx.x & 31;
#pragma unroll
            for (int pass = 0; pass < XT_BDIM * XT_BDIM / DET_THREADS; ++pass) {
                const int ty = pass * (DET_THREADS / XT_BDIM) + (threadIdx.x >> 5);
                const int y = XT_BDIM * by + ty;
#pragma unroll
                for (int tile = 0; tile < 3; tile += 2) {
                    const int x = XT_DATAX * bx + tx + XT_BDIM * tile;
                    const float pv = sh[XT_PLANE_OFF + 1 + tx + XT_BDIM * tile + XT_SDIMX * (1 + ty)];
                    const bool pass = x > 8 && x < w - 8 && y > 8 && y < h - 8 && fabsf(pv) > a.contrast;
                    const unsigned long long pm = __ballot(pass); // one LDS atomic per wave, not per pixel
                    if (pm) {
                        unsigned base = 0;
                        if (lane == 0) base = atomicAdd(&hot_n, (unsigned)__popcll(pm));
                        base = __shfl(base, 0);
                        if (pass)
                            hot[base + __popcll(pm & ((1ull << lane) - 1ull))] =
                                (unsigned)tx | ((unsigned)ty << 8) | ((unsigned)tile << 16);
                    }
                }
            }
            __syncthreads();
            const unsigned n_hot = hot_n;
            for (unsigned e0 = 0; e0 < n_hot; e0 += DET_THREADS) { // block-uniform trip count
                const unsigned e = e0 + threadIdx.x;
                bool hit = false;
                unsigned xy = 0;
                if (e < n_hot) {
                    const unsigned hv = hot[e];
                    const int htx = hv & 0xff, hty = (hv >> 8) & 0xff, tile = hv >> 16;
                    const int sy = 1 + hty;
                    int r = 0;
                    for (int tt = 0; tt <= tile; ++tt) { // the reference never resets its rank between tiles
                        const int sx = 1 + htx + XT_BDIM * tt;
                        const float cv = sh[XT_PLANE_OFF + sx + XT_SDIMX * sy];
#pragma unroll
                        for (int l = -1; l < 2; ++l)
#pragma unroll
                            for (int m = -1; m < 2; ++m)
#pragma unroll
                                for (int k = 0; k < 3; ++k)
                                    r += (cv >= sh[k * XT_PLANE_OFF + sx + l + XT_SDIMX * (sy + m)]) ? 1 : -1;
                    }
                    const int x = XT_DATAX * bx + htx + XT_BDIM * tile, y = XT_BDIM * by + hty;
                    xy = (unsigned)x | ((unsigned)y << 16);
                    hit = r == 9 && edge_ok(b3 + plane + x + (long)p * y, p, a.edge_limit);
                }
                // append, and announce the group once it holds `cap` points
                const unsigned long long hm = __ballot(hit);
                if (hm) {
                    const int leader = __ffsll((long long)hm) - 1;
                    const unsigned n_new = (unsigned)__popcll(hm);
                    unsigned base = 0;
                    if (lane == leader) {
                        base = atomicAdd(&counts[g], n_new);
                        if (base < (unsigned)a.cap && base + n_new >= (unsigned)a.cap)
                            atomicMax(full, (unsigned)(HL_MAX_GROUPS - g));
                    }
                    base = __shfl(base, leader);
                    const unsigned idx = base + __popcll(hm & ((1ull << lane) - 1ull));
                    if (hit && idx < (unsigned)a.cap) cand[(long)g * a.cap + idx] = xy;
                }
            }
        }
    }
    }
}

// ---------------------------------------------------------------------------
// Detector of the "lowe" profile (not in the reference): strict 26-neighbour
// extremum of the DoG, 3-D quadratic refinement, contrast and edge rejection on
// the DoG Hessian.  Single precision, one operation at a time in the order
// written (the build has contraction off), so the host restatement gives the
// same bits.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double det_exp2(double f)
{
    // exp(f ln 2) by its series in Horner form: |f ln 2| <= 1.39, 22 terms leave < 1e-19
    const double t = f * 0.69314718055994530942;
    double q = 1.0;
#pragma unroll
    for (int k = 22; k >= 1; --k) q = fma(q, t / (double)k, 1.0);
    return q;
}

// DoG planes c-1, c, c+1 at the 3x3 neighbourhood of (x, y): blur planes c-1..c+2,
// twelve 12-byte loads
__device__ __forceinline__ void lowe_cube(const float *__restrict__ blur0, long plane, int p, int c,
                                          int x, int y, float (&d)[3][3][3])
{
    const float *q = blur0 + (c - 1) * plane + x + (long)p * y;
    float v[4][3][3];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int m = -1; m < 2; ++m) {
            const float3u t = *reinterpret_cast<const float3u *>(q + s * plane + m * p - 1);
            v[s][m + 1][0] = t.a, v[s][m + 1][1] = t.b, v[s][m + 1][2] = t.c;
        }
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int l = 0; l < 3; ++l) d[s][m][l] = __fsub_rn(v[s][m][l], v[s + 1][m][l]);
}

struct LoweHit {
    unsigned xy; // sample the point was refined at
    float4 sub;  // x, y in input pixels, scale, DoG plane index (as an integer's bits)
};

// blur0: blur plane 0 of the octave.  (x, y): starting sample, i: scale index.
__device__ __forceinline__ bool lowe_candidate(const float *__restrict__ blur0, long plane, int w,
                                               int p, int h, int S, float T, float r, float pre,
                                               float sigma, float pix, int x, int y, int i,
                                               LoweHit &out)
{
    int c = i + 1;
    float d[3][3][3];
    // The pixel's own row of the three DoG planes first (four cache lines instead of
    // twelve): most candidates already fail against these eight neighbours.
    {
        const float *q = blur0 + (c - 1) * plane + x + (long)p * y;
        float3u t[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) t[s] = *reinterpret_cast<const float3u *>(q + s * plane - 1);
        float e[3][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            e[s][0] = __fsub_rn(t[s].a, t[s + 1].a);
            e[s][1] = __fsub_rn(t[s].b, t[s + 1].b);
            e[s][2] = __fsub_rn(t[s].c, t[s + 1].c);
        }
        const float v0 = e[1][1];
        if (!(fabsf(v0) > pre)) return false;
        bool above = true, below = true;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int l = 0; l < 3; ++l)
                if (s != 1 || l != 1) {
                    above = above && v0 > e[s][l];
                    below = below && v0 < e[s][l];
                }
        if (!above && !below) return false;
    }
    lowe_cube(blur0, plane, p, c, x, y, d);
    float v = d[1][1][1];
    bool above = true, below = true;
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int l = 0; l < 3; ++l)
                if (s != 1 || m != 1 || l != 1) {
                    above = above && v > d[s][m][l];
                    below = below && v < d[s][m][l];
                }
    if (!above && !below) return false;
    float Xx = 0.f, Xy = 0.f, Xs = 0.f, gx = 0.f, gy = 0.f, gs = 0.f, dxx = 0.f, dyy = 0.f, dxy = 0.f;
    bool ok = false;
    for (int it = 0; it < 5; ++it) {
        if (it) lowe_cube(blur0, plane, p, c, x, y, d);
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
        if (det == 0.0f) break;
        Xx = -(c00 * gx + c01 * gy + c02 * gs) / det;
        Xy = -(c01 * gx + c11 * gy + c12 * gs) / det;
        Xs = -(c02 * gx + c12 * gy + c22 * gs) / det;
        if (fabsf(Xx) < 0.5f && fabsf(Xy) < 0.5f && fabsf(Xs) < 0.5f) {
            ok = true;
            break;
        }
        if (!(fabsf(Xx) < 1e6f && fabsf(Xy) < 1e6f && fabsf(Xs) < 1e6f)) break; // also catches NaN
        x += (int)floorf(Xx + 0.5f);
        y += (int)floorf(Xy + 0.5f);
        c += (int)floorf(Xs + 0.5f);
        if (c < 1 || c > S || x <= 8 || x >= w - 8 || y <= 8 || y >= h - 8) break;
    }
    if (!ok) return false;
    const float dhat = v + 0.5f * (gx * Xx + gy * Xy + gs * Xs);
    if (fabsf(dhat) * (float)S < T) return false;
    const float tr = dxx + dyy, det2 = dxx * dyy - dxy * dxy;
    if (det2 <= 0.0f || tr * tr * r >= (r + 1.0f) * (r + 1.0f) * det2) return false;
    out.xy = (unsigned)x | ((unsigned)y << 16);
    out.sub.x = (float)((double)((float)x + Xx) * (double)pix);
    out.sub.y = (float)((double)((float)y + Xy) * (double)pix);
    out.sub.z = (float)((double)sigma * (double)pix * det_exp2((double)((float)(c - 1) + Xs) / (double)S));
    out.sub.w = __int_as_float(c);
    return true;
}

// wave-aggregated append of a point and its refined values
__device__ __forceinline__ void append_lowe(bool hit, const LoweHit &hv, int lane, unsigned *count,
                                            unsigned *list, float4 *sub, int cap)
{
    const unsigned long long m = __ballot(hit);
    if (m) {
        const int leader = __ffsll((long long)m) - 1;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(count, (unsigned)__popcll(m));
        base = __shfl(base, leader);
        const unsigned idx = base + __popcll(m & ((1ull << lane) - 1ull));
        if (hit && idx < (unsigned)cap) {
            list[idx] = hv.xy;
            sub[idx] = hv.sub;
        }
    }
}

// Queue-driven: one thread per candidate queued by pyramid_kernel.  The queues
// of all (octave, scale) groups of an image form one list of 64-candidate items,
// which the image's waves (gridDim.x blocks of one wave) take in turn: the launch
// has the same few thousand waves whatever the images hold, and each wave's share
// is even.  (A block per group and queue segment is mostly blocks that find
// nothing to do, and their dispatch alone outlasts the useful work.)
template <int S>
__global__ __launch_bounds__(64) void verify_lowe_kernel(
    const float *__restrict__ arena, long arena_stride, const unsigned *queue, long queue_stride,
    const unsigned *qcount, long qcount_stride, unsigned *cand, float4 *sub, long cand_stride,
    unsigned *counts, long counts_stride, hl_detect_args a)
{
    const int lane = threadIdx.x;
    const int z = blockIdx.y;
    qcount += z * qcount_stride;
    const int n_groups = a.n_oct * S;
    // first[g]: index of group g's first item (an overflowed queue has none: the dense
    // kernel takes that group).  Lane g counts group g, a wave scan adds them up.
    __shared__ unsigned first[HL_MAX_GROUPS + 1];
    {
        unsigned items = 0;
        if (lane < n_groups) {
            const unsigned n = qcount[lane];
            items = n > (unsigned)a.oct[lane / S].qcap ? 0u : (n + 63u) / 64u;
        }
#pragma unroll
        for (int off = 1; off < HL_MAX_GROUPS; off <<= 1) {
            const unsigned up = __shfl_up(items, off);
            if (lane >= off) items += up;
        }
        if (lane == 0) first[0] = 0;
        if (lane < HL_MAX_GROUPS) first[lane + 1] = items;
        __syncthreads();
    }
    const unsigned total = first[n_groups];
    int g = 0;
    for (unsigned item = blockIdx.x; item < total; item += gridDim.x) { // wave-uniform
        while (item >= first[g + 1]) ++g;
        const int oi = g / S, i = g - oi * S;
        const hl_octave oc = a.oct[oi];
        const unsigned n = qcount[g];
        const long plane = (long)oc.p * oc.h;
        const float *blur0 = arena + z * arena_stride + oc.blur_off;
        const unsigned e = (item - first[g]) * 64u + lane;
        bool hit = false;
        LoweHit hv;
        hv.xy = 0;
        hv.sub = float4{0.f, 0.f, 0.f, 0.f};
        if (e < n) {
            const unsigned xy = queue[z * queue_stride + oc.queue_off + (long)i * oc.qcap + e];
            hit = lowe_candidate(blur0, plane, oc.w, oc.p, oc.h, S, a.contrast, a.r_thresh, a.pre_thresh,
                                 a.sigma, a.pix[oi], (int)(xy & 0xffffu), (int)(xy >> 16), i, hv);
        }
        append_lowe(hit, hv, lane, counts + z * counts_stride + g, cand + z * cand_stride + (long)g * a.cap,
                    sub + z * cand_stride + (long)g * a.cap, a.cap);
    }
}

// Dense form of the same detector, for the groups whose queue overflowed (gate)
// or for everything (gate NULL).  A block walks over 96 x 32 pixel units.
template <int S>
__global__ __launch_bounds__(256) void detect_lowe_kernel(
    const float *__restrict__ arena, long arena_stride, unsigned *cand, float4 *sub, long cand_stride,
    unsigned *counts, long counts_stride, const unsigned *gate, long gate_stride, hl_detect_args a)
{
    const int lane = threadIdx.x & 63;
    const int z = blockIdx.z;
    cand += z * cand_stride;
    sub += z * cand_stride;
    counts += z * counts_stride;
    if (gate) {
        gate += z * gate_stride;
        bool any = false;
        for (int g = 0; g < a.n_oct * S; ++g) any = any || gate[g] > (unsigned)a.oct[g / S].qcap;
        if (!any) return; // block-uniform
    }
    for (int unit = blockIdx.x; unit < a.dense0[a.n_oct]; unit += gridDim.x) {
        const int oi = find_octave(a.dense0, a.n_oct, unit);
        const hl_octave oc = a.oct[oi];
        const int w = oc.w, h = oc.h, p = oc.p;
        const int xblocks = (w + XT_DATAX - 1) / XT_DATAX; // the units are the dense detector's 96 x 32 blocks
        const int u = unit - a.dense0[oi];
        const int x = threadIdx.x < XT_DATAX ? (u % xblocks) * XT_DATAX + threadIdx.x : w;
        const int yb = (u / xblocks) * 32;
        const long plane = (long)p * h;
        const float *blur0 = arena + z * arena_stride + oc.blur_off;
        for (int i = 0; i < S; ++i) {
            const int g = oi * S + i;
            if (gate && gate[g] <= (unsigned)oc.qcap) continue; // block-uniform
            for (int ry = 0; ry < 32; ++ry) { // wave-uniform trip count
                const int y = yb + ry;
                bool hit = false;
                LoweHit hv;
                hv.xy = 0;
                hv.sub = float4{0.f, 0.f, 0.f, 0.f};
                if (x > 8 && x < w - 8 && y > 8 && y < h - 8) {
                    // contrast, then the extremum along the row and along the column: each
                    // test is necessary and discards most of what reaches it, before the
                    // 27-cell cube is fetched
                    const float *c = blur0 + (i + 1) * plane + x + (long)p * y;
                    const float v = __fsub_rn(c[0], c[plane]);
                    if (fabsf(v) > a.pre_thresh) {
                        const float l = __fsub_rn(c[-1], c[plane - 1]), r = __fsub_rn(c[1], c[plane + 1]);
                        const bool is_max = v > l && v > r, is_min = v < l && v < r;
                        if (is_max || is_min) {
                            const float u = __fsub_rn(c[-p], c[plane - p]), d = __fsub_rn(c[p], c[plane + p]);
                            if ((is_max && v > u && v > d) || (is_min && v < u && v < d))
                                hit = lowe_candidate(blur0, plane, w, p, h, S, a.contrast, a.r_thresh,
                                                     a.pre_thresh, a.sigma, a.pix[oi], x, y, i, hv);
                        }
                    }
                }
                append_lowe(hit, hv, lane, &counts[g], cand + (long)g * a.cap, sub + (long)g * a.cap, a.cap);
            }
        }
    }
}

// First octave of the "lowe" profile with up-sampling: dst (2w x 2h) holds src
// at its even samples and the mean of the two or four even neighbours at the
// odd ones; the last odd row and column repeat the edge.  T = float or
// unsigned char source.
template <class T>
__global__ __launch_bounds__(256) void upsample2x_kernel(const T *__restrict__ src, int src_pitch,
                                                         long src_stride, float *__restrict__ dst,
                                                         int dst_pitch, long dst_stride, int w, int h)
{
    const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y;
    if (X >= 2 * w) return;
    src += blockIdx.z * src_stride;
    dst += blockIdx.z * dst_stride;
    const int x0 = X >> 1, y0 = Y >> 1;
    const int x1 = (X & 1) ? (x0 + 1 < w ? x0 + 1 : x0) : x0;
    const int y1 = (Y & 1) ? (y0 + 1 < h ? y0 + 1 : y0) : y0;
    const float a = (float)src[x0 + (long)src_pitch * y0], b = (float)src[x1 + (long)src_pitch * y0];
    const float c = (float)src[x0 + (long)src_pitch * y1], d = (float)src[x1 + (long)src_pitch * y1];
    dst[X + (long)dst_pitch * Y] = __fmul_rn(__fadd_rn(__fadd_rn(a, b), __fadd_rn(c, d)), 0.25f);
}

// ---------------------------------------------------------------------------
// Gradient sample (gradientKernel cudaSiftD.cu:349-356).  atan2 is evaluated
// in double with a fixed sequence of IEEE operations and rounded once to
// float, so the same source gives the same bits on the host.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double atan_small(double t) // |t| < 1/8
{
    const double s = t * t;
    double q = 1.0 / 17.0;
    q = fma(q, s, -1.0 / 15.0);
    q = fma(q, s, 1.0 / 13.0);
    q = fma(q, s, -1.0 / 11.0);
    q = fma(q, s, 1.0 / 9.0);
    q = fma(q, s, -1.0 / 7.0);
    q = fma(q, s, 1.0 / 5.0);
    q = fma(q, s, -1.0 / 3.0);
    return fma(t * s, q, t);
}

// With a = min/max of |x|,|y| and c the nearest of 0, 1/4, 1/2, 3/4, 1:
// atan(a) = atan(c) + atan(t), t = (a - c)/(1 + a c) = (min - c max)/(max + c min),
// |t| < 0.1245: one division and a series up to t^17.  `tab(j, c, base)` supplies
// c = j/4 and atan(c).  Evaluated without a branch: for x = y = 0 the quotient is
// 0/0 and the result is replaced by 0 afterwards.
template <class Tab>
__device__ __forceinline__ float det_atan2f_with(float yf, float xf, Tab tab)
{
    const double x = (double)xf, y = (double)yf;
    const double ax = fabs(x), ay = fabs(y);
    const double mx = ax > ay ? ax : ay;
    const double mn = ax > ay ? ay : ax;
    const double m8 = 8.0 * mn;
    const int j = (m8 >= mx) + (m8 >= 3.0 * mx) + (m8 >= 5.0 * mx) + (m8 >= 7.0 * mx);
    double c, base;
    tab(j, c, base);
    const double t = fma(-c, mx, mn) / fma(c, mn, mx);
    double r = base + atan_small(t);
    r = mx != 0.0 ? r : 0.0;
    if (ay > ax) r = 1.57079632679489662 - r;
    if (x < 0.0) r = 3.14159265358979323846 - r;
    if (y < 0.0) r = -r;
    return (float)r;
}

constexpr double ATAN_QUARTERS[5] = {0.0, 0.24497866312686414, 0.46364760900080609, 0.64350110879328437,
                                     0.78539816339744831};

__device__ float det_atan2f(float yf, float xf)
{
    return det_atan2f_with(yf, xf, [](int j, double &c, double &base) {
        c = 0.25 * (double)j;
        base = j == 0 ? ATAN_QUARTERS[0]
               : j == 1 ? ATAN_QUARTERS[1]
               : j == 2 ? ATAN_QUARTERS[2]
               : j == 3 ? ATAN_QUARTERS[3]
                        : ATAN_QUARTERS[4];
    });
}

__device__ __forceinline__ void grad_from(float up, float down, float left, float right, float &mag,
                                          float &dir);
__device__ __forceinline__ void grad_at(const float *src, int w, int p, int h, int x, int y,
                                        float &mag, float &dir)
{
    grad_from(zpad(src, w, p, h, x, y - 1), zpad(src, w, p, h, x, y + 1), zpad(src, w, p, h, x - 1, y),
              zpad(src, w, p, h, x + 1, y), mag, dir);
}
__device__ __forceinline__ void grad_from(float up, float down, float left, float right, float &mag,
                                          float &dir)
{
    const float dx = right - left;
    const float dy = down - up;
    mag = sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy))); // correctly rounded (the _rn intrinsic is the fast approximation)
    dir = (float)((180 / 3.14159265) * (double)det_atan2f(dy, dx)); // PI of utils.h:4
}
// the same with c and atan(c) read from a table of five (c, atan c) pairs in LDS
__device__ __forceinline__ void grad_from(float up, float down, float left, float right, const double2 *tab,
                                          float &mag, float &dir)
{
    const float dx = right - left;
    const float dy = down - up;
    mag = sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    const float rad = det_atan2f_with(dy, dx, [tab](int j, double &c, double &base) {
        const double2 e = tab[j];
        c = e.x;
        base = e.y;
    });
    dir = (float)((180 / 3.14159265) * (double)rad);
}

// cudaAssignBin cudaSiftD.h:35-41; theta is within [-360, 360]
__device__ __forceinline__ int assign_bin(float theta, float bin_range)
{
    theta = (theta < 0.f) ? (theta + 360.f) : theta;
    theta = (theta >= 360.f) ? (theta - 360.f) : theta; // fmodf(theta, 360) on [0, 720)
    return (int)floorf(__fdiv_rn(theta, bin_range));
}

struct SiftPointD {
    float xpos, ypos, scale, orientation;
    float data[128];
};

constexpr int REG = 23; // gradient region [x-11, x+11]: covers the rotated 16x16 patch
#ifndef KP_RSTRIDE
#define KP_RSTRIDE 23
#endif
constexpr int RSTRIDE = KP_RSTRIDE; // floats between rows of the region in LDS

// The descriptor's rotated 16x16 samples read, through their bilinear corners,
// only part of the 23x23 region, and which part depends on the orientation
// alone.  For each of the 32 orientations: the region entries outside the
// upright 16x16 patch (entries [4,19] x [4,19], which the orientation histogram
// needs anyway) that some sample can touch.  A sample's corner is
// floor(x + u) - x with the sum rounded to float; for x < 2^16 that rounding
// moves the sum by less than 2^-8, and the float evaluation of u by far less,
// so the cells of floor(u - 1/64) and floor(u + 1/64) cover every case.
constexpr int PATCH0 = 4, PATCH1 = 19; // the upright patch inside the region
constexpr int RING_MAX = 96;
struct RingTable {
    unsigned short n[32];
    unsigned short e[32][RING_MAX];
};
constexpr double cx_sin(double x) // |x| <= pi
{
    double term = x, sum = x;
    for (int k = 1; k < 18; ++k) {
        term *= -x * x / (double)((2 * k) * (2 * k + 1));
        sum += term;
    }
    return sum;
}
constexpr double cx_cos(double x)
{
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 18; ++k) {
        term *= -x * x / (double)((2 * k - 1) * (2 * k));
        sum += term;
    }
    return sum;
}
constexpr int cx_floor(double v)
{
    const int i = (int)v;
    return (double)i > v ? i - 1 : i;
}
constexpr int cx_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
constexpr RingTable make_ring_table()
{
    RingTable t{};
    const double pi = 3.14159265358979323846, margin = 1.0 / 64.0;
    for (int o = 0; o < 32; ++o) {
        double rad = 11.25 * o * pi / 180.0;
        if (rad > pi) rad -= 2.0 * pi;
        const double cs = cx_cos(rad), sn = cx_sin(rad);
        bool used[REG * REG] = {};
        for (int ty = 0; ty < 16; ++ty)
            for (int tx = 0; tx < 16; ++tx) {
                const double dx = tx - 7.5, dy = ty - 7.5;
                const double u = dx * cs + dy * sn, v = dy * cs - dx * sn;
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b) {
                        const int rx = cx_clamp(cx_floor(u + (a ? margin : -margin)) + 11, 0, REG - 2);
                        const int ry = cx_clamp(cx_floor(v + (b ? margin : -margin)) + 11, 0, REG - 2);
                        used[ry * REG + rx] = used[ry * REG + rx + 1] = true;
                        used[(ry + 1) * REG + rx] = used[(ry + 1) * REG + rx + 1] = true;
                    }
            }
        int n = 0;
        for (int e = 0; e < REG * REG; ++e) {
            const int ry = e / REG, rx = e - ry * REG;
            const bool in_patch = ry >= PATCH0 && ry <= PATCH1 && rx >= PATCH0 && rx <= PATCH1;
            if (used[e] && !in_patch) {
                if (n < RING_MAX) t.e[o][n] = (unsigned short)e;
                ++n;
            }
        }
        t.n[o] = (unsigned short)n;
    }
    return t;
}
constexpr RingTable RING_HOST = make_ring_table();
constexpr bool ring_fits()
{
    for (int o = 0; o < 32; ++o)
        if (RING_HOST.n[o] > RING_MAX) return false;
    return true;
}
static_assert(ring_fits(), "RING_MAX is too small for some orientation");
__device__ const RingTable g_ring = RING_HOST;

// One wave per keypoint.  Float sums whose order the reference leaves to
// atomics are taken in one fixed order built from per-lane partial sums.  Lane
// L owns the four samples (ty, 4*(L&3) .. 4*(L&3)+3) of row ty = L >> 2 and
// adds those that fall into one bin in ascending tx.  An orientation bin is
// (partials of lanes 0..31 added in ascending order) + (the same for lanes
// 32..63); a descriptor bin is ((P0 + P1) + P2) + P3 over the four rows of its
// cell.
constexpr int OH_STRIDE = 33; // orientation partials: 64 lanes x 32 bins, padded against bank conflicts
constexpr int DH_STRIDE = 9;  // descriptor partials: 64 lanes x 8 bins

// the value lane-sequential accumulation leaves in each sample's bin: sample k
// continues the running sum of the latest earlier sample with the same bin
__device__ __forceinline__ void chain4(const int (&b)[4], const float (&w)[4], float (&o)[4])
{
    o[0] = w[0];
    o[1] = __fadd_rn(b[1] == b[0] ? o[0] : 0.f, w[1]);
    o[2] = __fadd_rn(b[2] == b[1] ? o[1] : (b[2] == b[0] ? o[0] : 0.f), w[2]);
    o[3] = __fadd_rn(b[3] == b[2] ? o[2] : (b[3] == b[1] ? o[1] : (b[3] == b[0] ? o[0] : 0.f)), w[3]);
}

__device__ __forceinline__ double ATAN_QUARTERS_DEV(int j)
{
    return j == 0 ? ATAN_QUARTERS[0] : j == 1 ? ATAN_QUARTERS[1] : j == 2 ? ATAN_QUARTERS[2]
           : j == 3 ? ATAN_QUARTERS[3] : ATAN_QUARTERS[4];
}

// The four neighbours of region entry (rx, ry) of the keypoint at (x, y): the
// entry stands for pixel clamp(x - 11 + rx), clamp(y - 11 + ry), and a neighbour
// outside the image counts as 0 (gradientKernel cudaSiftD.cu:336-347).  The loads
// use clamped addresses; region_values replaces what lay outside.
template <bool INTERIOR>
__device__ __forceinline__ void region_loads(const float *bl, int w, int p, int h, int x, int y, int rx,
                                             int ry, float (&nb)[4])
{
    if (INTERIOR) {
        const float *c = bl + (x - 11 + rx) + (long)p * (y - 11 + ry);
        nb[0] = c[-p];
        nb[1] = c[p];
        nb[2] = c[-1];
        nb[3] = c[1];
    } else {
        const int px = min(max(x - 11 + rx, 0), w - 1);
        const int py = min(max(y - 11 + ry, 0), h - 1);
        const float *c = bl + px + (long)p * py;
        nb[0] = c[py > 0 ? -p : 0];
        nb[1] = c[py < h - 1 ? p : 0];
        nb[2] = c[px > 0 ? -1 : 0];
        nb[3] = c[px < w - 1 ? 1 : 0];
    }
}
template <bool INTERIOR>
__device__ __forceinline__ void region_values(int w, int h, int x, int y, int rx, int ry, float (&nb)[4])
{
    if (!INTERIOR) {
        const int px = min(max(x - 11 + rx, 0), w - 1);
        const int py = min(max(y - 11 + ry, 0), h - 1);
        nb[0] = py > 0 ? nb[0] : 0.f;
        nb[1] = py < h - 1 ? nb[1] : 0.f;
        nb[2] = px > 0 ? nb[2] : 0.f;
        nb[3] = px < w - 1 ? nb[3] : 0.f;
    }
}

template <bool INTERIOR>
__device__ __forceinline__ void region_patch(const float *bl, int w, int p, int h, int x, int y, int lane,
                                             const double2 *tab, float *s_mag, float *s_dir)
{
    float nb[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = lane + 64 * q;
        region_loads<INTERIOR>(bl, w, p, h, x, y, PATCH0 + (i & 15), PATCH0 + (i >> 4), nb[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = lane + 64 * q;
        const int rx = PATCH0 + (i & 15), ry = PATCH0 + (i >> 4);
        region_values<INTERIOR>(w, h, x, y, rx, ry, nb[q]);
        float m, d;
        grad_from(nb[q][0], nb[q][1], nb[q][2], nb[q][3], tab, m, d);
        s_mag[ry * RSTRIDE + rx] = m;
        s_dir[ry * RSTRIDE + rx] = d;
    }
}

template <bool INTERIOR>
__device__ __forceinline__ void region_ring(const float *bl, int w, int p, int h, int x, int y, int lane,
                                            int oidx, const double2 *tab, float *s_mag, float *s_dir)
{
    const int n = g_ring.n[oidx];
    for (int i0 = 0; i0 < n; i0 += 64) { // one or two turns
        const int i = i0 + lane;
        if (i < n) {
            const int e = g_ring.e[oidx][i];
            const int ry = e / REG, rx = e - ry * REG;
            float nb[4];
            region_loads<INTERIOR>(bl, w, p, h, x, y, rx, ry, nb);
            region_values<INTERIOR>(w, h, x, y, rx, ry, nb);
            float m, d;
            grad_from(nb[0], nb[1], nb[2], nb[3], tab, m, d);
            s_mag[ry * RSTRIDE + rx] = m;
            s_dir[ry * RSTRIDE + rx] = d;
        }
    }
}

__global__ __launch_bounds__(64) void keypoint_kernel(
    SiftPointD *pts, long pts_stride, unsigned char *meta, const float *arena, long arena_stride,
    const unsigned *cand, const float4 *sub, long cand_stride, const unsigned *counts,
    long counts_stride, const unsigned *base_count, hl_keypoint_args a)
{
    __shared__ float s_mag[REG * RSTRIDE + 3];
    __shared__ float s_dir[REG * RSTRIDE + 3];
    // partial sums: 32 lanes x 32 orientation bins (the two lane halves take turns),
    // later 64 lanes x 8 descriptor bins.  Kept small: the kernel runs as many waves as
    // its LDS allows, and it needs them to hide its memory round trips.
    __shared__ __attribute__((aligned(16))) float s_part[64 * 20];
    __shared__ double2 s_tab[5]; // (j/4, atan(j/4))

    const int lane = threadIdx.x;
    if (lane < 5) s_tab[lane] = double2{0.25 * (double)lane, ATAN_QUARTERS_DEV(lane)};
    const int z = blockIdx.y;
    pts += z * pts_stride;
    if (meta) meta += z * pts_stride * 2;
    arena += z * arena_stride;
    cand += z * cand_stride;
    if (sub) sub += z * cand_stride;
    counts += z * counts_stride;
    const unsigned base = base_count[z];
    const unsigned cap = (unsigned)a.cap;

    for (unsigned j = blockIdx.x;; j += gridDim.x) {
        if (base + j >= cap) break;
        int g = -1;
        unsigned start = 0;
        for (int gg = 0; gg < a.n_groups; ++gg) {
            const unsigned c = min(counts[gg], cap);
            if (j < start + c) {
                g = gg;
                break;
            }
            start += c;
        }
        if (g < 0) break;
        const unsigned local = j - start;
        const int oi = g / a.n_scales;
        int si = g - oi * a.n_scales;
        const hl_octave oc = a.oct[oi];
        const unsigned xy = cand[(long)g * cap + local];
        // "lowe" profile: position, scale and the plane the point ended on come refined
        float4 rf = float4{0.f, 0.f, 0.f, 0.f};
        if (sub) {
            rf = sub[(long)g * cap + local];
            si = __float_as_int(rf.w) - 1;
        }
        const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
        const int w = oc.w, h = oc.h, p = oc.p;
        const float *bl = arena + oc.blur_off + (long)(si + 1) * p * h; // blur plane si+1

        __syncthreads(); // previous keypoint's readers are done with the LDS arrays
#ifdef SIFT_POISON_REGION
        // test builds only: an entry that is read without having been computed for
        // this keypoint turns the descriptor into NaNs
        for (int e = lane; e < REG * RSTRIDE; e += 64) s_mag[e] = s_dir[e] = __builtin_nanf("");
        __syncthreads();
#endif
        // away from the border no address needs clamping and no value replacing
        const bool interior = x >= 12 && x + 12 <= w - 1 && y >= 12 && y + 12 <= h - 1;
        // the upright 16x16 patch first: four entries per lane, all 16 neighbour loads
        // issued before the first is used (one memory round trip)
        if (interior) region_patch<true>(bl, w, p, h, x, y, lane, s_tab, s_mag, s_dir);
        else region_patch<false>(bl, w, p, h, x, y, lane, s_tab, s_mag, s_dir);
        __syncthreads();

        // ---- orientation: 16x16 patch [x-7,x+8] x [y-7,y+8], 32 bins ----
        const int ty = lane >> 2, tx0 = 4 * (lane & 3);
        float hsum;
        {
            int ob[4];
            float ow[4], oo[4];
            const float wy = a.wnd[si * 16 + ty];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = (ty + 4) * RSTRIDE + tx0 + k + 4;
                ow[k] = __fmul_rn(__fmul_rn(s_mag[e], a.wnd[si * 16 + tx0 + k]), wy);
                ob[k] = assign_bin(s_dir[e], 11.25f);
            }
            chain4(ob, ow, oo);
            // lanes 0..31 leave their partials and every lane adds up its bin's column,
            // then the same for lanes 32..63: bin = (first half) + (second half)
            float hh[2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int q = 0; q < 5; ++q) reinterpret_cast<float4 *>(s_part)[lane + 64 * q] = float4{0.f, 0.f, 0.f, 0.f};
                __syncthreads(); // the zeroes are in place
                if ((lane >> 5) == half) {
#ifndef KP_EXP_NO_ORI_SCATTER // (experiment switches: attribute the LDS bank conflicts; results are wrong with them)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        s_part[(lane & 31) * OH_STRIDE + ob[k]] = oo[k]; // in order: the last write of a bin wins
#endif
                }
                __syncthreads();
                const float *col = s_part + (lane & 31);
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < 32; ++i) acc = __fadd_rn(acc, col[i * OH_STRIDE]);
                hh[half] = acc;
                __syncthreads(); // the column sums are taken before the next zeroing
            }
            hsum = __fadd_rn(hh[0], hh[1]);
        }
        float mx = hsum;
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 8));
        mx = fmaxf(mx, __shfl_xor(mx, 4));
        mx = fmaxf(mx, __shfl_xor(mx, 2));
        mx = fmaxf(mx, __shfl_xor(mx, 1));
        const unsigned long long mm = __ballot(lane < 32 && hsum == mx); // (lanes 32..63 hold the same 32 bins again)
        const int oidx = mm ? (__ffsll((long long)mm) - 1) : 0; // first maximum wins
        const float orientation = 11.25f * (float)oidx;
        // the entries around the patch that this orientation's descriptor samples touch
        if (interior) region_ring<true>(bl, w, p, h, x, y, lane, oidx, s_tab, s_mag, s_dir);
        else region_ring<false>(bl, w, p, h, x, y, lane, oidx, s_tab, s_mag, s_dir);
        __syncthreads();

        // ---- descriptor: rotated 16x16 samples, bilinear, 4x4 cells x 8 bins ----
        const float cs = a.cos32[oidx], sn = a.sin32[oidx];
        {
            int db[4];
            float dwt[4], dout[4];
            const float dy = (float)ty - 7.5f;
            const float wy = a.desc_wnd[ty];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int tx = tx0 + k;
                const float dx = (float)tx - 7.5f;
                const float u = __fadd_rn(__fmul_rn(dx, cs), __fmul_rn(dy, sn));
                const float v = __fsub_rn(__fmul_rn(dy, cs), __fmul_rn(dx, sn));
                const float px = __fadd_rn((float)x, u), py = __fadd_rn((float)y, v);
                const float fx = floorf(px), fy = floorf(py);
                const float ax = __fsub_rn(px, fx), ay = __fsub_rn(py, fy);
                // region entry r holds the value of pixel clamp(x-11+r), so the
                // clamp-to-edge addressing needs no further clamping here
                const int rx = min(max((int)fx - (x - 11), 0), REG - 2);
                const int ry = min(max((int)fy - (y - 11), 0), REG - 2);
#ifdef KP_EXP_NO_GATHER
                const int e = lane + 0 * (rx + ry);
#else
                const int e = ry * RSTRIDE + rx;
#endif
                float val[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float *pl = c ? s_dir : s_mag;
                    const float v00 = pl[e], v10 = pl[e + 1], v01 = pl[e + RSTRIDE], v11 = pl[e + RSTRIDE + 1];
                    const float top = __builtin_fmaf(ax, __fsub_rn(v10, v00), v00);
                    const float bot = __builtin_fmaf(ax, __fsub_rn(v11, v01), v01);
                    val[c] = __builtin_fmaf(ay, __fsub_rn(bot, top), top);
                }
                dwt[k] = __fmul_rn(__fmul_rn(val[0], a.desc_wnd[tx]), wy);
                db[k] = assign_bin(val[1], 45.f);
            }
            chain4(db, dwt, dout);
            // the orientation partials have been consumed (the ballot above joins the wave)
#pragma unroll
            for (int q = 0; q < 3; ++q) reinterpret_cast<float4 *>(s_part)[lane + 64 * q] = float4{0.f, 0.f, 0.f, 0.f};
            __syncthreads();
#ifndef KP_EXP_NO_DESC_SCATTER
#pragma unroll
            for (int k = 0; k < 4; ++k) s_part[lane * DH_STRIDE + db[k]] = dout[k];
#endif
        }
        __syncthreads();
        float hv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = lane + 64 * q;
            const int cy = e >> 5, cx = (e >> 3) & 3, b = e & 7;
            const float *row = s_part + ((16 * cy + cx) * DH_STRIDE + b); // lane (4*cy + r)*4 + cx
            float sum = row[0];
#pragma unroll
            for (int r = 1; r < 4; ++r) sum = __fadd_rn(sum, row[r * 4 * DH_STRIDE]);
            hv[q] = sum;
        }
        // L1 normalise, clamp at 0.2, L1 normalise (cudaSiftD.cu:186-200); the
        // 32-lane tree is shflsum (cudaSiftD.h:96-106), the four partial sums
        // are added in index order
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            float t0 = hv[0], t1 = hv[1];
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) {
                t0 = __fadd_rn(t0, __shfl_down(t0, off, 32));
                t1 = __fadd_rn(t1, __shfl_down(t1, off, 32));
            }
            float tot = __fadd_rn(__shfl(t0, 0), __shfl(t0, 32));
            tot = __fadd_rn(tot, __shfl(t1, 0));
            tot = __fadd_rn(tot, __shfl(t1, 32));
            hv[0] = __fdiv_rn(hv[0], tot);
            hv[1] = __fdiv_rn(hv[1], tot);
            if (pass == 0) {
                hv[0] = (hv[0] >= 0.2f) ? 0.2f : hv[0];
                hv[1] = (hv[1] >= 0.2f) ? 0.2f : hv[1];
            }
        }
        SiftPointD *out = pts + (base + j);
        out->data[lane] = hv[0];
        out->data[lane + 64] = hv[1];
        if (lane == 0) {
            out->xpos = sub ? rf.x : (float)x;
            out->ypos = sub ? rf.y : (float)y;
            out->scale = sub ? rf.z : a.scale_value[g];
            out->orientation = orientation;
            if (meta) {
                meta[2 * (base + j)] = (unsigned char)(a.first_octave + oi);
                meta[2 * (base + j) + 1] = (unsigned char)si;
            }
        }
    }
}

// 8-bit descriptors: min(255, round-half-up(512 v)), 128 per point
__global__ __launch_bounds__(128) void quantize_kernel(const SiftPointD *pts, int n, unsigned char *out)
{
    const int j = blockIdx.x;
    if (j >= n) return;
    const float q = floorf(__fadd_rn(__fmul_rn(pts[j].data[threadIdx.x], 512.0f), 0.5f));
    out[128L * j + threadIdx.x] = (unsigned char)(q > 255.0f ? 255.0f : q);
}

__global__ void finalize_counts_kernel(unsigned *base_count, unsigned *host_copy, const unsigned *counts,
                                        long counts_stride, int n_groups, int cap, int nimg)
{
    const int z = blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= nimg) return;
    unsigned long long tot = base_count[z];
    for (int g = 0; g < n_groups; ++g) tot += min(counts[z * counts_stride + g], (unsigned)cap);
    const unsigned v = (unsigned)min(tot, (unsigned long long)cap);
    base_count[z] = v;
    if (host_copy) host_copy[z] = v; // pinned host memory: there when the stream has drained
}

// zeroes two word ranges in one launch (the group counters of a launch set and its point counts)
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned *a, long na, unsigned *b, long nb)
{
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += step) {
        if (i < na) a[i] = 0u;
        else b[i - na] = 0u;
    }
}

// ---- dense stage kernels (one pixel per lane) -------------------------------
__global__ __launch_bounds__(256) void subtract_kernel(float *dst, const float *a, const float *b,
                                                       int w, int p, int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < w && y < h) dst[x + (long)p * y] = a[x + (long)p * y] - b[x + (long)p * y];
}

__global__ __launch_bounds__(256) void copy_kernel(float *dst, const float *src, int w, int p, int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < w && y < h) dst[x + (long)p * y] = src[x + (long)p * y];
}

__global__ __launch_bounds__(256) void hessian_kernel(float *dst, const float *src, int w, int p,
                                                      int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const long plane = (long)p * h;
    const float c = src[x + (long)p * y];
    float Dxx = -2.f * c;
    Dxx += zpad(src, w, p, h, x - 1, y);
    Dxx += zpad(src, w, p, h, x + 1, y);
    float Dyy = -2.f * c;
    Dyy += zpad(src, w, p, h, x, y - 1);
    Dyy += zpad(src, w, p, h, x, y + 1);
    float Dxy = c;
    Dxy += zpad(src, w, p, h, x - 1, y - 1);
    Dxy -= zpad(src, w, p, h, x - 1, y);
    Dxy -= zpad(src, w, p, h, x, y - 1);
    const long g = x + (long)p * y;
    dst[g] = Dxx;
    dst[g + plane] = Dyy;
    dst[g + 2 * plane] = Dxy;
}

__global__ __launch_bounds__(256) void gradient_kernel(float *dst, const float *src, int w, int p,
                                                       int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    float m, d;
    grad_at(src, w, p, h, x, y, m, d);
    const long g = x + (long)p * y;
    dst[g] = m;
    dst[g + (long)p * h] = d;
}

// 8-bit grey pixels to floats, four per lane where the row allows
__global__ __launch_bounds__(256) void u8_to_f32_kernel(float *__restrict__ dst, int dp, const unsigned char *__restrict__ src,
                                                        int sp, int w, int h)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int y = blockIdx.y;
    if (x >= w || y >= h) return;
    const unsigned char *s = src + (long)sp * y + x;
    float *d = dst + (long)dp * y + x;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (x + k < w) d[k] = (float)s[k];
}

// ---- static scratch -------------------------------------------------------------
// Scratch of the entry point that allocates nothing (sift_extract_noalloc): blur planes,
// counters, candidate queues and lists of ONE image.  Every GPU the library runs on holds
// its own copy of the array, at its own address, until the process ends.  In host code the
// array's name is only a handle for hipGetSymbolAddress: the address that kernels, memsets
// and copies use is the one looked up below, per device.
constexpr size_t SCRATCH_BYTES = 32u << 20;
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];

constexpr int SCRATCH_MAX_DEV = 16;
std::mutex g_scratch_mu;
void *g_scratch_ptr[SCRATCH_MAX_DEV]; // per device, nullptr until looked up

// ---- device copy ceiling ------------------------------------------------------
// A plain device-to-device copy, 16 bytes per lane, each block a contiguous 16 KB: the
// rate this GPU's memory system gives a kernel that only moves bytes.  Bandwidth-bound
// kernels are priced against it beside the data-sheet figure.
typedef float copy_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_ceiling_kernel(copy_f4 *__restrict__ out, const copy_f4 *__restrict__ in, long n)
{
    const long base = (long)blockIdx.x * 1024 + threadIdx.x;
    copy_f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (base + u * 256 < n) v[u] = in[base + u * 256];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (base + u * 256 < n) out[base + u * 256] = v[u];
}

// ---- timing -----------------------------------------------------------------
struct TimedLaunch {
    int kernel;
    hipEvent_t start, stop;
};
std::mutex g_tmu;
bool g_timing = false;
thread_local bool t_timing_held = false; // this thread's launches are not timed (timing creates events)
std::vector<TimedLaunch> g_pending;
double g_ms[8];
long g_launches[8];

struct ScopedTimer {
    int kernel;
    hipStream_t s;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool on;
    ScopedTimer(int k, hipStream_t st) : kernel(k), s(st), on(g_timing && !t_timing_held)
    {
        if (on) {
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, s);
        }
    }
    ~ScopedTimer()
    {
        if (on) {
            (void)hipEventRecord(e1, s);
            std::lock_guard<std::mutex> lk(g_tmu);
            g_pending.push_back({kernel, e0, e1});
        }
    }
};

void drain_timers()
{
    std::lock_guard<std::mutex> lk(g_tmu);
    for (auto &t : g_pending) {
        (void)hipEventSynchronize(t.stop);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
            g_ms[t.kernel] += ms;
            g_launches[t.kernel] += 1;
        }
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    g_pending.clear();
}

inline int chk(hipError_t e) { return e == hipSuccess ? 0 : -(int)e; }
inline int launch_status() { return chk(hipGetLastError()); }
inline int idivup(int a, int b) { return (a + b - 1) / b; }

int g_strip_override = 0; // > 0: rows per strip of the pyramid pass, for experiments
// columns per lane of the pyramid pass: 1 = pyramid_kernel; 2 = pyramid2_kernel for octaves at least
// g_pyr2_min_w wide (reference profile)
int g_pyr_cols = 1;
int g_pyr2_min_w = 512;
// cost of a row step of pyramid_plane_kernel in 0.01 us: a wave alone on its SIMD, and what each
// further wave per SIMD adds
constexpr long PLANE_STEP_ALONE = 19, PLANE_STEP_SHARED = 10;
constexpr long PLANE_SPLIT_MAX_PX = 1000000; // pixels of a launch (all its images) up to which the planes are dealt over waves

// Rows per strip of the fused pyramid pass.  A wave walks its strip row by row and needs 18
// rows of apron, so a strip of r rows costs r + 18 row steps.
// A launch larger than the chip holds at once runs in rounds of resident waves, and its time is
// rounds x steps per wave: tall strips when the batch alone fills the chip (1080p, 256 images:
// 2 strips of 540 rows = 558 steps per column instead of 6 x 198).
// A launch the chip holds at once (one image, the small octaves of a few) is a single round, and
// the time of a row step grows with the waves that share a SIMD: measured on MI355X over the five
// octaves of one 1080p image and strips of 1 to 32 rows, a step costs 0.45 us when the wave has
// its SIMD to itself and 0.19 us more per further wave per SIMD (1.02 us with all three resident).
// So fewer, taller strips than "as many waves as fit" are faster there: 16 rows instead of 11 for
// one 1080p image (28 against 31 us), 8 instead of 3 for its second octave (20 against 23 us).
// Costs are kept in units of 0.01 us per step.
int pyramid_strip_rows_search(int w, int h, int nimg, int n_scales, int cols_per_lane);

// The search walks up to h candidates (13 us of host time for the five octaves of a 1080p image,
// in front of the first launch of a call that takes 130 us in all): each host thread remembers
// its last answers.
int pyramid_strip_rows(int w, int h, int nimg, int n_scales, int cols_per_lane)
{
    if (g_strip_override > 0) return g_strip_override < h ? g_strip_override : h;
    struct Memo {
        int w, h, nimg, n_scales, cols, rows;
    };
    constexpr int N = 16;
    thread_local Memo memo[N] = {};
    thread_local int next = 0;
    for (int i = 0; i < N; ++i) {
        const Memo &m = memo[i];
        if (m.rows > 0 && m.w == w && m.h == h && m.nimg == nimg && m.n_scales == n_scales && m.cols == cols_per_lane)
            return m.rows;
    }
    const int rows = pyramid_strip_rows_search(w, h, nimg, n_scales, cols_per_lane);
    memo[next] = Memo{w, h, nimg, n_scales, cols_per_lane, rows};
    next = (next + 1) % N;
    return rows;
}

int pyramid_strip_rows_search(int w, int h, int nimg, int n_scales, int cols_per_lane)
{
    const long slots = 256L * 4 * (cols_per_lane == 2 ? 2 : (n_scales == 2 ? PYR_WAVES_S2 : PYR_WAVES_S3));
    const long simds = 256L * 4;
    const long gx = idivup(w, 64 * cols_per_lane);
    // rounds x steps
    long best_t = -1;
    int best_rows = h;
    for (int n = 1; n <= h; ++n) {
        const int rows = idivup(h, n);
        if (rows < 2 && n > 1) break;
        if (n > 1 && rows == idivup(h, n - 1)) continue; // same strips as the last candidate
        const long strips = idivup(h, rows);
        const long steps = rows + 2 * RADIUS;
        const long rounds = (gx * strips * nimg + slots - 1) / slots;
        const long t = rounds * steps;
        if (best_t < 0 || t < best_t) best_t = t, best_rows = rows;
        if (rounds > 64) break; // more strips only add apron rows from here on
    }
    if (gx * idivup(h, best_rows) * nimg > slots) return best_rows;
    // a single round: steps x the cost of a step at that many waves per SIMD
    best_t = -1;
    for (int n = 1; n <= h; ++n) {
        const int rows = idivup(h, n);
        if (n > 1 && rows == idivup(h, n - 1)) continue;
        const long waves = gx * idivup(h, rows) * nimg;
        if (waves > slots) break;
        const long t = (rows + 2 * RADIUS) * (45 + 19 * waves / simds);
        if (best_t < 0 || t < best_t) best_t = t, best_rows = rows;
    }
    return best_rows;
}

// Rows per strip of pyramid_plane_kernel (single-round launches only): steps x the cost of a step
// with that many waves per SIMD, in units of 0.01 us (measured over the octaves of one 1080p image
// and strips of 1 to 64 rows: 0.16-0.20 us for a wave alone on its SIMD, 0.10 us more per further
// wave per SIMD).
int g_pyr_split = 1; // 0: never use pyramid_plane_kernel
int plane_strip_rows(int w, int h, int nimg, int np)
{
    if (g_strip_override > 0) return g_strip_override < h ? g_strip_override : h;
    const long simds = 256L * 4;
    const long gx = idivup(w, 64);
    long best_t = -1;
    int best_rows = h;
    for (int n = 1; n <= h; ++n) {
        const int rows = idivup(h, n);
        if (n > 1 && rows == idivup(h, n - 1)) continue;
        const long waves = gx * idivup(h, rows) * nimg * np;
        if (waves > 8 * simds) break;
        const long t = (rows + 2 * RADIUS) * (PLANE_STEP_ALONE + PLANE_STEP_SHARED * waves / simds);
        if (best_t < 0 || t < best_t) best_t = t, best_rows = rows;
    }
    return best_rows;
}

} // namespace

extern "C" {

void hl_set_pyramid_split(int on) { g_pyr_split = on != 0; }
int hl_debug_strip_rows(int w, int h, int nimg, int n_scales, int kind)
{
    if (w < 1 || h < 1 || nimg < 1 || (n_scales != 2 && n_scales != 3)) return -1;
    if (kind == 0) return pyramid_strip_rows_search(w, h, nimg, n_scales, 1);
    if (kind == 1) return plane_strip_rows(w, h, nimg, n_scales + 3);
    if (kind == 2) return pyramid_strip_rows_search(w, h, nimg, n_scales, 2);
    return -1;
}
void hl_set_pyramid_strip_rows(int rows) { g_strip_override = rows; }
void hl_set_pyramid_columns(int cols, int min_width)
{
    g_pyr_cols = cols == 2 ? 2 : 1;
    if (min_width > 0) g_pyr2_min_w = min_width;
}

int hl_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int hl_set_device(int dev) { return chk(hipSetDevice(dev)); }
int hl_get_device(void)
{
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) return -1;
    return d;
}
int hl_malloc(void **p, size_t bytes) { return chk(hipMalloc(p, bytes)); }
int hl_malloc_pitch(void **p, size_t *pitch_bytes, size_t width_bytes, size_t height)
{
    return chk(hipMallocPitch(p, pitch_bytes, width_bytes, height));
}
int hl_free(void *p) { return chk(hipFree(p)); }
int hl_memset_async(void *p, int v, size_t bytes, void *stream)
{
    return chk(hipMemsetAsync(p, v, bytes, (hipStream_t)stream));
}
static hipMemcpyKind kind_of(int k)
{
    return k == 1 ? hipMemcpyHostToDevice : (k == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
}
int hl_memcpy_async(void *dst, const void *src, size_t bytes, int kind, void *stream)
{
    return chk(hipMemcpyAsync(dst, src, bytes, kind_of(kind), (hipStream_t)stream));
}
int hl_memcpy2d_async(void *dst, size_t dpitch, const void *src, size_t spitch, size_t wbytes,
                      size_t h, int kind, void *stream)
{
    return chk(hipMemcpy2DAsync(dst, dpitch, src, spitch, wbytes, h, kind_of(kind), (hipStream_t)stream));
}
int hl_stream_create(void **s)
{
    hipStream_t st;
    int r = chk(hipStreamCreate(&st));
    *s = (void *)st;
    return r;
}
int hl_stream_destroy(void *s) { return chk(hipStreamDestroy((hipStream_t)s)); }
int hl_stream_sync(void *s) { return chk(hipStreamSynchronize((hipStream_t)s)); }
int hl_device_sync(void) { return chk(hipDeviceSynchronize()); }
int hl_event_create(void **e)
{
    hipEvent_t ev;
    int r = chk(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    *e = (void *)ev;
    return r;
}
int hl_event_destroy(void *e) { return chk(hipEventDestroy((hipEvent_t)e)); }
int hl_event_record(void *e, void *stream) { return chk(hipEventRecord((hipEvent_t)e, (hipStream_t)stream)); }
int hl_event_sync(void *e) { return chk(hipEventSynchronize((hipEvent_t)e)); }
int hl_stream_wait_event(void *stream, void *e)
{
    return chk(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)e, 0));
}
int hl_host_alloc(void **p, size_t bytes) { return chk(hipHostMalloc(p, bytes, hipHostMallocDefault)); }
int hl_host_free(void *p) { return chk(hipHostFree(p)); }
const char *hl_error_string(int code) { return hipGetErrorString((hipError_t)(-code)); }

void hl_timing_enable(int on) { g_timing = on != 0; }
void hl_timing_hold(int on) { t_timing_held = on != 0; }
int hl_timing_read(int kernel, double *ms, long *launches)
{
    drain_timers();
    if (kernel < 0 || kernel >= 8) return -1;
    *ms = g_ms[kernel];
    *launches = g_launches[kernel];
    return 0;
}
void hl_timing_reset(void)
{
    drain_timers();
    for (int i = 0; i < 8; ++i) {
        g_ms[i] = 0;
        g_launches[i] = 0;
    }
}

size_t hl_scratch_bytes(void) { return SCRATCH_BYTES; }

int hl_scratch(void **ptr, size_t *bytes)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SCRATCH_MAX_DEV) return -1001;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) { // once per device, with that device current
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return chk(e);
        if (!p) return -1009;
        g_scratch_ptr[dev] = p;
    }
    *ptr = g_scratch_ptr[dev];
    *bytes = SCRATCH_BYTES;
    return 0;
}

int hl_copy_ceiling(size_t bytes, int repeats, double *gbps)
{
    if (bytes < (1u << 20) || repeats < 1 || !gbps) return -1001;
    const long n = (long)(bytes / 16);
    void *a = nullptr, *b = nullptr;
    int r = chk(hipMalloc(&a, (size_t)n * 16));
    if (!r) r = chk(hipMalloc(&b, (size_t)n * 16));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!r) r = chk(hipEventCreate(&e0));
    if (!r) r = chk(hipEventCreate(&e1));
    if (!r) r = chk(hipMemsetAsync(a, 1, (size_t)n * 16, nullptr));
    if (!r) r = chk(hipMemsetAsync(b, 0, (size_t)n * 16, nullptr));
    float best = 0.f;
    for (int it = 0; it <= repeats && !r; ++it) { // the first pass warms up
        (void)hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL(copy_ceiling_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, nullptr,
                           (copy_f4 *)b, (const copy_f4 *)a, n);
        (void)hipEventRecord(e1, nullptr);
        r = chk(hipEventSynchronize(e1));
        float ms = 0.f;
        if (!r) r = chk(hipEventElapsedTime(&ms, e0, e1));
        if (!r && it > 0 && (best == 0.f || ms < best)) best = ms;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (r) return r;
    *gbps = best > 0.f ? 2.0 * (double)n * 16.0 / ((double)best * 1e-3) / 1e9 : 0.0;
    return 0;
}

int hl_row_blur(const float *src, int src_pitch, long src_stride, float *tmp, long tmp_stride,
                int w, int p, int h, int n_scales, int nimg, const hl_taps *taps, void *stream)
{
    if (w <= 0 || h <= 0 || nimg <= 0 || p < w || (p & 3) || src_pitch < w) return -1001;
    const bool aligned = ((src_pitch & 3) == 0) && ((((size_t)src) & 15) == 0) && ((src_stride & 3) == 0);
    dim3 grid(idivup(w, 256), idivup(h, 4), nimg), block(256);
    hipStream_t s = (hipStream_t)stream;
    ScopedTimer t(0, s);
#define ROW_LAUNCH(NP)                                                                          \
    if (aligned)                                                                                \
        hipLaunchKernelGGL((row_blur_kernel<NP, true>), grid, block, 0, s, src, src_pitch,      \
                           src_stride, tmp, tmp_stride, w, p, h, *taps);                        \
    else                                                                                        \
        hipLaunchKernelGGL((row_blur_kernel<NP, false>), grid, block, 0, s, src, src_pitch,     \
                           src_stride, tmp, tmp_stride, w, p, h, *taps);
    if (n_scales == 2) {
        ROW_LAUNCH(5)
    } else if (n_scales == 3) {
        ROW_LAUNCH(6)
    } else {
        return -1002;
    }
#undef ROW_LAUNCH
    return launch_status();
}

int hl_col_blur_dog(const float *tmp, long tmp_stride, float *blur, long blur_stride, int all_planes,
                    float *dog, long dog_stride, float *next_base, long next_stride, int next_p,
                    int w, int p, int h, int n_scales, int nimg, const hl_taps *taps, void *stream)
{
    if (w <= 0 || h <= 0 || nimg <= 0 || p < w || !tmp || !blur) return -1001;
    if (!all_planes && !dog) return -1001;
    if (next_base && next_p < (w + 1) / 2) return -1003;
    // One wave per block.  Strip height 20*m - 18 output rows, so the unrolled
    // ring is used in whole turns: tall strips re-read fewer apron rows, short
    // strips give small octaves enough waves to cover the chip.
    int m = 6;
    while (m > 2 && (long)idivup(w, 64) * idivup(h, COL_RING * m - 2 * RADIUS) * nimg < 2048) --m;
    const int rows = g_strip_override > 0 ? (g_strip_override < h ? g_strip_override : h) : COL_RING * m - 2 * RADIUS;
    dim3 grid(idivup(w, 64), idivup(h, rows), nimg), block(64);
    hipStream_t s = (hipStream_t)stream;
    ScopedTimer t(1, s);
#define COL_LAUNCH(SS, ST)                                                                         \
    hipLaunchKernelGGL((col_blur_dog_kernel<SS, ST>), grid, block, 0, s, tmp, tmp_stride, blur,    \
                       blur_stride, dog, dog_stride, next_base, next_stride, next_p, w, p, h,     \
                       rows, *taps)
    if (n_scales == 2 && all_planes)
        COL_LAUNCH(2, true);
    else if (n_scales == 2)
        COL_LAUNCH(2, false);
    else if (n_scales == 3 && all_planes)
        COL_LAUNCH(3, true);
    else if (n_scales == 3)
        COL_LAUNCH(3, false);
    else
        return -1002;
#undef COL_LAUNCH
    return launch_status();
}

int hl_pyramid(const void *src, int src_w, int src_h, int src_pitch, long src_stride, int src_mode,
               float *blur, long blur_stride, int w, int p, int h,
               int n_scales, int nimg, const hl_taps *taps, float contrast, unsigned *queue,
               long queue_stride, unsigned *qcount, long qcount_stride, int qcap, int group0,
               int plain, void *stream)
{
    if (w <= 0 || h <= 0 || nimg <= 0 || !src || !blur || src_pitch < src_w) return -1001;
    if (p < w || (p & 63)) return -1005; // all 64 lanes of a block store: the pitch must cover them
    if (src_mode < 0 || src_mode > 2) return -1002;
    if (src_mode == 2 ? (w != (src_w + 1) / 2 || h != (src_h + 1) / 2) : (w != src_w || h != src_h))
        return -1006;
    // 32-bit byte offsets inside one image's source and blur planes
    if ((long)p * h * (n_scales + 3) * 4 >= (long)FZ_DROP || (long)src_pitch * src_h * (src_mode == 1 ? 1 : 4) > 0xffffffffL)
        return -1007;
    if (queue && (!qcount || qcap < 1 || w > 65535 || h > 65535)) return -1008;
    if (n_scales != 2 && n_scales != 3) return -1002;
    if (g_pyr_cols == 2 && !plain && n_scales == 2 && w >= g_pyr2_min_w) { // two columns per lane (its ring needs 200 registers at 2 scales; 3 do not fit)
        const int rows = pyramid_strip_rows(w, h, nimg, n_scales, 2);
        const dim3 tiles(idivup(w, 128), idivup(h, rows), nimg), block(64);
        hipStream_t s = (hipStream_t)stream;
        ScopedTimer t(4, s);
#define PYR2_LAUNCH(SS, MD)                                                                         \
    hipLaunchKernelGGL((pyramid2_kernel<SS, MD>), tiles, block, 0, s, (const float *)src, src_w,    \
                       src_h, src_pitch, src_stride, blur, blur_stride, w, p, h, rows, *taps,       \
                       contrast, queue, queue_stride, qcount, qcount_stride, qcap, group0,         \
                       (int)tiles.x, (int)tiles.y, (int)tiles.z)
        switch (src_mode) {
        case 0: PYR2_LAUNCH(2, 0); break;
        case 1: PYR2_LAUNCH(2, 1); break;
        default: PYR2_LAUNCH(2, 2); break;
        }
#undef PYR2_LAUNCH
        return launch_status();
    }
    if (!queue && g_pyr_split) {
        // A launch the chip holds in a single round and that feeds no candidate queue (one image at a
        // time, the stage entry): the planes dealt over waves.
        const int rows1 = pyramid_strip_rows(w, h, nimg, n_scales, 1);
        const long slots = 256L * 4 * (n_scales == 2 ? PYR_WAVES_S2 : PYR_WAVES_S3);
        // (a launch of more pixels is bound by the chip's FMA rate, not by its longest wave, and
        // staging every source row S+3 times costs more than the shorter row steps give back: one
        // 1080p image 30 us against 26, its second octave 16 against 18, its fifth 6 against 10)
        if ((long)idivup(w, 64) * idivup(h, rows1) * nimg <= slots && (long)w * h * nimg <= PLANE_SPLIT_MAX_PX) {
            const int np = n_scales + 3;
            const int rows = plane_strip_rows(w, h, nimg, np);
            const dim3 tiles(idivup(w, 64), idivup(h, rows), nimg), block(64);
            const dim3 grid(tiles.x * tiles.y * tiles.z * np);
            hipStream_t s = (hipStream_t)stream;
            ScopedTimer t(4, s);
#define PYRP_LAUNCH(MD)                                                                              \
    hipLaunchKernelGGL((pyramid_plane_kernel<MD>), grid, block, 0, s, (const float *)src, src_w,     \
                       src_h, src_pitch, src_stride, blur, blur_stride, w, p, h, rows, np, *taps,    \
                       (int)tiles.x, (int)tiles.y, (int)tiles.z)
            switch (src_mode) {
            case 0: PYRP_LAUNCH(0); break;
            case 1: PYRP_LAUNCH(1); break;
            default: PYRP_LAUNCH(2); break;
            }
#undef PYRP_LAUNCH
            return launch_status();
        }
    }
    const int rows = pyramid_strip_rows(w, h, nimg, n_scales, 1);
    const dim3 tiles(idivup(w, 64), idivup(h, rows), nimg), block(64);
#ifdef PYR_EXP_PERSIST
    const unsigned resident = 256u * 4u * (n_scales == 2 ? PYR_WAVES_S2 : PYR_WAVES_S3);
    const unsigned total = tiles.x * tiles.y * tiles.z;
    const dim3 grid(total < resident ? total : resident);
#else
    const dim3 grid = tiles;
#endif
    hipStream_t s = (hipStream_t)stream;
    ScopedTimer t(4, s);
#define PYR_LAUNCH_P(SS, MD, PL)                                                                   \
    hipLaunchKernelGGL((pyramid_kernel<SS, MD, PL>), grid, block, 0, s, (const float *)src, src_w, \
                       src_h, src_pitch, src_stride, blur, blur_stride, w, p, h, rows, *taps,      \
                       contrast, queue, queue_stride, qcount, qcount_stride, qcap, group0,        \
                       (int)tiles.x, (int)tiles.y, (int)tiles.z)
#define PYR_LAUNCH(SS, MD)                                                                         \
    do {                                                                                           \
        if (plain) PYR_LAUNCH_P(SS, MD, true);                                                     \
        else PYR_LAUNCH_P(SS, MD, false);                                                          \
    } while (0)
    if (n_scales != 2 && n_scales != 3) return -1002;
    switch (src_mode + 3 * (n_scales - 2)) {
    case 0: PYR_LAUNCH(2, 0); break;
    case 1: PYR_LAUNCH(2, 1); break;
    case 2: PYR_LAUNCH(2, 2); break;
    case 3: PYR_LAUNCH(3, 0); break;
    case 4: PYR_LAUNCH(3, 1); break;
    default: PYR_LAUNCH(3, 2); break;
    }
#undef PYR_LAUNCH
#undef PYR_LAUNCH_P
    return launch_status();
}

static int detect_args_ok(const hl_detect_args *a, int nimg)
{
    if (nimg <= 0 || a->n_oct < 1 || a->n_oct > HL_MAX_OCT || a->cap < 1) return 0;
    if (a->n_scales != 2 && a->n_scales != 3) return 0;
    for (int o = 0; o < a->n_oct; ++o) {
        const hl_octave *c = &a->oct[o];
        if (c->w < 1 || c->h < 1 || c->p < c->w || (c->p & 3) || c->w > 65535 || c->h > 65535) return 0;
        if ((c->blur_off & 3) || c->qcap < 1) return 0;
    }
    return 1;
}

int hl_verify(const float *arena, long arena_stride, const unsigned *queue, long queue_stride,
              const unsigned *qcount, long qcount_stride, unsigned *cand, long cand_stride,
              unsigned *counts, long counts_stride, const hl_detect_args *a, int nimg, void *stream)
{
    if (!detect_args_ok(a, nimg)) return -1001;
    // blocks per image: enough for a few thousand waves per launch, which the chip holds at once
    int per_image = 1024 / nimg;
    per_image = per_image < 2 ? 2 : (per_image > 64 ? 64 : per_image);
    dim3 grid(per_image, nimg), block(256);
    hipStream_t s = (hipStream_t)stream;
    ScopedTimer t(5, s);
    if (a->n_scales == 2)
        hipLaunchKernelGGL((verify_kernel<2>), grid, block, 0, s, arena, arena_stride, queue,
                           queue_stride, qcount, qcount_stride, cand, cand_stride, counts,
                           counts_stride, *a);
    else
        hipLaunchKernelGGL((verify_kernel<3>), grid, block, 0, s, arena, arena_stride, queue,
                           queue_stride, qcount, qcount_stride, cand, cand_stride, counts,
                           counts_stride, *a);
    return launch_status();
}

int hl_detect(const float *arena, long arena_stride, unsigned *cand, long cand_stride,
              unsigned *counts, long counts_stride, const unsigned *gate, long gate_stride,
              const hl_detect_args *a, int nimg, void *stream)
{
    if (!detect_args_ok(a, nimg) || (arena_stride & 3) || (((size_t)arena) & 15)) return -1001;
    // with a gate the kernel normally has nothing to do: few blocks per image, each
    // walking over the units; without one it is the whole detector and fills the chip
    // (a block walks over several pixel blocks also because one that has seen a group
    // reach the cap can drop the rest of its work)
    int gx = a->dense0[a->n_oct] < 4096 ? a->dense0[a->n_oct] : 4096;
    // (gated, 2048 blocks per launch measured best at 256 images of 1080p: 10 us instead of 22 for
    // the launch that finds nothing to do, 6.8 instead of 7.3 us per image when every image needs it)
    int per_image = (gate ? 2048 : 16384) / nimg;
    const int least = gate ? 4 : 32;
    per_image = per_image < least ? least : (per_image > 1024 ? 1024 : per_image);
    if (gx > per_image) gx = per_image;
    dim3 grid(gx, 1, nimg), block(DET_THREADS);
    // A cap of a few thousand points is normally reached inside the first group of an image
    // dense enough to need this kernel: then the groups are taken one after the other and the
    // work stops with the first (see the kernel); with a large cap every group is needed and
    // the blocks are walked once, all scales of an octave at a time.
    const int scale_major = gate && a->cap <= 4096; // (without a gate it is the whole detector, for sparse images too)
    hipStream_t s = (hipStream_t)stream;
    ScopedTimer t(2, s);
#define DET_LAUNCH(SS, SM)                                                                          \
    hipLaunchKernelGGL((detect_kernel<SS, SM>), grid, block, 0, s, arena, arena_stride, cand,       \
                       cand_stride, counts, counts_stride, gate, gate_stride, *a)
    if (a->n_scales == 2 && scale_major)
        DET_LAUNCH(2, true);
    else if (a->n_scales == 2)
        DET_LAUNCH(2, false);
    else if (scale_major)
        DET_LAUNCH(3, true);
    else
        DET_LAUNCH(3, false);
#undef DET_LAUNCH
    return launch_status();
}

int hl_verify_lowe(const float *arena, long arena_stride, const unsigned *queue, long queue_stride,
                   const unsigned *qcount, long qcount_stride, unsigned *cand, void *sub,
                   long cand_stride, unsigned *counts, long counts_stride, const hl_detect_args *a,
                   int nimg, void *stream)
{
    if (!detect_args_ok(a, nimg) || !sub) return -1001;
    // waves per image: enough for a few thousand waves per launch, which the chip holds at once
    int per_image = 4096 / nimg;
    per_image = per_image < 4 ? 4 : (per_image > 256 ? 256 : per_image);
    dim3 grid(per_image, nimg), block(64);
    hipStream_t s = (hipStream_t)stream;
    ScopedTimer t(5, s);
    if (a->n_scales == 2)
        hipLaunchKernelGGL((verify_lowe_kernel<2>), grid, block, 0, s, arena, arena_stride, queue,
                           queue_stride, qcount, qcount_stride, cand, (float4 *)sub, cand_stride,
                           counts, counts_stride, *a);
    else
        hipLaunchKernelGGL((verify_lowe_kernel<3>), grid, block, 0, s, arena, arena_stride, queue,
                           queue_stride, qcount, qcount_stride, cand, (float4 *)sub, cand_stride,
                           counts, counts_stride, *a);
    return launch_status();
}

int hl_detect_lowe(const float *arena, long arena_stride, unsigned *cand, void *sub, long cand_stride,
                   unsigned *counts, long counts_stride, const unsigned *gate, long gate_stride,
                   const hl_detect_args *a, int nimg, void *stream)
{
    if (!detect_args_ok(a, nimg) || !sub) return -1001;
    // (gated: about 2048 blocks per launch, as for the reference-profile detector)
    const int gated = 2048 / nimg < 4 ? 4 : (2048 / nimg > 32 ? 32 : 2048 / nimg);
    const int gx = gate ? gated : (a->dense0[a->n_oct] < 4096 ? a->dense0[a->n_oct] : 4096);
    dim3 grid(gx, 1, nimg), block(256);
    hipStream_t s = (hipStream_t)stream;
    ScopedTimer t(2, s);
    if (a->n_scales == 2)
        hipLaunchKernelGGL((detect_lowe_kernel<2>), grid, block, 0, s, arena, arena_stride, cand,
                           (float4 *)sub, cand_stride, counts, counts_stride, gate, gate_stride, *a);
    else
        hipLaunchKernelGGL((detect_lowe_kernel<3>), grid, block, 0, s, arena, arena_stride, cand,
                           (float4 *)sub, cand_stride, counts, counts_stride, gate, gate_stride, *a);
    return launch_status();
}

int hl_upsample2x(const void *src, int src_u8, int src_pitch, long src_stride, float *dst, int dst_pitch,
                  long dst_stride, int w, int h, int nimg, void *stream)
{
    if (!src || !dst || w < 1 || h < 1 || nimg < 1 || src_pitch < w || dst_pitch < 2 * w) return -1001;
    dim3 grid(idivup(2 * w, 256), 2 * h, nimg), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (src_u8)
        hipLaunchKernelGGL((upsample2x_kernel<unsigned char>), grid, block, 0, s, (const unsigned char *)src,
                           src_pitch, src_stride, dst, dst_pitch, dst_stride, w, h);
    else
        hipLaunchKernelGGL((upsample2x_kernel<float>), grid, block, 0, s, (const float *)src, src_pitch,
                           src_stride, dst, dst_pitch, dst_stride, w, h);
    return launch_status();
}

int hl_quantize(const void *pts, int n, unsigned char *out, void *stream)
{
    if (!pts || !out || n < 0) return -1001;
    if (n == 0) return 0;
    hipLaunchKernelGGL(quantize_kernel, dim3(n), dim3(128), 0, (hipStream_t)stream, (const SiftPointD *)pts, n, out);
    return launch_status();
}

int hl_keypoints(void *pts, long pts_stride_points, unsigned char *meta, const float *arena,
                 long arena_stride, const unsigned *cand, const void *sub, long cand_stride,
                 const unsigned *counts, long counts_stride, const unsigned *base_count,
                 const hl_keypoint_args *args, int nimg, void *stream)
{
    if (nimg <= 0 || args->n_groups <= 0 || args->n_groups > HL_MAX_GROUPS) return -1001;
    // one wave per keypoint, grid-stride: the count is only known on the device
    // waves per image: the launch should be a few tens of thousands of waves (blocks that
    // find no keypoint left still cost their dispatch), a single image still fills the chip
    int gx = 32768 / nimg;
    gx = gx < 16 ? 16 : (gx > 2048 ? 2048 : gx);
    if (gx > args->cap) gx = args->cap;
    if (gx < 1) gx = 1;
    dim3 grid(gx, nimg), block(64);
    hipStream_t s = (hipStream_t)stream;
    ScopedTimer t(3, s);
    hipLaunchKernelGGL(keypoint_kernel, grid, block, 0, s, (SiftPointD *)pts, pts_stride_points, meta,
                       arena, arena_stride, cand, (const float4 *)sub, cand_stride, counts, counts_stride,
                       base_count, *args);
    return launch_status();
}

int hl_finalize_counts(unsigned *base_count, unsigned *host_copy, const unsigned *counts, long counts_stride,
                       int n_groups, int cap, int nimg, void *stream)
{
    hipLaunchKernelGGL(finalize_counts_kernel, dim3(idivup(nimg, 64)), dim3(64), 0,
                       (hipStream_t)stream, base_count, host_copy, counts, counts_stride, n_groups, cap, nimg);
    return launch_status();
}

int hl_zero_words(unsigned *a, long na, unsigned *b, long nb, void *stream)
{
    if (na < 0 || nb < 0 || (na > 0 && !a) || (nb > 0 && !b)) return -1001;
    if (na + nb == 0) return 0;
    const long blocks = (na + nb + 1023) / 1024; // four words per thread
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0,
                       (hipStream_t)stream, a, na, b, nb);
    return launch_status();
}

int hl_u8_to_f32(float *dst, int dp, const unsigned char *src, int sp, int w, int h, void *stream)
{
    if (!dst || !src || w < 1 || h < 1 || dp < w || sp < w) return -1001;
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3(idivup(w, 1024), h), dim3(256), 0, (hipStream_t)stream, dst, dp, src, sp, w, h);
    return launch_status();
}

static dim3 px_grid(int w, int h) { return dim3(idivup(w, 64), idivup(h, 4)); }

int hl_subtract(float *dst, const float *a, const float *b, int w, int p, int h, void *stream)
{
    hipLaunchKernelGGL(subtract_kernel, px_grid(w, h), dim3(256), 0, (hipStream_t)stream, dst, a, b, w, p, h);
    return launch_status();
}
int hl_hessian(float *dst, const float *src, int w, int p, int h, void *stream)
{
    hipLaunchKernelGGL(hessian_kernel, px_grid(w, h), dim3(256), 0, (hipStream_t)stream, dst, src, w, p, h);
    return launch_status();
}
int hl_gradient(float *dst, const float *src, int w, int p, int h, void *stream)
{
    hipLaunchKernelGGL(gradient_kernel, px_grid(w, h), dim3(256), 0, (hipStream_t)stream, dst, src, w, p, h);
    return launch_status();
}
int hl_copy(float *dst, const float *src, int w, int p, int h, void *stream)
{
    hipLaunchKernelGGL(copy_kernel, px_grid(w, h), dim3(256), 0, (hipStream_t)stream, dst, src, w, p, h);
    return launch_status();
}

} // extern "C"
