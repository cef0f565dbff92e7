// OptFlow block pieces for gfx950: the step right after the
// cost volume at every level (reference: qpwcnet/core/non_layers.py:213-273).
//
//   SeparableConv2D(3x3, 'same') = depthwise 3x3 (no bias) -> pointwise 1x1 (+bias) -> Mish
//
// dwconv3x3: the depthwise half, bandwidth-bound (read C, write C floats per pixel).
//   * up to three channels-last SOURCES are read as one virtual concatenation, so
//     Flow/UpFlow's concat([cost, prv, flo]) (non_layers.py:336-338, 381-385) is never
//     materialised;
//   * optional Mish on load: the previous layer's activation is applied to the
//     pre-activation pointwise output while it is read (zero padding applies to the
//     activated tensor, as in the reference where Mish precedes the next 'same' conv);
//   * lane = one channel, consecutive lanes = consecutive channels of a pixel: every
//     wave load/store is a contiguous 256-byte run; a thread owns 4 consecutive pixels
//     and walks a strip of rows with a rolling 3x6 register window (1.5 loads and Mish
//     evaluations per output).
// The pointwise half is a plain GEMM and stays on the library (MFMA through rocBLAS).
//
// flow_head: Mish -> 1x1 conv 16->16 + bias -> Mish -> BatchNorm(inference) -> 3x3 conv
// 16->2 (no bias) -> * scale   (non_layers.py:238-254, 268-273) in one launch; the
// normalised 16-channel tile (+1 halo, zero outside the image) lives in LDS.
#include "optflow_common.h"
#include "split_bf16.h"

namespace qpwc {

constexpr int kDwPx = 4;    // consecutive pixels per thread

// thread = (4 consecutive pixels, 1 channel): per input row it loads 6 values, applies
// Mish once to each (1.5 evaluations per output instead of 3 for one pixel per thread),
// and keeps a rolling 3-row window; the raw values of the next row are requested one
// iteration ahead so that the loop does not wait on a load it has just issued.
template <typename T, bool ACT>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(DwSrc src, const float* __restrict__ weight,
                                                        T* __restrict__ out, int H, int W, int C,
                                                        int strips, int wq, int rows) {
    QPWC_FLOW_CHAIN_PRIO();
    const int64_t rowthreads = (int64_t)wq * C;  // (x-quad, channel) pairs of one row
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rowthreads) return;
    const int xq = (int)(idx / C), c = (int)(idx - (int64_t)xq * C);
    const int x0 = xq * kDwPx;
    const int strip = blockIdx.y % strips, b = blockIdx.y / strips;
    const int y0 = strip * rows;

    // which source holds channel c
    const T* p;
    int64_t ps;
    int cc;
    {
        const DwPick k = dwsrc_pick(src, c, C);
        p = (const T*)k.p; ps = k.ps; cc = k.cc;
    }
    p += (int64_t)b * H * W * ps + cc;

    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = weight[c * 9 + k];

    bool col_ok[kDwPx + 2];
#pragma unroll
    for (int j = 0; j < kDwPx + 2; ++j) col_ok[j] = x0 - 1 + j >= 0 && x0 - 1 + j < W;

    auto load_raw = [&](int y, float (&r)[kDwPx + 2]) {
        const bool row_ok = y >= 0 && y < H;
        const T* q = p + ((int64_t)y * W + (x0 - 1)) * ps;
#pragma unroll
        for (int j = 0; j < kDwPx + 2; ++j) r[j] = (row_ok && col_ok[j]) ? ld(q + (int64_t)j * ps) : 0.0f;
    };
    auto activate = [&](int y, float (&r)[kDwPx + 2]) {
        if (!ACT) return;
        const bool row_ok = y >= 0 && y < H;  // zero padding applies to the ACTIVATED tensor
#pragma unroll
        for (int j = 0; j < kDwPx + 2; ++j) r[j] = (row_ok && col_ok[j]) ? mishf(r[j]) : 0.0f;
    };

    float r0[kDwPx + 2], r1[kDwPx + 2], r2[kDwPx + 2], ahead[kDwPx + 2];
    load_raw(y0 - 1, r0);
    load_raw(y0, r1);
    load_raw(y0 + 1, ahead);
    activate(y0 - 1, r0);
    activate(y0, r1);
    T* o = out + ((int64_t)(b * H + y0) * W + x0) * C + c;
    const int64_t rowlen = (int64_t)W * C;
    const int yend = y0 + rows < H ? y0 + rows : H;
    for (int y = y0; y < yend; ++y) {
#pragma unroll
        for (int j = 0; j < kDwPx + 2; ++j) r2[j] = ahead[j];
        if (y + 1 < yend) load_raw(y + 2, ahead);  // in flight during this row's arithmetic
        activate(y + 1, r2);
#pragma unroll
        for (int i = 0; i < kDwPx; ++i) {
            float a = w[0] * r0[i];
            a = fmaf(w[1], r0[i + 1], a);
            a = fmaf(w[2], r0[i + 2], a);
            a = fmaf(w[3], r1[i], a);
            a = fmaf(w[4], r1[i + 1], a);
            a = fmaf(w[5], r1[i + 2], a);
            a = fmaf(w[6], r2[i], a);
            a = fmaf(w[7], r2[i + 1], a);
            a = fmaf(w[8], r2[i + 2], a);
            if (x0 + i < W) st(o + (int64_t)i * C, a);
        }
        o += rowlen;
#pragma unroll
        for (int j = 0; j < kDwPx + 2; ++j) {
            r0[j] = r1[j];
            r1[j] = r2[j];
        }
    }
}

// Single-source, aligned case (every SeparableConv after the first): a lane owns VEC
// consecutive channels, so loads and stores are VEC elements wide (fp16 with one channel per
// lane moves only 128 B per wave instruction).  Same 4-pixel x row-strip scheme.
template <typename T, int VEC>
struct VecIo;
template <>
struct VecIo<__half, 4> {
    static __device__ __forceinline__ void load(const __half* p, float (&v)[4]) {
        const float4 t = ld4(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(__half* p, const float (&v)[4]) {
        st4(p, make_float4(v[0], v[1], v[2], v[3]));
    }
};

template <typename T, int VEC, bool ACT>
__global__ __launch_bounds__(256) void dwconv3x3_vec_kernel(DwSrc src, const float* __restrict__ weight,
                                                            T* __restrict__ out, int H, int W, int C,
                                                            int strips, int wq, int rows) {
    const int cv = C / VEC;
    const int64_t rowthreads = (int64_t)wq * cv;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rowthreads) return;
    const int xq = (int)(idx / cv), c = (int)(idx - (int64_t)xq * cv) * VEC;
    const int x0 = xq * kDwPx;
    const int strip = blockIdx.y % strips, b = blockIdx.y / strips;
    const int y0 = strip * rows;
    // (round 4) up to three sources, each a multiple of VEC channels in VEC-element aligned pixels: the lane's VEC channels
    // lie in one of them (config 5's level-0 layer: 84 + 256 + 256 channels, 30 us on the one-channel-per-lane kernel)
    const DwPick pk = dwsrc_pick(src, c, C);
    const int64_t ps = pk.ps;
    const T* p = (const T*)pk.p + (int64_t)b * H * W * ps + pk.cc;

    float w[9][VEC];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) w[k][e] = weight[(c + e) * 9 + k];

    bool col_ok[kDwPx + 2];
#pragma unroll
    for (int j = 0; j < kDwPx + 2; ++j) col_ok[j] = x0 - 1 + j >= 0 && x0 - 1 + j < W;

    auto load_row = [&](int y, float (&r)[kDwPx + 2][VEC]) {
        const bool row_ok = y >= 0 && y < H;
        const T* q = p + ((int64_t)y * W + (x0 - 1)) * ps;
#pragma unroll
        for (int j = 0; j < kDwPx + 2; ++j) {
            if (row_ok && col_ok[j]) {
                VecIo<T, VEC>::load(q + (int64_t)j * ps, r[j]);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) r[j][e] = 0.0f;
            }
        }
    };
    auto activate = [&](int y, float (&r)[kDwPx + 2][VEC]) {
        if (!ACT) return;
        const bool row_ok = y >= 0 && y < H;
#pragma unroll
        for (int j = 0; j < kDwPx + 2; ++j)
#pragma unroll
            for (int e = 0; e < VEC; ++e) r[j][e] = (row_ok && col_ok[j]) ? mishf(r[j][e]) : 0.0f;
    };

    float r0[kDwPx + 2][VEC], r1[kDwPx + 2][VEC], r2[kDwPx + 2][VEC];
    load_row(y0 - 1, r0);
    load_row(y0, r1);
    activate(y0 - 1, r0);
    activate(y0, r1);
    T* o = out + ((int64_t)(b * H + y0) * W + x0) * C + c;
    const int64_t rowlen = (int64_t)W * C;
    const int yend = y0 + rows < H ? y0 + rows : H;
    for (int y = y0; y < yend; ++y) {
        load_row(y + 1, r2);
        activate(y + 1, r2);
#pragma unroll
        for (int i = 0; i < kDwPx; ++i) {
            float a[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float s = w[0][e] * r0[i][e];
                s = fmaf(w[1][e], r0[i + 1][e], s);
                s = fmaf(w[2][e], r0[i + 2][e], s);
                s = fmaf(w[3][e], r1[i][e], s);
                s = fmaf(w[4][e], r1[i + 1][e], s);
                s = fmaf(w[5][e], r1[i + 2][e], s);
                s = fmaf(w[6][e], r2[i][e], s);
                s = fmaf(w[7][e], r2[i + 1][e], s);
                s = fmaf(w[8][e], r2[i + 2][e], s);
                a[e] = s;
            }
            if (x0 + i < W) VecIo<T, VEC>::store(o + (int64_t)i * C, a);
        }
        o += rowlen;
#pragma unroll
        for (int j = 0; j < kDwPx + 2; ++j)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                r0[j][e] = r1[j][e];
                r1[j][e] = r2[j][e];
            }
    }
}

template <typename T, int VEC>
static void dwconv_vec_dispatch(const DwSrc& d, int act, const void* weight, void* out, int B, int H, int W,
                                int C, int strips, int wq, int rows, hipStream_t s) {
    const int64_t rowthreads = (int64_t)wq * (C / VEC);
    const dim3 grid((unsigned)((rowthreads + 255) / 256), (unsigned)(strips * B));
    if (act)
        hipLaunchKernelGGL((dwconv3x3_vec_kernel<T, VEC, true>), grid, dim3(256), 0, s, d,
                           (const float*)weight, (T*)out, H, W, C, strips, wq, rows);
    else
        hipLaunchKernelGGL((dwconv3x3_vec_kernel<T, VEC, false>), grid, dim3(256), 0, s, d,
                           (const float*)weight, (T*)out, H, W, C, strips, wq, rows);
}

template <typename T>
static void dwconv_dispatch(const DwSrc& d, int act, const void* weight, void* out, int H, int W, int C,
                            int strips, int wq, int rows, dim3 grid, hipStream_t s) {
    if (act)
        hipLaunchKernelGGL((dwconv3x3_kernel<T, true>), grid, dim3(256), 0, s, d, (const float*)weight,
                           (T*)out, H, W, C, strips, wq, rows);
    else
        hipLaunchKernelGGL((dwconv3x3_kernel<T, false>), grid, dim3(256), 0, s, d, (const float*)weight,
                           (T*)out, H, W, C, strips, wq, rows);
}

int dwconv3x3_launch(const void* const* srcs, const int* chans, const int64_t* strides, int n_src,
                     int act, const void* weight, void* out, int B, int H, int W, int dtype,
                     hipStream_t s) {
    DwSrc d;
    int C = 0;
    for (int i = 0; i < 3; ++i) {
        d.ptr[i] = i < n_src ? srcs[i] : nullptr;
        d.ch[i] = i < n_src ? chans[i] : 0;
        d.stride[i] = i < n_src ? strides[i] : 0;
        C += d.ch[i];
    }
    // rows per thread strip: long strips amortise the 2 halo rows on big images, short ones
    // give small (coarse-level) images enough threads and a short dependent-load chain
    const int rows = H >= 64 ? 8 : (H >= 32 ? 4 : 2);
    const int strips = (H + rows - 1) / rows;
    const int wq = (W + kDwPx - 1) / kDwPx;
    // fp16 storage, single aligned source: 4 channels per lane (measured +3 % on the fp16 step;
    // the fp32 analogue with 2 channels per lane measured 1.5 % slower than one per lane)
    bool vec_ok = dtype == QPWC_F16 && C % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 8 == 0;
    for (int i = 0; i < n_src; ++i)      // every source: a multiple of 4 channels in 8-byte aligned pixels
        vec_ok = vec_ok && d.ch[i] % 4 == 0 && d.stride[i] % 4 == 0 && reinterpret_cast<uintptr_t>(d.ptr[i]) % 8 == 0;
    if (vec_ok) {
        dwconv_vec_dispatch<__half, 4>(d, act, weight, out, B, H, W, C, strips, wq, rows, s);
        return check_launch("dwconv3x3_vec_kernel");
    }
    const int64_t rowthreads = (int64_t)wq * C;
    const dim3 grid((unsigned)((rowthreads + 255) / 256), (unsigned)(strips * B));
    if (dtype == QPWC_F32)
        dwconv_dispatch<float>(d, act, weight, out, H, W, C, strips, wq, rows, grid, s);
    else
        dwconv_dispatch<__half>(d, act, weight, out, H, W, C, strips, wq, rows, grid, s);
    return check_launch("dwconv3x3_kernel");
}

// ---------------------------------------------------------------------------
// sepconv3x3_fused: the WHOLE SeparableConv2D (depthwise 3x3 -> pointwise 1x1 + bias,
// pre-activation output) in one launch, fp32 (non_layers.py:223-231).  The depthwise result
// never goes to HBM: input read once (+27..41 % halo), output written once.
//
// Workgroup = 4 waves = an 8 x 16 pixel tile; channels in steps of 32:
//   stage   : the 10 x 18 halo tile of the (virtually concatenated, optionally Mish-activated)
//             input, 32 channels, global -> registers (prefetched one step ahead) -> LDS
//             `in_s` (pixel stride 40 floats: the depthwise b128 reads are conflict free);
//             the 32 x F slice of the pointwise weights -> `w_s`, the 32 x 9 depthwise taps -> `dw_s`;
//   depthw. : thread = (4 consecutive pixels, 4 channels): 18 ds_read_b128, 144 FMA, the result
//             -> `y_s` [128 px][32 ch] as 4 ds_write_b128;
//   pointw. : v_mfma_f32_16x16x4_f32, rows = output channels (A = w_s), cols = pixels (B = y_s);
//             a wave owns 32 pixels x all F outputs (2 x F/16 accumulators); per 16 channels it
//             reads 2 + F/16 operands (ds_read_b128: k-slot g of a lane owns channels 4g..4g+3)
//             for 8 x F/16 matrix instructions.  `y_s` / `w_s` rows are 128 B with the 16-byte
//             chunk c of row n at chunk c ^ (n >> 1) (conflict free, as in the cost volume).
// Software pipeline: the matrix work of step k and the depthwise convolution of step k+1 share one
// barrier interval (`y_s` double-buffered, inputs prefetched two steps ahead of their matrix work), two
// barriers per step.  79 KB LDS -> 2 workgroups per CU.  Ablation at L4 (B=8, C 128 -> F 64, 80 us):
// without the matrix instructions 54 us, without the global loads 67 us, without the depthwise
// arithmetic 73 us -- the layer is as much bound by staging 200 MB through 2 waves per SIMD as by the
// matrix pipe (31 us at its peak).
// pointwise weights: (F, Cpad) row-major with Cpad = ceil(C/32)*32, zero padded.
#ifdef QPWC_SC_STAMP
__device__ long long g_sc_stamps[4 * 64];
__device__ long long g_sc_census[4096 * 6];   // per workgroup: start, staged0, last-step begin, end, HW_ID, XCC_ID
#endif

__device__ __attribute__((aligned(16))) const float kScZeros[4] = {0.f, 0.f, 0.f, 0.f};

#ifndef QPWC_SC_W_AHEAD
#define QPWC_SC_W_AHEAD 1   // A/B (round 4): pointwise A operands read one block of eight matrix instructions ahead (F = 128)
#endif
#ifndef QPWC_SC_AGPR
#define QPWC_SC_AGPR 0   // A/B (round 4): accumulators pinned to AGPRs -- every matrix instruction then accumulates in place (no renamed
                         // result, 0 suspects in tools/mfma_war_lint.py) but the epilogue pays a v_accvgpr_read per value: 129-131 vs 123-126 us
#endif
#ifndef QPWC_SC_BUF_STORE
#define QPWC_SC_BUF_STORE 1   // A/B (round 4): output stores through a buffer descriptor (no per-store address arithmetic / branch)
#endif
typedef unsigned u32x4sc __attribute__((ext_vector_type(4)));
#ifndef QPWC_SC_BIAS_EARLY
#define QPWC_SC_BIAS_EARLY 1   // A/B (round 4): the last step's bias values requested before its matrix instructions
#endif
#ifndef QPWC_SC_ASYM_PRIO
#define QPWC_SC_ASYM_PRIO 0   // A/B: asymmetric wave priority inside the fused SeparableConv2D (see the kernel)
#endif
template <int F, bool ACT, bool VEC, bool ACT_OUT, bool RES = false>
__global__ __launch_bounds__(256, 2) void sepconv3x3_fused_kernel(
    DwSrc src, const float* __restrict__ dw, const float* __restrict__ pw, const float* __restrict__ bias,
    float* __restrict__ out, int H, int W, int C, int cpad, int tiles_x, int tiles_y, int slices, int n_work) {
    QPWC_FLOW_CHAIN_PRIO();
#if QPWC_SC_ASYM_PRIO
    // Two waves share a SIMD (one of each resident workgroup).  When both are in their matrix phase each runs at half
    // rate and they leave it together -- a stable lock step in which the matrix pipe idles while both do their
    // staging / depthwise / store work.  The wave in the odd hardware slot asks for more issue priority: it finishes its
    // matrix phase at full rate while the other waits, and from then on the two alternate.
    if (__builtin_amdgcn_s_getreg((4) | (0 << 6) | (3 << 11)) & 1) __builtin_amdgcn_s_setprio(3);   // HW_ID.wave_id bit 0
#endif
    // F = output channels of THIS workgroup.  slices > 1 (coarse levels: few tiles, many 32-channel steps): the
    // layer's slices * F outputs are split over `slices` workgroups per tile -- each repeats the (cheap)
    // depthwise convolution and takes 1 / slices of the matrix work, and the launch has slices x more
    // workgroups for a chip the tiles alone would leave mostly idle.
    constexpr int NFT = F / 16;
    constexpr int NST = VEC ? 6 : 23;   // staging loads per thread and step
    __shared__ __attribute__((aligned(16))) float in_s[kScNH * kScInPS];
    __shared__ __attribute__((aligned(16))) float y_s[2 * kScTH * kScTW * kScKC];
    __shared__ __attribute__((aligned(16))) float w_s[F * kScKC];
    __shared__ __attribute__((aligned(16))) float dw_s[9 * kScKC];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    // Work item v = (tile, output slice).  n_work == gridDim.x: one per workgroup (coarse levels, sliced launches).
    // n_work > gridDim.x (round 3, the big levels): RESIDENT workgroups walk the items blockIdx.x, + gridDim.x, ... --
    // the first staging request of the next tile is issued before this tile's last matrix step and its epilogue, so
    // a tile's exposed first load and its store-bound end (phase stamps: 5-7 k and 9-12 k cycles of a 60 k-cycle
    // workgroup) overlap the neighbour's.  b / X0 / Y0 are the tile being FETCHED, eb / eX0 / eY0 the one in the
    // matrix cores.
    int b, X0, Y0, fslice;
    auto locate = [&](int v) __attribute__((always_inline)) {
        const int wg = xcd_swizzle(v, n_work);
        const int tile = wg / slices;                 // the slices of a tile are neighbours (shared input)
        fslice = wg - tile * slices;
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y;
        b = tile / (tiles_x * tiles_y);
        X0 = tx * kScTW;
        Y0 = ty * kScTH;
    };
    locate(blockIdx.x);
    const int n = lane & 15, g = lane >> 4;
    const int FT = F * slices;                    // output pixel stride
    pw += (int64_t)fslice * F * cpad;
    bias += fslice * F;
    out += fslice * F;
#ifdef QPWC_SC_STAMP
    // diagnostic build only (make ab ABSRC=optflow ABFLAGS=-DQPWC_SC_STAMP): shader-clock stamps of one wave
    // of a few workgroups, read back through qpwc_debug_sc_stamps(); no stamp touches an output
    const bool stamp_on = lane == 0 && wave == 0 && (blockIdx.x % 509) == 7 && blockIdx.x / 509 < 4;
    long long* stamp_p = g_sc_stamps + (blockIdx.x / 509) * 64;
    int stamp_i = 0;
#define SC_STAMP() do { if (stamp_on && stamp_i < 62) stamp_p[stamp_i++] = __builtin_amdgcn_s_memtime(); } while (0)
    const bool census_on = lane == 0 && wave == 0 && blockIdx.x < 4096;
    long long* census_p = g_sc_census + (blockIdx.x < 4096 ? blockIdx.x : 0) * 6;
    if (census_on) {
        census_p[0] = __builtin_amdgcn_s_memtime();
        census_p[4] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
        census_p[5] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
    }
#define SC_CENSUS(i) do { if (census_on) census_p[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SC_CENSUS(i) do { } while (0)
#define SC_STAMP() do { } while (0)
#endif

    f32x4v acc[2][NFT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int i = 0; i < NFT; ++i) {
            acc[m][i] = f32x4v{0.f, 0.f, 0.f, 0.f};
#if QPWC_SC_AGPR
            asm volatile("" : "+a"(acc[m][i]));   // accumulators live in the AGPR half of the register file (see QPWC_SC_AGPR)
#endif
        }

    // ---- staging map --------------------------------------------------------------
    // generic: lane = channel (32 lanes = 128 contiguous bytes of one pixel), 8 halo pixels per trip
    // VEC    : 8 lanes x 16 B per pixel, 32 halo pixels per trip (one dense 16-byte-aligned source)
    const int sch = VEC ? 4 * (tid & 7) : (tid & 31);
    const int sps = VEC ? (tid >> 3) : (tid >> 5);
    constexpr int SPT = VEC ? 32 : 8;
    int goff[NST];       // pixel offset of halo pixel `it` inside image b (-1 = outside; H*W < 2^31)
    auto set_goff = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int hp = sps + SPT * it;
            const int hy = hp / kScHW, hx = hp - hy * kScHW;
            const int gy = Y0 - 1 + hy, gx = X0 - 1 + hx;
            goff[it] = (hp < kScNH && gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
        }
    };
    set_goff();
    float4 st4[VEC ? NST : 1];
    float4 wreg0, wreg1, wreg2, wreg3;   // F/32 of them in use (kept out of an array: no LDS promotion)
    float dreg[2];
    wreg0 = wreg1 = wreg2 = wreg3 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch_in = [&](int c0) {   // inputs + depthwise taps, global -> registers, step at channel c0
        const int c = c0 + sch;
        if (VEC) {
            // the quad of channels c..c+3 lies inside one source (every source but the last holds a
            // multiple of 4 channels): 16-byte loads there, element loads for a short last source
            // (Flow/UpFlow's 2-channel flow) -- the host checks alignment and strides
            const float* p = kScZeros;
            int ps = 0;
            int left = 0;   // channels of the source from c on
            if (c < C) {
                const DwPick k = dwsrc_pick(src, c, C);
                p = (const float*)k.p; ps = (int)k.ps; left = k.left;
                p += (int64_t)b * H * W * ps + k.cc;
            }
            {
                // branch-free: halo pixels outside the image and channels past C read a 16-byte block of zeros;
                // pixel offset x pixel stride is a 24-bit multiply (the launcher checks H*W < 2^24).
                // A short last source (Flow/UpFlow's 2-channel flow: `left` = 1..3 channels from c on) is read
                // with the SAME 16-byte load, moved back by 4 - left floats so that it ENDS at the source's last
                // channel (the front of the quad then holds the previous pixel's trailing channels); commit_in()
                // re-aligns and zero-fills.  The first pixels of the tensor have nothing (or too little) in front of them
                // and read forward instead (the launcher checks that the tensor holds >= 8 floats).
                // Element loads here made the wave wait for its prefetch in the middle of the matrix phase.
                const bool full = left >= 4, tail = left > 0 && left < 4;
                const int back = tail ? 4 - left : 0;
#pragma unroll
                for (int it = 0; it < NST; ++it) {
                    const bool ok = goff[it] >= 0 && (full || tail);
                    const float* q = ok ? p : kScZeros;
                    unsigned off = ok ? __umul24((unsigned)goff[it], (unsigned)ps) : 0u;
                    // pixels of image 0 that lie fewer than `back` floats behind the tensor's base (pixel 0; pixels
                    // 1 and 2 of a 1-channel source with stride 1 or 2) would start in front of it: they read forward
                    const bool fwd = b == 0 && off < (unsigned)back;
                    if (ok && tail && !fwd) off -= (unsigned)back;
                    st4[it] = ldg_f4(q + (int)off);   // global_load, not flat (optflow_common.h); |off| < 2^31 (launcher); negative for pixel 0 of image b > 0
                }
            }
        }
        // (the generic multi-source path loads its inputs in commit_in(): 23 more live registers across
        // the depthwise and matrix work would spill; the other workgroup of the CU covers the latency)
#pragma unroll
        for (int i = 0; i < 2; ++i) {        // depthwise taps of the 32 channels: 288 floats
            const int idx = tid + 256 * i;
            dreg[i] = (idx < kScKC * 9 && c0 * 9 + idx < C * 9) ? dw[c0 * 9 + idx] : 0.0f;
        }
    };
    auto fetch_w = [&](int c0) {    // pointwise slice: F rows x 8 chunks of 16 B, 256 chunks per trip
        const float* wp = pw + (int64_t)(tid >> 3) * cpad + c0 + 4 * (tid & 7);
        if (F >= 32 || (tid >> 3) < F) wreg0 = *reinterpret_cast<const float4*>(wp);
        if (F >= 64) wreg1 = *reinterpret_cast<const float4*>(wp + (int64_t)32 * cpad);
        if (F >= 128) {
            wreg2 = *reinterpret_cast<const float4*>(wp + (int64_t)64 * cpad);
            wreg3 = *reinterpret_cast<const float4*>(wp + (int64_t)96 * cpad);
        }
    };
    auto commit_in = [&](int c0) {  // registers -> in_s, dw_s
        if (VEC) {
            // the quad of a short last source was loaded ending at its last channel (fetch_in): move its
            // `left` channels to the front, zero the rest -- one wave-uniform branch, taken in one step per tile
            const int c = c0 + sch;
            const int nfull = src.ch[0] + src.ch[1] + src.ch[2];
            int left = 4;
            if (c < nfull) {
                const int e0 = src.ch[0], e1 = e0 + src.ch[1];
                left = (c < e0 ? e0 : (c < e1 ? e1 : nfull)) - c;
            }
            const bool tail = left < 4;
            const unsigned ps_tail = (unsigned)(src.ch[2] ? src.stride[2] : (src.ch[1] ? src.stride[1] : src.stride[0]));   // a tail is in the last source
            if (__builtin_amdgcn_ballot_w64(tail) != 0) {
#pragma unroll
                for (int it = 0; it < NST; ++it) {
                    const float4 v = st4[it];
                    // forward (or zero block): no shift -- the rule of fetch_in()
                    const bool fwd = goff[it] < 0 || (b == 0 && (int)__umul24((unsigned)goff[it], ps_tail) < 4 - left);
                    const int sh = fwd ? 0 : 4 - left;
                    const float a0 = sh == 0 ? v.x : (sh == 1 ? v.y : (sh == 2 ? v.z : v.w));
                    const float a1 = sh == 0 ? v.y : (sh == 1 ? v.z : v.w);
                    const float a2 = sh == 0 ? v.z : v.w;
                    if (tail) st4[it] = make_float4(a0, left > 1 ? a1 : 0.f, left > 2 ? a2 : 0.f, 0.f);
                }
            }
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int hp = sps + SPT * it;
                if (hp < kScNH) {
                    float4 v = st4[it];
                    if (ACT && goff[it] >= 0) v = make_float4(mishf(v.x), mishf(v.y), mishf(v.z), mishf(v.w));
                    *reinterpret_cast<float4*>(in_s + hp * kScInPS + sch) = v;
                }
            }
        } else {
            const int c = c0 + sch;
            const float* p = nullptr;
            int64_t ps = 0;
            if (c < C) {
                const DwPick k = dwsrc_pick(src, c, C);
                p = (const float*)k.p; ps = k.ps;
                p += (int64_t)b * H * W * ps + k.cc;
            }
            float st[NST];
#pragma unroll
            for (int it = 0; it < NST; ++it) st[it] = (p && goff[it] >= 0) ? ldg_f1(p + (int64_t)goff[it] * ps) : 0.0f;
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int hp = sps + SPT * it;
                if (hp < kScNH) in_s[hp * kScInPS + sch] = (ACT && goff[it] >= 0) ? mishf(st[it]) : st[it];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;   // tap k of channel ch -> dw_s[k][ch]
            if (idx < kScKC * 9) dw_s[(idx % 9) * kScKC + idx / 9] = dreg[i];
        }
    };
    auto commit_w = [&]() {
        // row f = tid>>3 (+32 i), chunk q = tid&7 at slot q ^ ((f & 15) >> 1); (f + 32 i) & 15 == f & 15
        const int f = tid >> 3, q = tid & 7;
        float* wd = w_s + f * kScKC + ((q ^ ((f & 15) >> 1)) << 2);
        if (F >= 32 || f < F) *reinterpret_cast<float4*>(wd) = wreg0;
        if (F >= 64) *reinterpret_cast<float4*>(wd + 32 * kScKC) = wreg1;
        if (F >= 128) {
            *reinterpret_cast<float4*>(wd + 64 * kScKC) = wreg2;
            *reinterpret_cast<float4*>(wd + 96 * kScKC) = wreg3;
        }
    };

    // ---- depthwise map: 4 channels (quad cq) x 4 consecutive pixels of one tile row ----
    const int cq = tid & 7, strip = tid >> 3;
    const int drow = strip >> 2, dxs = (strip & 3) * 4;
    // ---- operand map of the matrix phase ----
    const int sw = n >> 1;

    auto depthwise = [&](float* yd) {   // in_s -> yd
        float4 wq[9];   // tap k of this thread's 4 channels
#pragma unroll
        for (int k = 0; k < 9; ++k) wq[k] = *reinterpret_cast<const float4*>(dw_s + k * kScKC + 4 * cq);
        float4 a[4];
#pragma unroll
        for (int px = 0; px < 4; ++px) a[px] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            float4 r[6];
#pragma unroll
            for (int j = 0; j < 6; ++j)
                r[j] = *reinterpret_cast<const float4*>(in_s + ((drow + ky) * kScHW + dxs + j) * kScInPS + 4 * cq);
#pragma unroll
            for (int px = 0; px < 4; ++px)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 v = r[px + kx];
                    const float4 wk = wq[ky * 3 + kx];
                    a[px].x = fmaf(wk.x, v.x, a[px].x);
                    a[px].y = fmaf(wk.y, v.y, a[px].y);
                    a[px].z = fmaf(wk.z, v.z, a[px].z);
                    a[px].w = fmaf(wk.w, v.w, a[px].w);
                }
        }
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const int pix = drow * kScTW + dxs + px;
            *reinterpret_cast<float4*>(yd + pix * kScKC + ((cq ^ ((pix & 15) >> 1)) << 2)) = a[px];
        }
    };
    auto pointwise = [&](const float* ys) {   // D[f][px] += W[f][k] * y[px][k] on the matrix cores
        if (QPWC_SC_W_AHEAD && F >= 128) {   // (F = 64: 76.3 vs 73.3 us at L4 with it, F = 128: 124 vs 126)
            // Round 4: a block's A operand is read while the block BEFORE it multiplies (both B operands up front).  As the
            // compiler placed them, every eight matrix instructions ended in `ds_read_b128; s_waitcnt lgkmcnt(0)` for the
            // next eight: ~90 cycles of an idle matrix pipe per 256 (ISA of round 3's build).
            f32x4v yv[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    yv[u][m] = *reinterpret_cast<const f32x4v*>(ys + (32 * wave + 16 * m + n) * kScKC + (((4 * u + g) ^ sw) << 2));
            auto read_w = [&](int j) __attribute__((always_inline)) {   // j = u * NFT + ft
                const int u = j / NFT, ft = j - u * NFT;
                return *reinterpret_cast<const f32x4v*>(w_s + (16 * ft + n) * kScKC + (((4 * u + g) ^ sw) << 2));
            };
            f32x4v wv = read_w(0);
#pragma unroll
            for (int j = 0; j < 2 * NFT; ++j) {
                const int u = j / NFT, ft = j - u * NFT;
                f32x4v wn;
                if (j + 1 < 2 * NFT) wn = read_w(j + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        acc[m][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t], yv[u][m][t], acc[m][ft], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (j + 1 < 2 * NFT) wv = wn;
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int coff = ((4 * u + g) ^ sw) << 2;
            f32x4v yv[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
                yv[m] = *reinterpret_cast<const f32x4v*>(ys + (32 * wave + 16 * m + n) * kScKC + coff);
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft) {
                const f32x4v wv = *reinterpret_cast<const f32x4v*>(w_s + (16 * ft + n) * kScKC + coff);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        acc[m][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t], yv[m][t], acc[m][ft], 0, 0, 0);
            }
        }
    };

    // Software pipeline: the matrix work of step k and the depthwise convolution of step k+1 sit in the
    // same barrier interval (y_s double-buffered), so a wave's vector/LDS instructions issue behind its own
    // matrix instructions instead of in a phase of their own.
    //   A: y_s[k&1] complete, w_s and in_s free   -> commit weights(k), inputs(k+1)
    //   B: staged                                 -> prefetch, pointwise(k) || depthwise(k+1)
    const int nsteps = cpad / kScKC;
    constexpr int kYs = kScTH * kScTW * kScKC;
    constexpr bool EARLY_NEXT = F < 64;   // when the next tile's first request is issued (see the last matrix step)
    SC_STAMP();
    fetch_in(0);
    fetch_w(0);
    // resident form: the accumulators are zeroed right before a tile's first matrix step (one-shot form: once, at the
    // top of the kernel, where the compiler sinks the zeros into the first matrix instructions)
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < NFT; ++i) acc[m][i] = f32x4v{0.f, 0.f, 0.f, 0.f};
    };
    // RES (resident workgroups) is compiled in only for F <= 32: the loop keeps ~50 more registers alive (next tile's
    // staged inputs, loop invariants hoisted out of it), which the 64- and 128-output kernels do not have -- they
    // spilled 19 / 58 registers with it -- and the narrow layers are the latency-bound ones (2.5 TB/s at L4) anyway.
    int v = blockIdx.x;
    do {
        // (this tile's step-0 inputs and weights were requested by the prologue above / during the previous tile)
        const int eb = b, eX0 = X0, eY0 = Y0;     // the tile whose outputs this iteration produces
        const bool more = RES && v + (int)gridDim.x < n_work;
        auto next_tile_request = [&]() __attribute__((always_inline)) {
            // the staging registers are free from here to the end of the tile: request the next tile's first step
            locate(v + (int)gridDim.x);
            set_goff();
            fetch_in(0);
        };
        commit_in(0);
        __syncthreads();
        SC_STAMP();
        SC_CENSUS(1);
        if (nsteps > 1) fetch_in(kScKC);
        else if (more && EARLY_NEXT) next_tile_request();
        depthwise(y_s);
        SC_STAMP();
        for (int k = 0; k + 1 < nsteps; ++k) {
            __syncthreads();
            SC_STAMP();
            commit_w();
            commit_in((k + 1) * kScKC);
            __syncthreads();
            SC_STAMP();
            fetch_w((k + 1) * kScKC);
            if (k + 2 < nsteps) fetch_in((k + 2) * kScKC);
            else if (more && EARLY_NEXT) next_tile_request();
            if (RES && k == 0) zero_acc();
            pointwise(y_s + (k & 1) * kYs);
            depthwise(y_s + ((k + 1) & 1) * kYs);
            // issue order: every matrix instruction (32 cycles in its pipe) followed by its share of the
            // ~40 LDS reads and ~120 vector instructions of the depthwise convolution
            constexpr int kNM = 16 * NFT;
#pragma unroll
            for (int i = 0; i < kNM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, (40 + kNM - 1) / kNM, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (120 + kNM - 1) / kNM, 0);
            }
            SC_STAMP();
        }
        __syncthreads();
        SC_STAMP();
        commit_w();
        __syncthreads();
        SC_STAMP();
        SC_CENSUS(2);
        asm volatile("; last step: matrix work, block of 16 outputs by block, each block's bias + Mish + stores behind it");
        // The accumulators of output block ft are final after its 16 matrix instructions: its epilogue (bias,
        // Mish, two 16-byte stores) issues in the matrix pipe's shadow of block ft + 1 instead of after all of
        // them (phase stamps, F = 128: 4.7 k cycles of matrix work followed by 9-12 k cycles of epilogue).
        if (RES && nsteps == 1) zero_acc();
        {
            const float* ys = y_s + ((nsteps - 1) & 1) * kYs;
            f32x4v yv[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    yv[u][m] = *reinterpret_cast<const f32x4v*>(ys + (32 * wave + 16 * m + n) * kScKC + (((4 * u + g) ^ sw) << 2));
            float* orow[2];
            bool ook[2];
            int ooff[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int pix = 32 * wave + 16 * m + n;
                const int gy = eY0 + pix / kScTW, gx = eX0 + pix % kScTW;
                ook[m] = gy < H && gx < W;
                orow[m] = out + ((int64_t)(eb * H + gy) * W + gx) * FT + 4 * g;
                // buffer form: offset inside the tile's band of rows; a column past the image is an out-of-range offset
                // (a row past it is past the descriptor's end): the store is dropped, no branch, no 64-bit address
                ooff[m] = gx < W ? (((pix / kScTW) * W + gx) * FT + 4 * g) * 4 : (int)0x80000000;
            }
            // Round 4: outputs through a buffer descriptor over the tile's band of (at most) 8 rows.  With pointers the
            // compiler re-derived the 64-bit address of every store from (image, row, column) -- v_mad_u64_u32, two
            // v_mul_lo_u32, v_add3, two v_lshl_add_u64 and an exec-mask branch per store, 16 stores per lane -- to keep
            // two pointers out of the register budget during the matrix phase.
            const int band_rows = min(kScTH, H - eY0);
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
                out + (int64_t)(eb * H + eY0) * W * FT, 0, band_rows * W * FT * 4 - fslice * F * 4, 0x00020000);
            // Round 4: ALL of the lane's bias values are requested here, before the first matrix instruction of the step.
            // Loaded inside epilogue(ft), each block's load sat right in front of its use behind an `s_waitcnt vmcnt(0)` --
            // and on gfx950 vmcnt counts STORES too, so every block also waited for the previous block's two output stores
            // to be acknowledged: the last step of a tile took 11.5-12 k cycles against 5 k for the others (stamps of the
            // ping-pong lab kernel, which shares this epilogue).
            float4 bvs[QPWC_SC_BIAS_EARLY ? NFT : 1];
            if (QPWC_SC_BIAS_EARLY) {
#pragma unroll
                for (int ft = 0; ft < NFT; ++ft) bvs[ft] = *reinterpret_cast<const float4*>(bias + 16 * ft + 4 * g);
            }
            auto epilogue = [&](int ft) __attribute__((always_inline)) {
                const float4 bv = QPWC_SC_BIAS_EARLY ? bvs[QPWC_SC_BIAS_EARLY ? ft : 0]
                                                     : *reinterpret_cast<const float4*>(bias + 16 * ft + 4 * g);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    float4 z = make_float4(acc[m][ft][0] + bv.x, acc[m][ft][1] + bv.y, acc[m][ft][2] + bv.z,
                                           acc[m][ft][3] + bv.w);
                    // the activation applied once per output element instead of once per (halo) load of the next layer
                    if (ACT_OUT) z = make_float4(mishf(z.x), mishf(z.y), mishf(z.z), mishf(z.w));
                    if (QPWC_SC_BUF_STORE)
                        // (the block's 64 * ft bytes go into the instruction's immediate offset, NOT the scalar offset: with a
                        // REGISTER there hipcc assumes the store has read its data when the next instruction issues and lets
                        // the Mish of the next values overwrite v[4:7] right behind `buffer_store_dwordx4 v[4:7], .., s4 offen`
                        // -- on gfx950 it has not: channel 4g+1 of one block came out as garbage, run-to-run different)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4sc, z), ro, ooff[m] + 64 * ft, 0, 0);
                    else if (ook[m]) *reinterpret_cast<float4*>(orow[m] + 16 * ft) = z;
                }
            };
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft) {
                f32x4v wv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    wv[u] = *reinterpret_cast<const f32x4v*>(w_s + (16 * ft + n) * kScKC + (((4 * u + g) ^ sw) << 2));
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int m = 0; m < 2; ++m)
                            acc[m][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][t], yv[u][m][t], acc[m][ft], 0, 0, 0);
                // F >= 64: the request for the next tile's first step goes out here, once the last block's matrix
                // instructions are issued and the operand registers are dead (earlier, its 24 staging registers on
                // top of the 64 accumulators and the operands spilled 54 / 16 registers at F = 128 / 64)
                if (!EARLY_NEXT && ft == NFT - 1) {
                    __builtin_amdgcn_sched_barrier(0);   // the scheduler must not hoist these loads above the matrix work
                    if (more) next_tile_request();
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (ft > 0) epilogue(ft - 1);
            }
            SC_STAMP();
            epilogue(NFT - 1);
        }
        SC_CENSUS(3);
        SC_STAMP();
        // the next tile's first weight slice (an L2 hit) is requested only now: held across the epilogue together
        // with the staged inputs it cost 54 spilled registers at F = 128
        if (more) fetch_w(0);
        // (no barrier here: in_s / dw_s were last read before the two barriers around commit_w, and y_s / w_s are not
        // written again before the barrier that follows the next tile's commit_in(0))
    } while (RES && (v += (int)gridDim.x) < n_work);
}

#ifdef QPWC_SC_WS
#include "experimental/sepconv_role_split.inc"
#endif

#ifdef QPWC_SC_PP     // lab note (round 4): eight-wave ping-pong, make ab ABSRC=optflow ABFLAGS=-DQPWC_SC_PP=1
#include <type_traits>
#include "experimental/sepconv_pp.inc"
#endif
#ifdef QPWC_SC_FLAT   // lab note (round 4): parity-correct only by luck of the register allocation, slower -- never in the product build
#include "experimental/sepconv_flat.inc"
#endif

#ifndef QPWC_SC_SPLIT_ONE_ROUND
#define QPWC_SC_SPLIT_ONE_ROUND 0   // 0 = off; else the smallest F it applies to
#endif
#ifndef QPWC_SC_SLICE_TARGET
#define QPWC_SC_SLICE_TARGET 192   // split a layer's outputs over workgroups until the launch has this many
#endif
#ifndef QPWC_SC_RESIDENT
#define QPWC_SC_RESIDENT 512   // resident workgroups of the fp32 fused SeparableConv2D (0 = one workgroup per tile)
#endif
template <int F>
static void sepconv_dispatch(const DwSrc& d, int act, bool vec, const float* dw, const float* pw,
                             const float* bias, float* out, int H, int W, int C, int cpad, int tiles_x,
                             int tiles_y, dim3 grid, int slices, hipStream_t s) {
    // round 3: where a launch would be several rounds of one-shot workgroups (the big levels), 512 RESIDENT workgroups
    // (2 per CU, what the kernel's 79 KB of LDS and 248 registers allow) walk the tiles instead
    const int n_work = (int)grid.x;
    const bool resident = F <= 32 && QPWC_SC_RESIDENT > 0 && slices == 1 && n_work > QPWC_SC_RESIDENT;
    if (resident) grid.x = QPWC_SC_RESIDENT;
#define QPWC_SC_LAUNCH(ACT, VEC, AO)                                                                               \
    do {                                                                                                           \
        if (F <= 32 && resident)                                                                                   \
            hipLaunchKernelGGL((sepconv3x3_fused_kernel<F, ACT, VEC, AO, (F <= 32)>), grid, dim3(256), 0, s, d, dw, \
                               pw, bias, out, H, W, C, cpad, tiles_x, tiles_y, slices, n_work);                    \
        else                                                                                                       \
            hipLaunchKernelGGL((sepconv3x3_fused_kernel<F, ACT, VEC, AO, false>), grid, dim3(256), 0, s, d, dw, pw, \
                               bias, out, H, W, C, cpad, tiles_x, tiles_y, slices, n_work);                        \
    } while (0)
    const bool in_act = (act & 1) != 0, out_act = (act & 2) != 0;   // QPWC_MISH_ON_LOAD / _ON_STORE
    if (out_act) {
        if (in_act) { if (vec) QPWC_SC_LAUNCH(true, true, true); else QPWC_SC_LAUNCH(true, false, true); }
        else        { if (vec) QPWC_SC_LAUNCH(false, true, true); else QPWC_SC_LAUNCH(false, false, true); }
    } else {
        if (in_act) { if (vec) QPWC_SC_LAUNCH(true, true, false); else QPWC_SC_LAUNCH(true, false, false); }
        else        { if (vec) QPWC_SC_LAUNCH(false, true, false); else QPWC_SC_LAUNCH(false, false, false); }
    }
#undef QPWC_SC_LAUNCH
}

int sepconv3x3_launch(const void* const* srcs, const int* chans, const int64_t* strides, int n_src,
                      int act, const void* dw, const void* pw, const void* bias, void* out, int B, int H,
                      int W, int F, hipStream_t s) {
    DwSrc d;
    int C = 0;
    for (int i = 0; i < 3; ++i) {
        d.ptr[i] = i < n_src ? srcs[i] : nullptr;
        d.ch[i] = i < n_src ? chans[i] : 0;
        d.stride[i] = i < n_src ? strides[i] : 0;
        C += d.ch[i];
    }
    const int cpad = (C + kScKC - 1) / kScKC * kScKC;
    const int tiles_x = (W + kScTW - 1) / kScTW, tiles_y = (H + kScTH - 1) / kScTH;
    const int64_t nblk = (int64_t)tiles_x * tiles_y * B;
    if (nblk > INT32_MAX) {
        set_error("sepconv3x3: too many tiles");
        return QPWC_E_SHAPE;
    }
    if (QPWC_SC_BUF_STORE && (int64_t)kScTH * W * F * 4 >= 0x7fffffff) {   // the output descriptor spans a tile's band of rows
        set_error("sepconv3x3: 8 rows of outputs must stay below 2 GiB");
        return QPWC_E_SHAPE;
    }
    // few tiles (coarse pyramid levels): split the outputs over 2 / 4 / 8 workgroups per tile (slices of >= 16
    // outputs) until the launch has ~one workgroup per CU
    int slices = 1;
    while (nblk * slices < QPWC_SC_SLICE_TARGET && F / (slices * 2) >= 16) slices *= 2;
    // A/B (round 4): a launch of 257..512 tiles is ONE round of two workgroups per CU -- every workgroup's prologue and
    // epilogue exposed at the same time; with its outputs split over two workgroups per tile it is two rounds
    if (QPWC_SC_SPLIT_ONE_ROUND && slices == 1 && nblk > 256 && nblk <= 512 && F >= QPWC_SC_SPLIT_ONE_ROUND) slices = 2;
    const dim3 grid((unsigned)(nblk * slices));
    // 16-byte loads: every source but the last holds a multiple of 4 channels in 16-byte aligned
    // pixels; the last one either does too or is read element-wise (it must not straddle a quad
    // boundary with its predecessor, which the multiple-of-4 rule guarantees)
    bool vec = true;
    for (int i = 0; i < n_src; ++i) {
        const bool aligned = chans[i] % 4 == 0 && strides[i] % 4 == 0 && reinterpret_cast<uintptr_t>(srcs[i]) % 16 == 0;
        if (!aligned && i + 1 < n_src) vec = false;                  // only the last source may be odd
        if (!aligned && i + 1 == n_src && chans[i] >= 4) vec = false; // ... and only if it is a short tail
        // the tail quads of the first pixels (those fewer than 3 floats behind the base) are read forward: the tensor
        // must hold 4 floats from there
        if (!aligned && i + 1 == n_src && (int64_t)B * H * W * strides[i] < 8) vec = false;
        // the 16-byte path multiplies pixel offset and pixel stride as 24-bit integers into 32 bits and adds the
        // product as a signed 32-bit element offset
        if ((int64_t)H * W >= (1 << 24) || strides[i] >= (1 << 24) || (int64_t)H * W * strides[i] >= ((int64_t)1 << 31))
            vec = false;
    }
    const float *fdw = (const float*)dw, *fpw = (const float*)pw, *fb = (const float*)bias;
    if (F != 16 && F != 32 && F != 64 && F != 128) {
        set_error("sepconv3x3: unsupported filter count %d (16/32/64/128)", F);
        return QPWC_E_SHAPE;
    }
#ifdef QPWC_SC_WS
    {
        // role-split kernel: every source but the last in whole 16-byte chunks (the last may be a short tail),
        // no activation on load, byte offsets inside one image below 2^31
        bool ws = vec && slices == 1 && nblk >= QPWC_SC_WS && (act & 1) == 0;
        for (int i = 0; i < n_src; ++i)
            if (((int64_t)H * W * strides[i] + 4) * 4 >= 0x7fffffff) ws = false;
        if (ws) {
            const bool oa = (act & 2) != 0;
#define QPWC_WS_LAUNCH(FF, AO)                                                                           \
    hipLaunchKernelGGL((sepconv3x3_ws_kernel<FF, AO>), grid, dim3(512), 0, s, d, fdw, fpw, fb, (float*)out, \
                       B, H, W, C, cpad, tiles_x, tiles_y)
#define QPWC_WS_F(FF) do { if (oa) QPWC_WS_LAUNCH(FF, true); else QPWC_WS_LAUNCH(FF, false); } while (0)
            switch (F) {
                case 128: QPWC_WS_F(128); break;
                case 64: QPWC_WS_F(64); break;
                case 32: QPWC_WS_F(32); break;
                default: QPWC_WS_F(16); break;
            }
#undef QPWC_WS_F
#undef QPWC_WS_LAUNCH
            return check_launch("sepconv3x3_ws_kernel");
        }
    }
#endif
#ifdef QPWC_SC_PP
    if (QPWC_SC_PP && vec && slices == 1 && (F == 64 || F == 128) && nblk >= QPWC_SC_PP_MIN_TILES &&
        (int64_t)H * W * F * 4 < 0x7fffffff) {
        if (dry_run(F == 128 ? "sepconv3x3_pp_kernel<128>" : "sepconv3x3_pp_kernel<64>")) return QPWC_OK;
        if (F == 128)
            sepconv_pp_dispatch<128>(d, act, fdw, fpw, fb, (float*)out, H, W, C, cpad, tiles_x, tiles_y, (int)nblk, s);
        else
            sepconv_pp_dispatch<64>(d, act, fdw, fpw, fb, (float*)out, H, W, C, cpad, tiles_x, tiles_y, (int)nblk, s);
        return check_launch("sepconv3x3_pp_kernel");
    }
#endif
#ifdef QPWC_SC_FLAT
    // round 4: the wide layers of the big levels as ONE flat software pipeline per resident workgroup (sepconv_flat.inc)
    if (vec && slices == 1 && (F == 64 || F == 128) && nblk >= QPWC_SC_FLAT_MIN_TILES &&
        (int64_t)H * W * F * 4 < 0x7fffffff) {
        if (dry_run(F == 128 ? "sepconv3x3_flat_kernel<64> x 2 slices" : "sepconv3x3_flat_kernel<64>")) return QPWC_OK;
        sepconv_flat_dispatch<64>(d, act, fdw, fpw, fb, (float*)out, H, W, C, cpad, tiles_x, tiles_y, (int)nblk, F / 64, s);
        return check_launch("sepconv3x3_flat_kernel");
    }
#endif
    switch (F / slices) {   // outputs per workgroup
        case 128: sepconv_dispatch<128>(d, act, vec, fdw, fpw, fb, (float*)out, H, W, C, cpad, tiles_x, tiles_y, grid, slices, s); break;
        case 64: sepconv_dispatch<64>(d, act, vec, fdw, fpw, fb, (float*)out, H, W, C, cpad, tiles_x, tiles_y, grid, slices, s); break;
        case 32: sepconv_dispatch<32>(d, act, vec, fdw, fpw, fb, (float*)out, H, W, C, cpad, tiles_x, tiles_y, grid, slices, s); break;
        default: sepconv_dispatch<16>(d, act, vec, fdw, fpw, fb, (float*)out, H, W, C, cpad, tiles_x, tiles_y, grid, slices, s); break;
    }
    return check_launch("sepconv3x3_fused_kernel");
}

#include "sepconv_x3.inc"

// ---------------------------------------------------------------------------
// flow head.  params (device, fp32): w1[16][16] (out,in) | b1[16] | bn_scale[16] |
// bn_shift[16] | wf[3][3][16][2] (ky,kx,in,out)   = 256+16+16+16+288 = 592 floats
constexpr int kFhC = 16;
constexpr int kFhTile = 16;
constexpr int kFhParams = 592;  // == qpwc_flow_head_param_floats()

// Both convolutions run lane = (pixel n, channel quad g), n = lane & 15, g = lane >> 4:
//   1x1 conv : v_mfma_f32_16x16x4_f32, rows = the 16 outputs (A = W1: 4 registers, loaded once), cols =
//              16 pixels; a lane's B operand is Mish of its own 16-byte load of z, and the result lane
//              holds outputs 4g..4g+3 of pixel n -> + b1, Mish, BatchNorm -> one ds_write_b128 of h;
//   3x3 conv : each lane keeps the 72 weights of its 4 input channels, accumulates its partial
//              (fx, fy) over the 9 taps from ds_read_b128 of h, and the four quads of a pixel are
//              added with two wave shuffles.
// No weight streams through scalar registers any more (the 592 parameters, re-fetched by s_load
// for every pixel round, made the old form latency bound: 15 us at every level).
template <typename T>
__global__ __launch_bounds__(256, 2) void flow_head_kernel(const T* __restrict__ z,
                                                           const float* __restrict__ params,
                                                           T* __restrict__ out, int H, int W,
                                                           int tiles_x, int tiles_y, float scale,
                                                           int out_nchw) {
    QPWC_FLOW_CHAIN_PRIO();
    constexpr int TW = kFhTile + 2;
    constexpr int NH = TW * TW;                                        // 324 halo pixels
    __shared__ __attribute__((aligned(16))) float hs[(NH + 12) * kFhC];  // 21 groups of 16 pixels
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int x0 = tx * kFhTile, y0 = ty * kFhTile;
    const float* w1 = params;
    const float* b1 = params + 256;
    const float* bs = params + 272;
    const float* bt = params + 288;
    const float* wf = params + 304;
    const T* zb = z + (int64_t)b * H * W * kFhC;

    // ---- h = BN(mish(W1 mish(z) + b1)) for the tile + 1 halo; zero outside the image ----
    const f32x4v w1v = *reinterpret_cast<const f32x4v*>(w1 + n * kFhC + 4 * g);   // W1[f = n][4g..4g+3]
    const float4 b1v = *reinterpret_cast<const float4*>(b1 + 4 * g);
    const float4 bsv = *reinterpret_cast<const float4*>(bs + 4 * g);
    const float4 btv = *reinterpret_cast<const float4*>(bt + 4 * g);
    // the 72 weights of this lane's 4 input channels, wf[ky][kx][in][out] (in flight with the halo loads)
    float4 wq[9][2];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        wq[k][0] = *reinterpret_cast<const float4*>(wf + k * kFhC * 2 + 8 * g);       // (c0:x,y) (c1:x,y)
        wq[k][1] = *reinterpret_cast<const float4*>(wf + k * kFhC * 2 + 8 * g + 4);   // (c2:x,y) (c3:x,y)
    }
    // a wave takes halo groups wave, wave+4, ... (at most 6); all of its loads are issued before the first
    // use (one exposed memory latency per workgroup instead of one per group)
    constexpr int NG = (NH + 15) / 16, NGW = (NG + 3) / 4;
    float4 av[NGW];
    bool inb[NGW];
#pragma unroll
    for (int i = 0; i < NGW; ++i) {
        const int hp = 16 * (wave + 4 * i) + n;
        const int ly = hp / TW, lx = hp - ly * TW;
        const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
        inb[i] = hp < NH && gy >= 0 && gy < H && gx >= 0 && gx < W;
        av[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (inb[i]) av[i] = ld4(zb + ((int64_t)gy * W + gx) * kFhC + 4 * g);
    }
#pragma unroll
    for (int i = 0; i < NGW; ++i) {
        const int grp = wave + 4 * i;
        if (grp >= NG) break;   // wave-uniform
        const int hp = 16 * grp + n;
        float4 a = av[i];
        if (inb[i]) a = make_float4(mishf(a.x), mishf(a.y), mishf(a.z), mishf(a.w));
        f32x4v d = {0.f, 0.f, 0.f, 0.f};
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[0], a.x, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[1], a.y, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[2], a.z, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[3], a.w, d, 0, 0, 0);
        float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
        if (inb[i])
            h = make_float4(fmaf(mishf(d[0] + b1v.x), bsv.x, btv.x), fmaf(mishf(d[1] + b1v.y), bsv.y, btv.y),
                            fmaf(mishf(d[2] + b1v.z), bsv.z, btv.z), fmaf(mishf(d[3] + b1v.w), bsv.w, btv.w));
        *reinterpret_cast<float4*>(hs + hp * kFhC + 4 * g) = h;
    }
    __syncthreads();

    // ---- 3x3 conv 16 -> 2: a wave takes tile rows wave, wave+4, ...; lane = (column n, quad g) ----
    for (int ry = wave; ry < kFhTile; ry += 4) {
        float fx = 0.0f, fy = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 v = *reinterpret_cast<const float4*>(hs + ((ry + ky) * TW + n + kx) * kFhC + 4 * g);
                const float4 wa = wq[ky * 3 + kx][0], wb = wq[ky * 3 + kx][1];
                fx = fmaf(v.x, wa.x, fx); fy = fmaf(v.x, wa.y, fy);
                fx = fmaf(v.y, wa.z, fx); fy = fmaf(v.y, wa.w, fy);
                fx = fmaf(v.z, wb.x, fx); fy = fmaf(v.z, wb.y, fy);
                fx = fmaf(v.w, wb.z, fx); fy = fmaf(v.w, wb.w, fy);
            }
        fx += __shfl_xor(fx, 16); fy += __shfl_xor(fy, 16);
        fx += __shfl_xor(fx, 32); fy += __shfl_xor(fy, 32);
        const int gx = x0 + n, gy = y0 + ry;
        if (g == 0 && gx < W && gy < H) {
            if (out_nchw) {   // (B,2,H,W): the model's 'channels_first' output, no transposition launch
                T* o = out + ((int64_t)(b * 2) * H + gy) * W + gx;
                st(o, scale * fx);
                st(o + (int64_t)H * W, scale * fy);
            } else {
                T* o = out + ((int64_t)(b * H + gy) * W + gx) * 2;
                st(o, scale * fx);
                st(o + 1, scale * fy);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// x = Mish(x + bias[c]) in place on a channels-last tensor: the `activation='Mish'`
// epilogue of the reference's Conv2D / Conv2DTranspose blocks (non_layers.py:196-210,
// 390-449).  The library convolution runs without bias; bias add and activation are
// one bandwidth-bound pass instead of two.
template <typename T>
__global__ __launch_bounds__(256) void bias_mish_kernel(T* __restrict__ x,
                                                        const float* __restrict__ bias, int64_t n4,
                                                        int c4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = ld4(x + 4 * i);
        if (bias) {
            const float4 b = reinterpret_cast<const float4*>(bias)[i % c4];
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        v.x = mishf(v.x); v.y = mishf(v.y); v.z = mishf(v.z); v.w = mishf(v.w);
        st4(x + 4 * i, v);
    }
}

// Out-of-place form that also produces TensorFlow's 'SAME' padding for the next stride-2
// convolution (0 before, pad_h/pad_w after): dst is (B, H+pad_h, W+pad_w, C), its interior
// gets Mish(src + bias) and its border zeros -- the separate F.pad copy disappears.
template <typename T>
__global__ __launch_bounds__(256) void bias_mish_pad_kernel(const T* __restrict__ src,
                                                            const float* __restrict__ bias,
                                                            T* __restrict__ dst, int H, int W, int c4,
                                                            int Hp, int Wp, int64_t n4, int64_t dps) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = i % c4;
        int64_t p = i / c4;
        const int x = p % Wp;
        p /= Wp;
        const int y = p % Hp;
        const int b = p / Hp;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y < H && x < W) {
            v = ld4(src + ((((int64_t)b * H + y) * W + x) * c4 + c) * 4);
            if (bias) {
                const float4 bv = reinterpret_cast<const float4*>(bias)[c];
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
            v.x = mishf(v.x); v.y = mishf(v.y); v.z = mishf(v.z); v.w = mishf(v.w);
        }
        st4(dst + (i / c4) * dps + 4 * c, v);
    }
}

int bias_mish_pad_launch(const void* src, const void* bias, void* dst, int B, int H, int W, int C,
                         int pad_h, int pad_w, int64_t dst_pixel_stride, int dtype, hipStream_t s) {
    const int Hp = H + pad_h, Wp = W + pad_w;
    const int64_t n4 = (int64_t)B * Hp * Wp * (C / 4);
    const int64_t want = (n4 + 255) / 256;
    const dim3 grid((unsigned)(want < 16384 ? want : 16384));
    if (dtype == QPWC_F32)
        hipLaunchKernelGGL(bias_mish_pad_kernel<float>, grid, dim3(256), 0, s, (const float*)src,
                           (const float*)bias, (float*)dst, H, W, C / 4, Hp, Wp, n4, dst_pixel_stride);
    else
        hipLaunchKernelGGL(bias_mish_pad_kernel<__half>, grid, dim3(256), 0, s, (const __half*)src,
                           (const float*)bias, (__half*)dst, H, W, C / 4, Hp, Wp, n4, dst_pixel_stride);
    return check_launch("bias_mish_pad_kernel");
}

int bias_mish_launch(void* x, const void* bias, int64_t n_pixels, int C, int dtype, hipStream_t s) {
    const int64_t n4 = n_pixels * C / 4;
    const int64_t want = (n4 + 255) / 256;
    const unsigned grid = (unsigned)(want < 16384 ? want : 16384);
    if (dtype == QPWC_F32)
        hipLaunchKernelGGL(bias_mish_kernel<float>, dim3(grid), dim3(256), 0, s, (float*)x,
                           (const float*)bias, n4, C / 4);
    else
        hipLaunchKernelGGL(bias_mish_kernel<__half>, dim3(grid), dim3(256), 0, s, (__half*)x,
                           (const float*)bias, n4, C / 4);
    return check_launch("bias_mish_kernel");
}

// ---------------------------------------------------------------------------
// Upsample(scale): scale * UpSampling2D(2, 'bilinear') of a flow field (B,h,w,2)
// (non_layers.py:183-193; half-pixel centres, edge-clamped: src = max(0, (dst+.5)/2 - .5)).
// scale * (w00 v00 + w01 v01 + w10 v10 + w11 v11) as ONE explicit chain of fused multiply-adds: both upsampling kernels
// call it, so that they agree bit for bit whatever the compiler would contract on its own
__device__ __forceinline__ float2 bilerp_flow(float ly, float lx, float2 v00, float2 v01, float2 v10, float2 v11, float scale) {
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    float2 r;
    r.x = scale * fmaf(w11, v11.x, fmaf(w10, v10.x, fmaf(w01, v01.x, w00 * v00.x)));
    r.y = scale * fmaf(w11, v11.y, fmaf(w10, v10.y, fmaf(w01, v01.y, w00 * v00.y)));
    return r;
}

template <typename T>
__global__ __launch_bounds__(256) void upsample2x_flow_kernel(const T* __restrict__ in,
                                                              T* __restrict__ out, int B, int h,
                                                              int w, float scale, int in_nchw, int out_nchw) {
    QPWC_FLOW_CHAIN_PRIO();
    const int H = 2 * h, W = 2 * w;
    const int64_t total = (int64_t)B * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = idx % W;
        const int y = (idx / W) % H;
        const int b = idx / ((int64_t)W * H);
        const float sy = fmaxf(0.0f, (y + 0.5f) * 0.5f - 0.5f), sx = fmaxf(0.0f, (x + 0.5f) * 0.5f - 0.5f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + 1 < h ? y0 + 1 : h - 1, x1 = x0 + 1 < w ? x0 + 1 : w - 1;
        const float ly = sy - y0, lx = sx - x0;
        const T* p = in + (int64_t)b * h * w * 2;
        auto px = [&](int yy, int xx) {
            if (in_nchw) {   // (B,2,h,w): what flow_head writes for a 'channels_first' model
                const T* q = p + (int64_t)yy * w + xx;
                return make_float2(ld(q), ld(q + (int64_t)h * w));
            }
            const T* q = p + ((int64_t)yy * w + xx) * 2;
            return make_float2(ld(q), ld(q + 1));
        };
        const float2 v00 = px(y0, x0), v01 = px(y0, x1), v10 = px(y1, x0), v11 = px(y1, x1);
        const float2 r = bilerp_flow(ly, lx, v00, v01, v10, v11, scale);
        if (out_nchw) {
            T* o = out + ((int64_t)(b * 2) * H + y) * W + x;
            st(o, r.x);
            st(o + (int64_t)H * W, r.y);
        } else {
            st(out + 2 * idx, r.x);
            st(out + 2 * idx + 1, r.y);
        }
    }
}

// Channels-last in and out (every call of the channels_last network): a thread owns TWO horizontally adjacent output
// pixels -- the arithmetic of the kernel above (bilerp_flow: bit-identical results), with each input
// pixel read as one 8- / 4-byte vector and the pair written as one 16- / 8-byte store, 32-bit indices
// (round 3: step 1.1749 -> 1.1703 ms, config 5's 1.6226 -> 1.6115 ms).
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_flow_pair_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                                   int n_pairs, int h, int w, float scale) {
    QPWC_FLOW_CHAIN_PRIO();
    const int H = 2 * h;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n_pairs; idx += gridDim.x * 256) {
        const int j = idx % w;
        const int t = idx / w;
        const int y = t % H, b = t / H;
        const float sy = fmaxf(0.0f, (y + 0.5f) * 0.5f - 0.5f);
        const int y0 = (int)sy;
        const int y1 = y0 + 1 < h ? y0 + 1 : h - 1;
        const float ly = sy - y0;
        const T* p = in + (int64_t)b * h * w * 2;
        auto px = [&](int yy, int xx) __attribute__((always_inline)) {
            if constexpr (sizeof(T) == 4) {
                return *reinterpret_cast<const float2*>(p + (yy * w + xx) * 2);
            } else {
                return __half22float2(*reinterpret_cast<const __half2*>(p + (yy * w + xx) * 2));
            }
        };
        float2 r[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int x = 2 * j + e;
            const float sx = fmaxf(0.0f, (x + 0.5f) * 0.5f - 0.5f);
            const int x0 = (int)sx;
            const int x1 = x0 + 1 < w ? x0 + 1 : w - 1;
            const float lx = sx - x0;
            const float2 v00 = px(y0, x0), v01 = px(y0, x1), v10 = px(y1, x0), v11 = px(y1, x1);
            r[e] = bilerp_flow(ly, lx, v00, v01, v10, v11, scale);
        }
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float4*>(out + (int64_t)idx * 4) = make_float4(r[0].x, r[0].y, r[1].x, r[1].y);
        } else {
            const __half2 a = __floats2half2_rn(r[0].x, r[0].y), c = __floats2half2_rn(r[1].x, r[1].y);
            uint2 u;
            u.x = *reinterpret_cast<const unsigned*>(&a);
            u.y = *reinterpret_cast<const unsigned*>(&c);
            *reinterpret_cast<uint2*>(out + (int64_t)idx * 4) = u;
        }
    }
}

#ifndef QPWC_UPSAMPLE_PAIRS
#define QPWC_UPSAMPLE_PAIRS 1   // 0: every call on the one-pixel-per-thread kernel (A/B)
#endif
int upsample2x_flow_launch(const void* in, void* out, int B, int h, int w, float scale, int dtype,
                           int in_layout, int out_layout, hipStream_t s) {
    const int in_nchw = in_layout == QPWC_NCHW, out_nchw = out_layout == QPWC_NCHW;
    const int64_t n_pairs = (int64_t)B * 2 * h * w;
    const size_t es = dtype == QPWC_F32 ? 4 : 2;
    if (QPWC_UPSAMPLE_PAIRS && !in_nchw && !out_nchw && n_pairs * 4 < INT32_MAX &&
        reinterpret_cast<uintptr_t>(in) % (2 * es) == 0 && reinterpret_cast<uintptr_t>(out) % (4 * es) == 0) {
        const int64_t wantp = (n_pairs + 255) / 256;
        const dim3 gridp((unsigned)(wantp < 16384 ? wantp : 16384));
        if (dtype == QPWC_F32)
            hipLaunchKernelGGL(upsample2x_flow_pair_kernel<float>, gridp, dim3(256), 0, s, (const float*)in, (float*)out,
                               (int)n_pairs, h, w, scale);
        else
            hipLaunchKernelGGL(upsample2x_flow_pair_kernel<__half>, gridp, dim3(256), 0, s, (const __half*)in,
                               (__half*)out, (int)n_pairs, h, w, scale);
        return check_launch("upsample2x_flow_pair_kernel");
    }
    const int64_t total = (int64_t)B * 4 * h * w;
    const int64_t want = (total + 255) / 256;
    const dim3 grid((unsigned)(want < 8192 ? want : 8192));
    if (dtype == QPWC_F32)
        hipLaunchKernelGGL(upsample2x_flow_kernel<float>, grid, dim3(256), 0, s, (const float*)in,
                           (float*)out, B, h, w, scale, in_nchw, out_nchw);
    else
        hipLaunchKernelGGL(upsample2x_flow_kernel<__half>, grid, dim3(256), 0, s, (const __half*)in,
                           (__half*)out, B, h, w, scale, in_nchw, out_nchw);
    return check_launch("upsample2x_flow_kernel");
}

// ---------------------------------------------------------------------------
// Split(2) of the (B,H,W,6) input pair (pwcnet.py:229) + stacking of the two frames on the
// batch axis (the encoder weights are shared) + the 'SAME' padding of the first stride-2
// conv, in one pass: out (2B, H+pad_h, W+pad_w, 3), frame f of pair b at batch f*B + b.
template <typename T>
__global__ __launch_bounds__(256) void split_frames_pad_kernel(const T* __restrict__ in,
                                                               T* __restrict__ out, int B, int H, int W,
                                                               int Hp, int Wp, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = idx % 3;
        int64_t p = idx / 3;
        const int x = p % Wp;
        p /= Wp;
        const int y = p % Hp;
        const int b2 = p / Hp;
        const int f = b2 / B, b = b2 - f * B;
        float v = 0.0f;
        if (y < H && x < W) v = ld(in + (((int64_t)b * H + y) * W + x) * 6 + f * 3 + c);
        st(out + idx, v);
    }
}

int split_frames_pad_launch(const void* in, void* out, int B, int H, int W, int pad_h, int pad_w, int dtype,
                            hipStream_t s) {
    const int Hp = H + pad_h, Wp = W + pad_w;
    const int64_t total = (int64_t)2 * B * Hp * Wp * 3;
    const int64_t want = (total + 255) / 256;
    const dim3 grid((unsigned)(want < 32768 ? want : 32768));
    if (dtype == QPWC_F32)
        hipLaunchKernelGGL(split_frames_pad_kernel<float>, grid, dim3(256), 0, s, (const float*)in,
                           (float*)out, B, H, W, Hp, Wp, total);
    else
        hipLaunchKernelGGL(split_frames_pad_kernel<__half>, grid, dim3(256), 0, s, (const __half*)in,
                           (__half*)out, B, H, W, Hp, Wp, total);
    return check_launch("split_frames_pad_kernel");
}

// ---------------------------------------------------------------------------
// optflow_tail: the last two SeparableConv2D (64 -> 32 -> 16), the flow head and the scale of OptFlow
// (qpwcnet/core/non_layers.py:223-231, 238-254, 268-273) in ONE launch, for the coarse pyramid levels.
//
// There the four launches it replaces (two fused SeparableConv2D, flow head; 5-16 us each in the step, next to the
// decoder's chip-filling launches) are bound by their own start-up latency, not by their work.  A workgroup owns
// an 8 x 8 pixel tile of the flow and recomputes the halo every 3x3 stage needs: z2 (input, 64 channels,
// Mish-activated by its producer) on 14 x 14 pixels -> z3 = Mish(sepconv3) on 12 x 12 -> z4 = sepconv4 on 10 x 10
// -> h = BN(Mish(W1 Mish(z4) + b1)) on 10 x 10 -> flow = scale * conv3x3(h) on 8 x 8.  2.25 x the matrix work of
// the first of those layers -- irrelevant at these sizes -- for one prologue instead of four.
// Every intermediate is ZERO outside the image (Keras 'same' padding pads each layer's input).
//   phase A (2 steps of 32 input channels): stage z2 -> depthwise 3x3 (thread = 4 channels x 5 pixels) -> y3 ->
//            pointwise on the matrix cores, rows = outputs, cols = 16 pixels (9 groups x 2 output blocks over 4 waves)
//   phase B: depthwise 3x3 on z3 -> y4 -> pointwise 32 -> 16 (7 pixel groups)
//   phase C: the accumulator layout (pixel n, outputs 4g..4g+3) IS the next operand layout: Mish, W1 on the matrix
//            cores, Mish, BatchNorm, to LDS
//   phase D: 3x3 conv 16 -> 2 as in flow_head_kernel (lane = pixel x channel quad, two wave shuffles)
// LDS 80 KB (fits beside one of the decoder's 74 KB workgroups): z2 tile (pixel stride 40 floats) | y3 | z3 (stride
// 40) | weights; y4 aliases the z2 tile, h aliases y3.
constexpr int kTlT = 8;                                  // flow tile
constexpr int kTlR2 = kTlT + 6, kTlR3 = kTlT + 4, kTlR4 = kTlT + 2;
constexpr int kTlN2 = kTlR2 * kTlR2, kTlN3 = kTlR3 * kTlR3, kTlN4 = kTlR4 * kTlR4;   // 196, 144, 100 pixels
constexpr int kTlG3 = kTlN3 / 16, kTlG4 = (kTlN4 + 15) / 16;                          // 9, 7 pixel groups
constexpr int kTlPS = 40;                                // floats per pixel of the depthwise inputs

template <bool ACT_IN>
__global__ __launch_bounds__(256, 2) void optflow_tail_kernel(
    const float* __restrict__ z2, const float* __restrict__ dw3, const float* __restrict__ pw3,
    const float* __restrict__ b3, const float* __restrict__ dw4, const float* __restrict__ pw4,
    const float* __restrict__ b4, const float* __restrict__ head, float* __restrict__ out, int H, int W,
    int tiles_x, int tiles_y, float scale, int out_nchw) {
    QPWC_FLOW_CHAIN_PRIO();
    __shared__ __attribute__((aligned(16))) float in_s[kTlN2 * kTlPS];          // z2 tile; later y4 [112][32]
    __shared__ __attribute__((aligned(16))) float y3_s[kTlN3 * 32];             // y3; later h [112][16]
    __shared__ __attribute__((aligned(16))) float z3_s[kTlN3 * kTlPS];
    __shared__ __attribute__((aligned(16))) float w3_s[32 * 32];
    __shared__ __attribute__((aligned(16))) float w4_s[16 * 32];
    __shared__ __attribute__((aligned(16))) float dw_s[9 * 32];
    __shared__ __attribute__((aligned(16))) float wf_s[9 * 32];                 // taps of the final 3 x 3 convolution
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4, sw = n >> 1;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int X0 = tx * kTlT, Y0 = ty * kTlT;
    const float* zb = z2 + (int64_t)b * H * W * 64;
    const int cq = tid & 7, pr = tid >> 3;      // depthwise map: channel quad, pixel row of 32

    // Round 3: EVERY global load of the workgroup is issued here, before the first barrier -- both 32-channel steps of
    // the input tile, both weight slices, all depthwise taps, biases, flow-head parameters.  The kernel used to pay one
    // exposed L2 / HBM round trip per phase (step 0, step 1, layer-4 taps, head parameters, the 18 tap loads of the final
    // 3 x 3 convolution): five to six of them in a 17 us workgroup whose arithmetic is a few us.
    // (wr as an array of two was not promoted to registers: it went to scratch memory behind an s_waitcnt vmcnt(0),
    // which undid the prefetch -- two named variables instead)
    float4 st[2][7], wr0, wr1;
    float dr[2][2], dr4[2];
    auto request_step = [&](auto step_c) __attribute__((always_inline)) {
        constexpr int s = decltype(step_c)::value;
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int idx = tid + 256 * it;
            const int hp = idx >> 3, q = idx & 7;
            const int gy = Y0 - 3 + hp / kTlR2, gx = X0 - 3 + hp % kTlR2;
            st[s][it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (hp < kTlN2 && gy >= 0 && gy < H && gx >= 0 && gx < W)
                st[s][it] = *reinterpret_cast<const float4*>(zb + ((int64_t)gy * W + gx) * 64 + 32 * s + 4 * q);
        }
        (s ? wr1 : wr0) = *reinterpret_cast<const float4*>(pw3 + (tid >> 3) * 64 + 32 * s + 4 * (tid & 7));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            dr[s][i] = idx < 288 ? dw3[32 * s * 9 + idx] : 0.0f;
        }
    };
    request_step(std::integral_constant<int, 0>{});
    request_step(std::integral_constant<int, 1>{});
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        dr4[i] = idx < 288 ? dw4[idx] : 0.0f;
    }
    // pointwise 32 -> 16 weights: 16 rows x 8 chunks, chunk q of row f at q ^ ((f & 15) >> 1)
    if (tid < 128) {
        const int f = tid >> 3, q = tid & 7;
        *reinterpret_cast<float4*>(w4_s + f * 32 + ((q ^ ((f & 15) >> 1)) << 2)) =
            *reinterpret_cast<const float4*>(pw4 + f * 32 + 4 * q);
    }
    // the 3 x 3 x 16 x 2 taps of the final convolution (head + 304 ...): 288 floats
    if (tid < 72) *reinterpret_cast<float4*>(wf_s + 4 * tid) = *reinterpret_cast<const float4*>(head + 304 + 4 * tid);
    // flow-head parameters (as flow_head_kernel): w1[16][16] | b1 | bn_scale | bn_shift | wf[3][3][16][2]
    const f32x4v w1v = *reinterpret_cast<const f32x4v*>(head + n * 16 + 4 * g);
    const float4 b1v = *reinterpret_cast<const float4*>(head + 256 + 4 * g);
    const float4 bsv = *reinterpret_cast<const float4*>(head + 272 + 4 * g);
    const float4 btv = *reinterpret_cast<const float4*>(head + 288 + 4 * g);
    const float4 b4v = *reinterpret_cast<const float4*>(b4 + 4 * g);
    float4 b3v[2];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) b3v[ft] = *reinterpret_cast<const float4*>(b3 + 16 * ft + 4 * g);

    // ================= phase A: z3 = Mish(pointwise3(depthwise3(z2)) + b3) on the 12 x 12 region ==============
    f32x4v acc3[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) acc3[i][ft] = f32x4v{0.f, 0.f, 0.f, 0.f};
    // one 32-channel step of phase A; the step index is a compile-time constant so that st / wr / dr stay in registers
    auto phase_a_step = [&](auto step_c) __attribute__((always_inline)) {
        constexpr int s = decltype(step_c)::value;
        if (s) __syncthreads();   // the previous step's operand reads are done
        // ---- stage the 14 x 14 x 32 input tile, the 32 x 32 weight slice and the 32 x 9 depthwise taps ----
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int idx = tid + 256 * it;
            const int hp = idx >> 3, q = idx & 7;
            if (hp < kTlN2) {
                float4 v = st[s][it];
                if (ACT_IN) v = make_float4(mishf(v.x), mishf(v.y), mishf(v.z), mishf(v.w));   // mish(0) == 0
                *reinterpret_cast<float4*>(in_s + hp * kTlPS + 4 * q) = v;
            }
        }
        {
            const int f = tid >> 3, q = tid & 7;
            *reinterpret_cast<float4*>(w3_s + f * 32 + ((q ^ ((f & 15) >> 1)) << 2)) = s ? wr1 : wr0;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            if (idx < 288) dw_s[(idx % 9) * 32 + idx / 9] = dr[s][i];
        }
        __syncthreads();
        // ---- depthwise 3x3 on the 12 x 12 region: thread = channel quad cq, pixels pr + 32 j ----
        {
            float4 wq[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) wq[k] = *reinterpret_cast<const float4*>(dw_s + k * 32 + 4 * cq);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int p = pr + 32 * j;
                if (p < kTlN3) {
                    const int r = p / kTlR3, c = p - r * kTlR3;
                    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float4 v = *reinterpret_cast<const float4*>(in_s + ((r + ky) * kTlR2 + c + kx) * kTlPS + 4 * cq);
                            const float4 wk = wq[ky * 3 + kx];
                            a.x = fmaf(wk.x, v.x, a.x); a.y = fmaf(wk.y, v.y, a.y);
                            a.z = fmaf(wk.z, v.z, a.z); a.w = fmaf(wk.w, v.w, a.w);
                        }
                    *reinterpret_cast<float4*>(y3_s + p * 32 + ((cq ^ ((p & 15) >> 1)) << 2)) = a;
                }
            }
        }
        __syncthreads();
        // ---- pointwise: wave w owns pixel groups w, w + 4, w + 8 (< 9), both output blocks ----
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int coff = ((4 * u + g) ^ sw) << 2;
            f32x4v wv[2];
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) wv[ft] = *reinterpret_cast<const f32x4v*>(w3_s + (16 * ft + n) * 32 + coff);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int grp = wave + 4 * i;
                if (grp < kTlG3) {   // wave-uniform
                    const f32x4v yv = *reinterpret_cast<const f32x4v*>(y3_s + (16 * grp + n) * 32 + coff);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int ft = 0; ft < 2; ++ft)
                            acc3[i][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ft][t], yv[t], acc3[i][ft], 0, 0, 0);
                }
            }
        }
    };
    phase_a_step(std::integral_constant<int, 0>{});
    phase_a_step(std::integral_constant<int, 1>{});
    // ---- z3 = Mish(acc + b3), zero outside the image, to LDS (pixel stride 40) ----
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int grp = wave + 4 * i;
        if (grp < kTlG3) {
            const int p = 16 * grp + n;
            const int gy = Y0 - 2 + p / kTlR3, gx = X0 - 2 + p % kTlR3;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                const float4 bv = b3v[ft];
                float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                if (in) z = make_float4(mishf(acc3[i][ft][0] + bv.x), mishf(acc3[i][ft][1] + bv.y),
                                        mishf(acc3[i][ft][2] + bv.z), mishf(acc3[i][ft][3] + bv.w));
                *reinterpret_cast<float4*>(z3_s + p * kTlPS + 16 * ft + 4 * g) = z;
            }
        }
    }
    // depthwise taps of layer 4 (loaded at the top of the kernel)
    {
        __syncthreads();   // z3 complete; y3 / w3 / dw_s / in_s free
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            if (idx < 288) dw_s[(idx % 9) * 32 + idx / 9] = dr4[i];
        }
    }
    __syncthreads();
    // ================= phase B: z4 = pointwise4(depthwise4(z3)) + b4 on the 10 x 10 region ==================
    float* y4_s = in_s;   // [112][32]
    {
        float4 wq[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wq[k] = *reinterpret_cast<const float4*>(dw_s + k * 32 + 4 * cq);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = pr + 32 * j;
            if (p < 16 * kTlG4) {
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p < kTlN4) {
                    const int r = p / kTlR4, c = p - r * kTlR4;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float4 v = *reinterpret_cast<const float4*>(z3_s + ((r + ky) * kTlR3 + c + kx) * kTlPS + 4 * cq);
                            const float4 wk = wq[ky * 3 + kx];
                            a.x = fmaf(wk.x, v.x, a.x); a.y = fmaf(wk.y, v.y, a.y);
                            a.z = fmaf(wk.z, v.z, a.z); a.w = fmaf(wk.w, v.w, a.w);
                        }
                }
                *reinterpret_cast<float4*>(y4_s + p * 32 + ((cq ^ ((p & 15) >> 1)) << 2)) = a;
            }
        }
    }
    __syncthreads();
    float* h_s = y3_s;    // [112][16]
    {
        // (flow-head parameters w1v / b1v / bsv / btv / b4v: loaded at the top of the kernel)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int grp = wave + 4 * i;
            if (grp < kTlG4) {   // wave-uniform
                f32x4v z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int coff = ((4 * u + g) ^ sw) << 2;
                    const f32x4v wv = *reinterpret_cast<const f32x4v*>(w4_s + n * 32 + coff);
                    const f32x4v yv = *reinterpret_cast<const f32x4v*>(y4_s + (16 * grp + n) * 32 + coff);
#pragma unroll
                    for (int t = 0; t < 4; ++t) z4 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t], yv[t], z4, 0, 0, 0);
                }
                // ============ phase C: h = BN(Mish(W1 Mish(z4 + b4) + b1)), zero outside the image ============
                const float a0 = mishf(z4[0] + b4v.x), a1 = mishf(z4[1] + b4v.y), a2 = mishf(z4[2] + b4v.z),
                            a3 = mishf(z4[3] + b4v.w);
                f32x4v d = {0.f, 0.f, 0.f, 0.f};
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[0], a0, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[1], a1, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[2], a2, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[3], a3, d, 0, 0, 0);
                const int p = 16 * grp + n;
                const int gy = Y0 - 1 + p / kTlR4, gx = X0 - 1 + p % kTlR4;
                float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p < kTlN4 && gy >= 0 && gy < H && gx >= 0 && gx < W)
                    h = make_float4(fmaf(mishf(d[0] + b1v.x), bsv.x, btv.x), fmaf(mishf(d[1] + b1v.y), bsv.y, btv.y),
                                    fmaf(mishf(d[2] + b1v.z), bsv.z, btv.z), fmaf(mishf(d[3] + b1v.w), bsv.w, btv.w));
                *reinterpret_cast<float4*>(h_s + p * 16 + 4 * g) = h;
            }
        }
    }
    __syncthreads();
    // ================= phase D: flow = scale * conv3x3(h), 16 -> 2; wave w = tile rows 2 w, 2 w + 1 ==========
    {
        const float* wf = wf_s;   // staged at the top of the kernel
        const int p = 16 * wave + n, r = p >> 3, c = p & 7;
        float fx = 0.0f, fy = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 v = *reinterpret_cast<const float4*>(h_s + ((r + ky) * kTlR4 + c + kx) * 16 + 4 * g);
                const float4 wa = *reinterpret_cast<const float4*>(wf + (ky * 3 + kx) * 32 + 8 * g);
                const float4 wb = *reinterpret_cast<const float4*>(wf + (ky * 3 + kx) * 32 + 8 * g + 4);
                fx = fmaf(v.x, wa.x, fx); fy = fmaf(v.x, wa.y, fy);
                fx = fmaf(v.y, wa.z, fx); fy = fmaf(v.y, wa.w, fy);
                fx = fmaf(v.z, wb.x, fx); fy = fmaf(v.z, wb.y, fy);
                fx = fmaf(v.w, wb.z, fx); fy = fmaf(v.w, wb.w, fy);
            }
        fx += __shfl_xor(fx, 16); fy += __shfl_xor(fy, 16);
        fx += __shfl_xor(fx, 32); fy += __shfl_xor(fy, 32);
        const int gy = Y0 + r, gx = X0 + c;
        if (g == 0 && gy < H && gx < W) {
            if (out_nchw) {
                float* o = out + ((int64_t)(b * 2) * H + gy) * W + gx;
                o[0] = scale * fx;
                o[(int64_t)H * W] = scale * fy;
            } else {
                float* o = out + ((int64_t)(b * H + gy) * W + gx) * 2;
                o[0] = scale * fx;
                o[1] = scale * fy;
            }
        }
    }
}

int optflow_tail_launch(const void* z2, const void* dw3, const void* pw3, const void* b3, const void* dw4,
                        const void* pw4, const void* b4, const void* head, void* out, int B, int H, int W,
                        float scale, int act_in, int out_layout, hipStream_t s) {
    const int tiles_x = (W + kTlT - 1) / kTlT, tiles_y = (H + kTlT - 1) / kTlT;
    const int64_t nblk = (int64_t)tiles_x * tiles_y * B;
    if (nblk > INT32_MAX) {
        set_error("optflow_tail: too many tiles");
        return QPWC_E_SHAPE;
    }
    const dim3 grid((unsigned)nblk);
    const int nchw = out_layout == QPWC_NCHW;
#define QPWC_TL(A)                                                                                             \
    hipLaunchKernelGGL(optflow_tail_kernel<A>, grid, dim3(256), 0, s, (const float*)z2, (const float*)dw3,     \
                       (const float*)pw3, (const float*)b3, (const float*)dw4, (const float*)pw4,              \
                       (const float*)b4, (const float*)head, (float*)out, H, W, tiles_x, tiles_y, scale, nchw)
    if (act_in) QPWC_TL(true); else QPWC_TL(false);
#undef QPWC_TL
    return check_launch("optflow_tail_kernel");
}

int flow_head_param_floats() { return kFhParams; }

// ---------------------------------------------------------------------------
// flow head + the Upsample(x2) that always follows it (pwcnet.py:55,60), one launch (round 4): the 16 x 16 flow tile is
// computed WITH the one-pixel rim the bilinear upsampling reads (18 x 18 flow pixels from a 20 x 20 tile of h: 1.23 x the
// 1x1 stage, 1.27 x the 3x3 stage of a kernel that is bound by its launch, not by its arithmetic), kept in LDS after the
// rounding of the store, and the tile's 32 x 32 upsampled pixels are written from there with bilerp_flow() -- the same
// chain of fused multiply-adds on the same stored values as upsample2x_flow_pair_kernel: bit-identical to the two launches.
// A rim pixel inside the image is computed by this tile and by its neighbour from the same operands in the same order;
// one outside the image is never read (edge clamp).  Channels-last only.
constexpr int kFuF = kFhTile + 2;      // 18: flow region
constexpr int kFuH = kFhTile + 4;      // 20: h region
template <typename T>
__global__ __launch_bounds__(256, 2) void flow_head_up_kernel(const T* __restrict__ z, const float* __restrict__ params,
                                                              T* __restrict__ out, T* __restrict__ out_up,
                                                              float* __restrict__ out_up_f32, int H, int W,
                                                              int tiles_x, int tiles_y, float scale, float up_scale) {
    QPWC_FLOW_CHAIN_PRIO();
    constexpr int NH = kFuH * kFuH;                                    // 400 = 25 groups of 16 pixels
    constexpr int NF = kFuF * kFuF;                                    // 324 flow pixels
    __shared__ __attribute__((aligned(16))) float hs[NH * kFhC];
    __shared__ __attribute__((aligned(8))) float2 fl[NF];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int x0 = tx * kFhTile, y0 = ty * kFhTile;
    const float* w1 = params;
    const float* b1 = params + 256;
    const float* bs = params + 272;
    const float* bt = params + 288;
    const float* wf = params + 304;
    const T* zb = z + (int64_t)b * H * W * kFhC;

    // ---- h on the 20 x 20 region (flow_head_kernel's arithmetic) ----
    const f32x4v w1v = *reinterpret_cast<const f32x4v*>(w1 + n * kFhC + 4 * g);
    const float4 b1v = *reinterpret_cast<const float4*>(b1 + 4 * g);
    const float4 bsv = *reinterpret_cast<const float4*>(bs + 4 * g);
    const float4 btv = *reinterpret_cast<const float4*>(bt + 4 * g);
    float4 wq[9][2];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        wq[k][0] = *reinterpret_cast<const float4*>(wf + k * kFhC * 2 + 8 * g);
        wq[k][1] = *reinterpret_cast<const float4*>(wf + k * kFhC * 2 + 8 * g + 4);
    }
    constexpr int NG = NH / 16, NGW = (NG + 3) / 4;                    // 25 groups, at most 7 per wave
    float4 av[NGW];
    bool inb[NGW];
#pragma unroll
    for (int i = 0; i < NGW; ++i) {
        const int grp = wave + 4 * i;
        const int hp = 16 * grp + n;
        const int ly = hp / kFuH, lx = hp - ly * kFuH;
        const int gy = y0 - 2 + ly, gx = x0 - 2 + lx;
        inb[i] = grp < NG && gy >= 0 && gy < H && gx >= 0 && gx < W;
        av[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (inb[i]) av[i] = ld4(zb + ((int64_t)gy * W + gx) * kFhC + 4 * g);
    }
#pragma unroll
    for (int i = 0; i < NGW; ++i) {
        const int grp = wave + 4 * i;
        if (grp >= NG) break;   // wave-uniform
        const int hp = 16 * grp + n;
        float4 a = av[i];
        if (inb[i]) a = make_float4(mishf(a.x), mishf(a.y), mishf(a.z), mishf(a.w));
        f32x4v d = {0.f, 0.f, 0.f, 0.f};
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[0], a.x, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[1], a.y, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[2], a.z, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[3], a.w, d, 0, 0, 0);
        float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
        if (inb[i])
            h = make_float4(fmaf(mishf(d[0] + b1v.x), bsv.x, btv.x), fmaf(mishf(d[1] + b1v.y), bsv.y, btv.y),
                            fmaf(mishf(d[2] + b1v.z), bsv.z, btv.z), fmaf(mishf(d[3] + b1v.w), bsv.w, btv.w));
        *reinterpret_cast<float4*>(hs + hp * kFhC + 4 * g) = h;
    }
    __syncthreads();

    // ---- 3x3 conv 16 -> 2 on the 18 x 18 flow region: groups of 16 flow pixels, lane = (pixel n of the group, quad g) ----
    constexpr int NFG = (NF + 15) / 16;                                // 21 groups
    for (int fg = wave; fg < NFG; fg += 4) {
        const int fp_raw = 16 * fg + n;
        const int fp = fp_raw < NF ? fp_raw : NF - 1;                  // (the last group's spare lanes repeat a pixel)
        const int fyl = fp / kFuF, fxl = fp - fyl * kFuF;
        float fx = 0.0f, fy = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 v = *reinterpret_cast<const float4*>(hs + ((fyl + ky) * kFuH + fxl + kx) * kFhC + 4 * g);
                const float4 wa = wq[ky * 3 + kx][0], wb = wq[ky * 3 + kx][1];
                fx = fmaf(v.x, wa.x, fx); fy = fmaf(v.x, wa.y, fy);
                fx = fmaf(v.y, wa.z, fx); fy = fmaf(v.y, wa.w, fy);
                fx = fmaf(v.z, wb.x, fx); fy = fmaf(v.z, wb.y, fy);
                fx = fmaf(v.w, wb.z, fx); fy = fmaf(v.w, wb.w, fy);
            }
        fx += __shfl_xor(fx, 16); fy += __shfl_xor(fy, 16);
        fx += __shfl_xor(fx, 32); fy += __shfl_xor(fy, 32);
        if (g == 0 && fp_raw < NF) {
            // the value as the level's flow tensor holds it (one rounding for fp16 storage): what the upsampling reads
            T sx, sy;
            st(&sx, scale * fx);
            st(&sy, scale * fy);
            fl[fp] = make_float2(ld(&sx), ld(&sy));
            const int gx = x0 - 1 + fxl, gy = y0 - 1 + fyl;
            if (fxl >= 1 && fxl <= kFhTile && fyl >= 1 && fyl <= kFhTile && gx < W && gy < H) {
                T* o = out + ((int64_t)(b * H + gy) * W + gx) * 2;
                o[0] = sx;
                o[1] = sy;
            }
        }
    }
    __syncthreads();

    // ---- the tile's 32 x 32 upsampled pixels, two horizontally adjacent ones per thread and trip ----
    const int H2 = 2 * H, W2 = 2 * W;
#pragma unroll
    for (int r = 0; r < (2 * kFhTile * kFhTile) / 256; ++r) {
        const int pi = tid + 256 * r;
        const int yl = pi / kFhTile, j = pi - yl * kFhTile;
        const int Y = 2 * y0 + yl, X = 2 * x0 + 2 * j;
        if (Y >= H2 || X >= W2) continue;
        const float sy = fmaxf(0.0f, (Y + 0.5f) * 0.5f - 0.5f);
        const int yy0 = (int)sy;
        const int yy1 = yy0 + 1 < H ? yy0 + 1 : H - 1;
        const float ly = sy - yy0;
        const int r0 = (yy0 - (y0 - 1)) * kFuF, r1 = (yy1 - (y0 - 1)) * kFuF;
        float2 res[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int x = X + e;
            const float sx = fmaxf(0.0f, (x + 0.5f) * 0.5f - 0.5f);
            const int xx0 = (int)sx;
            const int xx1 = xx0 + 1 < W ? xx0 + 1 : W - 1;
            const float lx = sx - xx0;
            const int c0 = xx0 - (x0 - 1), c1 = xx1 - (x0 - 1);
            res[e] = bilerp_flow(ly, lx, fl[r0 + c0], fl[r0 + c1], fl[r1 + c0], fl[r1 + c1], up_scale);
        }
        T* o = out_up + ((int64_t)(b * H2 + Y) * W2 + X) * 2;
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float4*>(o) = make_float4(res[0].x, res[0].y, res[1].x, res[1].y);
        } else {
            const __half2 a = __floats2half2_rn(res[0].x, res[0].y), c = __floats2half2_rn(res[1].x, res[1].y);
            uint2 u;
            u.x = *reinterpret_cast<const unsigned*>(&a);
            u.y = *reinterpret_cast<const unsigned*>(&c);
            *reinterpret_cast<uint2*>(o) = u;
            // the SAME values as fp32 (what `out_up.float()` holds): the next level's warp takes fp32 coordinates whatever
            // the storage type, and that cast was a launch of its own per level (4.7 us each, config 5)
            if (out_up_f32 != nullptr) {
                const float2 fa = __half22float2(a), fc = __half22float2(c);
                *reinterpret_cast<float4*>(out_up_f32 + ((int64_t)(b * H2 + Y) * W2 + X) * 2) = make_float4(fa.x, fa.y, fc.x, fc.y);
            }
        }
    }
}

int flow_head_up_launch(const void* z, const void* params, void* out, void* out_up, void* out_up_f32, int B, int H, int W,
                        float scale, float up_scale, int dtype, hipStream_t s) {
    const int tiles_x = (W + kFhTile - 1) / kFhTile, tiles_y = (H + kFhTile - 1) / kFhTile;
    const dim3 grid((unsigned)(tiles_x * tiles_y * B));
    if (dtype == QPWC_F32)
        hipLaunchKernelGGL(flow_head_up_kernel<float>, grid, dim3(256), 0, s, (const float*)z, (const float*)params,
                           (float*)out, (float*)out_up, (float*)nullptr, H, W, tiles_x, tiles_y, scale, up_scale);
    else
        hipLaunchKernelGGL(flow_head_up_kernel<__half>, grid, dim3(256), 0, s, (const __half*)z, (const float*)params,
                           (__half*)out, (__half*)out_up, (float*)out_up_f32, H, W, tiles_x, tiles_y, scale, up_scale);
    return check_launch("flow_head_up_kernel");
}

int flow_head_launch(const void* z, const void* params, void* out, int B, int H, int W, float scale,
                     int dtype, int out_layout, hipStream_t s) {
    const int out_nchw = out_layout == QPWC_NCHW;
    const int tiles_x = (W + kFhTile - 1) / kFhTile, tiles_y = (H + kFhTile - 1) / kFhTile;
    const dim3 grid((unsigned)(tiles_x * tiles_y * B));
    if (dtype == QPWC_F32)
        hipLaunchKernelGGL(flow_head_kernel<float>, grid, dim3(256), 0, s, (const float*)z,
                           (const float*)params, (float*)out, H, W, tiles_x, tiles_y, scale, out_nchw);
    else
        hipLaunchKernelGGL(flow_head_kernel<__half>, grid, dim3(256), 0, s, (const __half*)z,
                           (const float*)params, (__half*)out, H, W, tiles_x, tiles_y, scale, out_nchw);
    return check_launch("flow_head_kernel");
}

}  // namespace qpwc

#ifdef QPWC_SC_STAMP
extern "C" int qpwc_debug_sc_census(long long* out, int n) {
    if (n > 4096 * 6) n = 4096 * 6;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(qpwc::g_sc_census), n * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
extern "C" int qpwc_debug_sc_stamps(long long* out, int n) {
    if (n > 4 * 64) n = 4 * 64;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(qpwc::g_sc_stamps), n * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
#endif
