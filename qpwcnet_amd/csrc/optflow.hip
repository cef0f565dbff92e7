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
#include "common.h"

namespace qpwc {

// mish(x) = x * tanh(softplus(x)) = x * t / (t + 2),  t = e^x (e^x + 2); x > 20 -> x
// (torch's softplus threshold).  ~2 ulp with the fast exp/div.
__device__ __forceinline__ float mishf(float x) {
    const float e = __expf(fminf(x, 20.0f));
    const float t = e * (e + 2.0f);
    const float m = x * __fdividef(t, t + 2.0f);
    return x > 20.0f ? x : m;
}

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct DwSrc {
    const void* ptr[3];
    int ch[3];          // channels taken from each source (0 = unused)
    int64_t stride[3];  // floats per pixel of each source
};

constexpr int kDwPx = 4;    // consecutive pixels per thread

// thread = (4 consecutive pixels, 1 channel): per input row it loads 6 values, applies
// Mish once to each (1.5 evaluations per output instead of 3 for one pixel per thread),
// and keeps a rolling 3-row window; the raw values of the next row are requested one
// iteration ahead so that the loop does not wait on a load it has just issued.
template <typename T, bool ACT>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(DwSrc src, const float* __restrict__ weight,
                                                        T* __restrict__ out, int H, int W, int C,
                                                        int strips, int wq, int rows) {
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
    if (c < src.ch[0]) {
        p = (const T*)src.ptr[0]; ps = src.stride[0]; cc = c;
    } else if (c < src.ch[0] + src.ch[1]) {
        p = (const T*)src.ptr[1]; ps = src.stride[1]; cc = c - src.ch[0];
    } else {
        p = (const T*)src.ptr[2]; ps = src.stride[2]; cc = c - src.ch[0] - src.ch[1];
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
__global__ __launch_bounds__(256) void dwconv3x3_vec_kernel(const T* __restrict__ in, int64_t ps,
                                                            const float* __restrict__ weight,
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
    const T* p = in + (int64_t)b * H * W * ps + c;

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
        hipLaunchKernelGGL((dwconv3x3_vec_kernel<T, VEC, true>), grid, dim3(256), 0, s, (const T*)d.ptr[0],
                           d.stride[0], (const float*)weight, (T*)out, H, W, C, strips, wq, rows);
    else
        hipLaunchKernelGGL((dwconv3x3_vec_kernel<T, VEC, false>), grid, dim3(256), 0, s, (const T*)d.ptr[0],
                           d.stride[0], (const float*)weight, (T*)out, H, W, C, strips, wq, rows);
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
    const bool vec_ok = dtype == QPWC_F16 && n_src == 1 && d.stride[0] % 4 == 0 && C % 4 == 0 &&
                        reinterpret_cast<uintptr_t>(d.ptr[0]) % 8 == 0 &&
                        reinterpret_cast<uintptr_t>(out) % 8 == 0;
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
// pre-activation output) in one launch, fp32.  The depthwise result never goes to HBM:
// per 16-channel chunk a workgroup (4 waves, 8x8 pixel tile) stages the 10x10 halo tile
// of the (virtually concatenated, optionally Mish-activated) input in LDS, computes the
// depthwise outputs into an LDS operand tile, and feeds them with the matching slice of
// the pointwise weights to v_mfma_f32_16x16x4_f32 (rows = output channels, cols = the
// wave's 16 pixels).  k-slot g of a lane owns channels 4g..4g+3 of the chunk, so every
// operand is one ds_read_b128.
// pointwise weights: (F, Cpad) row-major with Cpad = ceil(C/16)*16, zero padded.
constexpr int kScKC = 16;               // channels per chunk
constexpr int kScTile = 8;              // 8x8 pixels per workgroup
constexpr int kScHalo = kScTile + 2;
constexpr int kScLd = 20;               // padded row stride (floats) of the operand tiles

template <int F, bool ACT>
__global__ __launch_bounds__(256) void sepconv3x3_fused_kernel(
    DwSrc src, const float* __restrict__ dw, const float* __restrict__ pw, const float* __restrict__ bias,
    float* __restrict__ out, int H, int W, int C, int cpad, int tiles_x, int tiles_y) {
    constexpr int NFT = F / 16;
    __shared__ __attribute__((aligned(16))) float in_s[kScHalo * kScHalo * kScKC];
    __shared__ __attribute__((aligned(16))) float y_s[64 * kScLd];
    __shared__ __attribute__((aligned(16))) float w_s[F * kScLd];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int X0 = tx * kScTile, Y0 = ty * kScTile;
    const int n = lane & 15, g = lane >> 4;

    f32x4v acc[NFT];
#pragma unroll
    for (int i = 0; i < NFT; ++i) acc[i] = f32x4v{0.f, 0.f, 0.f, 0.f};

    const int dc = tid & 15;       // channel of this thread inside a chunk (staging + depthwise)
    const int dp = tid >> 4;       // 0..15
    for (int c0 = 0; c0 < cpad; c0 += kScKC) {
        // ---- stage the halo tile of this chunk (Mish applied once per element) ---------
        const int c = c0 + dc;
        const float* p = nullptr;
        int64_t ps = 0;
        if (c < C) {
            int cc;
            if (c < src.ch[0]) {
                p = (const float*)src.ptr[0]; ps = src.stride[0]; cc = c;
            } else if (c < src.ch[0] + src.ch[1]) {
                p = (const float*)src.ptr[1]; ps = src.stride[1]; cc = c - src.ch[0];
            } else {
                p = (const float*)src.ptr[2]; ps = src.stride[2]; cc = c - src.ch[0] - src.ch[1];
            }
            p += (int64_t)b * H * W * ps + cc;
        }
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int hp = dp + 16 * it;  // halo pixel 0..99
            if (hp < kScHalo * kScHalo) {
                const int hy = hp / kScHalo, hx = hp - hy * kScHalo;
                const int gy = Y0 - 1 + hy, gx = X0 - 1 + hx;
                float v = 0.0f;
                if (p && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                    v = p[((int64_t)gy * W + gx) * ps];
                    if (ACT) v = mishf(v);
                }
                in_s[hp * kScKC + dc] = v;
            }
        }
        // pointwise weights of this chunk: w_s[f][0..15] = pw[f][c0..c0+15]
        for (int i = tid; i < F * 4; i += 256) {
            const int f = i >> 2, q = i & 3;
            *reinterpret_cast<float4*>(w_s + f * kScLd + 4 * q) =
                *reinterpret_cast<const float4*>(pw + (int64_t)f * cpad + c0 + 4 * q);
        }
        float wk[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k] = c < C ? dw[c * 9 + k] : 0.0f;
        __syncthreads();
        // ---- depthwise: 4 pixels per thread for channel dc ----------------------------
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pix = dp + 16 * j;  // 0..63
            const int py = pix >> 3, px = pix & 7;
            float a = 0.0f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    a = fmaf(wk[ky * 3 + kx], in_s[((py + ky) * kScHalo + px + kx) * kScKC + dc], a);
            y_s[pix * kScLd + dc] = a;
        }
        __syncthreads();
        // ---- pointwise on the matrix cores: D[f][px] += W[f][k] * y[px][k] --------------
        const f32x4v yv = *reinterpret_cast<const f32x4v*>(y_s + (16 * wave + n) * kScLd + 4 * g);
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) {
            const f32x4v wv = *reinterpret_cast<const f32x4v*>(w_s + (16 * ft + n) * kScLd + 4 * g);
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t], yv[t], acc[ft], 0, 0, 0);
        }
        __syncthreads();  // operand tiles are rewritten by the next chunk
    }
    // ---- bias + store: lane = pixel n of the wave's 16, 4 consecutive outputs 4g..4g+3 ----
    const int pix = 16 * wave + n;
    const int gy = Y0 + (pix >> 3), gx = X0 + (pix & 7);
    if (gy < H && gx < W) {
        float* o = out + ((int64_t)(b * H + gy) * W + gx) * F;
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + 16 * ft + 4 * g);
            *reinterpret_cast<float4*>(o + 16 * ft + 4 * g) =
                make_float4(acc[ft][0] + bv.x, acc[ft][1] + bv.y, acc[ft][2] + bv.z, acc[ft][3] + bv.w);
        }
    }
}

template <int F>
static void sepconv_dispatch(const DwSrc& d, int act, const float* dw, const float* pw, const float* bias,
                             float* out, int H, int W, int C, int cpad, int tiles_x, int tiles_y,
                             dim3 grid, hipStream_t s) {
    if (act)
        hipLaunchKernelGGL((sepconv3x3_fused_kernel<F, true>), grid, dim3(256), 0, s, d, dw, pw, bias, out,
                           H, W, C, cpad, tiles_x, tiles_y);
    else
        hipLaunchKernelGGL((sepconv3x3_fused_kernel<F, false>), grid, dim3(256), 0, s, d, dw, pw, bias, out,
                           H, W, C, cpad, tiles_x, tiles_y);
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
    const int tiles_x = (W + kScTile - 1) / kScTile, tiles_y = (H + kScTile - 1) / kScTile;
    const dim3 grid((unsigned)(tiles_x * tiles_y * B));
    const float *fdw = (const float*)dw, *fpw = (const float*)pw, *fb = (const float*)bias;
    switch (F) {
        case 128: sepconv_dispatch<128>(d, act, fdw, fpw, fb, (float*)out, H, W, C, cpad, tiles_x, tiles_y, grid, s); break;
        case 64: sepconv_dispatch<64>(d, act, fdw, fpw, fb, (float*)out, H, W, C, cpad, tiles_x, tiles_y, grid, s); break;
        case 32: sepconv_dispatch<32>(d, act, fdw, fpw, fb, (float*)out, H, W, C, cpad, tiles_x, tiles_y, grid, s); break;
        case 16: sepconv_dispatch<16>(d, act, fdw, fpw, fb, (float*)out, H, W, C, cpad, tiles_x, tiles_y, grid, s); break;
        default: set_error("sepconv3x3: unsupported filter count %d (16/32/64/128)", F); return QPWC_E_SHAPE;
    }
    return check_launch("sepconv3x3_fused_kernel");
}

// ---------------------------------------------------------------------------
// flow head.  params (device, fp32): w1[16][16] (out,in) | b1[16] | bn_scale[16] |
// bn_shift[16] | wf[3][3][16][2] (ky,kx,in,out)   = 256+16+16+16+288 = 592 floats
constexpr int kFhC = 16;
constexpr int kFhTile = 16;
constexpr int kFhParams = 592;  // == qpwc_flow_head_param_floats()

template <typename T>
__global__ __launch_bounds__(256, 4) void flow_head_kernel(const T* __restrict__ z,
                                                           const float* __restrict__ params,
                                                           T* __restrict__ out, int H, int W,
                                                           int tiles_x, int tiles_y, float scale) {
    constexpr int TW = kFhTile + 2;
    __shared__ __attribute__((aligned(16))) float hs[TW * TW * kFhC];  // 18*18*16*4 = 20.7 KB
    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int x0 = tx * kFhTile, y0 = ty * kFhTile;
    // parameters are read with compile-time offsets from the (uniform) kernel argument:
    // scalar loads, the weights sit in SGPRs as FMA operands
    const float* w1 = params;
    const float* b1 = params + 256;
    const float* bs = params + 272;
    const float* bt = params + 288;
    const float* wf = params + 304;
    const T* zb = z + (int64_t)b * H * W * kFhC;

    // stage h3 = BN(mish(W1 mish(z) + b1)) for the tile + 1 halo; zero outside the image
    for (int p = tid; p < TW * TW; p += 256) {
        const int ly = p / TW, lx = p - ly * TW;
        const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
        float h[kFhC];
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const T* q = zb + ((int64_t)gy * W + gx) * kFhC;
            float a[kFhC];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 v = ld4(q + 4 * i);
                a[4 * i] = mishf(v.x); a[4 * i + 1] = mishf(v.y);
                a[4 * i + 2] = mishf(v.z); a[4 * i + 3] = mishf(v.w);
            }
#pragma unroll
            for (int o = 0; o < kFhC; ++o) {
                float s = b1[o];
#pragma unroll
                for (int i = 0; i < kFhC; ++i) s = fmaf(w1[o * kFhC + i], a[i], s);
                h[o] = fmaf(mishf(s), bs[o], bt[o]);
            }
        } else {
#pragma unroll
            for (int o = 0; o < kFhC; ++o) h[o] = 0.0f;
        }
        float4* d = reinterpret_cast<float4*>(hs + p * kFhC);
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = make_float4(h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3]);
    }
    __syncthreads();

    const int lx = tid % kFhTile, ly = tid / kFhTile;
    const int gx = x0 + lx, gy = y0 + ly;
    if (gx >= W || gy >= H) return;
    float fx = 0.0f, fy = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float4* q = reinterpret_cast<const float4*>(hs + ((ly + ky) * TW + lx + kx) * kFhC);
            const float* wk = wf + (ky * 3 + kx) * kFhC * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 v = q[i];
                fx = fmaf(v.x, wk[(4 * i) * 2], fx);     fy = fmaf(v.x, wk[(4 * i) * 2 + 1], fy);
                fx = fmaf(v.y, wk[(4 * i + 1) * 2], fx); fy = fmaf(v.y, wk[(4 * i + 1) * 2 + 1], fy);
                fx = fmaf(v.z, wk[(4 * i + 2) * 2], fx); fy = fmaf(v.z, wk[(4 * i + 2) * 2 + 1], fy);
                fx = fmaf(v.w, wk[(4 * i + 3) * 2], fx); fy = fmaf(v.w, wk[(4 * i + 3) * 2 + 1], fy);
            }
        }
    T* o = out + ((int64_t)(b * H + gy) * W + gx) * 2;
    st(o, scale * fx);
    st(o + 1, scale * fy);
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
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_flow_kernel(const T* __restrict__ in,
                                                              T* __restrict__ out, int B, int h,
                                                              int w, float scale) {
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
            const T* q = p + ((int64_t)yy * w + xx) * 2;
            return make_float2(ld(q), ld(q + 1));
        };
        const float2 v00 = px(y0, x0), v01 = px(y0, x1), v10 = px(y1, x0), v11 = px(y1, x1);
        const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
        float2 r;
        r.x = scale * (w00 * v00.x + w01 * v01.x + w10 * v10.x + w11 * v11.x);
        r.y = scale * (w00 * v00.y + w01 * v01.y + w10 * v10.y + w11 * v11.y);
        st(out + 2 * idx, r.x);
        st(out + 2 * idx + 1, r.y);
    }
}

int upsample2x_flow_launch(const void* in, void* out, int B, int h, int w, float scale, int dtype,
                           hipStream_t s) {
    const int64_t total = (int64_t)B * 4 * h * w;
    const int64_t want = (total + 255) / 256;
    const dim3 grid((unsigned)(want < 8192 ? want : 8192));
    if (dtype == QPWC_F32)
        hipLaunchKernelGGL(upsample2x_flow_kernel<float>, grid, dim3(256), 0, s, (const float*)in,
                           (float*)out, B, h, w, scale);
    else
        hipLaunchKernelGGL(upsample2x_flow_kernel<__half>, grid, dim3(256), 0, s, (const __half*)in,
                           (__half*)out, B, h, w, scale);
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

int flow_head_param_floats() { return kFhParams; }

int flow_head_launch(const void* z, const void* params, void* out, int B, int H, int W, float scale,
                     int dtype, hipStream_t s) {
    const int tiles_x = (W + kFhTile - 1) / kFhTile, tiles_y = (H + kFhTile - 1) / kFhTile;
    const dim3 grid((unsigned)(tiles_x * tiles_y * B));
    if (dtype == QPWC_F32)
        hipLaunchKernelGGL(flow_head_kernel<float>, grid, dim3(256), 0, s, (const float*)z,
                           (const float*)params, (float*)out, H, W, tiles_x, tiles_y, scale);
    else
        hipLaunchKernelGGL(flow_head_kernel<__half>, grid, dim3(256), 0, s, (const __half*)z,
                           (const float*)params, (__half*)out, H, W, tiles_x, tiles_y, scale);
    return check_launch("flow_head_kernel");
}

}  // namespace qpwc
