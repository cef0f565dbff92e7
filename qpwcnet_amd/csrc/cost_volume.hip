// CostVolume / CostVolumeV2 forward for gfx950 (MI355X, CDNA4, wave64).
//
// Reference semantics: qpwcnet/core/layers.py:72-100 (CostVolume.call) ==
// qpwcnet/core/layers.py:128-132 (CostVolumeV2.call) by the reference's own
// invariant (qpwcnet/app/test/test_cvol_equal.py:25):
//   out[b,y,x,i*d+j] = lrelu( mean_c prv[b,y,x,c] * nxt0[b,y+i-r,x+j-r,c] )
//
// Fast path (NHWC, C % 4 == 0, r == 4): one workgroup owns a TH x TW pixel tile.
//   * per channel chunk of CC floats, the nxt tile with its r-pixel halo and the
//     prv tile are staged into LDS with 16-byte coalesced loads (zero fill = the
//     reference's ZeroPadding2D); optionally the nxt tile is produced by the
//     WarpV2 bilinear gather on the fly (UpFlow front end, non_layers.py:377-380);
//   * a thread owns 4 consecutive pixels of one row and ONE displacement row
//     (9 column displacements): 36 fp32 accumulators, and per 4 channels it reads
//     4 prv + 12 nxt float4 from LDS for 144 FMAs (each nxt value feeds up to 4
//     pixels, each prv value 9 displacements);
//   * LDS image: pixel = CC/4+1 16-byte slots (odd), row stride == 1 (mod 4) slots,
//     which makes the ds_read_b128 of a 4x4 (quad,row) lane group conflict free;
//   * the 81 results of a pixel are produced by 9 different waves, so they are
//     transposed through LDS (aliasing the input tiles) and leave as fully
//     coalesced row stores of TW*81 contiguous floats.
// HBM-bound by design: algorithmic bytes B*H*W*(2C+81)*4, see DESIGN.md.
#include <stdlib.h>

#include "common.h"

namespace qpwc {

template <int R, int TH, int TW, int CC>
struct CvCfg {
    static constexpr int D = 2 * R + 1;
    static constexpr int DD = D * D;
    static constexpr int QW = TW / 4;        // pixel quads per tile row
    static constexpr int NQ = TH * QW;       // quads per tile
    static constexpr int NT = D * NQ;        // threads: one per (quad, displacement row)
    static constexpr int NCH = CC / 4;       // float4 chunks per pixel per stage
    static constexpr int P = NCH + 1;        // slots per pixel (odd -> conflict-free quads)
    static constexpr int NH = TH + 2 * R;
    static constexpr int NW = TW + 2 * R;
    static constexpr int pad1(int v) { return v + ((1 - (v % 4)) + 4) % 4; }  // >= v, == 1 (mod 4)
    static constexpr int RSN = pad1(NW * P);
    static constexpr int RSP = pad1(TW * P);
    static constexpr int NXT_SLOTS = NH * RSN;
    static constexpr int PRV_SLOTS = TH * RSP;
    static constexpr int IN_BYTES = (NXT_SLOTS + PRV_SLOTS) * 16;
    static constexpr int RP = (TH >= 8) ? TH / 2 : TH;  // tile rows per output pass
    static constexpr int NPASS = TH / RP;
    static constexpr int OUT_BYTES = RP * TW * DD * 4;
    static constexpr int LDS_BYTES = IN_BYTES > OUT_BYTES ? IN_BYTES : OUT_BYTES;
    static_assert(TW % 4 == 0 && CC % 8 == 0 && TH % RP == 0, "tile shape");
    static_assert(NT <= 1024, "workgroup too large");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <typename T, int R, int TH, int TW, int CC, bool FUSE_WARP>
__global__ __launch_bounds__((CvCfg<R, TH, TW, CC>::NT)) void cost_volume_tiled_kernel(
    const T* __restrict__ prv, const T* __restrict__ nxt, const float* __restrict__ flo,
    T* __restrict__ out, int H, int W, int C, int tiles_x, int tiles_y, int64_t out_pix_stride,
    float slope) {
    using Cfg = CvCfg<R, TH, TW, CC>;
    constexpr int D = Cfg::D, DD = Cfg::DD, NT = Cfg::NT, NCH = Cfg::NCH, P = Cfg::P;
    constexpr int NH = Cfg::NH, NW = Cfg::NW, RSN = Cfg::RSN, RSP = Cfg::RSP;

    // static LDS (up to 160 KiB per workgroup on gfx950): no launch-time attribute needed
    __shared__ __attribute__((aligned(16))) char smem[Cfg::LDS_BYTES];
    float4* nxt_s = reinterpret_cast<float4*>(smem);
    float4* prv_s = nxt_s + Cfg::NXT_SLOTS;

    const int tid = threadIdx.x;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x0 = tx * TW, y0 = ty * TH;

    const int q = tid % Cfg::NQ;
    const int dyi = tid / Cfg::NQ;  // displacement row 0..D-1  (dy = dyi - R)
    const int qx = q % Cfg::QW;
    const int qy = q / Cfg::QW;

    const T* prv_b = prv + (int64_t)b * H * W * C;
    const T* nxt_b = nxt + (int64_t)b * H * W * C;

    float acc[4][D];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) acc[i][j] = 0.0f;

    for (int c0 = 0; c0 < C; c0 += CC) {
        __syncthreads();  // everyone is done reading the previous chunk
        // ---- stage nxt (+halo), zero outside the image --------------------
        for (int it = tid; it < NH * NW * NCH; it += NT) {
            const int ch = it % NCH;
            const int pix = it / NCH;
            const int col = pix % NW;
            const int row = pix / NW;
            const int gy = y0 - R + row, gx = x0 - R + col;
            const int c = c0 + 4 * ch;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W && c < C) {
                if (FUSE_WARP) {
                    const float* f = flo + ((int64_t)(b * H + gy) * W + gx) * 2;
                    const Taps t = taps_clamp(gy, gx, f[0], f[1], H, W);
                    const float4 tl = ld4(nxt_b + ((int64_t)t.y0 * W + t.x0) * C + c);
                    const float4 tr = ld4(nxt_b + ((int64_t)t.y0 * W + t.x1) * C + c);
                    const float4 bl = ld4(nxt_b + ((int64_t)t.y1 * W + t.x0) * C + c);
                    const float4 br = ld4(nxt_b + ((int64_t)t.y1 * W + t.x1) * C + c);
                    v = blend4<QPWC_WARP_CLAMP>(t, tl, tr, bl, br);
                } else {
                    v = ld4(nxt_b + ((int64_t)gy * W + gx) * C + c);
                }
            }
            nxt_s[row * RSN + col * P + ch] = v;
        }
        // ---- stage prv -----------------------------------------------------
        for (int it = tid; it < TH * TW * NCH; it += NT) {
            const int ch = it % NCH;
            const int pix = it / NCH;
            const int col = pix % TW;
            const int row = pix / TW;
            const int gy = y0 + row, gx = x0 + col;
            const int c = c0 + 4 * ch;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy < H && gx < W && c < C) v = ld4(prv_b + ((int64_t)gy * W + gx) * C + c);
            prv_s[row * RSP + col * P + ch] = v;
        }
        __syncthreads();
        // ---- 4 pixels x 9 column displacements per thread -----------------
        const float4* pb = prv_s + qy * RSP + (4 * qx) * P;
        const float4* nb = nxt_s + (qy + dyi) * RSN + (4 * qx) * P;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            float4 p[4], n[4 + D - 1];
#pragma unroll
            for (int i = 0; i < 4; ++i) p[i] = pb[i * P + j];
#pragma unroll
            for (int m = 0; m < 4 + D - 1; ++m) n[m] = nb[m * P + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int dx = 0; dx < D; ++dx) {
                    float a = acc[i][dx];
                    a = fmaf(p[i].x, n[i + dx].x, a);
                    a = fmaf(p[i].y, n[i + dx].y, a);
                    a = fmaf(p[i].z, n[i + dx].z, a);
                    a = fmaf(p[i].w, n[i + dx].w, a);
                    acc[i][dx] = a;
                }
        }
    }

    // ---- epilogue: mean, LeakyReLU, transpose through LDS, coalesced stores --
    const float cf = (float)C;
    float* ost = reinterpret_cast<float*>(smem);
    T* out_b = out + (int64_t)b * H * W * out_pix_stride;
    for (int pass = 0; pass < Cfg::NPASS; ++pass) {
        __syncthreads();  // tiles (pass 0) / previous pass fully consumed
        if (qy / Cfg::RP == pass) {
            const int ly = qy % Cfg::RP;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int dx = 0; dx < D; ++dx)
                    ost[(ly * TW + 4 * qx + i) * DD + dyi * D + dx] = lrelu(acc[i][dx] / cf, slope);
        }
        __syncthreads();
        for (int e = tid; e < Cfg::RP * TW * DD; e += NT) {
            const int k = e % DD;
            const int pix = e / DD;
            const int px = pix % TW;
            const int ly = pix / TW;
            const int gy = y0 + pass * Cfg::RP + ly, gx = x0 + px;
            if (gy < H && gx < W) st(out_b + ((int64_t)gy * W + gx) * out_pix_stride + k, ost[e]);
        }
    }
}

// Generic fallback: any C, any search range, both layouts, fp32/fp16 storage.
// One thread per output element, indexed in memory order so stores coalesce.
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void cost_volume_generic_kernel(
    const T* __restrict__ prv, const T* __restrict__ nxt, T* __restrict__ out, int B, int H, int W,
    int C, int r, int64_t out_pix_stride, float slope) {
    const int d = 2 * r + 1, DD = d * d;
    const int64_t total = (int64_t)B * H * W * DD;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        int b, y, x, k;
        if (LAYOUT == QPWC_NHWC) {
            k = idx % DD;
            int64_t p = idx / DD;
            x = p % W; p /= W;
            y = p % H;
            b = p / H;
        } else {
            int64_t p = idx;
            x = p % W; p /= W;
            y = p % H; p /= H;
            k = p % DD;
            b = p / DD;
        }
        const int yy = y + k / d - r, xx = x + k % d - r;
        float s = 0.0f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            if (LAYOUT == QPWC_NHWC) {
                const T* pp = prv + ((int64_t)(b * H + y) * W + x) * C;
                const T* np = nxt + ((int64_t)(b * H + yy) * W + xx) * C;
                for (int c = 0; c < C; ++c) s = fmaf(ld(pp + c), ld(np + c), s);
            } else {
                const int64_t plane = (int64_t)H * W;
                const T* pp = prv + (int64_t)b * C * plane + (int64_t)y * W + x;
                const T* np = nxt + (int64_t)b * C * plane + (int64_t)yy * W + xx;
                for (int c = 0; c < C; ++c) s = fmaf(ld(pp + c * plane), ld(np + c * plane), s);
            }
        }
        const float v = lrelu(s / (float)C, slope);
        if (LAYOUT == QPWC_NHWC)
            st(out + ((int64_t)(b * H + y) * W + x) * out_pix_stride + k, v);
        else
            st(out + idx, v);
    }
}

template <typename T, int TH, int TW, int CC, bool FUSE>
static int launch_tiled(const T* prv, const T* nxt, const float* flo, T* out, int B, int H, int W,
                        int C, int64_t ops, float slope, hipStream_t s) {
    using Cfg = CvCfg<4, TH, TW, CC>;
    auto kern = cost_volume_tiled_kernel<T, 4, TH, TW, CC, FUSE>;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int64_t nblk = (int64_t)tiles_x * tiles_y * B;
    if (nblk > INT32_MAX) {
        set_error("grid too large");
        return QPWC_E_SHAPE;
    }
    if (dry_run(FUSE ? "cost_volume_tiled_kernel<fused>" : "cost_volume_tiled_kernel")) return QPWC_OK;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(Cfg::NT), 0, s, prv, nxt, flo,
                       out, H, W, C, tiles_x, tiles_y, ops, slope);
    return check_launch("cost_volume_tiled_kernel");
}

template <typename T, bool FUSE>
static int dispatch_tiled(const T* prv, const T* nxt, const float* flo, T* out, int B, int H, int W,
                          int C, int64_t ops, float slope, hipStream_t s) {
    // Small images: a 16x16 tile would leave most of the chip idle.
    if ((int64_t)B * ((H + 15) / 16) * ((W + 15) / 16) >= 256 || (H >= 16 && W >= 16 && H * W >= 2048))
        return launch_tiled<T, 16, 16, 16, FUSE>(prv, nxt, flo, out, B, H, W, C, ops, slope, s);
    return launch_tiled<T, 8, 8, 16, FUSE>(prv, nxt, flo, out, B, H, W, C, ops, slope, s);
}

template <typename T>
static int cost_volume_impl(const T* prv, const T* nxt, const float* flo, T* out, int B, int H,
                            int W, int C, int r, int layout, int64_t ops, float slope, bool fuse,
                            hipStream_t s) {
    const bool fast = layout == QPWC_NHWC && r == 4 && C % 4 == 0 &&
                      (reinterpret_cast<uintptr_t>(prv) % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(nxt) % 16 == 0);
    if (fuse) {
        if (!fast) {
            set_error("fused warp+cost volume needs NHWC, search_range 4, C %% 4 == 0 (got C=%d r=%d)",
                      C, r);
            return QPWC_E_SHAPE;
        }
        return dispatch_tiled<T, true>(prv, nxt, flo, out, B, H, W, C, ops, slope, s);
    }
    if (fast) return dispatch_tiled<T, false>(prv, nxt, nullptr, out, B, H, W, C, ops, slope, s);
    if (dry_run("cost_volume_generic_kernel")) return QPWC_OK;
    const int DD = (2 * r + 1) * (2 * r + 1);
    const int64_t total = (int64_t)B * H * W * DD;
    const int64_t want = (total + 255) / 256;
    const unsigned grid = (unsigned)(want < 65536 ? want : 65536);
    if (layout == QPWC_NHWC)
        hipLaunchKernelGGL((cost_volume_generic_kernel<T, QPWC_NHWC>), dim3(grid), dim3(256), 0, s,
                           prv, nxt, out, B, H, W, C, r, ops, slope);
    else
        hipLaunchKernelGGL((cost_volume_generic_kernel<T, QPWC_NCHW>), dim3(grid), dim3(256), 0, s,
                           prv, nxt, out, B, H, W, C, r, ops, slope);
    return check_launch("cost_volume_generic_kernel");
}

int cost_volume_mfma_launch(const void* prv, const void* nxt, const void* flo, void* out, int B, int H, int W,
                            int C, int dtype, int64_t ops, float slope, int pad84, bool* pads_written,
                            hipStream_t s);

// channels 81..83 of an 84-float pixel := 0 (see cost_volume_launch, pad84)
template <typename T>
__global__ __launch_bounds__(256) void zero_pads_kernel(T* __restrict__ out, int64_t n_pixels) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pixels * 3;
         i += (int64_t)gridDim.x * blockDim.x)
        st(out + (i / 3) * 84 + 81 + (i % 3), 0.0f);
}

// pad84: the caller passes ops == 84 with channel offset 0 and wants channels 81..83 zeroed (an
// 84-channel cost volume whose pixels are 16-byte aligned, for vector loads downstream).
int cost_volume_launch(const void* prv, const void* nxt, const void* flo, void* out, int B, int H,
                       int W, int C, int r, int layout, int dtype, int64_t ops, float slope,
                       bool fuse, bool pad84, hipStream_t s) {
    auto zero_pads = [&]() {
        if (g_dry_run) return;
        const int64_t npx = (int64_t)B * H * W;
        const unsigned grid = (unsigned)((npx * 3 + 255) / 256 < 4096 ? (npx * 3 + 255) / 256 : 4096);
        if (dtype == QPWC_F32)
            hipLaunchKernelGGL(zero_pads_kernel<float>, dim3(grid), dim3(256), 0, s, (float*)out, npx);
        else
            hipLaunchKernelGGL(zero_pads_kernel<__half>, dim3(grid), dim3(256), 0, s, (__half*)out, npx);
    };
    if (layout == QPWC_NHWC && r == 4) {
        bool pads_written = false;
        const int rc = cost_volume_mfma_launch(prv, nxt, fuse ? flo : nullptr, out, B, H, W, C, dtype, ops, slope,
                                               pad84 ? 1 : 0, &pads_written, s);
        if (rc != 1) {  // 1 = shape not eligible for the matrix-core path
            if (rc == QPWC_OK && pad84 && !pads_written) zero_pads();
            return rc;
        }
    }
    if (pad84) zero_pads();
    if (dtype == QPWC_F32)
        return cost_volume_impl<float>((const float*)prv, (const float*)nxt, (const float*)flo,
                                       (float*)out, B, H, W, C, r, layout, ops, slope, fuse, s);
    return cost_volume_impl<__half>((const __half*)prv, (const __half*)nxt, (const float*)flo,
                                    (__half*)out, B, H, W, C, r, layout, ops, slope, fuse, s);
}

}  // namespace qpwc
