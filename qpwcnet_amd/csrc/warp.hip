// Warp / WarpV2 forward for gfx950 (MI355X, CDNA4, wave64).
//
// Reference semantics:
//   mode CLAMP  : WarpV2.call, qpwcnet/core/layers.py:177-186 =
//                 tfa.image.dense_image_warp(img, -flo[..., ::-1]); algorithm documented in-tree
//                 at qpwcnet/core/warp.py:156-211 (clamp-to-border bilinear).
//   mode TFWARP : Warp.call -> tf_warp, qpwcnet/core/warp.py:63-153 (+ :8-47).
// Both sample img at (y + flo[...,1], x + flo[...,0]).
//
// Pure gather, ~1 flop/byte: HBM/L2 bound.  NHWC fast path: C/4 consecutive lanes
// cover one pixel's channel vector with 16-byte loads (the four corner gathers
// and the store are each contiguous per pixel), flow is read once per lane
// group from L1.  Compiled with -ffp-contract=off so every multiply/add rounds
// separately, as in the reference's op-by-op graph.
#include "common.h"

namespace qpwc {

// element strides of the flow tensor for (b, y, x, channel); 0 = broadcast
struct FloStrides {
    int64_t b, y, x, c;
};

template <typename T, int MODE>
__global__ __launch_bounds__(256) void warp_nhwc_vec4_kernel(const T* __restrict__ img,
                                                             const float* __restrict__ flo,
                                                             T* __restrict__ out, int B, int H,
                                                             int W, int C, FloStrides fs) {
    const int nch = C >> 2;
    const int64_t total = (int64_t)B * H * W * nch;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = idx % nch;
        int64_t p = idx / nch;
        const int x = p % W;
        p /= W;
        const int y = p % H;
        const int b = p / H;
        const float* f = flo + b * fs.b + y * fs.y + x * fs.x;
        const float fx = f[0], fy = f[fs.c];
        const Taps t = make_taps<MODE>(y, x, fx, fy, H, W);
        const T* ib = img + (int64_t)b * H * W * C + 4 * ch;
        const float4 tl = ld4(ib + ((int64_t)t.y0 * W + t.x0) * C);
        const float4 tr = ld4(ib + ((int64_t)t.y0 * W + t.x1) * C);
        const float4 bl = ld4(ib + ((int64_t)t.y1 * W + t.x0) * C);
        const float4 br = ld4(ib + ((int64_t)t.y1 * W + t.x1) * C);
        st4(out + (((int64_t)(b * H + y) * W + x) * C + 4 * ch), blend4<MODE>(t, tl, tr, bl, br));
    }
}

// Generic: any C, both layouts; one thread per element in memory order.
template <typename T, int MODE, int LAYOUT>
__global__ __launch_bounds__(256) void warp_generic_kernel(const T* __restrict__ img,
                                                           const float* __restrict__ flo,
                                                           T* __restrict__ out, int B, int H, int W,
                                                           int C, FloStrides fs) {
    const int64_t total = (int64_t)B * H * W * C;
    const int64_t plane = (int64_t)H * W;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        int b, y, x, c;
        int64_t p = idx;
        if (LAYOUT == QPWC_NHWC) {
            c = p % C; p /= C;
            x = p % W; p /= W;
            y = p % H;
            b = p / H;
        } else {
            x = p % W; p /= W;
            y = p % H; p /= H;
            c = p % C;
            b = p / C;
        }
        const float* f = flo + b * fs.b + y * fs.y + x * fs.x;
        const float fx = f[0], fy = f[fs.c];
        const Taps t = make_taps<MODE>(y, x, fx, fy, H, W);
        float tl, tr, bl, br;
        if (LAYOUT == QPWC_NHWC) {
            const T* ib = img + (int64_t)b * plane * C + c;
            tl = ld(ib + ((int64_t)t.y0 * W + t.x0) * C);
            tr = ld(ib + ((int64_t)t.y0 * W + t.x1) * C);
            bl = ld(ib + ((int64_t)t.y1 * W + t.x0) * C);
            br = ld(ib + ((int64_t)t.y1 * W + t.x1) * C);
        } else {
            const T* ib = img + ((int64_t)b * C + c) * plane;
            tl = ld(ib + (int64_t)t.y0 * W + t.x0);
            tr = ld(ib + (int64_t)t.y0 * W + t.x1);
            bl = ld(ib + (int64_t)t.y1 * W + t.x0);
            br = ld(ib + (int64_t)t.y1 * W + t.x1);
        }
        st(out + idx, blend<MODE>(t, tl, tr, bl, br));
    }
}

template <typename T, int MODE>
static int warp_impl(const T* img, const float* flo, T* out, int B, int H, int W, int C,
                     FloStrides fs, int layout, hipStream_t s) {
    const bool fast = layout == QPWC_NHWC && C % 4 == 0 &&
                      reinterpret_cast<uintptr_t>(img) % 16 == 0 &&
                      reinterpret_cast<uintptr_t>(out) % 16 == 0;
    const int64_t total = fast ? (int64_t)B * H * W * (C / 4) : (int64_t)B * H * W * C;
    const int64_t want = (total + 255) / 256;
    const unsigned grid = (unsigned)(want < (1 << 20) ? want : (1 << 20));
    if (fast)
        hipLaunchKernelGGL((warp_nhwc_vec4_kernel<T, MODE>), dim3(grid), dim3(256), 0, s, img, flo,
                           out, B, H, W, C, fs);
    else if (layout == QPWC_NHWC)
        hipLaunchKernelGGL((warp_generic_kernel<T, MODE, QPWC_NHWC>), dim3(grid), dim3(256), 0, s,
                           img, flo, out, B, H, W, C, fs);
    else
        hipLaunchKernelGGL((warp_generic_kernel<T, MODE, QPWC_NCHW>), dim3(grid), dim3(256), 0, s,
                           img, flo, out, B, H, W, C, fs);
    return check_launch("warp kernel");
}

int warp_launch(const void* img, const void* flo, void* out, int B, int H, int W, int C,
                int flo_bcast_mask, int layout, int dtype, int mode, hipStream_t s) {
    // dense strides over the non-broadcast dims of the flow tensor
    const int fb = (flo_bcast_mask & QPWC_BCAST_B) ? 1 : B;
    const int fh = (flo_bcast_mask & QPWC_BCAST_H) ? 1 : H;
    const int fw = (flo_bcast_mask & QPWC_BCAST_W) ? 1 : W;
    FloStrides fs;
    if (layout == QPWC_NHWC) {  // (fb, fh, fw, 2)
        fs.c = 1;
        fs.x = fw > 1 ? 2 : 0;
        fs.y = fh > 1 ? (int64_t)fw * 2 : 0;
        fs.b = fb > 1 ? (int64_t)fh * fw * 2 : 0;
    } else {                    // (fb, 2, fh, fw)
        fs.c = (int64_t)fh * fw;
        fs.x = fw > 1 ? 1 : 0;
        fs.y = fh > 1 ? fw : 0;
        fs.b = fb > 1 ? (int64_t)2 * fh * fw : 0;
    }
    const float* f = (const float*)flo;
    if (dtype == QPWC_F32) {
        if (mode == QPWC_WARP_CLAMP)
            return warp_impl<float, QPWC_WARP_CLAMP>((const float*)img, f, (float*)out, B, H, W, C,
                                                     fs, layout, s);
        return warp_impl<float, QPWC_WARP_TFWARP>((const float*)img, f, (float*)out, B, H, W, C, fs,
                                                  layout, s);
    }
    if (mode == QPWC_WARP_CLAMP)
        return warp_impl<__half, QPWC_WARP_CLAMP>((const __half*)img, f, (__half*)out, B, H, W, C,
                                                  fs, layout, s);
    return warp_impl<__half, QPWC_WARP_TFWARP>((const __half*)img, f, (__half*)out, B, H, W, C, fs,
                                               layout, s);
}

}  // namespace qpwc
