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
#include <type_traits>

#include "common.h"

namespace qpwc {

// element strides of the flow tensor for (b, y, x, channel); 0 = broadcast
struct FloStrides {
    int64_t b, y, x, c;
};

// VEC consecutive channels of one pixel as fp32: 16-byte loads for fp32 x 4 and fp16 x 8, 8-byte for fp16 x 4
template <int VEC>
struct ChanVec {
    float v[VEC];
};
__device__ __forceinline__ ChanVec<4> ldvec4(const float* p) {
    const float4 a = ld4(p);
    return ChanVec<4>{{a.x, a.y, a.z, a.w}};
}
__device__ __forceinline__ ChanVec<4> ldvec4(const __half* p) {
    const float4 a = ld4(p);
    return ChanVec<4>{{a.x, a.y, a.z, a.w}};
}
__device__ __forceinline__ ChanVec<8> ldvec8(const __half* p) {
    const uint4 raw = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
    ChanVec<8> r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float2 f = __half22float2(*reinterpret_cast<const __half2*>(&w[i]));
        r.v[2 * i] = f.x;
        r.v[2 * i + 1] = f.y;
    }
    return r;
}
__device__ __forceinline__ void stvec(float* p, const ChanVec<4>& a) {
    st4(p, make_float4(a.v[0], a.v[1], a.v[2], a.v[3]));
}
__device__ __forceinline__ void stvec(__half* p, const ChanVec<4>& a) {
    st4(p, make_float4(a.v[0], a.v[1], a.v[2], a.v[3]));
}
__device__ __forceinline__ void stvec(__half* p, const ChanVec<8>& a) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __half2 h = __floats2half2_rn(a.v[2 * i], a.v[2 * i + 1]);
        w[i] = *reinterpret_cast<const unsigned*>(&h);
    }
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}
template <int VEC, typename T>
__device__ __forceinline__ ChanVec<VEC> ldvec(const T* p) {
    if constexpr (VEC == 8) return ldvec8(p);
    else return ldvec4(p);
}

// VEC = 4: fp32 (16-byte accesses) and fp16 with C % 8 != 0 (8-byte); VEC = 8: fp16 with C % 8 == 0 -- a lane moves
// 16 bytes per access like the fp32 kernel does (round 3: the fp16 kernel with 8-byte accesses ran at 0.35 of the HBM
// peak at config 5's L4 against 0.54-0.57 for fp32).  The kernel keeps its round-1 symbol name for VEC = 4.
template <typename T, int MODE, int VEC>
__device__ __forceinline__ void warp_nhwc_body(const T* __restrict__ img, const float* __restrict__ flo,
                                               T* __restrict__ out, int B, int H, int W, int C, FloStrides fs) {
    const int nch = C / VEC;
    const int64_t total = (int64_t)B * H * W * nch;
    // two independent items per thread and trip: 2 flow reads, then 8 corner gathers in flight
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // Round 4: workgroups are dealt round-robin over the 8 XCDs; in plain grid order eight neighbouring 32- / 64-pixel
    // runs land on eight different L2s and every XCD gathers from everywhere in the image -- the request-size
    // counters (TCC_EA0_RDREQ_128B: all of a launch's fabric reads are 128-byte lines) showed 1.92 x the compulsory
    // read bytes for the fp16 kernel at config 5's L4 (64-byte pixels: each line fetch brings a neighbour pixel that
    // another XCD then fetches again) and 1.18 x for fp32.  An XCD now takes a CONTIGUOUS run of the launch's items
    // (xcd_swizzle: bijective, same items, same arithmetic), so neighbouring rows meet in one L2.
    const int64_t bid = xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    for (int64_t idx0 = bid * blockDim.x + threadIdx.x; idx0 < total; idx0 += 2 * stride) {
        const int64_t idx1 = idx0 + stride;
        const bool two = idx1 < total;
        int ch[2], x[2], y[2], b[2];
        float fx[2], fy[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int64_t idx = (k == 0 || two) ? (k == 0 ? idx0 : idx1) : idx0;
            ch[k] = idx % nch;
            int64_t p = idx / nch;
            x[k] = p % W;
            p /= W;
            y[k] = p % H;
            b[k] = p / H;
            const float* f = flo + b[k] * fs.b + y[k] * fs.y + x[k] * fs.x;
            fx[k] = f[0];
            fy[k] = f[fs.c];
        }
        ChanVec<VEC> tl[2], tr[2], bl[2], br[2];
        Taps t[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            t[k] = make_taps<MODE>(y[k], x[k], fx[k], fy[k], H, W);
            const T* ib = img + (int64_t)b[k] * H * W * C + VEC * ch[k];
            tl[k] = ldvec<VEC>(ib + ((int64_t)t[k].y0 * W + t[k].x0) * C);
            tr[k] = ldvec<VEC>(ib + ((int64_t)t[k].y0 * W + t[k].x1) * C);
            bl[k] = ldvec<VEC>(ib + ((int64_t)t[k].y1 * W + t[k].x0) * C);
            br[k] = ldvec<VEC>(ib + ((int64_t)t[k].y1 * W + t[k].x1) * C);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (k == 0 || two) {
                ChanVec<VEC> o;
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    o.v[e] = blend<MODE>(t[k], tl[k].v[e], tr[k].v[e], bl[k].v[e], br[k].v[e]);
                stvec(out + (((int64_t)(b[k] * H + y[k]) * W + x[k]) * C + VEC * ch[k]), o);
            }
    }
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void warp_nhwc_vec4_kernel(const T* __restrict__ img,
                                                             const float* __restrict__ flo,
                                                             T* __restrict__ out, int B, int H,
                                                             int W, int C, FloStrides fs) {
    QPWC_FLOW_CHAIN_PRIO();
    warp_nhwc_body<T, MODE, 4>(img, flo, out, B, H, W, C, fs);
}

template <int MODE>
__global__ __launch_bounds__(256) void warp_nhwc_half8_kernel(const __half* __restrict__ img,
                                                              const float* __restrict__ flo,
                                                              __half* __restrict__ out, int B, int H, int W,
                                                              int C, FloStrides fs) {
    QPWC_FLOW_CHAIN_PRIO();
    warp_nhwc_body<__half, MODE, 8>(img, flo, out, B, H, W, C, fs);
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
    const bool half8 = fast && std::is_same<T, __half>::value && C % 8 == 0;
    const int64_t total = fast ? (int64_t)B * H * W * (C / (half8 ? 8 : 4)) : (int64_t)B * H * W * C;
    // fast path: two items per thread; every CU gets >= 8 workgroups before the grid is halved
    const int64_t want = fast && total >= 2 * 256 * 2048 ? (total + 511) / 512 : (total + 255) / 256;
    const unsigned grid = (unsigned)(want < (1 << 20) ? want : (1 << 20));
    if (half8)
        hipLaunchKernelGGL((warp_nhwc_half8_kernel<MODE>), dim3(grid), dim3(256), 0, s, (const __half*)img, flo,
                           (__half*)out, B, H, W, C, fs);
    else if (fast)
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
