// Inverse flow and occlusion map for gfx950.
//
// Reference semantics:
//   invert flow : inv_flow = -tf_warp(flow, flow)         qpwcnet/core/occlusion.py:85,
//                                                          qpwcnet/app/test/test_invert_flow.py:47
//   occlusion   : estimate_occlusion_map                   qpwcnet/core/occlusion.py:27-118
//       oob[p]  = p + flow[p] leaves the image                       (:60-62, :74-75)
//       idx3[p] = clip(int32(p + inv_flow[p]))  (truncation)         (:86-92)
//       map3    = tensor_scatter_nd_min(ones, idx3, zeros)           (:94-95)
//       out     = max(oob, map3)                                     (:98)
//   flow channel 0 = x (dj), channel 1 = y (di)  (:59 `dj, di = tf.unstack(flow)`).
//
// A pixel that no idx3 lands on keeps map3 = 1 and therefore out = 1 whatever its oob; a pixel t
// that is hit gets out[t] = oob[t].  So: fill with 1, then every source pixel writes oob[target]
// to its target -- all writers of one target store the same value, the result does not depend on
// the order (no atomics needed).  Both passes are ~12 B per pixel: HBM/launch-latency bound.
#include "common.h"

namespace qpwc {

template <typename T, int LAYOUT>
struct FlowField {
    const T* p;
    int H, W;
    __device__ __forceinline__ float2 at(int b, int y, int x) const {
        if (LAYOUT == QPWC_NHWC) {
            const T* q = p + (((int64_t)b * H + y) * W + x) * 2;
            return make_float2(ld(q), ld(q + 1));
        }
        const int64_t plane = (int64_t)H * W;
        const T* q = p + (int64_t)b * 2 * plane + (int64_t)y * W + x;
        return make_float2(ld(q), ld(q + plane));
    }
};

// -tf_warp(flow, flow) at one pixel (warp.py:100-151 on the 2-channel image `flow`)
template <typename T, int LAYOUT>
__device__ __forceinline__ float2 inverse_flow_at(const FlowField<T, LAYOUT>& f, int b, int y, int x) {
    const float2 fl = f.at(b, y, x);
    const Taps t = taps_tfwarp(y, x, fl.x, fl.y, f.H, f.W);
    const float2 tl = f.at(b, t.y0, t.x0), tr = f.at(b, t.y0, t.x1);
    const float2 bl = f.at(b, t.y1, t.x0), br = f.at(b, t.y1, t.x1);
    return make_float2(-blend<QPWC_WARP_TFWARP>(t, tl.x, tr.x, bl.x, br.x),
                       -blend<QPWC_WARP_TFWARP>(t, tl.y, tr.y, bl.y, br.y));
}

template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void invert_flow_kernel(const T* __restrict__ flow,
                                                          T* __restrict__ out, int B, int H, int W) {
    const FlowField<T, LAYOUT> f{flow, H, W};
    const int64_t total = (int64_t)B * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = idx % W;
        const int y = (idx / W) % H;
        const int b = idx / ((int64_t)W * H);
        const float2 v = inverse_flow_at(f, b, y, x);
        if (LAYOUT == QPWC_NHWC) {
            st(out + idx * 2, v.x);
            st(out + idx * 2 + 1, v.y);
        } else {
            const int64_t plane = (int64_t)H * W;
            T* q = out + (int64_t)b * 2 * plane + (int64_t)y * W + x;
            st(q, v.x);
            st(q + plane, v.y);
        }
    }
}

__global__ __launch_bounds__(256) void fill_ones_kernel(float* __restrict__ out, int64_t n) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += (int64_t)gridDim.x * blockDim.x)
        out[idx] = 1.0f;
}

template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void occlusion_scatter_kernel(const T* __restrict__ flow,
                                                                float* __restrict__ out, int B, int H,
                                                                int W) {
#pragma clang fp contract(off)
    const FlowField<T, LAYOUT> f{flow, H, W};
    const int64_t total = (int64_t)B * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = idx % W;
        const int y = (idx / W) % H;
        const int b = idx / ((int64_t)W * H);
        const float2 inv = inverse_flow_at(f, b, y, x);
        const int ti = clampi((int)((float)y + inv.y), 0, H - 1);   // int32 cast truncates (:88)
        const int tj = clampi((int)((float)x + inv.x), 0, W - 1);
        const float2 ft = f.at(b, ti, tj);
        const float i2 = (float)ti + ft.y, j2 = (float)tj + ft.x;
        const bool oob = i2 < 0.0f || i2 >= (float)H || j2 < 0.0f || j2 >= (float)W;
        out[((int64_t)b * H + ti) * W + tj] = oob ? 1.0f : 0.0f;
    }
}

static unsigned grid_for(int64_t total) {
    const int64_t want = (total + 255) / 256;
    return (unsigned)(want < (1 << 20) ? want : (1 << 20));
}

template <typename T>
static int invert_flow_impl(const T* flow, T* out, int B, int H, int W, int layout, hipStream_t s) {
    const unsigned grid = grid_for((int64_t)B * H * W);
    if (layout == QPWC_NHWC)
        hipLaunchKernelGGL((invert_flow_kernel<T, QPWC_NHWC>), dim3(grid), dim3(256), 0, s, flow, out, B, H, W);
    else
        hipLaunchKernelGGL((invert_flow_kernel<T, QPWC_NCHW>), dim3(grid), dim3(256), 0, s, flow, out, B, H, W);
    return check_launch("invert_flow kernel");
}

int invert_flow_launch(const void* flow, void* out, int B, int H, int W, int layout, int dtype,
                       hipStream_t s) {
    if (dtype == QPWC_F32) return invert_flow_impl((const float*)flow, (float*)out, B, H, W, layout, s);
    return invert_flow_impl((const __half*)flow, (__half*)out, B, H, W, layout, s);
}

template <typename T>
static int occlusion_impl(const T* flow, float* out, int B, int H, int W, int layout, hipStream_t s) {
    const int64_t total = (int64_t)B * H * W;
    const unsigned grid = grid_for(total);
    hipLaunchKernelGGL(fill_ones_kernel, dim3(grid), dim3(256), 0, s, out, total);
    if (layout == QPWC_NHWC)
        hipLaunchKernelGGL((occlusion_scatter_kernel<T, QPWC_NHWC>), dim3(grid), dim3(256), 0, s, flow, out, B, H, W);
    else
        hipLaunchKernelGGL((occlusion_scatter_kernel<T, QPWC_NCHW>), dim3(grid), dim3(256), 0, s, flow, out, B, H, W);
    return check_launch("occlusion kernels");
}

int occlusion_launch(const void* flow, void* out, int B, int H, int W, int layout, int dtype,
                     hipStream_t s) {
    if (dtype == QPWC_F32) return occlusion_impl((const float*)flow, (float*)out, B, H, W, layout, s);
    return occlusion_impl((const __half*)flow, (float*)out, B, H, W, layout, s);
}

}  // namespace qpwc
