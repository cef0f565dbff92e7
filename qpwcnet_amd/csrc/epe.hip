// End-point error reduction (qpwcnet/app/optical_flow/train.py:247-253):
//   mean over (b,y,x) of || y_true - y_pred ||_2 along the 2-channel flow axis.
// Deterministic two-stage sum: kEpeBlocks partials, then one block folds them.
#include "common.h"

namespace qpwc {

constexpr int kEpeBlocks = 256;
constexpr int kEpeThreads = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    v = wave_sum(v);
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float r = 0.0f;
    if (wid == 0) {
        r = lane < (int)(blockDim.x >> 6) ? red[lane] : 0.0f;
        r = wave_sum(r);
    }
    return r;  // valid in thread 0
}

template <int LAYOUT>
__global__ __launch_bounds__(kEpeThreads) void epe_partial_kernel(const float* __restrict__ a,
                                                                  const float* __restrict__ b,
                                                                  float* __restrict__ partial,
                                                                  int64_t npix, int64_t plane) {
    __shared__ float red[kEpeThreads / 64];
    float s = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
         i += (int64_t)gridDim.x * blockDim.x) {
        float dx, dy;
        if (LAYOUT == QPWC_NHWC) {
            const float2 va = reinterpret_cast<const float2*>(a)[i];
            const float2 vb = reinterpret_cast<const float2*>(b)[i];
            dx = va.x - vb.x;
            dy = va.y - vb.y;
        } else {
            const int64_t bi = i / plane, r = i % plane;
            const int64_t o = bi * 2 * plane + r;
            dx = a[o] - b[o];
            dy = a[o + plane] - b[o + plane];
        }
        s += sqrtf(dx * dx + dy * dy);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(kEpeThreads) void epe_final_kernel(const float* __restrict__ partial,
                                                                float* __restrict__ out, int n,
                                                                float inv_npix) {
    __shared__ float red[kEpeThreads / 64];
    float s = (int)threadIdx.x < n ? partial[threadIdx.x] : 0.0f;
    s = block_sum(s, red);
    if (threadIdx.x == 0) *out = s * inv_npix;
}

// All levels of the multi-scale EPE (FlowMseLoss, qpwcnet/train/loss.py:56-67) in two
// launches instead of 2 per level: blockIdx.y = level, blockIdx.x = partial sum.
constexpr int kEpeMaxLevels = 8;
constexpr int kEpeMultiBlocks = 256;
struct EpeLevels {
    const float* a[kEpeMaxLevels];
    const float* b[kEpeMaxLevels];
    int64_t npix[kEpeMaxLevels];
};

__global__ __launch_bounds__(kEpeThreads) void epe_multi_partial_kernel(EpeLevels lv,
                                                                       float* __restrict__ partial) {
    __shared__ float red[kEpeThreads / 64];
    const int l = blockIdx.y;
    const float2* a = reinterpret_cast<const float2*>(lv.a[l]);
    const float2* b = reinterpret_cast<const float2*>(lv.b[l]);
    const int64_t n = lv.npix[l];
    float s = 0.0f;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    // two pixels per 16-byte load, four loads of each array in flight (the finest level is 1 M pixels
    // for 65 k threads: one 8-byte load at a time made this pass latency bound)
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) % 16 == 0) {
        const float4* a4 = reinterpret_cast<const float4*>(a);
        const float4* b4 = reinterpret_cast<const float4*>(b);
        const int64_t n2 = n / 2;
        int64_t i = tid;
        for (; i + 3 * nthr < n2; i += 4 * nthr) {
            float4 va[4], vb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                va[k] = a4[i + k * nthr];
                vb[k] = b4[i + k * nthr];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dx0 = va[k].x - vb[k].x, dy0 = va[k].y - vb[k].y;
                const float dx1 = va[k].z - vb[k].z, dy1 = va[k].w - vb[k].w;
                s += sqrtf(dx0 * dx0 + dy0 * dy0);
                s += sqrtf(dx1 * dx1 + dy1 * dy1);
            }
        }
        for (; i < n2; i += nthr) {
            const float4 va = a4[i], vb = b4[i];
            const float dx0 = va.x - vb.x, dy0 = va.y - vb.y, dx1 = va.z - vb.z, dy1 = va.w - vb.w;
            s += sqrtf(dx0 * dx0 + dy0 * dy0);
            s += sqrtf(dx1 * dx1 + dy1 * dy1);
        }
        if ((n & 1) && tid == 0) {
            const float2 va = a[n - 1], vb = b[n - 1];
            const float dx = va.x - vb.x, dy = va.y - vb.y;
            s += sqrtf(dx * dx + dy * dy);
        }
    } else {
        for (int64_t i = tid; i < n; i += nthr) {
            const float2 va = a[i], vb = b[i];
            const float dx = va.x - vb.x, dy = va.y - vb.y;
            s += sqrtf(dx * dx + dy * dy);
        }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[l * kEpeMultiBlocks + blockIdx.x] = s;
}

__global__ __launch_bounds__(kEpeThreads) void epe_multi_final_kernel(const float* __restrict__ partial,
                                                                      float* __restrict__ out,
                                                                      EpeLevels lv) {
    __shared__ float red[kEpeThreads / 64];
    const int l = blockIdx.x;
    float s = (int)threadIdx.x < kEpeMultiBlocks ? partial[l * kEpeMultiBlocks + threadIdx.x] : 0.0f;
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[l] = s / (float)lv.npix[l];
}

int epe_multi_workspace_floats() { return kEpeMaxLevels * kEpeMultiBlocks; }

int epe_multi_launch(const void* const* a, const void* const* b, const int64_t* npix, int n_levels,
                     float* out, float* ws, hipStream_t s) {
    EpeLevels lv;
    for (int i = 0; i < kEpeMaxLevels; ++i) {
        lv.a[i] = i < n_levels ? (const float*)a[i] : nullptr;
        lv.b[i] = i < n_levels ? (const float*)b[i] : nullptr;
        lv.npix[i] = i < n_levels ? npix[i] : 0;
    }
    hipLaunchKernelGGL(epe_multi_partial_kernel, dim3(kEpeMultiBlocks, n_levels), dim3(kEpeThreads), 0, s,
                       lv, ws);
    int rc = check_launch("epe_multi_partial_kernel");
    if (rc != QPWC_OK) return rc;
    hipLaunchKernelGGL(epe_multi_final_kernel, dim3(n_levels), dim3(kEpeThreads), 0, s, ws, out, lv);
    return check_launch("epe_multi_final_kernel");
}

int epe_workspace_floats() { return kEpeBlocks; }

int epe_launch(const float* a, const float* b, float* out, float* ws, int B, int H, int W,
               int layout, hipStream_t s) {
    const int64_t npix = (int64_t)B * H * W;
    const int64_t plane = (int64_t)H * W;
    if (layout == QPWC_NHWC)
        hipLaunchKernelGGL((epe_partial_kernel<QPWC_NHWC>), dim3(kEpeBlocks), dim3(kEpeThreads), 0,
                           s, a, b, ws, npix, plane);
    else
        hipLaunchKernelGGL((epe_partial_kernel<QPWC_NCHW>), dim3(kEpeBlocks), dim3(kEpeThreads), 0,
                           s, a, b, ws, npix, plane);
    int rc = check_launch("epe_partial_kernel");
    if (rc != QPWC_OK) return rc;
    hipLaunchKernelGGL(epe_final_kernel, dim3(1), dim3(kEpeThreads), 0, s, ws, out, kEpeBlocks,
                       1.0f / (float)npix);
    return check_launch("epe_final_kernel");
}

}  // namespace qpwc
