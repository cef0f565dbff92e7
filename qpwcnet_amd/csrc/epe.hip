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
    int64_t plane[kEpeMaxLevels];   // 0: (B,H,W,2) pixels; > 0: (B,2,H,W) with H*W = plane
    int b_f16[kEpeMaxLevels];       // the prediction `b` of this level is fp16 (a: always fp32)
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
    const int64_t plane = lv.plane[l];
    // two pixels per 16-byte load, four loads of each array in flight (the finest level is 1 M pixels
    // for 65 k threads: one 8-byte load at a time made this pass latency bound)
    if (lv.b_f16[l]) {   // fp16 predictions (fp16-storage network): converted here instead of by a pass of their own
        const __half* bh = reinterpret_cast<const __half*>(lv.b[l]);
        const float* af = lv.a[l];
        if (plane > 0) {
            for (int64_t i = tid; i < n; i += nthr) {
                const int64_t img = i / plane, r = i - img * plane;
                const int64_t o = img * 2 * plane + r;
                const float dx = af[o] - __half2float(bh[o]), dy = af[o + plane] - __half2float(bh[o + plane]);
                s += sqrtf(dx * dx + dy * dy);
            }
        } else {
            int64_t i = tid;
            for (; i + 3 * nthr < n; i += 4 * nthr) {
                float2 va[4];
                __half2 vb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    va[k] = a[i + k * nthr];
                    vb[k] = *reinterpret_cast<const __half2*>(bh + 2 * (i + k * nthr));
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float2 fb = __half22float2(vb[k]);
                    const float dx = va[k].x - fb.x, dy = va[k].y - fb.y;
                    s += sqrtf(dx * dx + dy * dy);
                }
            }
            for (; i < n; i += nthr) {
                const float2 va = a[i], fb = __half22float2(*reinterpret_cast<const __half2*>(bh + 2 * i));
                const float dx = va.x - fb.x, dy = va.y - fb.y;
                s += sqrtf(dx * dx + dy * dy);
            }
        }
    } else if (plane > 0) {   // 'channels_first' flows: x and y planes, lanes walk the pixels of a plane
        const float* af = lv.a[l];
        const float* bf = lv.b[l];
        for (int64_t i = tid; i < n; i += nthr) {
            const int64_t img = i / plane, r = i - img * plane;
            const int64_t o = img * 2 * plane + r;
            const float dx = af[o] - bf[o], dy = af[o + plane] - bf[o + plane];
            s += sqrtf(dx * dx + dy * dy);
        }
    } else if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) % 16 == 0) {
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

int epe_multi_launch(const void* const* a, const void* const* b, const int64_t* npix, const int64_t* plane,
                     const int* pred_dtype, int n_levels, float* out, float* ws, hipStream_t s) {
    EpeLevels lv;
    for (int i = 0; i < kEpeMaxLevels; ++i) {
        lv.b_f16[i] = (i < n_levels && pred_dtype && pred_dtype[i] == QPWC_F16) ? 1 : 0;
        lv.a[i] = i < n_levels ? (const float*)a[i] : nullptr;
        lv.b[i] = i < n_levels ? (const float*)b[i] : nullptr;
        lv.npix[i] = i < n_levels ? npix[i] : 0;
        lv.plane[i] = (i < n_levels && plane) ? plane[i] : 0;
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

// ---------------------------------------------------------------------------
// cost_volume_to_flow (qpwcnet/core/vis.py:9-34): per pixel the displacement of the strongest
// correlation, imax = argmax_k cvol[..., k] (first maximum, as tf.argmax), q = sqrt(D),
// di = floor(imax / q) - (q-1)/2, dj = imax - floor(imax / q) * q - (q-1)/2, all in fp32 as the
// reference computes them; out = (di, dj) = (row, column) displacement -- the decode that pins the
// channel order i0 * 9 + j0 of the cost volume.
// NHWC: 16 lanes per pixel scan channels l, l+16, ... (64-byte coalesced runs) and reduce with
// shuffles; NCHW: one thread per pixel walks the D planes (coalesced across pixels).
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void cost_volume_to_flow_kernel(const T* __restrict__ cvol,
                                                                  float* __restrict__ flow, int64_t npix,
                                                                  int64_t plane, int D, int64_t pix_stride,
                                                                  float q) {
    if (LAYOUT == QPWC_NHWC) {
        const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
        const int l = threadIdx.x & 15;
        float best = -INFINITY;
        int arg = 0x7fffffff;
        if (p < npix) {
            const T* src = cvol + p * pix_stride;
            for (int k = l; k < D; k += 16) {
                const float v = ld<T>(src + k);
                if (v > best || arg == 0x7fffffff) { best = v; arg = k; }   // strictly greater: first maximum wins
            }
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off, 16);
            const int oa = __shfl_xor(arg, off, 16);
            if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        }
        if (p < npix && l == 0) {
            const float im = (float)arg;
            const float fi = floorf(im / q);
            const float dj = im - fi * q;
            const float half = (q - 1.0f) / 2.0f;
            *reinterpret_cast<float2*>(flow + 2 * p) = make_float2(fi - half, dj - half);
        }
    } else {
        const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (p >= npix) return;
        const int64_t b = p / plane, r = p - b * plane;
        const T* src = cvol + b * D * plane + r;
        float best = ld<T>(src);
        int arg = 0;
        for (int k = 1; k < D; ++k) {
            const float v = ld<T>(src + (int64_t)k * plane);
            if (v > best) { best = v; arg = k; }
        }
        const float im = (float)arg;
        const float fi = floorf(im / q);
        const float dj = im - fi * q;
        const float half = (q - 1.0f) / 2.0f;
        flow[(b * 2) * plane + r] = fi - half;
        flow[(b * 2 + 1) * plane + r] = dj - half;
    }
}

int cost_volume_to_flow_launch(const void* cvol, float* flow, int B, int H, int W, int D, int64_t pix_stride,
                               int layout, int dtype, hipStream_t s) {
    const int64_t npix = (int64_t)B * H * W, plane = (int64_t)H * W;
    const float q = sqrtf((float)D);
    const int64_t threads = layout == QPWC_NHWC ? npix * 16 : npix;
    const int64_t nblk = (threads + 255) / 256;
    if (nblk > INT32_MAX) {
        set_error("cost_volume_to_flow: too many pixels");
        return QPWC_E_SHAPE;
    }
    const dim3 grid((unsigned)nblk);
#define QPWC_CVF(T, L)                                                                                      \
    hipLaunchKernelGGL((cost_volume_to_flow_kernel<T, L>), grid, dim3(256), 0, s, (const T*)cvol, flow, npix, \
                       plane, D, pix_stride, q)
    if (dtype == QPWC_F32) { if (layout == QPWC_NHWC) QPWC_CVF(float, QPWC_NHWC); else QPWC_CVF(float, QPWC_NCHW); }
    else                   { if (layout == QPWC_NHWC) QPWC_CVF(__half, QPWC_NHWC); else QPWC_CVF(__half, QPWC_NCHW); }
#undef QPWC_CVF
    return check_launch("cost_volume_to_flow_kernel");
}

}  // namespace qpwc
