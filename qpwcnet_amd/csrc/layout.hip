// NCHW <-> NHWC conversion at the boundary of the hot path (gfx950).
//
// The reference reads its layout from tf.keras.backend.image_data_format() (qpwcnet/core/layers.py:41,146)
// and its inference script defaults to 'channels_first' (app/optical_flow/test_infer.py:52).  Every kernel
// of this library works on channels-last pixels (a pixel's channel vector is one contiguous run: that is
// what the matrix-core operand loads and the 16-byte gathers need), so a dense (B,C,H,W) tensor crosses the
// boundary through this transposition: 64 pixels x 32 channels per workgroup through LDS, both global sides
// in runs of >= 128 bytes (NCHW side: 64 consecutive pixels of one channel plane; NHWC side: 32
// consecutive channels of one pixel), LDS rows padded to 65 floats (conflict free both ways).
#include "common.h"

namespace qpwc {

constexpr int kLtPix = 64, kLtCh = 32;

// grid: (pixel tiles of one image, channel chunks, B)
template <typename T, bool TO_NHWC>
__global__ __launch_bounds__(256) void layout_transpose_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                              int C, int64_t HW) {
    __shared__ float tile[kLtCh][kLtPix + 1];
    const int tid = threadIdx.x;
    const int64_t p0 = (int64_t)blockIdx.x * kLtPix;
    const int c0 = blockIdx.y * kLtCh;
    const int64_t b = blockIdx.z;
    const T* ib = in + b * C * HW;
    T* ob = out + b * C * HW;
    // plane-side map: lane = pixel (64 consecutive), 4 channels per pass; pixel-side map: lane = channel
    // (32 consecutive), 8 pixels per pass
    const int ppx = tid & 63, pcr = tid >> 6;
    const int qch = tid & 31, qpr = tid >> 5;
    if (TO_NHWC) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = 4 * k + pcr;
            float v = 0.0f;
            if (c0 + c < C && p0 + ppx < HW) v = ld<T>(ib + (int64_t)(c0 + c) * HW + p0 + ppx);
            tile[c][ppx] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int px = 8 * k + qpr;
            if (c0 + qch < C && p0 + px < HW) st<T>(ob + (p0 + px) * C + c0 + qch, tile[qch][px]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int px = 8 * k + qpr;
            float v = 0.0f;
            if (c0 + qch < C && p0 + px < HW) v = ld<T>(ib + (p0 + px) * C + c0 + qch);
            tile[qch][px] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = 4 * k + pcr;
            if (c0 + c < C && p0 + ppx < HW) st<T>(ob + (int64_t)(c0 + c) * HW + p0 + ppx, tile[c][ppx]);
        }
    }
}

int layout_transpose_launch(const void* in, void* out, int B, int H, int W, int C, int to_layout, int dtype,
                            hipStream_t s) {
    const int64_t HW = (int64_t)H * W;
    const int64_t tiles = (HW + kLtPix - 1) / kLtPix;
    const int chunks = (C + kLtCh - 1) / kLtCh;
    if (tiles > INT32_MAX || chunks > 65535 || B > 65535) {
        set_error("layout transpose: grid too large (tiles=%lld chunks=%d B=%d)", (long long)tiles, chunks, B);
        return QPWC_E_SHAPE;
    }
    const dim3 grid((unsigned)tiles, (unsigned)chunks, (unsigned)B);
#define QPWC_LT(T, TO)                                                                                   \
    hipLaunchKernelGGL((layout_transpose_kernel<T, TO>), grid, dim3(256), 0, s, (const T*)in, (T*)out, C, HW)
    if (dtype == QPWC_F32) { if (to_layout == QPWC_NHWC) QPWC_LT(float, true); else QPWC_LT(float, false); }
    else                   { if (to_layout == QPWC_NHWC) QPWC_LT(__half, true); else QPWC_LT(__half, false); }
#undef QPWC_LT
    return check_launch("layout_transpose_kernel");
}

}  // namespace qpwc
