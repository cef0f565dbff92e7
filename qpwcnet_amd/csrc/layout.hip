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

// ---------------------------------------------------------------------------
// Copy of a channels-last (B,H,W,C) view into another one, both with their own batch / row / pixel strides and
// contiguous channels: the skip half of the decoder's concat([UpConv(x), skip]) (qpwcnet/core/pwcnet.py:186-195) --
// source = the un-padded view of the encoder's zero-bordered output, destination = channels F.. of the concat
// buffer.  One 16-byte chunk per thread and trip, four trips in flight; a pixel's chunks sit in neighbouring lanes.
__global__ __launch_bounds__(256) void copy_pixels_kernel(const char* __restrict__ src, char* __restrict__ dst,
                                                         int H, int W, int nq, int64_t total, int64_t sb,
                                                         int64_t sy, int64_t sx, int64_t db, int64_t dy,
                                                         int64_t dx) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    auto offs = [&](int64_t i, int64_t& so, int64_t& d_o) {
        const int q = (int)(i % nq);
        int64_t p = i / nq;
        const int x = (int)(p % W);
        p /= W;
        const int y = (int)(p % H);
        const int64_t b = p / H;
        so = b * sb + y * sy + x * sx + 16 * q;
        d_o = b * db + y * dy + x * dx + 16 * q;
    };
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * nthr < total; i += 4 * nthr) {
        int64_t so[4], d_o[4];
        f4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) offs(i + k * nthr, so[k], d_o[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f4*>(src + so[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<f4*>(dst + d_o[k]) = v[k];
    }
    for (; i < total; i += nthr) {
        int64_t so, d_o;
        offs(i, so, d_o);
        *reinterpret_cast<f4*>(dst + d_o) = *reinterpret_cast<const f4*>(src + so);
    }
}

// strides in BYTES here (the C-ABI takes elements)
int copy_pixels_launch(const void* src, void* dst, int B, int H, int W, int64_t row_bytes, int64_t sb, int64_t sy,
                       int64_t sx, int64_t db, int64_t dy, int64_t dx, hipStream_t s) {
    const int nq = (int)(row_bytes / 16);
    const int64_t total = (int64_t)B * H * W * nq;
    const int64_t want = (total + 4 * 256 - 1) / (4 * 256);
    const dim3 grid((unsigned)(want < 1 ? 1 : (want > 16384 ? 16384 : want)));
    hipLaunchKernelGGL(copy_pixels_kernel, grid, dim3(256), 0, s, (const char*)src, (char*)dst, H, W, nq, total, sb, sy,
                       sx, db, dy, dx);
    return check_launch("copy_pixels_kernel");
}

}  // namespace qpwc
