// extern "C" boundary of libqpwc_hip.so (declared in include/qpwc.h).
// Validates arguments, then enqueues on the caller's stream.  No allocation,
// no synchronisation, no global state besides a thread-local error string.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace qpwc {

static thread_local char g_err[512] = "";
thread_local const char* g_dry_kernel = nullptr;
thread_local bool g_dry_run = false;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return QPWC_E_LAUNCH;
    }
    return QPWC_OK;
}

int cost_volume_launch(const void* prv, const void* nxt, const void* flo, void* out, int B, int H,
                       int W, int C, int r, int layout, int dtype, int64_t ops, float slope,
                       bool fuse, bool pad84, hipStream_t s);
int warp_launch(const void* img, const void* flo, void* out, int B, int H, int W, int C,
                int flo_bcast_mask, int layout, int dtype, int mode, hipStream_t s);
int epe_workspace_floats();
int epe_launch(const float* a, const float* b, float* out, float* ws, int B, int H, int W,
               int layout, hipStream_t s);

int epe_multi_workspace_floats();
int epe_multi_launch(const void* const* a, const void* const* b, const int64_t* npix, const int64_t* plane,
                     const int* pred_dtype, int n_levels, float* out, float* ws, hipStream_t s);
int layout_transpose_launch(const void* in, void* out, int B, int H, int W, int C, int to_layout, int dtype,
                            hipStream_t s);
int copy_pixels_launch(const void* src, void* dst, int B, int H, int W, int64_t row_bytes, int64_t sb, int64_t sy,
                       int64_t sx, int64_t db, int64_t dy, int64_t dx, hipStream_t s);
int dwconv3x3_launch(const void* const* srcs, const int* chans, const int64_t* strides, int n_src,
                     int act, const void* weight, void* out, int B, int H, int W, int dtype,
                     hipStream_t s);
int cost_volume_to_flow_launch(const void* cvol, float* flow, int B, int H, int W, int D, int64_t pix_stride,
                               int layout, int dtype, hipStream_t s);
int sepconv3x3_f16_launch(const void* const* srcs, const int* chans, const int64_t* strides, int n_src, int act,
                          const void* dw, const void* pw, const void* bias, void* out, int B, int H, int W,
                          int F, hipStream_t s);
int sepconv3x3_x3_launch(const void* const* srcs, const int* chans, const int64_t* strides, int n_src, int act,
                         const void* dw, const void* pw3, const void* bias, void* out, int B, int H, int W, int F,
                         hipStream_t s);
int sepconv3x3_launch(const void* const* srcs, const int* chans, const int64_t* strides, int n_src,
                      int act, const void* dw, const void* pw, const void* bias, void* out, int B, int H,
                      int W, int F, hipStream_t s);
int flow_head_launch(const void* z, const void* params, void* out, int B, int H, int W, float scale,
                     int dtype, int out_layout, hipStream_t s);
int flow_head_param_floats();
int pointwise_bias_launch(const void* y, const void* w, const void* bias, void* out, int64_t M, int C, int cpad, int F,
                          hipStream_t s);
int flow_head_up_launch(const void* z, const void* params, void* out, void* out_up, void* out_up_f32, int B, int H, int W,
                        float scale, float up_scale, int dtype, hipStream_t s);
int optflow_tail_launch(const void* z2, const void* dw3, const void* pw3, const void* b3, const void* dw4,
                        const void* pw4, const void* b4, const void* head, void* out, int B, int H, int W,
                        float scale, int act_in, int out_layout, hipStream_t s);
int upsample2x_flow_launch(const void* in, void* out, int B, int h, int w, float scale, int dtype,
                           int in_layout, int out_layout, hipStream_t s);
int bias_mish_pad_launch(const void* src, const void* bias, void* dst, int B, int H, int W, int C,
                         int pad_h, int pad_w, int64_t dst_pixel_stride, int dtype, hipStream_t s);
int split_frames_pad_launch(const void* in, void* out, int B, int H, int W, int pad_h, int pad_w, int dtype,
                            hipStream_t s);
int invert_flow_launch(const void* flow, void* out, int B, int H, int W, int layout, int dtype,
                       hipStream_t s);
int occlusion_launch(const void* flow, void* out, int B, int H, int W, int layout, int dtype,
                     hipStream_t s);
int upconv4x4s2_mish_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W, int C,
                            int F, int out_pixel_stride, hipStream_t s, const void* skip = nullptr, int64_t skip_bs = 0,
                            int64_t skip_rs = 0, int64_t skip_ps = 0);
int upconv4x4s2_mish_f16_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W, int C,
                            int F, int out_pixel_stride, hipStream_t s, const void* skip = nullptr, int64_t skip_bs = 0,
                            int64_t skip_rs = 0, int64_t skip_ps = 0);
int conv3x3s2_mish_any_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                              int CI, hipStream_t s);
int conv3x3s2_mish_f16_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                              int CI, hipStream_t s);
int conv3x3_mish_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                        int C, int pad_h, int pad_w, hipStream_t s);
int conv3x3_mish_x3_launch(const void* x, const void* w3, const void* bias, void* out, int B, int H, int W, int C,
                           int pad_h, int pad_w, hipStream_t s);
int split_bf16x3_launch(const void* src, void* out, int64_t n, hipStream_t s);
int conv3x3s2_mish_x3_launch(const void* x, const void* w3, const void* bias, void* out, int B, int H, int W, int CI,
                             hipStream_t s);
int upconv4x4s2_mish_x3_launch(const void* x, const void* w3, const void* bias, void* out, int B, int H, int W, int C,
                               int F, int out_pixel_stride, hipStream_t s);
int conv3x3_mish_f16_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                        int C, int pad_h, int pad_w, hipStream_t s);
int first_conv_mish_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                           int layout, int dtype, hipStream_t s);
int conv3x3s2_mish_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                          hipStream_t s);
int bias_mish_launch(void* x, const void* bias, int64_t n_pixels, int C, int dtype, hipStream_t s);

// 16-byte-per-lane streaming copy: the box's achievable HBM ceiling (read + write) for bench.py's roofline
// block.  tools/micro/copybench.hip on MI355X: one float4 per thread over a one-shot grid with non-temporal
// loads and stores 6.5 TB/s (plain 6.1; 4 float4 per thread over a grid-strided loop 5.5-6.2; hipMemcpyAsync 5.3).
typedef unsigned int copy_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void device_copy_kernel(const copy_u32x4* __restrict__ src,
                                                         copy_u32x4* __restrict__ dst, int64_t n16) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static size_t esize(int dtype) { return dtype == QPWC_F16 ? 2 : 4; }

static bool overlaps(const void* a, size_t na, const void* b, size_t nb) {
    const uintptr_t a0 = (uintptr_t)a, b0 = (uintptr_t)b;
    return a0 < b0 + nb && b0 < a0 + na;
}

static int check_common(int B, int H, int W, int C, int layout, int dtype) {
    if (layout != QPWC_NHWC && layout != QPWC_NCHW)
        return fail(QPWC_E_LAYOUT, "Unsupported data format : %d", layout);
    if (dtype != QPWC_F32 && dtype != QPWC_F16) return fail(QPWC_E_DTYPE, "unsupported dtype %d", dtype);
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0)
        return fail(QPWC_E_SHAPE, "non-positive extent B=%d H=%d W=%d C=%d", B, H, W, C);
    return QPWC_OK;
}

static int cost_volume_checked(const void* prv, const void* nxt, const void* flo, void* out, int B,
                               int H, int W, int C, int r, int layout, int dtype, float slope,
                               int64_t ops, int64_t off, bool strided, bool fuse, void* stream) {
    if (!prv || !nxt || !out || (fuse && !flo)) return fail(QPWC_E_NULL, "null pointer argument");
    int rc = check_common(B, H, W, C, layout, dtype);
    if (rc != QPWC_OK) return rc;
    if (r < 0 || r > 16) return fail(QPWC_E_RANGE, "search_range %d outside [0,16]", r);
    if (fuse && (H < 2 || W < 2))
        return fail(QPWC_E_SHAPE, "warp needs H,W >= 2 (got %dx%d)", H, W);
    const int64_t DD = (int64_t)(2 * r + 1) * (2 * r + 1);
    if (!strided) {
        ops = DD;
        off = 0;
    } else if (off < 0 || ops < off + DD) {
        return fail(QPWC_E_STRIDE, "out_pixel_stride %lld cannot hold %lld channels at offset %lld",
                    (long long)ops, (long long)DD, (long long)off);
    }
    const size_t es = esize(dtype);
    if ((uintptr_t)prv % es || (uintptr_t)nxt % es || (uintptr_t)out % es ||
        (fuse && (uintptr_t)flo % 4))
        return fail(QPWC_E_ALIGN, "pointer not aligned to its element size");
    const size_t n_in = (size_t)B * H * W * C * es;
    const size_t n_out = ((size_t)B * H * W * ops) * es;
    if (overlaps(out, n_out, prv, n_in) || overlaps(out, n_out, nxt, n_in) ||
        (fuse && overlaps(out, n_out, flo, (size_t)B * H * W * 2 * 4)))
        return fail(QPWC_E_ALIAS, "out overlaps an input");
    char* o = (char*)out + (size_t)off * es;
    // 84-float pixels holding the 81 channels at offset 0: the 3 pad channels are written as zeros
    const bool pad84 = strided && r == 4 && ops == 84 && off == 0;
    return cost_volume_launch(prv, nxt, flo, o, B, H, W, C, r, layout, dtype, ops, slope, fuse, pad84,
                              (hipStream_t)stream);
}

}  // namespace qpwc

using namespace qpwc;

extern "C" {

int qpwc_version(void) { return QPWC_VERSION; }

int qpwc_layout_transpose_fwd(const void* in, void* out, int B, int H, int W, int C, int to_layout, int dtype,
                              void* stream) {
    if (!in || !out) return fail(QPWC_E_NULL, "null pointer argument");
    const int rc = check_common(B, H, W, C, to_layout, dtype);
    if (rc) return rc;
    const size_t es = esize(dtype), n = (size_t)B * H * W * C * es;
    if ((uintptr_t)in % es || (uintptr_t)out % es) return fail(QPWC_E_ALIGN, "pointer not aligned to its element size");
    if (overlaps(out, n, in, n)) return fail(QPWC_E_ALIAS, "out overlaps in");
    return layout_transpose_launch(in, out, B, H, W, C, to_layout, dtype, (hipStream_t)stream);
}

int qpwc_copy_pixels_fwd(const void* src, void* dst, int B, int H, int W, int C, const int64_t* src_strides,
                         const int64_t* dst_strides, int dtype, void* stream) {
    if (!src || !dst || !src_strides || !dst_strides) return fail(QPWC_E_NULL, "null pointer argument");
    if (dtype != QPWC_F32 && dtype != QPWC_F16) return fail(QPWC_E_DTYPE, "unsupported dtype %d", dtype);
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return fail(QPWC_E_SHAPE, "non-positive extent B=%d H=%d W=%d C=%d", B, H, W, C);
    const int64_t es = dtype == QPWC_F32 ? 4 : 2;
    if ((C * es) % 16) return fail(QPWC_E_SHAPE, "a pixel's %d channels must be whole 16-byte chunks", C);
    for (int i = 0; i < 3; ++i) {
        if (src_strides[i] < (i == 2 ? C : 0) || dst_strides[i] < (i == 2 ? C : 0))
            return fail(QPWC_E_STRIDE, "stride %d smaller than what it spans", i);
        if ((src_strides[i] * es) % 16 || (dst_strides[i] * es) % 16)
            return fail(QPWC_E_STRIDE, "strides must be multiples of 16 bytes");
    }
    if (((uintptr_t)src | (uintptr_t)dst) % 16) return fail(QPWC_E_ALIGN, "pointers must be 16-byte aligned");
    const size_t sext = (size_t)(((B - 1) * src_strides[0] + (H - 1) * src_strides[1] + (W - 1) * src_strides[2] + C) * es);
    const size_t dext = (size_t)(((B - 1) * dst_strides[0] + (H - 1) * dst_strides[1] + (W - 1) * dst_strides[2] + C) * es);
    if (overlaps(dst, dext, src, sext)) return fail(QPWC_E_ALIAS, "dst overlaps src");
    return copy_pixels_launch(src, dst, B, H, W, C * es, src_strides[0] * es, src_strides[1] * es, src_strides[2] * es,
                              dst_strides[0] * es, dst_strides[1] * es, dst_strides[2] * es, (hipStream_t)stream);
}

// One wave that stamps (shader-clock counter, 100 MHz real-time counter) pairs while it sleeps: the sustained clock of
// the chip under whatever runs beside it is delta(s_memtime) / delta(s_memrealtime) x 100 MHz
// (MI355X_MICROARCH.md, constants table).  Bounded: n_samples stamps, then the wave ends.
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* __restrict__ out, int n_samples,
                                                         int sleeps_per_sample) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < n_samples; ++i) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        const unsigned long long r = __builtin_amdgcn_s_memrealtime();
        out[2 * i] = t;
        out[2 * i + 1] = r;
        for (int k = 0; k < sleeps_per_sample; ++k) __builtin_amdgcn_s_sleep(127);
    }
}

int qpwc_clock_probe(void* out_pairs, int n_samples, int sleeps_per_sample, void* stream) {
    if (!out_pairs) return fail(QPWC_E_NULL, "null pointer argument");
    if (n_samples <= 0 || n_samples > (1 << 20) || sleeps_per_sample < 0 || sleeps_per_sample > 4096)
        return fail(QPWC_E_SHAPE, "clock probe: 1..2^20 samples of 0..4096 sleeps");
    if ((uintptr_t)out_pairs % 8) return fail(QPWC_E_ALIGN, "clock probe buffer must be 8-byte aligned");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out_pairs,
                       n_samples, sleeps_per_sample);
    return check_launch("clock_probe_kernel");
}

int qpwc_device_copy(const void* src, void* dst, int64_t bytes, void* stream) {
    if (!src || !dst) return fail(QPWC_E_NULL, "null pointer argument");
    if (bytes <= 0 || bytes % 16) return fail(QPWC_E_SHAPE, "bytes must be a positive multiple of 16");
    if (((uintptr_t)src | (uintptr_t)dst) % 16) return fail(QPWC_E_ALIGN, "pointers must be 16-byte aligned");
    if (overlaps(dst, (size_t)bytes, src, (size_t)bytes)) return fail(QPWC_E_ALIAS, "dst overlaps src");
    const int64_t n16 = bytes / 16;
    const int64_t want = (n16 + 255) / 256;
    if (want > INT32_MAX) return fail(QPWC_E_SHAPE, "copy of %lld bytes needs too many workgroups", (long long)bytes);
    const unsigned grid = (unsigned)want;
    hipLaunchKernelGGL(device_copy_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const copy_u32x4*)src, (copy_u32x4*)dst, n16);
    return check_launch("device_copy_kernel");
}

const char* qpwc_last_error(void) { return g_err; }

const char* qpwc_strerror(int code) {
    switch (code) {
        case QPWC_OK: return "ok";
        case QPWC_E_NULL: return "null pointer";
        case QPWC_E_LAYOUT: return "unsupported data format";
        case QPWC_E_DTYPE: return "unsupported dtype";
        case QPWC_E_SHAPE: return "bad shape";
        case QPWC_E_RANGE: return "bad search range";
        case QPWC_E_MODE: return "unknown warp mode";
        case QPWC_E_ALIAS: return "output aliases an input";
        case QPWC_E_LAUNCH: return "kernel launch failed";
        case QPWC_E_ALIGN: return "misaligned pointer";
        case QPWC_E_STRIDE: return "bad output stride/offset";
        case QPWC_E_NODEVICE: return "no HIP device";
        default: return "unknown error";
    }
}

int qpwc_cost_volume_fwd(const void* prv, const void* nxt, void* out, int B, int H, int W, int C,
                         int search_range, int layout, int dtype, float lrelu_slope, void* stream) {
    return cost_volume_checked(prv, nxt, nullptr, out, B, H, W, C, search_range, layout, dtype,
                               lrelu_slope, 0, 0, false, false, stream);
}

int qpwc_cost_volume_fwd_strided(const void* prv, const void* nxt, void* out, int B, int H, int W,
                                 int C, int search_range, int dtype, float lrelu_slope,
                                 int64_t out_pixel_stride, int64_t out_channel_offset, void* stream) {
    return cost_volume_checked(prv, nxt, nullptr, out, B, H, W, C, search_range, QPWC_NHWC, dtype,
                               lrelu_slope, out_pixel_stride, out_channel_offset, true, false,
                               stream);
}

int qpwc_warp_cost_volume_fwd(const void* prv, const void* nxt, const void* flo, void* out, int B,
                              int H, int W, int C, int search_range, int dtype, float lrelu_slope,
                              int64_t out_pixel_stride, int64_t out_channel_offset, void* stream) {
    return cost_volume_checked(prv, nxt, flo, out, B, H, W, C, search_range, QPWC_NHWC, dtype,
                               lrelu_slope, out_pixel_stride, out_channel_offset, true, true, stream);
}

const char* qpwc_cost_volume_kernel(int B, int H, int W, int C, int search_range, int layout, int dtype,
                                    int64_t out_pixel_stride, int fused) {
    if (check_common(B, H, W, C, layout, dtype) != QPWC_OK || search_range < 0 || search_range > 16 ||
        (fused && (H < 2 || W < 2 || layout != QPWC_NHWC)))
        return "";
    const int DD = (2 * search_range + 1) * (2 * search_range + 1);
    if (out_pixel_stride <= 0) out_pixel_stride = DD;
    // the launchers only look at the alignment of their operands: a 4096-aligned placeholder stands for
    // "aligned as torch allocates"; nothing is dereferenced or enqueued while g_dry_run is set
    const void* p = reinterpret_cast<const void*>((uintptr_t)4096);
    g_dry_kernel = "";
    g_dry_run = true;
    const bool pad84 = layout == QPWC_NHWC && out_pixel_stride == 84 && DD == 81;
    const int rc = cost_volume_launch(p, p, fused ? p : nullptr, const_cast<void*>(p), B, H, W, C, search_range, layout,
                                      dtype, out_pixel_stride, 0.1f, fused != 0, pad84, nullptr);
    g_dry_run = false;
    return rc == QPWC_OK ? g_dry_kernel : "";
}

int qpwc_warp_fwd(const void* img, const void* flo, void* out, int B, int H, int W, int C,
                  int flo_bcast_mask, int layout, int dtype, int mode, void* stream) {
    if (!img || !flo || !out) return fail(QPWC_E_NULL, "null pointer argument");
    int rc = check_common(B, H, W, C, layout, dtype);
    if (rc != QPWC_OK) return rc;
    if (mode != QPWC_WARP_CLAMP && mode != QPWC_WARP_TFWARP)
        return fail(QPWC_E_MODE, "unknown warp mode %d", mode);
    if (mode == QPWC_WARP_CLAMP && (H < 2 || W < 2))
        return fail(QPWC_E_SHAPE, "Grid must be at least 2x2 (got %dx%d)", H, W);
    if (flo_bcast_mask & ~(QPWC_BCAST_B | QPWC_BCAST_H | QPWC_BCAST_W))
        return fail(QPWC_E_SHAPE, "bad flo_bcast_mask %d", flo_bcast_mask);
    const size_t es = esize(dtype);
    if ((uintptr_t)img % es || (uintptr_t)out % es || (uintptr_t)flo % 4)
        return fail(QPWC_E_ALIGN, "pointer not aligned to its element size");
    const size_t n = (size_t)B * H * W * C * es;
    const size_t nf = (size_t)((flo_bcast_mask & QPWC_BCAST_B) ? 1 : B) *
                      ((flo_bcast_mask & QPWC_BCAST_H) ? 1 : H) *
                      ((flo_bcast_mask & QPWC_BCAST_W) ? 1 : W) * 2 * 4;
    if (overlaps(out, n, img, n) || overlaps(out, n, flo, nf))
        return fail(QPWC_E_ALIAS, "out overlaps an input");
    return warp_launch(img, flo, out, B, H, W, C, flo_bcast_mask, layout, dtype, mode,
                       (hipStream_t)stream);
}

int qpwc_epe_workspace_floats(void) { return epe_workspace_floats(); }

int qpwc_epe_fwd(const void* y_true, const void* y_pred, void* out_mean, void* workspace, int B,
                 int H, int W, int layout, void* stream) {
    if (!y_true || !y_pred || !out_mean || !workspace) return fail(QPWC_E_NULL, "null pointer argument");
    int rc = check_common(B, H, W, 2, layout, QPWC_F32);
    if (rc != QPWC_OK) return rc;
    if ((uintptr_t)y_true % 8 || (uintptr_t)y_pred % 8 || (uintptr_t)out_mean % 4 ||
        (uintptr_t)workspace % 4)
        return fail(QPWC_E_ALIGN, "flow pointers must be 8-byte aligned");
    return epe_launch((const float*)y_true, (const float*)y_pred, (float*)out_mean,
                      (float*)workspace, B, H, W, layout, (hipStream_t)stream);
}

int qpwc_dwconv3x3_fwd(const void* const* src, const int* src_channels,
                       const int64_t* src_pixel_stride, int n_src, int mish_on_load,
                       const void* weight, void* out, int B, int H, int W, int dtype, void* stream) {
    if (!src || !src_channels || !src_pixel_stride || !weight || !out)
        return fail(QPWC_E_NULL, "null pointer argument");
    if (dtype != QPWC_F32 && dtype != QPWC_F16) return fail(QPWC_E_DTYPE, "unsupported dtype %d", dtype);
    const size_t es = esize(dtype);
    if (n_src < 1 || n_src > 3) return fail(QPWC_E_SHAPE, "n_src %d outside [1,3]", n_src);
    if (B <= 0 || H <= 0 || W <= 0) return fail(QPWC_E_SHAPE, "non-positive extent B=%d H=%d W=%d", B, H, W);
    int64_t C = 0;
    for (int i = 0; i < n_src; ++i) {
        if (!src[i]) return fail(QPWC_E_NULL, "null source %d", i);
        if (src_channels[i] <= 0 || src_pixel_stride[i] < src_channels[i])
            return fail(QPWC_E_STRIDE, "source %d: %d channels at pixel stride %lld", i,
                        src_channels[i], (long long)src_pixel_stride[i]);
        if ((uintptr_t)src[i] % es) return fail(QPWC_E_ALIGN, "source %d not element aligned", i);
        C += src_channels[i];
    }
    if ((uintptr_t)out % es || (uintptr_t)weight % 4) return fail(QPWC_E_ALIGN, "pointer not element aligned");
    const size_t n_out = (size_t)B * H * W * C * es;
    for (int i = 0; i < n_src; ++i)
        if (overlaps(out, n_out, src[i], (((size_t)B * H * W - 1) * src_pixel_stride[i] + src_channels[i]) * es))
            return fail(QPWC_E_ALIAS, "out overlaps source %d", i);
    if ((int64_t)W * C > INT32_MAX) return fail(QPWC_E_SHAPE, "row too long");
    return dwconv3x3_launch(src, src_channels, src_pixel_stride, n_src, mish_on_load, weight, out, B,
                            H, W, dtype, (hipStream_t)stream);
}

int qpwc_sepconv3x3_fwd(const void* const* src, const int* src_channels,
                        const int64_t* src_pixel_stride, int n_src, int mish_flags, const void* dw,
                        const void* pw, const void* bias, void* out, int B, int H, int W, int F,
                        void* stream) {
    if (!src || !src_channels || !src_pixel_stride || !dw || !pw || !bias || !out)
        return fail(QPWC_E_NULL, "null pointer argument");
    if (n_src < 1 || n_src > 3) return fail(QPWC_E_SHAPE, "n_src %d outside [1,3]", n_src);
    if (B <= 0 || H <= 0 || W <= 0) return fail(QPWC_E_SHAPE, "non-positive extent B=%d H=%d W=%d", B, H, W);
    if (F != 16 && F != 32 && F != 64 && F != 128) return fail(QPWC_E_SHAPE, "F=%d not in {16,32,64,128}", F);
    for (int i = 0; i < n_src; ++i) {
        if (!src[i]) return fail(QPWC_E_NULL, "null source %d", i);
        if (src_channels[i] <= 0 || src_pixel_stride[i] < src_channels[i])
            return fail(QPWC_E_STRIDE, "source %d: %d channels at pixel stride %lld", i,
                        src_channels[i], (long long)src_pixel_stride[i]);
        if ((uintptr_t)src[i] % 4) return fail(QPWC_E_ALIGN, "source %d not 4-byte aligned", i);
        if (overlaps(out, (size_t)B * H * W * F * 4, src[i],
                     (((size_t)B * H * W - 1) * src_pixel_stride[i] + src_channels[i]) * 4))
            return fail(QPWC_E_ALIAS, "out overlaps source %d", i);
    }
    if ((uintptr_t)out % 16 || (uintptr_t)pw % 16 || (uintptr_t)bias % 16 || (uintptr_t)dw % 4)
        return fail(QPWC_E_ALIGN, "out, pw, bias must be 16-byte aligned");
    if (mish_flags < 0 || mish_flags > 3) return fail(QPWC_E_SHAPE, "mish_flags %d outside [0,3]", mish_flags);
    return sepconv3x3_launch(src, src_channels, src_pixel_stride, n_src, mish_flags, dw, pw, bias, out, B,
                             H, W, F, (hipStream_t)stream);
}

int qpwc_sepconv3x3_x3_fwd(const void* const* src, const int* src_channels, const int64_t* src_pixel_stride,
                           int n_src, int mish_flags, const void* dw, const void* pw3, const void* bias, void* out,
                           int B, int H, int W, int F, void* stream) {
    if (!src || !src_channels || !src_pixel_stride || !dw || !pw3 || !bias || !out)
        return fail(QPWC_E_NULL, "null pointer argument");
    if (n_src < 1 || n_src > 3) return fail(QPWC_E_SHAPE, "n_src %d outside [1,3]", n_src);
    if (B <= 0 || H <= 0 || W <= 0) return fail(QPWC_E_SHAPE, "non-positive extent B=%d H=%d W=%d", B, H, W);
    if (F != 16 && F != 32 && F != 64 && F != 128) return fail(QPWC_E_SHAPE, "F=%d not in {16,32,64,128}", F);
    for (int i = 0; i < n_src; ++i) {
        if (!src[i]) return fail(QPWC_E_NULL, "null source %d", i);
        if (src_channels[i] <= 0 || src_pixel_stride[i] < src_channels[i])
            return fail(QPWC_E_STRIDE, "source %d: %d channels at pixel stride %lld", i, src_channels[i],
                        (long long)src_pixel_stride[i]);
        if ((uintptr_t)src[i] % 4) return fail(QPWC_E_ALIGN, "source %d not 4-byte aligned", i);
        if (overlaps(out, (size_t)B * H * W * F * 4, src[i],
                     (((size_t)B * H * W - 1) * src_pixel_stride[i] + src_channels[i]) * 4))
            return fail(QPWC_E_ALIAS, "out overlaps source %d", i);
    }
    if ((uintptr_t)out % 16 || (uintptr_t)pw3 % 16 || (uintptr_t)bias % 16 || (uintptr_t)dw % 4)
        return fail(QPWC_E_ALIGN, "out, pw3, bias must be 16-byte aligned");
    if (mish_flags < 0 || mish_flags > 3) return fail(QPWC_E_SHAPE, "mish_flags %d outside [0,3]", mish_flags);
    return sepconv3x3_x3_launch(src, src_channels, src_pixel_stride, n_src, mish_flags, dw, pw3, bias, out, B, H, W, F,
                                (hipStream_t)stream);
}

int qpwc_sepconv3x3_f16_fwd(const void* const* src, const int* src_channels, const int64_t* src_pixel_stride,
                            int n_src, int mish_flags, const void* dw, const void* pw, const void* bias,
                            void* out, int B, int H, int W, int F, void* stream) {
    if (!src || !src_channels || !src_pixel_stride || !dw || !pw || !bias || !out)
        return fail(QPWC_E_NULL, "null pointer argument");
    if (n_src < 1 || n_src > 3) return fail(QPWC_E_SHAPE, "n_src %d outside [1,3]", n_src);
    if (B <= 0 || H <= 0 || W <= 0) return fail(QPWC_E_SHAPE, "non-positive extent B=%d H=%d W=%d", B, H, W);
    if (F != 16 && F != 32 && F != 64 && F != 128) return fail(QPWC_E_SHAPE, "F=%d not in {16,32,64,128}", F);
    for (int i = 0; i < n_src; ++i) {
        if (!src[i]) return fail(QPWC_E_NULL, "null source %d", i);
        if (src_channels[i] <= 0 || src_pixel_stride[i] < src_channels[i])
            return fail(QPWC_E_STRIDE, "source %d: %d channels at pixel stride %lld", i, src_channels[i],
                        (long long)src_pixel_stride[i]);
        // 8-byte loads of 4 channels: every source but a short (< 4 channel) last one
        const bool tail = i + 1 == n_src && src_channels[i] < 4;
        if (!tail && (src_channels[i] % 4 || src_pixel_stride[i] % 4 || (uintptr_t)src[i] % 8))
            return fail(QPWC_E_ALIGN, "source %d: %d channels at stride %lld must be multiples of 4, 8-byte aligned",
                        i, src_channels[i], (long long)src_pixel_stride[i]);
        if ((uintptr_t)src[i] % 2) return fail(QPWC_E_ALIGN, "source %d not element aligned", i);
        if (overlaps(out, (size_t)B * H * W * F * 2, src[i],
                     (((size_t)B * H * W - 1) * src_pixel_stride[i] + src_channels[i]) * 2))
            return fail(QPWC_E_ALIAS, "out overlaps source %d", i);
        if ((int64_t)H * W * src_pixel_stride[i] > INT32_MAX) return fail(QPWC_E_SHAPE, "image too large");
    }
    if ((uintptr_t)out % 16 || (uintptr_t)pw % 16 || (uintptr_t)bias % 16 || (uintptr_t)dw % 4)
        return fail(QPWC_E_ALIGN, "out, pw, bias must be 16-byte aligned");
    if (mish_flags < 0 || mish_flags > 3) return fail(QPWC_E_SHAPE, "mish_flags %d outside [0,3]", mish_flags);
    return sepconv3x3_f16_launch(src, src_channels, src_pixel_stride, n_src, mish_flags, dw, pw, bias, out, B,
                                 H, W, F, (hipStream_t)stream);
}

int qpwc_pointwise_bias_fwd(const void* y, const void* weight, const void* bias, void* out, int64_t M, int C, int F,
                            void* stream) {
    if (!y || !weight || !bias || !out) return fail(QPWC_E_NULL, "null pointer argument");
    if (M <= 0 || C <= 0) return fail(QPWC_E_SHAPE, "non-positive extent M=%lld C=%d", (long long)M, C);
    if (F != 16 && F != 32 && F != 64 && F != 128 && F != 256) return fail(QPWC_E_SHAPE, "F=%d not in {16,32,64,128,256}", F);
    if (M * (int64_t)(C > F ? C : F) >= INT32_MAX) return fail(QPWC_E_SHAPE, "M x max(C, F) must stay below 2^31");
    if ((uintptr_t)y % 16 || (uintptr_t)weight % 16 || (uintptr_t)bias % 16 || (uintptr_t)out % 16)
        return fail(QPWC_E_ALIGN, "y, weight, bias, out must be 16-byte aligned");
    const int cpad = (C + 31) / 32 * 32;
    if (overlaps(out, (size_t)M * F * 4, y, (size_t)M * C * 4)) return fail(QPWC_E_ALIAS, "out overlaps y");
    return pointwise_bias_launch(y, weight, bias, out, M, C, cpad, F, (hipStream_t)stream);
}

int qpwc_flow_head_param_floats(void) { return flow_head_param_floats(); }

int qpwc_flow_head_up_fwd(const void* z, const void* params, void* out, void* out_up, void* out_up_f32, int B, int H, int W,
                          float scale, float up_scale, int dtype, void* stream) {
    if (!z || !params || !out || !out_up) return fail(QPWC_E_NULL, "null pointer argument");
    if (dtype != QPWC_F32 && dtype != QPWC_F16) return fail(QPWC_E_DTYPE, "unsupported dtype %d", dtype);
    const size_t es = esize(dtype);
    if (B <= 0 || H <= 0 || W <= 0) return fail(QPWC_E_SHAPE, "non-positive extent B=%d H=%d W=%d", B, H, W);
    if ((int64_t)B * 4 * H * W * 2 >= INT32_MAX) return fail(QPWC_E_SHAPE, "upsampled flow too large for 32-bit pixel indices");
    if ((uintptr_t)z % (4 * es) || (uintptr_t)out % (2 * es) || (uintptr_t)out_up % (4 * es) || (uintptr_t)params % 4)
        return fail(QPWC_E_ALIGN, "z and out_up must be aligned to 4 elements, out to 2");
    const size_t nz = (size_t)B * H * W * 16 * es, no = (size_t)B * H * W * 2 * es, nu = 4 * no;
    if (overlaps(out, no, z, nz) || overlaps(out_up, nu, z, nz) || overlaps(out_up, nu, out, no))
        return fail(QPWC_E_ALIAS, "out / out_up overlap z or each other");
    if (out_up_f32) {
        if (dtype != QPWC_F16) return fail(QPWC_E_DTYPE, "out_up_f32 is the fp32 copy of an fp16 out_up: pass NULL for fp32 storage");
        if ((uintptr_t)out_up_f32 % 16) return fail(QPWC_E_ALIGN, "out_up_f32 must be 16-byte aligned");
        if (overlaps(out_up_f32, (size_t)B * 4 * H * W * 2 * 4, z, nz) || overlaps(out_up_f32, (size_t)B * 4 * H * W * 2 * 4, out, no) ||
            overlaps(out_up_f32, (size_t)B * 4 * H * W * 2 * 4, out_up, nu))
            return fail(QPWC_E_ALIAS, "out_up_f32 overlaps another operand");
    }
    return flow_head_up_launch(z, params, out, out_up, out_up_f32, B, H, W, scale, up_scale, dtype, (hipStream_t)stream);
}

int qpwc_flow_head_fwd(const void* z, const void* params, void* out, int B, int H, int W, float scale,
                       int dtype, int out_layout, void* stream) {
    if (!z || !params || !out) return fail(QPWC_E_NULL, "null pointer argument");
    if (out_layout != QPWC_NHWC && out_layout != QPWC_NCHW)
        return fail(QPWC_E_LAYOUT, "Unsupported data format : %d", out_layout);
    if (dtype != QPWC_F32 && dtype != QPWC_F16) return fail(QPWC_E_DTYPE, "unsupported dtype %d", dtype);
    const size_t es = esize(dtype);
    if (B <= 0 || H <= 0 || W <= 0) return fail(QPWC_E_SHAPE, "non-positive extent B=%d H=%d W=%d", B, H, W);
    if ((uintptr_t)z % (4 * es) || (uintptr_t)out % (2 * es) || (uintptr_t)params % 4)
        return fail(QPWC_E_ALIGN, "z must be aligned to 4 elements, out to 2");
    if (overlaps(out, (size_t)B * H * W * 2 * es, z, (size_t)B * H * W * 16 * es))
        return fail(QPWC_E_ALIAS, "out overlaps z");
    return flow_head_launch(z, params, out, B, H, W, scale, dtype, out_layout, (hipStream_t)stream);
}

int qpwc_optflow_tail_fwd(const void* z2, const void* dw3, const void* pw3, const void* b3, const void* dw4,
                          const void* pw4, const void* b4, const void* head_params, void* out, int B, int H, int W,
                          float scale, int mish_on_load, int out_layout, void* stream) {
    if (!z2 || !dw3 || !pw3 || !b3 || !dw4 || !pw4 || !b4 || !head_params || !out)
        return fail(QPWC_E_NULL, "null pointer argument");
    if (out_layout != QPWC_NHWC && out_layout != QPWC_NCHW)
        return fail(QPWC_E_LAYOUT, "Unsupported data format : %d", out_layout);
    if (B <= 0 || H <= 0 || W <= 0) return fail(QPWC_E_SHAPE, "non-positive extent B=%d H=%d W=%d", B, H, W);
    if ((uintptr_t)z2 % 16 || (uintptr_t)pw3 % 16 || (uintptr_t)pw4 % 16 || (uintptr_t)b3 % 16 || (uintptr_t)b4 % 16 ||
        (uintptr_t)head_params % 16 || (uintptr_t)dw3 % 4 || (uintptr_t)dw4 % 4 || (uintptr_t)out % 8)
        return fail(QPWC_E_ALIGN, "z2, pw3, pw4, b3, b4, head_params must be 16-byte aligned, out 8-byte");
    if (overlaps(out, (size_t)B * H * W * 2 * 4, z2, (size_t)B * H * W * 64 * 4)) return fail(QPWC_E_ALIAS, "out overlaps z2");
    return optflow_tail_launch(z2, dw3, pw3, b3, dw4, pw4, b4, head_params, out, B, H, W, scale, mish_on_load,
                               out_layout, (hipStream_t)stream);
}

int qpwc_bias_mish_fwd(void* x, const void* bias, int64_t n_pixels, int C, int dtype, void* stream) {
    if (!x) return fail(QPWC_E_NULL, "null pointer argument");
    if (dtype != QPWC_F32 && dtype != QPWC_F16) return fail(QPWC_E_DTYPE, "unsupported dtype %d", dtype);
    if (n_pixels <= 0 || C <= 0 || C % 4) return fail(QPWC_E_SHAPE, "need n_pixels > 0 and C %% 4 == 0 (C=%d)", C);
    if ((uintptr_t)x % (4 * esize(dtype)) || (uintptr_t)bias % 16)
        return fail(QPWC_E_ALIGN, "x must be aligned to 4 elements, bias to 16 bytes");
    return bias_mish_launch(x, bias, n_pixels, C, dtype, (hipStream_t)stream);
}

int qpwc_bias_mish_pad_fwd(const void* src, const void* bias, void* dst, int B, int H, int W, int C,
                           int pad_h, int pad_w, int64_t dst_pixel_stride, int dtype, void* stream) {
    if (!src || !dst) return fail(QPWC_E_NULL, "null pointer argument");
    if (dtype != QPWC_F32 && dtype != QPWC_F16) return fail(QPWC_E_DTYPE, "unsupported dtype %d", dtype);
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || pad_h < 0 || pad_w < 0)
        return fail(QPWC_E_SHAPE, "bad shape B=%d H=%d W=%d C=%d pad=%d,%d", B, H, W, C, pad_h, pad_w);
    if (dst_pixel_stride < C || dst_pixel_stride % 4)
        return fail(QPWC_E_STRIDE, "dst_pixel_stride %lld must be >= C and a multiple of 4",
                    (long long)dst_pixel_stride);
    const size_t es = esize(dtype);
    if ((uintptr_t)src % (4 * es) || (uintptr_t)dst % (4 * es) || (uintptr_t)bias % 16)
        return fail(QPWC_E_ALIGN, "src/dst must be aligned to 4 elements, bias to 16 bytes");
    if (overlaps(dst, (size_t)B * (H + pad_h) * (W + pad_w) * dst_pixel_stride * es, src,
                 (size_t)B * H * W * C * es))
        return fail(QPWC_E_ALIAS, "dst overlaps src");
    return bias_mish_pad_launch(src, bias, dst, B, H, W, C, pad_h, pad_w, dst_pixel_stride, dtype,
                                (hipStream_t)stream);
}

int qpwc_split_frames_pad_fwd(const void* in, void* out, int B, int H, int W, int pad_h, int pad_w,
                              int dtype, void* stream) {
    if (!in || !out) return fail(QPWC_E_NULL, "null pointer argument");
    if (dtype != QPWC_F32 && dtype != QPWC_F16) return fail(QPWC_E_DTYPE, "unsupported dtype %d", dtype);
    if (B <= 0 || H <= 0 || W <= 0 || pad_h < 0 || pad_w < 0)
        return fail(QPWC_E_SHAPE, "bad shape B=%d H=%d W=%d pad=%d,%d", B, H, W, pad_h, pad_w);
    const size_t es = esize(dtype);
    if ((uintptr_t)in % es || (uintptr_t)out % es) return fail(QPWC_E_ALIGN, "pointer not element aligned");
    if (overlaps(out, (size_t)2 * B * (H + pad_h) * (W + pad_w) * 3 * es, in, (size_t)B * H * W * 6 * es))
        return fail(QPWC_E_ALIAS, "out overlaps in");
    return split_frames_pad_launch(in, out, B, H, W, pad_h, pad_w, dtype, (hipStream_t)stream);
}

int qpwc_upsample2x_flow_fwd(const void* in, void* out, int B, int h, int w, float scale, int dtype,
                             int in_layout, int out_layout, void* stream) {
    if (!in || !out) return fail(QPWC_E_NULL, "null pointer argument");
    if ((in_layout != QPWC_NHWC && in_layout != QPWC_NCHW) || (out_layout != QPWC_NHWC && out_layout != QPWC_NCHW))
        return fail(QPWC_E_LAYOUT, "Unsupported data format : %d / %d", in_layout, out_layout);
    if (dtype != QPWC_F32 && dtype != QPWC_F16) return fail(QPWC_E_DTYPE, "unsupported dtype %d", dtype);
    const size_t es = esize(dtype);
    if (B <= 0 || h <= 0 || w <= 0) return fail(QPWC_E_SHAPE, "non-positive extent B=%d h=%d w=%d", B, h, w);
    if ((uintptr_t)in % es || (uintptr_t)out % es) return fail(QPWC_E_ALIGN, "flow pointers must be element aligned");
    if (overlaps(out, (size_t)B * 4 * h * w * 2 * es, in, (size_t)B * h * w * 2 * es))
        return fail(QPWC_E_ALIAS, "out overlaps in");
    return upsample2x_flow_launch(in, out, B, h, w, scale, dtype, in_layout, out_layout, (hipStream_t)stream);
}

int qpwc_epe_multi_workspace_floats(void) { return epe_multi_workspace_floats(); }

int qpwc_epe_multi_fwd(const void* const* y_true, const void* const* y_pred, const int64_t* n_pixels,
                       const int64_t* plane_pixels, int n_levels, void* out_means, void* workspace,
                       void* stream) {
    if (!y_true || !y_pred || !n_pixels || !out_means || !workspace)
        return fail(QPWC_E_NULL, "null pointer argument");
    if (n_levels < 1 || n_levels > 8) return fail(QPWC_E_SHAPE, "n_levels %d outside [1,8]", n_levels);
    for (int i = 0; i < n_levels; ++i) {
        if (!y_true[i] || !y_pred[i]) return fail(QPWC_E_NULL, "null flow pointer at level %d", i);
        if (n_pixels[i] <= 0) return fail(QPWC_E_SHAPE, "level %d has no pixels", i);
        const bool planar = plane_pixels && plane_pixels[i] > 0;
        if (planar && n_pixels[i] % plane_pixels[i])
            return fail(QPWC_E_SHAPE, "level %d: %lld pixels are not whole planes of %lld", i,
                        (long long)n_pixels[i], (long long)plane_pixels[i]);
        if ((uintptr_t)y_true[i] % (planar ? 4 : 8) || (uintptr_t)y_pred[i] % (planar ? 4 : 8))
            return fail(QPWC_E_ALIGN, "flow pointers must be 8-byte (pixels) / 4-byte (planes) aligned");
    }
    return epe_multi_launch(y_true, y_pred, n_pixels, plane_pixels, nullptr, n_levels, (float*)out_means,
                            (float*)workspace, (hipStream_t)stream);
}

int qpwc_epe_multi_mixed_fwd(const void* const* y_true, const void* const* y_pred, const int64_t* n_pixels,
                             const int64_t* plane_pixels, const int* pred_dtype, int n_levels, void* out_means,
                             void* workspace, void* stream) {
    if (!y_true || !y_pred || !n_pixels || !pred_dtype || !out_means || !workspace)
        return fail(QPWC_E_NULL, "null pointer argument");
    if (n_levels < 1 || n_levels > 8) return fail(QPWC_E_SHAPE, "n_levels %d outside [1,8]", n_levels);
    for (int i = 0; i < n_levels; ++i) {
        if (!y_true[i] || !y_pred[i]) return fail(QPWC_E_NULL, "null flow pointer at level %d", i);
        if (pred_dtype[i] != QPWC_F32 && pred_dtype[i] != QPWC_F16)
            return fail(QPWC_E_DTYPE, "level %d: unsupported prediction dtype %d", i, pred_dtype[i]);
        if (n_pixels[i] <= 0) return fail(QPWC_E_SHAPE, "level %d has no pixels", i);
        const bool planar = plane_pixels && plane_pixels[i] > 0;
        if (planar && n_pixels[i] % plane_pixels[i])
            return fail(QPWC_E_SHAPE, "level %d: %lld pixels are not whole planes of %lld", i,
                        (long long)n_pixels[i], (long long)plane_pixels[i]);
        const int pes = pred_dtype[i] == QPWC_F16 ? 2 : 4;
        if ((uintptr_t)y_true[i] % (planar ? 4 : 8) || (uintptr_t)y_pred[i] % (planar ? pes : 2 * pes))
            return fail(QPWC_E_ALIGN, "flow pointers must be aligned to a pixel (channels-last) / an element (planes)");
    }
    return epe_multi_launch(y_true, y_pred, n_pixels, plane_pixels, pred_dtype, n_levels, (float*)out_means,
                            (float*)workspace, (hipStream_t)stream);
}

int qpwc_cost_volume_to_flow_fwd(const void* cvol, void* flow, int B, int H, int W, int D,
                                 int64_t pixel_stride, int layout, int dtype, void* stream) {
    if (!cvol || !flow) return fail(QPWC_E_NULL, "null pointer argument");
    const int rc = check_common(B, H, W, D, layout, dtype);
    if (rc) return rc;
    if (layout == QPWC_NHWC ? pixel_stride < D : pixel_stride != D)
        return fail(QPWC_E_STRIDE, "pixel stride %lld for %d channels", (long long)pixel_stride, D);
    if ((uintptr_t)cvol % esize(dtype) || (uintptr_t)flow % 8)
        return fail(QPWC_E_ALIGN, "cvol must be element aligned, flow 8-byte aligned");
    if (overlaps(flow, (size_t)B * H * W * 8, cvol, (size_t)B * H * W * pixel_stride * esize(dtype)))
        return fail(QPWC_E_ALIAS, "flow overlaps cvol");
    return cost_volume_to_flow_launch(cvol, (float*)flow, B, H, W, D, pixel_stride, layout, dtype,
                                      (hipStream_t)stream);
}

static int check_flow_args(const void* flow, const void* out, int B, int H, int W, int layout, int dtype,
                           size_t out_bytes) {
    if (!flow || !out) return fail(QPWC_E_NULL, "null pointer argument");
    const int rc = check_common(B, H, W, 2, layout, dtype);
    if (rc) return rc;
    const size_t es = esize(dtype);
    if ((uintptr_t)flow % es || (uintptr_t)out % es)
        return fail(QPWC_E_ALIGN, "pointer not element aligned");
    if ((int64_t)B * H * W * 2 >= ((int64_t)1 << 40)) return fail(QPWC_E_SHAPE, "flow too large");
    if (overlaps(out, out_bytes, flow, (size_t)B * H * W * 2 * es)) return fail(QPWC_E_ALIAS, "out overlaps flow");
    return QPWC_OK;
}

int qpwc_invert_flow_fwd(const void* flow, void* out, int B, int H, int W, int layout, int dtype,
                         void* stream) {
    const int rc = check_flow_args(flow, out, B, H, W, layout, dtype,
                                   (size_t)(B > 0 ? B : 0) * (H > 0 ? H : 0) * (W > 0 ? W : 0) * 2 * esize(dtype));
    if (rc) return rc;
    return invert_flow_launch(flow, out, B, H, W, layout, dtype, (hipStream_t)stream);
}

int qpwc_occlusion_fwd(const void* flow, void* out, int B, int H, int W, int layout, int dtype,
                       void* stream) {
    const int rc = check_flow_args(flow, out, B, H, W, layout, dtype,
                                   (size_t)(B > 0 ? B : 0) * (H > 0 ? H : 0) * (W > 0 ? W : 0) * 4);
    if (rc) return rc;
    if ((uintptr_t)out % 4) return fail(QPWC_E_ALIGN, "out must be 4-byte aligned");
    return occlusion_launch(flow, out, B, H, W, layout, dtype, (hipStream_t)stream);
}

int qpwc_conv3x3_mish_fwd(const void* x, const void* weight, const void* bias, void* out, int B, int H,
                          int W, int C, int pad_h, int pad_w, void* stream) {
    if (!x || !weight || !bias || !out) return fail(QPWC_E_NULL, "null pointer argument");
    if (C != 16 && C != 32 && C != 64 && C != 128 && C != 256)
        return fail(QPWC_E_SHAPE, "C=%d not in {16,32,64,128,256}", C);
    if (B <= 0 || H <= 0 || W <= 0 || pad_h < 0 || pad_w < 0 || pad_h > 8 || pad_w > 8)
        return fail(QPWC_E_SHAPE, "bad shape B=%d H=%d W=%d pad=%d,%d", B, H, W, pad_h, pad_w);
    if ((uintptr_t)x % 16 || (uintptr_t)weight % 16 || (uintptr_t)bias % 16 || (uintptr_t)out % 16)
        return fail(QPWC_E_ALIGN, "x, weight, bias, out must be 16-byte aligned");
    if (overlaps(out, (size_t)B * (H + pad_h) * (W + pad_w) * C * 4, x, (size_t)B * H * W * C * 4))
        return fail(QPWC_E_ALIAS, "out overlaps x");
    return conv3x3_mish_launch(x, weight, bias, out, B, H, W, C, pad_h, pad_w, (hipStream_t)stream);
}

int qpwc_conv3x3_mish_x3_fwd(const void* x, const void* weight3, const void* bias, void* out, int B, int H,
                             int W, int C, int pad_h, int pad_w, void* stream) {
    if (!x || !weight3 || !bias || !out) return fail(QPWC_E_NULL, "null pointer argument");
    if (C != 16 && C != 32 && C != 64 && C != 128 && C != 256)
        return fail(QPWC_E_SHAPE, "C=%d not in {16,32,64,128,256}", C);
    if (B <= 0 || H <= 0 || W <= 0 || pad_h < 0 || pad_w < 0 || pad_h > 8 || pad_w > 8)
        return fail(QPWC_E_SHAPE, "bad shape B=%d H=%d W=%d pad=%d,%d", B, H, W, pad_h, pad_w);
    if ((uintptr_t)x % 16 || (uintptr_t)weight3 % 16 || (uintptr_t)bias % 16 || (uintptr_t)out % 16)
        return fail(QPWC_E_ALIGN, "x, weight3, bias, out must be 16-byte aligned");
    if (overlaps(out, (size_t)B * (H + pad_h) * (W + pad_w) * C * 4, x, (size_t)B * H * W * C * 4))
        return fail(QPWC_E_ALIAS, "out overlaps x");
    return conv3x3_mish_x3_launch(x, weight3, bias, out, B, H, W, C, pad_h, pad_w, (hipStream_t)stream);
}

int qpwc_split_bf16x3_fwd(const void* src, void* out, long long n, void* stream) {
    if (!src || !out) return fail(QPWC_E_NULL, "null pointer argument");
    if (n <= 0 || n > ((long long)1 << 40)) return fail(QPWC_E_SHAPE, "n=%lld out of range", n);
    if ((uintptr_t)src % 4 || (uintptr_t)out % 2) return fail(QPWC_E_ALIGN, "src must be 4-byte, out 2-byte aligned");
    if (overlaps(out, (size_t)n * 6, src, (size_t)n * 4)) return fail(QPWC_E_ALIAS, "out overlaps src");
    return split_bf16x3_launch(src, out, (int64_t)n, (hipStream_t)stream);
}

int qpwc_conv3x3_mish_f16_fwd(const void* x, const void* weight, const void* bias, void* out, int B, int H,
                              int W, int C, int pad_h, int pad_w, void* stream) {
    if (!x || !weight || !bias || !out) return fail(QPWC_E_NULL, "null pointer argument");
    if (C != 16 && C != 32 && C != 64 && C != 128 && C != 256)
        return fail(QPWC_E_SHAPE, "C=%d not in {16,32,64,128,256}", C);
    if (B <= 0 || H <= 0 || W <= 0 || pad_h < 0 || pad_w < 0 || pad_h > 8 || pad_w > 8)
        return fail(QPWC_E_SHAPE, "bad shape B=%d H=%d W=%d pad=%d,%d", B, H, W, pad_h, pad_w);
    if ((uintptr_t)x % 16 || (uintptr_t)weight % 16 || (uintptr_t)bias % 16 || (uintptr_t)out % 16)
        return fail(QPWC_E_ALIGN, "x, weight, bias, out must be 16-byte aligned");
    if (overlaps(out, (size_t)B * (H + pad_h) * (W + pad_w) * C * 2, x, (size_t)B * H * W * C * 2))
        return fail(QPWC_E_ALIAS, "out overlaps x");
    return conv3x3_mish_f16_launch(x, weight, bias, out, B, H, W, C, pad_h, pad_w, (hipStream_t)stream);
}

int qpwc_first_conv_mish_fwd(const void* pairs, const void* weight, const void* bias, void* out, int B,
                             int H, int W, int layout, void* stream) {
    if (!pairs || !weight || !bias || !out) return fail(QPWC_E_NULL, "null pointer argument");
    if (layout != QPWC_NHWC && layout != QPWC_NCHW) return fail(QPWC_E_LAYOUT, "Unsupported data format : %d", layout);
    if (B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1))
        return fail(QPWC_E_SHAPE, "B=%d H=%d W=%d: H and W must be even and >= 2", B, H, W);
    if ((uintptr_t)pairs % 8 || (uintptr_t)weight % 4 || (uintptr_t)bias % 16 || (uintptr_t)out % 16)
        return fail(QPWC_E_ALIGN, "pairs must be 8-byte, bias and out 16-byte aligned");
    if (overlaps(out, (size_t)2 * B * (H / 2) * (W / 2) * 16 * 4, pairs, (size_t)B * H * W * 6 * 4))
        return fail(QPWC_E_ALIAS, "out overlaps pairs");
    return first_conv_mish_launch(pairs, weight, bias, out, B, H, W, layout, QPWC_F32, (hipStream_t)stream);
}

int qpwc_first_conv_mish_f16_fwd(const void* pairs, const void* weight, const void* bias, void* out, int B,
                                 int H, int W, int layout, void* stream) {
    if (!pairs || !weight || !bias || !out) return fail(QPWC_E_NULL, "null pointer argument");
    if (layout != QPWC_NHWC && layout != QPWC_NCHW) return fail(QPWC_E_LAYOUT, "Unsupported data format : %d", layout);
    if (B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1))
        return fail(QPWC_E_SHAPE, "B=%d H=%d W=%d: H and W must be even and >= 2", B, H, W);
    if ((uintptr_t)pairs % 4 || (uintptr_t)weight % 4 || (uintptr_t)bias % 16 || (uintptr_t)out % 8)
        return fail(QPWC_E_ALIGN, "pairs must be 4-byte, bias 16-byte and out 8-byte aligned");
    if (overlaps(out, (size_t)2 * B * (H / 2) * (W / 2) * 16 * 2, pairs, (size_t)B * H * W * 6 * 2))
        return fail(QPWC_E_ALIAS, "out overlaps pairs");
    return first_conv_mish_launch(pairs, weight, bias, out, B, H, W, layout, QPWC_F16, (hipStream_t)stream);
}

int qpwc_conv3x3s2_mish_fwd(const void* x_padded, const void* weight, const void* bias, void* out, int B,
                            int H, int W, void* stream) {
    if (!x_padded || !weight || !bias || !out) return fail(QPWC_E_NULL, "null pointer argument");
    if (B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1))
        return fail(QPWC_E_SHAPE, "B=%d H=%d W=%d: H and W must be even and >= 2", B, H, W);
    if ((uintptr_t)x_padded % 16 || (uintptr_t)weight % 16 || (uintptr_t)bias % 16 || (uintptr_t)out % 16)
        return fail(QPWC_E_ALIGN, "x, weight, bias, out must be 16-byte aligned");
    if (overlaps(out, (size_t)B * (H / 2) * (W / 2) * 32 * 4, x_padded, (size_t)B * (H + 1) * (W + 1) * 16 * 4))
        return fail(QPWC_E_ALIAS, "out overlaps x");
    return conv3x3s2_mish_launch(x_padded, weight, bias, out, B, H, W, (hipStream_t)stream);
}

int qpwc_conv3x3s2_mish_x3_fwd(const void* x_padded, const void* weight3, const void* bias, void* out, int B,
                               int H, int W, int C_in, void* stream) {
    if (!x_padded || !weight3 || !bias || !out) return fail(QPWC_E_NULL, "null pointer argument");
    if (C_in != 32 && C_in != 64 && C_in != 128) return fail(QPWC_E_SHAPE, "C_in=%d not in {32,64,128}", C_in);
    if (B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1))
        return fail(QPWC_E_SHAPE, "B=%d H=%d W=%d: H and W must be even and >= 2", B, H, W);
    if ((uintptr_t)x_padded % 16 || (uintptr_t)weight3 % 16 || (uintptr_t)bias % 16 || (uintptr_t)out % 16)
        return fail(QPWC_E_ALIGN, "x, weight3, bias, out must be 16-byte aligned");
    if (overlaps(out, (size_t)B * (H / 2) * (W / 2) * 2 * C_in * 4, x_padded,
                 (size_t)B * (H + 1) * (W + 1) * C_in * 4))
        return fail(QPWC_E_ALIAS, "out overlaps x");
    return conv3x3s2_mish_x3_launch(x_padded, weight3, bias, out, B, H, W, C_in, (hipStream_t)stream);
}

int qpwc_conv3x3s2_mish_c_fwd(const void* x_padded, const void* weight, const void* bias, void* out, int B,
                              int H, int W, int C_in, void* stream) {
    if (!x_padded || !weight || !bias || !out) return fail(QPWC_E_NULL, "null pointer argument");
    if (C_in != 16 && C_in != 32 && C_in != 64 && C_in != 128)
        return fail(QPWC_E_SHAPE, "C_in=%d not in {16,32,64,128}", C_in);
    if (B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1))
        return fail(QPWC_E_SHAPE, "B=%d H=%d W=%d: H and W must be even and >= 2", B, H, W);
    if ((uintptr_t)x_padded % 16 || (uintptr_t)weight % 16 || (uintptr_t)bias % 16 || (uintptr_t)out % 16)
        return fail(QPWC_E_ALIGN, "x, weight, bias, out must be 16-byte aligned");
    if (overlaps(out, (size_t)B * (H / 2) * (W / 2) * 2 * C_in * 4, x_padded,
                 (size_t)B * (H + 1) * (W + 1) * C_in * 4))
        return fail(QPWC_E_ALIAS, "out overlaps x");
    return conv3x3s2_mish_any_launch(x_padded, weight, bias, out, B, H, W, C_in, (hipStream_t)stream);
}

int qpwc_conv3x3s2_mish_f16_fwd(const void* x_padded, const void* weight, const void* bias, void* out, int B,
                                int H, int W, int C_in, void* stream) {
    if (!x_padded || !weight || !bias || !out) return fail(QPWC_E_NULL, "null pointer argument");
    if (C_in != 16 && C_in != 32 && C_in != 64 && C_in != 128)
        return fail(QPWC_E_SHAPE, "C_in=%d not in {16,32,64,128}", C_in);
    if (B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1))
        return fail(QPWC_E_SHAPE, "B=%d H=%d W=%d: H and W must be even and >= 2", B, H, W);
    if ((uintptr_t)x_padded % 16 || (uintptr_t)weight % 16 || (uintptr_t)bias % 16 || (uintptr_t)out % 16)
        return fail(QPWC_E_ALIGN, "x, weight, bias, out must be 16-byte aligned");
    if (overlaps(out, (size_t)B * (H / 2) * (W / 2) * 2 * C_in * 2, x_padded,
                 (size_t)B * (H + 1) * (W + 1) * C_in * 2))
        return fail(QPWC_E_ALIAS, "out overlaps x");
    return conv3x3s2_mish_f16_launch(x_padded, weight, bias, out, B, H, W, C_in, (hipStream_t)stream);
}

int qpwc_upconv4x4s2_mish_x3_fwd(const void* x, const void* weight3, const void* bias, void* out, int B, int H, int W,
                                 int C, int F, int64_t out_pixel_stride, void* stream) {
    if (!x || !weight3 || !bias || !out) return fail(QPWC_E_NULL, "null pointer argument");
    if (C != 64 && C != 128 && C != 256) return fail(QPWC_E_SHAPE, "C=%d not in {64,128,256}", C);
    if (F <= 0 || F % 16) return fail(QPWC_E_SHAPE, "F=%d must be a positive multiple of 16", F);
    if (B <= 0 || H <= 0 || W <= 0) return fail(QPWC_E_SHAPE, "non-positive extent B=%d H=%d W=%d", B, H, W);
    if (out_pixel_stride < F || out_pixel_stride % 4 || out_pixel_stride > (1 << 20))
        return fail(QPWC_E_STRIDE, "out_pixel_stride %lld must be >= F and a multiple of 4", (long long)out_pixel_stride);
    if ((uintptr_t)x % 16 || (uintptr_t)weight3 % 16 || (uintptr_t)bias % 16 || (uintptr_t)out % 16)
        return fail(QPWC_E_ALIGN, "x, weight3, bias, out must be 16-byte aligned");
    if (overlaps(out, (size_t)B * 4 * H * W * out_pixel_stride * 4, x, (size_t)B * H * W * C * 4))
        return fail(QPWC_E_ALIAS, "out overlaps x");
    return upconv4x4s2_mish_x3_launch(x, weight3, bias, out, B, H, W, C, F, (int)out_pixel_stride, (hipStream_t)stream);
}

int qpwc_upconv4x4s2_mish_fwd(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                              int C, int F, int64_t out_pixel_stride, void* stream) {
    if (!x || !weight || !bias || !out) return fail(QPWC_E_NULL, "null pointer argument");
    if (C != 64 && C != 128 && C != 256) return fail(QPWC_E_SHAPE, "C=%d not in {64,128,256}", C);
    if (F <= 0 || F % 16) return fail(QPWC_E_SHAPE, "F=%d must be a positive multiple of 16", F);
    if (B <= 0 || H <= 0 || W <= 0) return fail(QPWC_E_SHAPE, "non-positive extent B=%d H=%d W=%d", B, H, W);
    if (out_pixel_stride < F || out_pixel_stride % 4 || out_pixel_stride > (1 << 20))
        return fail(QPWC_E_STRIDE, "out_pixel_stride %lld must be >= F and a multiple of 4", (long long)out_pixel_stride);
    if ((uintptr_t)x % 16 || (uintptr_t)weight % 16 || (uintptr_t)bias % 16 || (uintptr_t)out % 16)
        return fail(QPWC_E_ALIGN, "x, weight, bias, out must be 16-byte aligned");
    if (overlaps(out, (size_t)B * 4 * H * W * out_pixel_stride * 4, x, (size_t)B * H * W * C * 4))
        return fail(QPWC_E_ALIAS, "out overlaps x");
    return upconv4x4s2_mish_launch(x, weight, bias, out, B, H, W, C, F, (int)out_pixel_stride, (hipStream_t)stream);
}

int qpwc_upconv4x4s2_mish_f16_fwd(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                                  int C, int F, int64_t out_pixel_stride, void* stream) {
    if (!x || !weight || !bias || !out) return fail(QPWC_E_NULL, "null pointer argument");
    if (C != 64 && C != 128 && C != 256) return fail(QPWC_E_SHAPE, "C=%d not in {64,128,256}", C);
    if (F <= 0 || F % 16) return fail(QPWC_E_SHAPE, "F=%d must be a positive multiple of 16", F);
    if (B <= 0 || H <= 0 || W <= 0) return fail(QPWC_E_SHAPE, "non-positive extent B=%d H=%d W=%d", B, H, W);
    if (out_pixel_stride < F || out_pixel_stride % 4 || out_pixel_stride > (1 << 20))
        return fail(QPWC_E_STRIDE, "out_pixel_stride %lld must be >= F and a multiple of 4", (long long)out_pixel_stride);
    if ((uintptr_t)x % 16 || (uintptr_t)weight % 16 || (uintptr_t)bias % 16 || (uintptr_t)out % 8)
        return fail(QPWC_E_ALIGN, "x, weight, bias must be 16-byte aligned, out 8-byte");
    if (overlaps(out, (size_t)B * 4 * H * W * out_pixel_stride * 2, x, (size_t)B * H * W * C * 2))
        return fail(QPWC_E_ALIAS, "out overlaps x");
    return upconv4x4s2_mish_f16_launch(x, weight, bias, out, B, H, W, C, F, (int)out_pixel_stride, (hipStream_t)stream);
}

// UpConv + the skip half of concat([up, skip]) in one launch (round 4; include/qpwc.h)
static int upconv_cat_common(const void* x, const void* weight, const void* bias, const void* skip, int64_t skip_bs,
                             int64_t skip_rs, int64_t skip_ps, void* out, int B, int H, int W, int C, int F,
                             int64_t out_pixel_stride, void* stream, int esize) {
    if (!x || !weight || !bias || !out || !skip) return fail(QPWC_E_NULL, "null pointer argument");
    if (C != 64 && C != 128 && C != 256) return fail(QPWC_E_SHAPE, "C=%d not in {64,128,256}", C);
    if (F <= 0 || F % 16) return fail(QPWC_E_SHAPE, "F=%d must be a positive multiple of 16", F);
    if (B <= 0 || H <= 0 || W <= 0) return fail(QPWC_E_SHAPE, "non-positive extent B=%d H=%d W=%d", B, H, W);
    if (out_pixel_stride < 2 * (int64_t)F || out_pixel_stride % 4 || out_pixel_stride > (1 << 20))
        return fail(QPWC_E_STRIDE, "out_pixel_stride %lld must be >= 2 F and a multiple of 4", (long long)out_pixel_stride);
    if (skip_ps < F || skip_ps % 4 || skip_rs % 4 || skip_bs % 4 || skip_rs < 2 * (int64_t)W * skip_ps ||
        skip_bs < 2 * (int64_t)H * skip_rs)
        return fail(QPWC_E_STRIDE, "skip strides (%lld, %lld, %lld) must describe (B, 2H, 2W, >= F) in multiples of 4 elements",
                    (long long)skip_bs, (long long)skip_rs, (long long)skip_ps);
    const int oa = esize == 4 ? 16 : 8;
    if ((uintptr_t)x % 16 || (uintptr_t)weight % 16 || (uintptr_t)bias % 16 || (uintptr_t)out % oa || (uintptr_t)skip % oa)
        return fail(QPWC_E_ALIGN, "x, weight, bias must be 16-byte aligned, out and skip %d-byte", oa);
    const size_t out_bytes = (size_t)B * 4 * H * W * out_pixel_stride * esize;
    if (overlaps(out, out_bytes, x, (size_t)B * H * W * C * esize)) return fail(QPWC_E_ALIAS, "out overlaps x");
    if (overlaps(out, out_bytes, skip, (size_t)B * skip_bs * esize)) return fail(QPWC_E_ALIAS, "out overlaps skip");
    if (esize == 4)
        return upconv4x4s2_mish_launch(x, weight, bias, out, B, H, W, C, F, (int)out_pixel_stride, (hipStream_t)stream, skip,
                                       skip_bs, skip_rs, skip_ps);
    return upconv4x4s2_mish_f16_launch(x, weight, bias, out, B, H, W, C, F, (int)out_pixel_stride, (hipStream_t)stream, skip,
                                       skip_bs, skip_rs, skip_ps);
}

int qpwc_upconv4x4s2_mish_cat_fwd(const void* x, const void* weight, const void* bias, const void* skip,
                                  int64_t skip_batch_stride, int64_t skip_row_stride, int64_t skip_pixel_stride, void* out,
                                  int B, int H, int W, int C, int F, int64_t out_pixel_stride, void* stream) {
    return upconv_cat_common(x, weight, bias, skip, skip_batch_stride, skip_row_stride, skip_pixel_stride, out, B, H, W, C, F,
                             out_pixel_stride, stream, 4);
}

int qpwc_upconv4x4s2_mish_cat_f16_fwd(const void* x, const void* weight, const void* bias, const void* skip,
                                      int64_t skip_batch_stride, int64_t skip_row_stride, int64_t skip_pixel_stride, void* out,
                                      int B, int H, int W, int C, int F, int64_t out_pixel_stride, void* stream) {
    return upconv_cat_common(x, weight, bias, skip, skip_batch_stride, skip_row_stride, skip_pixel_stride, out, B, H, W, C, F,
                             out_pixel_stride, stream, 2);
}

}  // extern "C"
