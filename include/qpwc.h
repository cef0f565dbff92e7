/* qpwc.h -- C ABI of libqpwc_hip.so: the MI355X (gfx950) hot path of qpwcnet.
 *
 * The reference (yycho0108/qpwcnet) has no FFI layer of its own: its boundary
 * is the Python Keras-layer call() surface, and one level down the registered
 * TensorFlow op of tensorflow-addons.  Each entry point below names the
 * reference interface it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer;
 *   - the caller owns all memory; kernels never allocate, free or retain;
 *   - `out` must not overlap an input;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*, NULL = the
 *     default stream) and the call returns without synchronising;
 *   - return value 0 = enqueued, negative = QPWC_E_* (nothing was enqueued);
 *     qpwc_last_error() returns a thread-local description of the last failure.
 */
#ifndef QPWC_H_
#define QPWC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QPWC_VERSION 200 /* 0.2.0 */

/* layout of every image-like tensor of one call */
#define QPWC_NHWC 0 /* 'channels_last'  (B,H,W,C) */
#define QPWC_NCHW 1 /* 'channels_first' (B,C,H,W) */

/* storage dtype; arithmetic is always fp32 */
#define QPWC_F32 0
#define QPWC_F16 1

/* warp border semantics */
#define QPWC_WARP_CLAMP 0  /* WarpV2: tfa dense_image_warp, clamp-to-border */
#define QPWC_WARP_TFWARP 1 /* Warp  : qpwcnet tf_warp (truncate, clip, raw weights) */

/* flo_bcast_mask bits: set = that flow dimension has extent 1 and is broadcast */
#define QPWC_BCAST_B 1
#define QPWC_BCAST_H 2
#define QPWC_BCAST_W 4

#define QPWC_OK 0
#define QPWC_E_NULL (-1)     /* null pointer */
#define QPWC_E_LAYOUT (-2)   /* 'Unsupported data format' (layers.py:19-29) */
#define QPWC_E_DTYPE (-3)    /* unsupported dtype code */
#define QPWC_E_SHAPE (-4)    /* non-positive extent, or H/W < 2 for clamp warp (warp.py:182-184) */
#define QPWC_E_RANGE (-5)    /* search_range < 0 or too large */
#define QPWC_E_MODE (-6)     /* unknown warp mode */
#define QPWC_E_ALIAS (-7)    /* out overlaps an input */
#define QPWC_E_LAUNCH (-8)   /* hipGetLastError() after launch; see qpwc_last_error() */
#define QPWC_E_ALIGN (-9)    /* pointer not aligned to its element size */
#define QPWC_E_STRIDE (-10)  /* output stride/offset cannot hold the result */
#define QPWC_E_NODEVICE (-11)/* no HIP device / runtime error before launch */

int qpwc_version(void);
const char* qpwc_last_error(void);
const char* qpwc_strerror(int code);
/* Static description of the build.  The product library selects kernels from the call's arguments
 * only (no environment variable changes what runs); the `make experimental` build of the same ABI
 * keeps A/B switches and reports "EXPERIMENTAL" here. */
const char* qpwc_build_info(void);

/* Layout conversion at the boundary: out = in with the same logical (B,H,W,C) content in `to_layout`
 * (in is in the other layout).  The reference picks its layout from image_data_format()
 * (qpwcnet/core/layers.py:41,146; NCHW handled by transposing in and out, layers.py:179-183); the hot-path
 * kernels are channels-last, so a dense 'channels_first' tensor crosses the boundary through this kernel. */
int qpwc_layout_transpose_fwd(const void* in, void* out, int B, int H, int W, int C, int to_layout, int dtype,
                              void* stream);

/* dst view = src view for two channels-last (B,H,W,C) views with their own strides (elements, order batch / row /
 * pixel; channels contiguous): the skip half of the decoder's concat([UpConv(x), skip])
 * (qpwcnet/core/pwcnet.py:186-195), which the reference leaves to tf.concat.  C * element size and every stride
 * must be whole 16-byte units, both base pointers 16-byte aligned, the two extents disjoint. */
int qpwc_copy_pixels_fwd(const void* src, void* dst, int B, int H, int W, int C, const int64_t* src_strides,
                         const int64_t* dst_strides, int dtype, void* stream);

/* Measurement aid ("measure the achievable ceiling on the box with a device copy
 * kernel"): dst[0..bytes) = src[0..bytes), 16 B per lane, bytes % 16 == 0, both 16-byte aligned.
 * Not part of the reference's surface. */
int qpwc_device_copy(const void* src, void* dst, int64_t bytes, void* stream);

/* Measurement aid: ONE wave writes n_samples pairs (s_memtime = shader-clock ticks, s_memrealtime = 100 MHz ticks) into
 * out_pairs (2 * n_samples uint64), sleeping sleeps_per_sample x s_sleep 127 between stamps, then ends.  Launched on a
 * stream of its own beside the work under test, delta(s_memtime) / delta(s_memrealtime) x 100 MHz is the shader clock
 * the chip sustains under that work (bench.py: `sustained_clock_mhz`).  Not part of the reference's surface. */
int qpwc_clock_probe(void* out_pairs, int n_samples, int sleeps_per_sample, void* stream);

/* CostVolume / CostVolumeV2 forward.
 * Replaces: CostVolume.call           qpwcnet/core/layers.py:72-100
 *           CostVolumeV2.call         qpwcnet/core/layers.py:128-132, i.e. the op
 *           tfa CorrelationCost(kernel_size=1, max_displacement=r, stride_1=1,
 *           stride_2=1, pad=r, data_format) + leaky_relu   (layers.py:124-125,131)
 *           and their functor twins   qpwcnet/core/non_layers.py:72-104, 119-123.
 *   d = 2*search_range+1
 *   out[b,y,x,i*d+j] = lrelu( (1/C) * sum_c prv[b,y,x,c] * nxt[b,y+i-r,x+j-r,c] ),
 *   nxt taken as zero outside the image; lrelu(v) = v > 0 ? v : slope*v.
 * prv, nxt: (B,H,W,C) or (B,C,H,W) dense; out: (B,H,W,d*d) or (B,d*d,H,W) dense. */
int qpwc_cost_volume_fwd(const void* prv, const void* nxt, void* out,
                         int B, int H, int W, int C, int search_range,
                         int layout, int dtype, float lrelu_slope, void* stream);

/* Same computation, writing into a wider channels-last buffer: pixel p's d*d
 * results go to out + p*out_pixel_stride + out_channel_offset (in elements).
 * This is how Flow/UpFlow's concat([cost, prv, ...]) (non_layers.py:332-338,
 * 381-385) is fed without a copy.  NHWC only.
 * With out_pixel_stride == 84 and out_channel_offset == 0 the three pad channels 81..83 of every
 * pixel are written as zeros: an 84-channel cost volume whose pixels start on 16-byte boundaries (the
 * vector-load layout the fused SeparableConv2D consumes). */
int qpwc_cost_volume_fwd_strided(const void* prv, const void* nxt, void* out,
                                 int B, int H, int W, int C, int search_range,
                                 int dtype, float lrelu_slope,
                                 int64_t out_pixel_stride, int64_t out_channel_offset,
                                 void* stream);

/* Warp / WarpV2 forward.
 * Replaces: WarpV2.call  qpwcnet/core/layers.py:177-186 (non_layers.py:147-158):
 *             tfa.image.dense_image_warp(img, -flo[..., ::-1])  -> mode CLAMP
 *           Warp.call    qpwcnet/core/layers.py:166-168 -> tf_warp,
 *             qpwcnet/core/warp.py:63-153                       -> mode TFWARP
 * Samples img at (y + flo[...,1], x + flo[...,0]) bilinearly.
 * img/out: (B,H,W,C) or (B,C,H,W) of `dtype`; flo: same layout with 2 channels
 * (x, y), ALWAYS fp32 (coordinates need the precision); flow dims flagged in flo_bcast_mask have extent 1 (e.g. a (1,1,1,2) flow,
 * qpwcnet/app/optical_flow/test_warp.py:32) and flo is dense over the rest. */
int qpwc_warp_fwd(const void* img, const void* flo, void* out,
                  int B, int H, int W, int C, int flo_bcast_mask,
                  int layout, int dtype, int mode, void* stream);

/* UpFlow front end in one launch (non_layers.py:377-385):
 *   nxt_w = WarpV2(nxt, flo); cost = CostVolumeV2(prv, nxt_w)
 * without materialising nxt_w.  NHWC, mode CLAMP, flo dense (B,H,W,2) fp32,
 * search_range 4, C % 4 == 0.  Output addressing as in
 * qpwc_cost_volume_fwd_strided (pass stride d*d, offset 0 for a dense result).
 * PERFORMANCE NOTE: this entry point always computes what it is asked.  Where the matrix-core
 * kernels do not apply (qpwc_warp_cost_volume_kernel() names the choice: anything but
 * "cost_volume_mfma_lds*" means fewer than 256 regions of 8x8 pixels, C % 32 != 0, or a generic
 * shape) it runs the LDS-tiled vector kernel, which is SLOWER than calling qpwc_warp_fwd +
 * qpwc_cost_volume_fwd (B=8, 16x32x256: 94 us against 5.6 + 7.8 us).  A caller that wants the
 * faster of the two asks qpwc_warp_cost_volume_kernel() first, as qpwcnet_amd/non_layers.py does. */
int qpwc_warp_cost_volume_fwd(const void* prv, const void* nxt, const void* flo, void* out,
                              int B, int H, int W, int C, int search_range,
                              int dtype, float lrelu_slope,
                              int64_t out_pixel_stride, int64_t out_channel_offset,
                              void* stream);

/* Which kernel qpwc_cost_volume_fwd[_strided] (fused == 0) or qpwc_warp_cost_volume_fwd (fused != 0) launches for
 * this shape with 16-byte aligned operands: the launchers' own selection rules run without enqueuing anything
 * (host only, no GPU call).  out_pixel_stride <= 0 means dense.  Returns a static string -- one of
 * "cost_volume_mfma_kernel", "cost_volume_mfma_lds_kernel", "cost_volume_mfma_lds_kernel<true>",
 * "cost_volume_mfma_lds8x16_warp_kernel", "cost_volume_mfma_lds16_kernel<true>",
 * "cost_volume_mfma_lds_f16_kernel", "cost_volume_mfma_lds_f16_kernel<true>", "cost_volume_tiled_kernel",
 * "cost_volume_tiled_kernel<fused>", "cost_volume_generic_kernel" -- or "" for arguments the entry point refuses.
 * No reference counterpart: a caller's aid for the fused-or-pair decision (see the note above) and for profiles. */
const char* qpwc_cost_volume_kernel(int B, int H, int W, int C, int search_range, int layout, int dtype,
                                    int64_t out_pixel_stride, int fused);

/* End-point error (qpwcnet/app/optical_flow/train.py:247-253):
 *   *out_mean = mean over (b,y,x) of || y_true - y_pred ||_2 over the 2 flow channels.
 * fp32 flows of shape (B,H,W,2) / (B,2,H,W).  workspace: >= qpwc_epe_workspace_floats()
 * floats of device scratch owned by the caller (deterministic two-stage sum). */
int qpwc_epe_workspace_floats(void);
int qpwc_epe_fwd(const void* y_true, const void* y_pred, void* out_mean, void* workspace,
                 int B, int H, int W, int layout, void* stream);

/* Multi-scale form (FlowMseLoss.call, qpwcnet/train/loss.py:56-67): the EPE of n_levels
 * (<= 8) fp32 flow pairs y_true[i], y_pred[i] with n_pixels[i] = B*h_i*w_i each, in two
 * launches.  plane_pixels: NULL or all zero = channels-last (B,h,w,2) flows; plane_pixels[i] =
 * h_i*w_i > 0 = 'channels_first' (B,2,h,w) flows at that level.  out_means: n_levels floats;
 * workspace: >= qpwc_epe_multi_workspace_floats() floats. */
int qpwc_epe_multi_workspace_floats(void);
int qpwc_epe_multi_fwd(const void* const* y_true, const void* const* y_pred, const int64_t* n_pixels,
                       const int64_t* plane_pixels, int n_levels, void* out_means, void* workspace,
                       void* stream);
/* The same with a storage dtype per prediction (pred_dtype[i] = QPWC_F32 / QPWC_F16; y_true stays fp32): the
 * fp16-storage network's flows are converted inside the reduction instead of by a pass of their own. */
int qpwc_epe_multi_mixed_fwd(const void* const* y_true, const void* const* y_pred, const int64_t* n_pixels,
                             const int64_t* plane_pixels, const int* pred_dtype, int n_levels, void* out_means,
                             void* workspace, void* stream);

/* cost_volume_to_flow (qpwcnet/core/vis.py:9-34): flow[b,y,x] = (di, dj), the (row, column) displacement
 * of the first maximum over the D = d*d channels of a cost volume: imax = argmax_k, q = sqrt(D),
 * di = floor(imax/q) - (q-1)/2, dj = imax - floor(imax/q)*q - (q-1)/2 in fp32 like the reference.
 * cvol: (B,H,W,D) read at pixel_stride elements per pixel (>= D: the 84-channel padded volume decodes
 * in place) or (B,D,H,W) (pixel_stride == D); fp32 or fp16.  flow: fp32 (B,H,W,2) / (B,2,H,W).
 * NaN-free input assumed (tf.argmax's NaN ordering is not reproduced). */
int qpwc_cost_volume_to_flow_fwd(const void* cvol, void* flow, int B, int H, int W, int D,
                                 int64_t pixel_stride, int layout, int dtype, void* stream);

/* ---- OptFlow block, the step right after the cost volume at every level
 * (reference qpwcnet/core/non_layers.py:213-273) ---------- */

/* Depthwise half of SeparableConv2D(3x3, 'same', depth_multiplier 1, no depthwise
 * bias) (non_layers.py:223-231): out[b,y,x,c] = sum_{ky,kx} w[c,ky,kx] * in0[b,y+ky-1,x+kx-1,c],
 * in0 = (mish_on_load ? Mish(in) : in) inside the image and 0 outside.
 * `in` is the channel-wise concatenation of n_src (1..3) channels-last fp32 sources:
 * source i contributes src_channels[i] channels read at src[i] + pixel*src_pixel_stride[i]
 * (elements; all sources and `out` share the storage `dtype`, weights/params stay fp32,
 * arithmetic is fp32) -- Flow/UpFlow's concat([cost, prv, flo]) (non_layers.py:336-338,381-385) is
 * never materialised.  weight: (C,3,3) fp32, C = sum(src_channels); out: (B,H,W,C) dense. */
int qpwc_dwconv3x3_fwd(const void* const* src, const int* src_channels,
                       const int64_t* src_pixel_stride, int n_src, int mish_on_load,
                       const void* weight, void* out, int B, int H, int W, int dtype, void* stream);

/* The whole SeparableConv2D(3x3,'same') before its activation, fp32, in one launch:
 *   out[b,y,x,f] = bias[f] + sum_c pw[f,c] * (depthwise3x3(in0))[b,y,x,c]
 * with `in`/`in0` as in qpwc_dwconv3x3_fwd (1..3 sources); mish_flags: bit 0 = Mish on load (the
 * input is a pre-activation tensor), bit 1 = store Mish(out) (the SeparableConv2D's own
 * `activation='Mish'`, non_layers.py:226, applied once per element instead of at the next load); the
 * depthwise result stays on chip (LDS -> matrix cores).  dw: (C,3,3); pw: (F, Cpad) row-major,
 * Cpad = ceil(C/32)*32, zero padded; bias: (F); F in {16,32,64,128}; out: (B,H,W,F) dense. */
int qpwc_sepconv3x3_fwd(const void* const* src, const int* src_channels,
                        const int64_t* src_pixel_stride, int n_src, int mish_flags,
                        const void* dw, const void* pw, const void* bias, void* out,
                        int B, int H, int W, int F, void* stream);

/* qpwc_sepconv3x3_fwd with the pointwise products on the bf16 matrix instructions ("bf16x3", csrc/split_bf16.h: both
 * operands split into three bf16 values, six partial products, fp32 accumulation; depthwise 3x3 in fp32 as before).
 * pw3: the (F, Cpad) fp32 matrix of qpwc_sepconv3x3_fwd split by qpwc_split_bf16x3_fwd = (3, F, Cpad) bf16.
 * Sources: every source but the last a multiple of 4 channels in 16-byte aligned pixels (a last source of fewer than 4
 * channels is allowed), H*W < 2^24 -- QPWC_E_ALIGN otherwise (use qpwc_sepconv3x3_fwd then). */
int qpwc_sepconv3x3_x3_fwd(const void* const* src, const int* src_channels, const int64_t* src_pixel_stride,
                           int n_src, int mish_flags, const void* dw, const void* pw3, const void* bias, void* out,
                           int B, int H, int W, int F, void* stream);

/* The same SeparableConv2D for fp16 storage (BASELINE configs[4]; the reference itself has no
 * fp16 path; layer: non_layers.py:223-231): sources and `out` fp16, dw (C,3,3) and bias (F) fp32, pw (F, Cpad)
 * fp16 with Cpad = ceil(C/32)*32, zero padded.  Depthwise in fp32 on the fp16 input, rounded to fp16
 * once, pointwise on the f16 matrix cores with fp32 accumulation; mish_flags as above.  Sources as in
 * qpwc_sepconv3x3_fwd, each a multiple of 4 channels in 8-byte aligned pixels (a last source of fewer
 * than 4 channels is read element-wise); one dense source of C % 8 == 0 channels in 16-byte aligned
 * pixels takes 16-byte loads. */
int qpwc_sepconv3x3_f16_fwd(const void* const* src, const int* src_channels,
                            const int64_t* src_pixel_stride, int n_src, int mish_flags,
                            const void* dw, const void* pw, const void* bias, void* out,
                            int B, int H, int W, int F, void* stream);

/* Tail of OptFlow.__call__ (non_layers.py:238-254, 268-273) on the 16-channel
 * pre-activation output z (B,H,W,16) of the last SeparableConv's pointwise conv:
 *   flow = scale * conv3x3_{16->2, no bias, 'same'}( BN( Mish( W1 * Mish(z) + b1 ) ) )
 * params (device fp32, qpwc_flow_head_param_floats() = 592 floats):
 *   w1[16][16] (out,in) | b1[16] | bn_scale[16] | bn_shift[16] | wf[3][3][16][2] (ky,kx,in,out)
 * with bn_scale = gamma/sqrt(var+eps), bn_shift = beta - mean*bn_scale.
 * out: (B,H,W,2) for out_layout QPWC_NHWC, (B,2,H,W) for QPWC_NCHW (a 'channels_first' model's flow
 * output, written by the kernel itself instead of by a transposition launch). */
int qpwc_flow_head_param_floats(void);
int qpwc_flow_head_fwd(const void* z, const void* params, void* out, int B, int H, int W,
                       float scale, int dtype, int out_layout, void* stream);

/* Pointwise half of a SeparableConv2D whose depthwise half ran as qpwc_dwconv3x3_fwd (non_layers.py:223-231; the wide
 * first OptFlow layer of the coarsest levels, which stay split): out (M, F) = y (M, C) . weight^T + bias, fp32, on the
 * matrix cores, M = B*H*W pixels.  weight: (F, ceil(C/32)*32) row-major, zero padded (the layout qpwc_sepconv3x3_fwd
 * takes); F in {16,32,64,128,256}.  Measured slower than hipBLASLt at the step's shapes: the network keeps the
 * library GEMM for these two layers; this entry point serves callers that have none. */
int qpwc_pointwise_bias_fwd(const void* y, const void* weight, const void* bias, void* out, int64_t M, int C, int F,
                            void* stream);

/* qpwc_flow_head_fwd (channels-last) AND the Upsample(scale = up_scale) that follows it in pwcnet.py:55,60 in one launch:
 * out (B,H,W,2) as there, out_up (B,2H,2W,2) = up_scale * UpSampling2D(2, 'bilinear')(out) -- bit for bit what
 * qpwc_upsample2x_flow_fwd returns for `out` (the tile's one-pixel rim is computed by the same workgroup).
 * out_up_f32 (fp16 storage only, else NULL): the values of out_up once more as fp32 (B,2H,2W,2) -- the coordinates the next
 * level's WarpV2 takes, which would otherwise be a cast launch of their own. */
int qpwc_flow_head_up_fwd(const void* z, const void* params, void* out, void* out_up, void* out_up_f32, int B, int H, int W,
                          float scale, float up_scale, int dtype, void* stream);

/* The tail of OptFlow.__call__ in one launch, for small images (coarse pyramid levels), fp32 channels-last:
 *   z3   = Mish(SeparableConv2D_3(z2))    64 -> 32   (non_layers.py:223-231, third of the four)
 *   z4   = SeparableConv2D_4(z3)          32 -> 16
 *   flow = scale * conv3x3( BN( Mish( W1 * Mish(z4) + b1 ) ) )   (non_layers.py:238-254, 268-273)
 * z2: (B,H,W,64), the second SeparableConv2D's output, ALREADY Mish-activated (mish_on_load = 0) or
 * pre-activation (mish_on_load = 1).  dw3 (64,9), pw3 (32,64) row-major, b3 (32); dw4 (32,9), pw4 (16,32), b4 (16);
 * head_params as qpwc_flow_head_fwd.  out: (B,H,W,2) / (B,2,H,W) by out_layout.  Same arithmetic as
 * qpwc_sepconv3x3_fwd x 2 + qpwc_flow_head_fwd (fp32 matrix cores for the pointwise parts). */
int qpwc_optflow_tail_fwd(const void* z2, const void* dw3, const void* pw3, const void* b3, const void* dw4,
                          const void* pw4, const void* b4, const void* head_params, void* out, int B, int H, int W,
                          float scale, int mish_on_load, int out_layout, void* stream);

/* x = Mish(x + bias[c]) in place, channels-last fp32 (n_pixels, C), C % 4 == 0, bias may
 * be NULL: the `activation='Mish'` epilogue of the reference's Conv2D / Conv2DTranspose /
 * SeparableConv2D blocks (non_layers.py:196-210, 223-231, 390-449; mish.py:27-28). */
int qpwc_bias_mish_fwd(void* x, const void* bias, int64_t n_pixels, int C, int dtype, void* stream);

/* Out-of-place form that also lays down TensorFlow's 'SAME' padding for a following
 * stride-2 3x3 convolution (0 before, pad_h / pad_w after; non_layers.py:402-409):
 * dst (B, H+pad_h, W+pad_w, *): interior = Mish(src + bias[c]), border = 0.  src (B,H,W,C).
 * dst pixels are dst_pixel_stride elements apart (>= C, multiple of 4) and `dst` already points at
 * the first of the C destination channels: with pad 0 and a wider buffer this writes one
 * half of the decoder's concat([up, skip]) (pwcnet.py:186-195) in place. */
int qpwc_bias_mish_pad_fwd(const void* src, const void* bias, void* dst, int B, int H, int W, int C,
                           int pad_h, int pad_w, int64_t dst_pixel_stride, int dtype, void* stream);

/* Split(2) of the (B,H,W,6) input pair (pwcnet.py:229), the two frames stacked on the batch
 * axis (frame f of pair b at f*B + b; the encoder weights are shared, pwcnet.py:145-162) and
 * zero-padded by pad_h/pad_w at the far edges ('SAME' padding of the first stride-2 conv):
 * out (2B, H+pad_h, W+pad_w, 3). */
int qpwc_split_frames_pad_fwd(const void* in, void* out, int B, int H, int W, int pad_h, int pad_w,
                              int dtype, void* stream);

/* Upsample(scale) of a flow field (non_layers.py:183-193; pwcnet.py:55,60):
 * out (B,2h,2w,2) = scale * bilinear x2 upsampling (half-pixel centres, edge clamp) of
 * in (B,h,w,2); in_layout / out_layout QPWC_NCHW read (B,2,h,w) / write (B,2,2h,2w) instead. */
int qpwc_upsample2x_flow_fwd(const void* in, void* out, int B, int h, int w, float scale, int dtype,
                             int in_layout, int out_layout, void* stream);

/* inv_flow = -tf_warp(flow, flow) (occlusion.py:85; app/test/test_invert_flow.py:47): the flow
 * field sampled at its own targets with the tf_warp rules (warp.py:63-153), negated.
 * flow, out: (B,H,W,2) [NHWC] or (B,2,H,W) [NCHW], storage `dtype`, arithmetic fp32. */
int qpwc_invert_flow_fwd(const void* flow, void* out, int B, int H, int W, int layout, int dtype,
                         void* stream);

/* estimate_occlusion_map (occlusion.py:27-118): out (B,H,W) fp32, 1 where the pixel leaves the
 * image under `flow` or is the target of no pixel under the inverse flow above
 * (idx3 = clip(int32(p + inv_flow[p])), tensor_scatter_nd_min of zeros into ones), else 0.
 * Two launches (fill, scatter); deterministic. */
int qpwc_occlusion_fwd(const void* flow, void* out, int B, int H, int W, int layout, int dtype,
                       void* stream);

/* One 3x3 stride-1 Conv2D(padding='same', activation='Mish') of the encoder's DownConv blocks
 * (conv_aa / conv_b, non_layers.py:410-449 with use_normalizer=False, pwcnet.py:146) for C_in = C_out = C
 * in {16, 32, 64, 128, 256}, channels-last fp32: out (B, H+pad_h, W+pad_w, C), interior = Mish(conv3x3(x) + bias),
 * border (the 'SAME' padding of a following stride-2 convolution, non_layers.py:402-409) = 0.
 * weight: (9, C, C) fp32 = [ky*3+kx][out][in]; x: (B,H,W,C); all pointers 16-byte aligned. */
int qpwc_conv3x3_mish_fwd(const void* x, const void* weight, const void* bias, void* out, int B, int H,
                          int W, int C, int pad_h, int pad_w, void* stream);

/* The same fp32 layer with its products on the bf16 matrix instructions ("bf16x3", csrc/split_bf16.h): both
 * operands are split (exactly) into three bf16 values, six partial products per product, fp32 accumulation -- what is
 * dropped per product is about one fp32 rounding at worst (2^-23), 2^-28 on average.  x, bias, out as qpwc_conv3x3_mish_fwd; weight3: the (9, C, C) fp32 taps split
 * by qpwc_split_bf16x3_fwd = (3, 9, C, C) bf16. */
int qpwc_conv3x3_mish_x3_fwd(const void* x, const void* weight3, const void* bias, void* out, int B, int H,
                             int W, int C, int pad_h, int pad_w, void* stream);

/* src (n) fp32 -> out (3, n) bf16: out[0] = bf16(src), out[1] = bf16(src - out[0]), out[2] = bf16(src - out[0] -
 * out[1]), round to nearest even; src[i] == out[0][i] + out[1][i] + out[2][i] exactly (parts below the smallest normal
 * fp32 flush to zero).
 * Operands must be finite and below ~3.39e38 in magnitude: bf16(a) overflows to Inf above that and the residual
 * a - bf16(a) becomes NaN (the fp32-instruction entry points have no such restriction).
 * The weight operands of the *_x3_fwd entry points. */
int qpwc_split_bf16x3_fwd(const void* src, void* out, long long n, void* stream);

/* The same layer for fp16 storage (BASELINE configs[4]; the reference itself has no fp16 path): x, weight
 * ((9, C, C) = [ky*3+kx][out][in]) and out fp16, bias fp32; fp32 accumulation on the fp16 matrix instructions,
 * bias + Mish in fp32, one rounding to fp16 at the store.  Same shapes, padding and alignment rules. */
int qpwc_conv3x3_mish_f16_fwd(const void* x, const void* weight, const void* bias, void* out, int B, int H,
                              int W, int C, int pad_h, int pad_w, void* stream);

/* First encoder layer on the raw input pair: Split(2) (pwcnet.py:229) + both frames stacked on the
 * batch axis (shared encoder weights, pwcnet.py:145-162) + enc.0.conv_a = Conv2D(3 -> 16, 3x3, stride 2,
 * padding='same' [TensorFlow: 0 before, 1 after for even H, W], activation='Mish')
 * (non_layers.py:402-409):  pairs (B,H,W,6) fp32 (layout QPWC_NHWC) or (B,6,H,W) (QPWC_NCHW, the
 * reference's inference default, app/optical_flow/test_infer.py:52), H and W even  ->  out
 * (2B, H/2, W/2, 16) channels-last, frame f of pair b at f*B + b.
 * weight: (9, 16, 4) fp32 = [ky*3+kx][out][in, slot 3 = 0]; bias (16). */
int qpwc_first_conv_mish_fwd(const void* pairs, const void* weight, const void* bias, void* out, int B,
                             int H, int W, int layout, void* stream);

/* The same layer for fp16 storage (BASELINE configs[4]): pairs and out fp16, weight and bias fp32 as above (the
 * 27-term products are exact in fp32), one rounding at the store; pairs 4-byte, out 8-byte aligned. */
int qpwc_first_conv_mish_f16_fwd(const void* pairs, const void* weight, const void* bias, void* out, int B,
                                 int H, int W, int layout, void* stream);

/* conv_a of the second encoder level: Conv2D(16 -> 32, 3x3, stride 2, padding='same', activation='Mish')
 * (non_layers.py:402-409) on the zero-bordered output of qpwc_conv3x3_mish_fwd (pad 1, 1):
 * x_padded (B, H+1, W+1, 16) fp32 with H, W even (row H and column W zero = TensorFlow's 'SAME' padding)
 * -> out (B, H/2, W/2, 32).  weight: (9, 32, 16) fp32 = [ky*3+kx][out][in]; bias (32). */
int qpwc_conv3x3s2_mish_fwd(const void* x_padded, const void* weight, const void* bias, void* out, int B,
                            int H, int W, void* stream);

/* The same layer for C_in in {16, 32, 64, 128} -> 2 C_in outputs (conv_a of encoder levels 2..5):
 * x_padded (B, H+1, W+1, C_in), weight (9, 2 C_in, C_in), bias (2 C_in), out (B, H/2, W/2, 2 C_in). */
int qpwc_conv3x3s2_mish_c_fwd(const void* x_padded, const void* weight, const void* bias, void* out, int B,
                              int H, int W, int C_in, void* stream);

/* qpwc_conv3x3s2_mish_c_fwd for C_in in {32, 64, 128} with the products on the bf16 matrix instructions ("bf16x3",
 * csrc/split_bf16.h): weight3 = the (9, 2 C_in, C_in) fp32 taps split by qpwc_split_bf16x3_fwd = (3, 9, 2 C_in, C_in) bf16. */
int qpwc_conv3x3s2_mish_x3_fwd(const void* x_padded, const void* weight3, const void* bias, void* out, int B,
                               int H, int W, int C_in, void* stream);

/* The same layers for fp16 storage (BASELINE configs[4]): x_padded, weight ((9, 2 C_in, C_in)) and out fp16, bias
 * fp32, fp32 accumulation, one rounding at the store. */
int qpwc_conv3x3s2_mish_f16_fwd(const void* x_padded, const void* weight, const void* bias, void* out, int B,
                                int H, int W, int C_in, void* stream);

/* UpConv of the decoder (non_layers.py:196-210): Conv2DTranspose(F, 4x4, strides 2, padding='same') + bias +
 * Mish of x (B,H,W,C), C in {64,128,256}, F % 16 == 0, written into channels [0, F) of `out`
 * (B, 2H, 2W, *) whose pixels are out_pixel_stride floats apart -- with out_pixel_stride = F + C_skip this is
 * the `up` half of the decoder's concat([up, skip]) (pwcnet.py:186-195) in place.
 * weight: (16, F, C) fp32 = [ky*4+kx][out][in] (torch ConvTranspose2d weight (C,F,4,4) permuted (2,3,1,0)). */
int qpwc_upconv4x4s2_mish_fwd(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                              int C, int F, int64_t out_pixel_stride, void* stream);

/* qpwc_upconv4x4s2_mish_fwd with the products on the bf16 matrix instructions ("bf16x3", csrc/split_bf16.h):
 * weight3 = the (16, F, C) fp32 taps split by qpwc_split_bf16x3_fwd = (3, 16, F, C) bf16; everything else as there. */
int qpwc_upconv4x4s2_mish_x3_fwd(const void* x, const void* weight3, const void* bias, void* out, int B, int H, int W,
                                 int C, int F, int64_t out_pixel_stride, void* stream);

/* The same layer for fp16 storage (BASELINE configs[4]): x, weight ((16, F, C)) and out fp16, bias fp32, fp32
 * accumulation, one rounding at the store; out_pixel_stride in elements (halves), out 8-byte aligned. */
int qpwc_upconv4x4s2_mish_f16_fwd(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                                  int C, int F, int64_t out_pixel_stride, void* stream);

/* qpwc_upconv4x4s2_mish_fwd AND the skip half of the decoder's concat([up, skip]) (pwcnet.py:186-195) in one launch:
 * channels [0, F) of `out` as there, and channels [F, 2F) = the first F channels of `skip` (B, 2H, 2W, >= F), whose batch /
 * row / pixel strides are given in elements (multiples of 4; the skip may be the interior of a zero-bordered buffer).
 * out_pixel_stride >= 2F; out must not overlap x or skip.  fp32: every pointer 16-byte aligned. */
int qpwc_upconv4x4s2_mish_cat_fwd(const void* x, const void* weight, const void* bias, const void* skip,
                                  int64_t skip_batch_stride, int64_t skip_row_stride, int64_t skip_pixel_stride, void* out,
                                  int B, int H, int W, int C, int F, int64_t out_pixel_stride, void* stream);

/* The same for fp16 storage: x, weight, skip and out fp16 (out, skip 8-byte aligned), bias fp32. */
int qpwc_upconv4x4s2_mish_cat_f16_fwd(const void* x, const void* weight, const void* bias, const void* skip,
                                      int64_t skip_batch_stride, int64_t skip_row_stride, int64_t skip_pixel_stride, void* out,
                                      int B, int H, int W, int C, int F, int64_t out_pixel_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QPWC_H_ */
