/* CPU restatement of the qpwcnet CostVolume + Warp hot path, plain C.
 *
 * TEST INFRASTRUCTURE ONLY: linked/called by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg.  Never by the product path.
 *
 * PARITY UNPINNED: the reference holds no golden vectors and TensorFlow /
 * tensorflow-addons cannot run offline; pinned by the in-tree reference source
 * and analytic known answers only (see oracle/__init__.py).
 *
 * Reference lines restated (paths relative to the reference checkout):
 *   cost volume : qpwcnet/core/layers.py:72-100   (CostVolume.call)
 *   Warp   (V1) : qpwcnet/core/warp.py:63-153     (tf_warp) + :8-47
 *   WarpV2      : qpwcnet/core/layers.py:177-186 + qpwcnet/core/warp.py:156-211
 *                 (tfa dense_image_warp / interpolate_bilinear, clamp-to-border)
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off keeps every multiply and add separately rounded, like the
 * reference's unfused TF op graph.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* element offset of (b,y,x,c); layout 0 = NHWC, 1 = NCHW */
static inline size_t idx(int layout, int H, int W, int C, int b, int y, int x, int c) {
    if (layout == 0) return (((size_t)b * H + y) * W + x) * C + c;
    return (((size_t)b * C + c) * H + y) * W + x;
}

/* out[b,y,x,i0*d+j0] = lrelu( mean_c prv[b,y,x,c] * pad_nxt[b,y+i0,x+j0,c] )
 * layers.py:80-81 loop order (i0 rows outer, j0 cols inner), :94 mean, :99 lrelu.
 * acc_double != 0 accumulates in double (high-precision answer), else float. */
int oracle_cost_volume(const float* prv, const float* nxt, float* out,
                       int B, int H, int W, int C, int r, int layout, int acc_double) {
    const int d = 2 * r + 1, D = d * d;
    if (layout != 0 && layout != 1) return -1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int i0 = 0; i0 < d; ++i0)
                    for (int j0 = 0; j0 < d; ++j0) {
                        const int yy = y + i0 - r, xx = x + j0 - r;
                        float res;
                        if (yy < 0 || yy >= H || xx < 0 || xx >= W) {
                            res = 0.0f; /* zero padding, layers.py:50-51,77 */
                        } else if (acc_double) {
                            double s = 0.0;
                            for (int c = 0; c < C; ++c)
                                s += (double)prv[idx(layout, H, W, C, b, y, x, c)] *
                                     (double)nxt[idx(layout, H, W, C, b, yy, xx, c)];
                            res = (float)(s / (double)C);
                        } else {
                            float s = 0.0f;
                            for (int c = 0; c < C; ++c)
                                s += prv[idx(layout, H, W, C, b, y, x, c)] *
                                     nxt[idx(layout, H, W, C, b, yy, xx, c)];
                            res = s / (float)C;
                        }
                        res = res > 0.0f ? res : res * 0.1f;
                        out[idx(layout, H, W, D, b, y, x, i0 * d + j0)] = res;
                    }
    return 0;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* flow element (b,y,x,ch) with explicit element strides, 0 = broadcast dim */
static inline float flo_at(const float* flo, const int64_t* fs, int b, int y, int x, int ch) {
    return flo[(size_t)b * fs[0] + (size_t)y * fs[1] + (size_t)x * fs[2] + (size_t)ch * fs[3]];
}

/* mode 0 = WarpV2 (tfa clamp-to-border), mode 1 = Warp (tf_warp).
 * fs = element strides of the flow tensor for (b,y,x,channel). */
int oracle_warp(const float* img, const float* flo, float* out,
                int B, int H, int W, int C, const int64_t* fs, int layout, int mode) {
    if (layout != 0 && layout != 1) return -1;
    if (mode == 0 && (H < 2 || W < 2)) return -2; /* warp.py:182-184 */
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const float fx = flo_at(flo, fs, b, y, x, 0);
                const float fy = flo_at(flo, fs, b, y, x, 1);
                if (mode == 0) {
                    /* query = grid - (-flo[..., ::-1]) = (y + fy, x + fx) */
                    const float qy = (float)y - (-fy);
                    const float qx = (float)x - (-fx);
                    float fl_y = fminf(fmaxf(0.0f, floorf(qy)), (float)(H - 2));
                    float fl_x = fminf(fmaxf(0.0f, floorf(qx)), (float)(W - 2));
                    const int y0 = (int)fl_y, x0 = (int)fl_x;
                    const int y1 = y0 + 1, x1 = x0 + 1;
                    float ay = fminf(fmaxf(0.0f, qy - fl_y), 1.0f);
                    float ax = fminf(fmaxf(0.0f, qx - fl_x), 1.0f);
                    for (int c = 0; c < C; ++c) {
                        const float tl = img[idx(layout, H, W, C, b, y0, x0, c)];
                        const float tr = img[idx(layout, H, W, C, b, y0, x1, c)];
                        const float bl = img[idx(layout, H, W, C, b, y1, x0, c)];
                        const float br = img[idx(layout, H, W, C, b, y1, x1, c)];
                        const float top = ax * (tr - tl) + tl;
                        const float bot = ax * (br - bl) + bl;
                        out[idx(layout, H, W, C, b, y, x, c)] = ay * (bot - top) + top;
                    }
                } else {
                    const float xf = (float)x + fx;              /* warp.py:102 */
                    const float yf = (float)y + fy;
                    int x0 = (int)xf, y0 = (int)yf;              /* warp.py:115,117 truncation */
                    int x1 = x0 + 1, y1 = y0 + 1;
                    x0 = clampi(x0, 0, W - 1);                   /* warp.py:121-124 */
                    x1 = clampi(x1, 0, W - 1);
                    y0 = clampi(y0, 0, H - 1);
                    y1 = clampi(y1, 0, H - 1);
                    const float wa = ((float)x1 - xf) * ((float)y1 - yf); /* warp.py:139-142 */
                    const float wb = ((float)x1 - xf) * (yf - (float)y0);
                    const float wc = (xf - (float)x0) * ((float)y1 - yf);
                    const float wd = (xf - (float)x0) * (yf - (float)y0);
                    for (int c = 0; c < C; ++c) {
                        const float Ia = img[idx(layout, H, W, C, b, y0, x0, c)];
                        const float Ib = img[idx(layout, H, W, C, b, y1, x0, c)];
                        const float Ic = img[idx(layout, H, W, C, b, y0, x1, c)];
                        const float Id = img[idx(layout, H, W, C, b, y1, x1, c)];
                        out[idx(layout, H, W, C, b, y, x, c)] =
                            ((wa * Ia + wb * Ib) + wc * Ic) + wd * Id; /* warp.py:151 */
                    }
                }
            }
    return 0;
}

/* mean over (b,y,x) of the L2 norm over the 2 flow channels, train.py:247-253.
 * Inputs NHWC (B,H,W,2). */
double oracle_epe(const float* a, const float* b, int B, int H, int W) {
    const size_t n = (size_t)B * H * W;
    double s = 0.0;
#pragma omp parallel for reduction(+ : s)
    for (size_t i = 0; i < n; ++i) {
        const double dx = (double)a[2 * i] - (double)b[2 * i];
        const double dy = (double)a[2 * i + 1] - (double)b[2 * i + 1];
        s += sqrt(dx * dx + dy * dy);
    }
    return s / (double)n;
}

/* tensorflow-addons CorrelationCost (the op behind CostVolumeV2, layers.py:124-132), CPU
 * functor restated loop for loop from its published algorithm (tensorflow_addons, version
 * unpinned by the reference; see oracle/tfa_ref.py for the statement and its caveats).
 * fp32 accumulation in the op's order (kernel rows, kernel cols, channels), then /= K.
 * The op's output is NCHW; out_layout 0 applies the Keras layer's transpose to NHWC.
 * in_layout: 0 = NHWC, 1 = NCHW.  apply_lrelu != 0 adds CostVolumeV2's leaky_relu(0.1). */
int oracle_correlation_cost(const float* a, const float* b, float* out, int N, int H, int W, int C,
                            int kernel_size, int max_displacement, int stride_1, int stride_2,
                            int pad, int in_layout, int out_layout, int apply_lrelu) {
    if ((in_layout != 0 && in_layout != 1) || (out_layout != 0 && out_layout != 1)) return -1;
    if (kernel_size % 2 != 1 || stride_1 < 1 || stride_2 < 1) return -2;
    const int kernel_rad = (kernel_size - 1) / 2;
    const int border = max_displacement + kernel_rad;
    const int oH = (int)ceil((double)(H + 2 * pad - 2 * border) / (double)stride_1);
    const int oW = (int)ceil((double)(W + 2 * pad - 2 * border) / (double)stride_1);
    if (oH < 1 || oW < 1) return -3;
    const int disp_rad = max_displacement / stride_2;
    const int disp_size = 2 * disp_rad + 1, oC = disp_size * disp_size;
    const float K = (float)(kernel_size * kernel_size * C);
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < oH; ++h) {
            const int h1 = (h - pad) * stride_1 + max_displacement + kernel_rad;
            for (int w = 0; w < oW; ++w) {
                const int w1 = (w - pad) * stride_1 + max_displacement + kernel_rad;
                for (int tj = -disp_rad; tj <= disp_rad; ++tj)
                    for (int ti = -disp_rad; ti <= disp_rad; ++ti) {
                        const int tc = (tj + disp_rad) * disp_size + (ti + disp_rad);
                        const int w2 = w1 + ti * stride_2, h2 = h1 + tj * stride_2;
                        float acc = 0.0f;
                        for (int j = -kernel_rad; j <= kernel_rad; ++j) {
                            if (h1 + j < 0 || h1 + j >= H || h2 + j < 0 || h2 + j >= H) continue;
                            for (int i = -kernel_rad; i <= kernel_rad; ++i) {
                                if (w1 + i < 0 || w1 + i >= W || w2 + i < 0 || w2 + i >= W) continue;
                                for (int c = 0; c < C; ++c)
                                    acc += a[idx(in_layout, H, W, C, n, h1 + j, w1 + i, c)] *
                                           b[idx(in_layout, H, W, C, n, h2 + j, w2 + i, c)];
                            }
                        }
                        acc /= K;
                        if (apply_lrelu) acc = acc > 0.0f ? acc : acc * 0.1f;
                        out[idx(out_layout, oH, oW, oC, n, h, w, tc)] = acc;
                    }
            }
        }
    return 0;
}

int oracle_version(void) { return 2; }
