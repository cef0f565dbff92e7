// Shared helpers for the gfx950 hot-path kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qpwc.h"

namespace qpwc {

constexpr int kNumXcd = 8;  // MI355X: 8 XCDs, each with a private L2

// A/B (round 4): Mish with an explicit `x > 20 ? x : m` per value (1) or without (0: the clamped exponent already gives
// m = x to 2 ulp there; one compare + one select less per activation, ~160 M activations per step)
#ifndef QPWC_MISH_SELECT
#define QPWC_MISH_SELECT 0
#endif

// qpwc_cost_volume_kernel(): the launchers of the cost-volume family run their selection rules with this
// set and report the kernel they WOULD launch instead of launching it (one rule, never a mirrored copy).
extern thread_local const char* g_dry_kernel;
extern thread_local bool g_dry_run;
inline bool dry_run(const char* name) {
    if (g_dry_run) g_dry_kernel = name;
    return g_dry_run;
}

// The flow chain (cost volume, warp, OptFlow, upsample: the critical path of a forward) shares the chip with
// the decoder's chip-filling launches on a second hardware queue; its waves ask for issue priority over
// co-resident decoder waves.  (Stream priorities do not survive hipGraph capture on this ROCm.)
#ifndef QPWC_FLOW_PRIO
#define QPWC_FLOW_PRIO 2
#endif
#define QPWC_FLOW_CHAIN_PRIO() do { if (QPWC_FLOW_PRIO > 0) __builtin_amdgcn_s_setprio(QPWC_FLOW_PRIO); } while (0)

// Blocks are dealt round-robin over the XCDs (block b and b+8 share one), so
// give every XCD a contiguous run of tiles: neighbouring tiles share their halo
// through one L2.  Bijective for any grid size; affects speed only.
__device__ __forceinline__ int xcd_swizzle(int bid, int nblk) {
    const int xcd = bid % kNumXcd;
    const int idx = bid / kNumXcd;
    const int base = nblk / kNumXcd;
    const int rem = nblk % kNumXcd;
    return xcd * base + (xcd < rem ? xcd : rem) + idx;
}

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.0f ? v : v * slope; }

template <typename T>
__device__ __forceinline__ float ld(const T* p);
template <>
__device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld<__half>(const __half* p) { return __half2float(*p); }

template <typename T>
__device__ __forceinline__ void st(T* p, float v);
template <>
__device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void st<__half>(__half* p, float v) { *p = __float2half_rn(v); }

// 4 consecutive channels as fp32, from fp32 or fp16 storage (16 B / 8 B loads).
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const __half* p) {
    const uint2 raw = *reinterpret_cast<const uint2*>(p);
    const __half2 a = *reinterpret_cast<const __half2*>(&raw.x);
    const __half2 b = *reinterpret_cast<const __half2*>(&raw.y);
    const float2 fa = __half22float2(a), fb = __half22float2(b);
    return make_float4(fa.x, fa.y, fb.x, fb.y);
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(__half* p, float4 v) {
    const __half2 a = __floats2half2_rn(v.x, v.y);
    const __half2 b = __floats2half2_rn(v.z, v.w);
    uint2 raw;
    raw.x = *reinterpret_cast<const uint32_t*>(&a);
    raw.y = *reinterpret_cast<const uint32_t*>(&b);
    *reinterpret_cast<uint2*>(p) = raw;
}

// Bilinear sampling set-up shared by the warp kernels.  All arithmetic in fp32
// with separately rounded multiplies and adds (fp contract off in these helpers).
struct Taps {
    int y0, y1, x0, x1;
    float w00, w01, w10, w11;  // tfwarp: weights of (y0,x0),(y0,x1),(y1,x0),(y1,x1)
    float ay, ax;              // clamp : lerp factors
};

// WarpV2: tfa dense_image_warp(img, -flo[..., ::-1]) -> query (y+fy, x+fx),
// interpolate_bilinear clamp-to-border (reference warp.py:157-185,207).
__device__ __forceinline__ Taps taps_clamp(int y, int x, float fx, float fy, int H, int W) {
#pragma clang fp contract(off)
    Taps t;
    const float qy = (float)y - (-fy);
    const float qx = (float)x - (-fx);
    const float fl_y = fminf(fmaxf(0.0f, floorf(qy)), (float)(H - 2));
    const float fl_x = fminf(fmaxf(0.0f, floorf(qx)), (float)(W - 2));
    t.y0 = (int)fl_y;
    t.x0 = (int)fl_x;
    t.y1 = t.y0 + 1;
    t.x1 = t.x0 + 1;
    t.ay = fminf(fmaxf(0.0f, qy - fl_y), 1.0f);
    t.ax = fminf(fmaxf(0.0f, qx - fl_x), 1.0f);
    // NaN flow: fmaxf(0, NaN) = 0 -> taps stay in range, alpha = 0.
    t.w00 = t.w01 = t.w10 = t.w11 = 0.0f;
    return t;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Warp (V1): tf_warp, reference warp.py:100-151.  int cast truncates toward
// zero; corners are clipped; weights use the clipped corners and the raw query.
__device__ __forceinline__ Taps taps_tfwarp(int y, int x, float fx, float fy, int H, int W) {
#pragma clang fp contract(off)
    Taps t;
    const float xf = (float)x + fx;
    const float yf = (float)y + fy;
    int x0 = (int)xf, y0 = (int)yf;  // v_cvt_i32_f32: truncation, saturating, NaN -> 0
    // x0 + 1 must not wrap for a saturated x0
    int x1 = x0 == INT32_MAX ? x0 : x0 + 1;
    int y1 = y0 == INT32_MAX ? y0 : y0 + 1;
    x0 = clampi(x0, 0, W - 1);
    x1 = clampi(x1, 0, W - 1);
    y0 = clampi(y0, 0, H - 1);
    y1 = clampi(y1, 0, H - 1);
    t.x0 = x0; t.x1 = x1; t.y0 = y0; t.y1 = y1;
    const float x0f = (float)x0, x1f = (float)x1, y0f = (float)y0, y1f = (float)y1;
    t.w00 = (x1f - xf) * (y1f - yf);  // wa : (x0,y0)
    t.w10 = (x1f - xf) * (yf - y0f);  // wb : (x0,y1)
    t.w01 = (xf - x0f) * (y1f - yf);  // wc : (x1,y0)
    t.w11 = (xf - x0f) * (yf - y0f);  // wd : (x1,y1)
    t.ay = t.ax = 0.0f;
    return t;
}

template <int MODE>
__device__ __forceinline__ Taps make_taps(int y, int x, float fx, float fy, int H, int W) {
    if (MODE == QPWC_WARP_CLAMP) return taps_clamp(y, x, fx, fy, H, W);
    return taps_tfwarp(y, x, fx, fy, H, W);
}

// combine the four corner values: tl=(y0,x0) tr=(y0,x1) bl=(y1,x0) br=(y1,x1)
template <int MODE>
__device__ __forceinline__ float blend(const Taps& t, float tl, float tr, float bl, float br) {
#pragma clang fp contract(off)
    if (MODE == QPWC_WARP_CLAMP) {
        const float top = t.ax * (tr - tl) + tl;
        const float bot = t.ax * (br - bl) + bl;
        return t.ay * (bot - top) + top;
    }
    // tf.add_n([wa*Ia, wb*Ib, wc*Ic, wd*Id]) with Ia=(x0,y0) Ib=(x0,y1) Ic=(x1,y0) Id=(x1,y1)
    return ((t.w00 * tl + t.w10 * bl) + t.w01 * tr) + t.w11 * br;
}

template <int MODE>
__device__ __forceinline__ float4 blend4(const Taps& t, float4 tl, float4 tr, float4 bl, float4 br) {
    return make_float4(blend<MODE>(t, tl.x, tr.x, bl.x, br.x), blend<MODE>(t, tl.y, tr.y, bl.y, br.y),
                       blend<MODE>(t, tl.z, tr.z, bl.z, br.z), blend<MODE>(t, tl.w, tr.w, bl.w, br.w));
}

// ---- host side ----------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

}  // namespace qpwc
