// Shared by the OptFlow translation units (optflow.hip: fp32 kernels; sepconv_f16.hip: fp16-storage SeparableConv2D).
#pragma once
#include "common.h"

namespace qpwc {

// mish(x) = x * tanh(softplus(x)) = x * t / (t + 2),  t = e^x (e^x + 2); x > 20 -> x
// (torch's softplus threshold).  ~2 ulp with the fast exp/div.
__device__ __forceinline__ float mishf(float x) {
    // v_exp_f32 / v_rcp_f32 directly: hipcc lowers __expf with denormal range handling and __fdividef to
    // the full IEEE division sequence (div_scale / div_fmas / div_fixup), ~28 instructions per value
    const float e = __builtin_amdgcn_exp2f(fminf(x, 20.0f) * 1.4426950408889634f);
    const float t = e * (e + 2.0f);
    const float m = x * (t * __builtin_amdgcn_rcpf(t + 2.0f));
    return x > 20.0f ? x : m;
}

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct DwSrc {
    const void* ptr[3];
    int ch[3];          // channels taken from each source (0 = unused)
    int64_t stride[3];  // floats per pixel of each source
};

// tile / step geometry of the fused SeparableConv2D kernels (fp32 and fp16 storage)
constexpr int kScKC = 32;               // channels per step
constexpr int kScTH = 8, kScTW = 16;    // pixel tile
constexpr int kScHH = kScTH + 2, kScHW = kScTW + 2;
constexpr int kScNH = kScHH * kScHW;    // 180 halo pixels
constexpr int kScInPS = 40;             // floats per halo pixel in in_s

}  // namespace qpwc
