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
#if QPWC_MISH_SELECT
    return x > 20.0f ? x : m;
#else
    return m;   // x > 20: e is clamped, t / (t + 2) rounds to 1 +- 1 ulp, m = x to 2 ulp -- no compare + select per value
#endif
}

typedef float f32x4v __attribute__((ext_vector_type(4)));

#ifndef QPWC_DWSRC_REGS
#define QPWC_DWSRC_REGS 1   // A/B (round 4): see dwsrc_in_registers()
#endif

struct DwSrc {
    const void* ptr[3];
    int ch[3];          // channels taken from each source (0 = unused)
    int64_t stride[3];  // floats per pixel of each source
};

// Round 4: a DwSrc arrives by value, i.e. in the kernel-argument segment, and `cond ? src.ptr[0] : src.ptr[1]` under a
// per-lane condition is folded by the optimiser into ONE load through a per-lane selected ADDRESS of the field: a
// `global_load_dwordx2` from the argument segment, `s_waitcnt vmcnt(0)` (which also drains every load issued before it),
// 64-bit address arithmetic on the result, and only then the data loads -- two dependent memory round trips in front of
// every step's request, in every kernel that reads a virtual concat (found in the ISA behind the step's second barrier;
// it is the "request phase of 2.8-5.9 k cycles" of the phase stamps).  The fields pass through an opaque scalar-register
// barrier -- no: both a local struct copy and nine scalars behind an asm barrier ended up in SCRATCH memory, indexed the same
// way (the branches' loads are merged into one load through a selected address before the locals are promoted to
// registers).  What works is to have no branch at all: dwsrc_pick() below.
// which source of the virtual concat holds channel c -- BRANCH-FREE: all nine fields are read unconditionally (scalar
// loads of kernel arguments) and combined with masks; with QPWC_DWSRC_REGS = 0 the old if / else-if chain (A/B)
struct DwPick {
    const void* p;   // base pointer of the source
    int64_t ps;      // its pixel stride (elements)
    int cc;          // channel inside the source
    int left;        // channels of the source from c on (C - c in the last one)
};
__device__ __forceinline__ DwPick dwsrc_pick(const DwSrc& s, int c, int C) {
    DwPick r;
#if QPWC_DWSRC_REGS
    const uint64_t p0 = (uint64_t)s.ptr[0], p1 = (uint64_t)s.ptr[1], p2 = (uint64_t)s.ptr[2];
    const int64_t s0 = s.stride[0], s1 = s.stride[1], s2 = s.stride[2];
    const int e0 = s.ch[0], e1 = e0 + s.ch[1];
    const bool in0 = c < e0, in1 = !in0 && c < e1;
    const uint64_t m0 = 0ull - (uint64_t)in0, m1 = 0ull - (uint64_t)in1, m2 = ~(m0 | m1);
    r.p = (const void*)((p0 & m0) | (p1 & m1) | (p2 & m2));
    r.ps = (int64_t)(((uint64_t)s0 & m0) | ((uint64_t)s1 & m1) | ((uint64_t)s2 & m2));
    r.cc = c - ((e0 & (int)m1) | (e1 & (int)m2));
    r.left = ((e0 & (int)m0) | (e1 & (int)m1) | (C & (int)m2)) - c;
#else
    if (c < s.ch[0]) {
        r.p = s.ptr[0]; r.ps = s.stride[0]; r.cc = c; r.left = s.ch[0] - c;
    } else if (c < s.ch[0] + s.ch[1]) {
        r.p = s.ptr[1]; r.ps = s.stride[1]; r.cc = c - s.ch[0]; r.left = s.ch[0] + s.ch[1] - c;
    } else {
        r.p = s.ptr[2]; r.ps = s.stride[2]; r.cc = c - s.ch[0] - s.ch[1]; r.left = C - c;
    }
#endif
    return r;
}

// ---- loads that are GLOBAL, said so (round 4) ----
// A pointer that comes out of a by-value struct (DwSrc) or out of a select against a __device__ constant is a GENERIC
// pointer to the compiler: it emits flat_load_*, and a flat load counts on lgkmcnt as well as on vmcnt -- so every
// `s_waitcnt lgkmcnt(0)` in front of an LDS operand (the first ds_read of a matrix or depthwise phase) also waits for
// the whole prefetch that was issued a moment earlier.  The fused SeparableConv2D kernels requested a step's inputs
// "one step ahead" since round 1 and paid their full memory latency at the top of every step: the SUM of memory time
// and matrix time that every profile of these kernels showed (found in round 4 with s_memtime stamps around the
// request: 4.3-5.4 k cycles for nine load instructions).  These helpers cast to address space 1: global_load_*, vmcnt
// only.
#ifndef QPWC_LDG_GLOBAL
#define QPWC_LDG_GLOBAL 1   // 0 = generic pointers again (A/B builds: what rounds 1-3 ran)
#endif
#if QPWC_LDG_GLOBAL
#define QPWC_GLOBAL_AS __attribute__((address_space(1)))
#else
#define QPWC_GLOBAL_AS
#endif
typedef float qpwc_f32x4g __attribute__((ext_vector_type(4)));
typedef unsigned qpwc_u32x4g __attribute__((ext_vector_type(4)));
typedef unsigned qpwc_u32x2g __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ldg_f4(const float* p) {
    const qpwc_f32x4g v = *(const QPWC_GLOBAL_AS qpwc_f32x4g*)p;
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float ldg_f1(const float* p) { return *(const QPWC_GLOBAL_AS float*)p; }
__device__ __forceinline__ uint4 ldg_u4(const void* p) {
    const qpwc_u32x4g v = *(const QPWC_GLOBAL_AS qpwc_u32x4g*)p;
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint2 ldg_u2(const void* p) {
    const qpwc_u32x2g v = *(const QPWC_GLOBAL_AS qpwc_u32x2g*)p;
    return make_uint2(v[0], v[1]);
}
__device__ __forceinline__ unsigned ldg_u1(const void* p) { return *(const QPWC_GLOBAL_AS unsigned*)p; }
__device__ __forceinline__ unsigned short ldg_h1(const void* p) { return *(const QPWC_GLOBAL_AS unsigned short*)p; }

// tile / step geometry of the fused SeparableConv2D kernels (fp32 and fp16 storage)
constexpr int kScKC = 32;               // channels per step
constexpr int kScTH = 8, kScTW = 16;    // pixel tile
constexpr int kScHH = kScTH + 2, kScHW = kScTW + 2;
constexpr int kScNH = kScHH * kScHW;    // 180 halo pixels
constexpr int kScInPS = 40;             // floats per halo pixel in in_s

}  // namespace qpwc
