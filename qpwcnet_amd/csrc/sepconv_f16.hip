// fp16-storage SeparableConv2D of OptFlow (BASELINE configs[4]): its own translation unit since round 4 (the fp32
// kernels of optflow.hip take two minutes to compile; the two now build in parallel).
#ifndef QPWC_DWSRC_REGS
#define QPWC_DWSRC_REGS 0   // the branch-free source selection of optflow_common.h measured +-0 .. +2 % here (first layer, L4 B=32:
                            // 242-248 vs 238-244 us; config 5 step 1.619-1.624 vs 1.609-1.611 ms): this file keeps the if-chain
#endif
#include "optflow_common.h"

namespace qpwc {

// ---------------------------------------------------------------------------
// fp16-storage form of the fused SeparableConv2D (BASELINE configs[4]): either one dense source whose
// pixels are 16-byte aligned runs of a multiple of 8 channels (OptFlow's layers 2..4, WIDE) or the
// virtual concat of up to three sources in 8-byte aligned runs of 4 channels (the first layer's
// [cost 81 + 3 zero pads | prv | flo]; a last source of fewer than 4 channels is read element-wise); fp16
// in and out, the depthwise 3x3 in fp32 on the staged (and, on request, Mish-activated) tile, its result
// rounded to fp16 -- the same rounding point as the depthwise kernel + fp16 GEMM it replaces -- and
// the pointwise conv as ONE v_mfma_f32_16x16x32_f16 (fp32 accumulate) per accumulator and step.
// y_s / w_s rows are 64 B (32 halves) with the 16-byte chunk c of row n at chunk c ^ ((n >> 2) & 2),
// as in the fp16 cost volume.  pw: (F, Cpad) fp16, Cpad = ceil(C/32)*32; dw, bias fp32.
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
constexpr int kDwHoistC = 128;      // widest layer (padded channels) whose taps are staged whole: 4.5 KB of LDS (64 outputs: 3 workgroups per CU still fit)
#ifndef QPWC_SC16_HOIST_TAPS
#define QPWC_SC16_HOIST_TAPS 0      // A/B (round 4): measured +-0 on the wide layers, -4 % on the first layer (L4 B=32: 259-264 vs 248 us): off
#endif
#ifndef QPWC_SC16_DEEP
#define QPWC_SC16_DEEP 0            // A/B (round 4): two staged steps in flight in the one-shot form -- measured +-0 (L4 B=32: 258.3 vs 261.2, 171.4 vs 172.8 us): the kernel is not bound by its bytes in flight
#endif
#ifndef QPWC_SC16_BIAS_EARLY
#define QPWC_SC16_BIAS_EARLY 1      // A/B (round 4): bias values requested before the last matrix step
#endif
#ifndef QPWC_SC16_WIDE_STORES
#define QPWC_SC16_WIDE_STORES 0     // A/B (round 4): 16-byte output stores through v_permlane16_swap -- parity-green, +-0 (config 5 step 1.679 vs 1.656 ms with both switches on): off
#endif

template <int F, bool ACT, bool ACT_OUT, bool WIDE, bool RES = false>
__global__ __launch_bounds__(256, (RES && WIDE && F <= 32) ? 3 : 2) void sepconv3x3_fused_f16_kernel(   // resident narrow layers: 3 workgroups per CU (<= 168 registers)
    DwSrc src, const float* __restrict__ dw, const __half* __restrict__ pw, const float* __restrict__ bias,
    __half* __restrict__ out, int H, int W, int C, int cpad, int tiles_x, int tiles_y, int n_work) {
    constexpr int NFT = F / 16;
    // WIDE : one dense source, 4 lanes x 16 B (8 channels) per halo pixel, 64 halo pixels per trip
    // else : up to three sources (virtual concat), 8 lanes x 8 B (4 channels) per pixel, 32 per trip;
    //        every source but the last holds a multiple of 4 channels in 8-byte aligned pixels
    constexpr int NST = WIDE ? 3 : 6;
    constexpr int SPT = WIDE ? 64 : 32;
    __shared__ __attribute__((aligned(16))) float in_s[kScNH * kScInPS];
    __shared__ __attribute__((aligned(16))) __half y_s[2 * kScTH * kScTW * kScKC];   // double-buffered by step
    __shared__ __attribute__((aligned(16))) __half w_s[F * kScKC];
    // Round 4: the depthwise taps of the WHOLE layer are staged once (per tile; per workgroup in the resident form) with
    // 16-byte loads where they fit (<= 128 channels: every layer of L4, layers 2-4 of L3) instead of two dword loads per thread and
    // 32-channel step -- a third of the kernel's load instructions, and its busiest unit is the texture addresser
    // (23-34 cycles per wave-level memory instruction whatever its width, profiles/r03_pmc_sepconv_f16.txt).
    __shared__ __attribute__((aligned(16))) float dw_s[9 * kDwHoistC];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const bool hoist = QPWC_SC16_HOIST_TAPS && cpad <= kDwHoistC;
    const int dws = hoist ? kDwHoistC : kScKC;      // channel stride of a tap row in dw_s
    // RES (round 3, as in the fp32 kernel): resident workgroups walk the tiles blockIdx.x, + gridDim.x, ... and request
    // the next tile's first step before this tile's last matrix step; b / X0 / Y0 = the tile being FETCHED.
    int b, X0, Y0;
    auto locate = [&](int v) __attribute__((always_inline)) {
        const int tile = xcd_swizzle(v, n_work);
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y;
        b = tile / (tiles_x * tiles_y);
        X0 = tx * kScTW;
        Y0 = ty * kScTH;
    };
    locate(blockIdx.x);
    const int n = lane & 15, g = lane >> 4;

    f32x4v acc[2][NFT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int i = 0; i < NFT; ++i) acc[m][i] = f32x4v{0.f, 0.f, 0.f, 0.f};

    const int sch = WIDE ? 8 * (tid & 3) : 4 * (tid & 7);
    const int sps = WIDE ? (tid >> 2) : (tid >> 3);
    int goff[NST];
    auto set_goff = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int hp = sps + SPT * it;
            const int hy = hp / kScHW, hx = hp - hy * kScHW;
            const int gy = Y0 - 1 + hy, gx = X0 - 1 + hx;
            goff[it] = (hp < kScNH && gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
        }
    };
    set_goff();
    // one step's staged inputs (+ its taps where they are not staged whole).  Round 4: the one-shot form keeps TWO sets
    // in flight (DEEP): the kernel's HBM rate is (bytes in flight per CU) / (memory latency under load) -- two
    // workgroups x one 11.5 KB request each per ~2.5 us round trip = the 2.0-2.3 TB/s it measured -- so step k + 3 is
    // requested when step k + 1 is committed, into the register set that commit frees (12 more registers).
    struct Stage {
        uint4 st[WIDE ? NST : 1];
        uint2 st2[WIDE ? 1 : NST];
        float dreg[2];
    };
    constexpr bool DEEP = QPWC_SC16_DEEP && !RES;
    Stage sa, sb;
    uint4 wreg0, wreg1;
    wreg0 = wreg1 = make_uint4(0, 0, 0, 0);
    auto fetch_in = [&](int c0, Stage& S) __attribute__((always_inline)) {   // inputs and depthwise taps of the step at channel c0: global -> registers
        auto& st = S.st;
        auto& st2 = S.st2;
        auto& dreg = S.dreg;
        const int c = c0 + sch;
        if (WIDE) {
            const __half* sb = (const __half*)src.ptr[0] + (int64_t)b * H * W * src.stride[0];
#pragma unroll
            for (int it = 0; it < NST; ++it)
                st[it] = (goff[it] >= 0 && c < C)
                             ? ldg_u4(sb + (int64_t)goff[it] * src.stride[0] + c)   // global_load, not flat (optflow_common.h)
                             : make_uint4(0, 0, 0, 0);
        } else {
            const __half* p = nullptr;
            int ps = 0, left = 0;   // channels of the source from c on
            if (c < C) {
                const DwPick k = dwsrc_pick(src, c, C);
                p = (const __half*)k.p; ps = (int)k.ps; left = k.left;
                p += (int64_t)b * H * W * ps + k.cc;
            }
            if (left >= 4) {
#pragma unroll
                for (int it = 0; it < NST; ++it)
                    st2[it] = goff[it] >= 0 ? ldg_u2(p + (int64_t)goff[it] * ps)
                                            : make_uint2(0, 0);
            } else if (left == 2 && (ps & 1) == 0 && (reinterpret_cast<uintptr_t>(p) & 3) == 0) {
                // Flow/UpFlow's 2-channel flow: ONE 4-byte load per pixel
#pragma unroll
                for (int it = 0; it < NST; ++it)
                    st2[it] = goff[it] >= 0
                                  ? make_uint2(ldg_u1(p + (int64_t)goff[it] * ps), 0u)
                                  : make_uint2(0, 0);
            } else {   // any other short last source: element loads
#pragma unroll
                for (int it = 0; it < NST; ++it) {
                    unsigned short h[4] = {0, 0, 0, 0};
                    if (goff[it] >= 0 && left > 0) {
                        const unsigned short* q = reinterpret_cast<const unsigned short*>(p + (int64_t)goff[it] * ps);
                        h[0] = ldg_h1(q);
                        if (left > 1) h[1] = ldg_h1(q + 1);
                        if (left > 2) h[2] = ldg_h1(q + 2);
                    }
                    st2[it] = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
                }
            }
        }
        if (!hoist) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + 256 * i;
                dreg[i] = (idx < kScKC * 9 && c0 * 9 + idx < C * 9) ? dw[c0 * 9 + idx] : 0.0f;
            }
        }
    };
    auto stage_all_taps = [&]() {   // dw (C, 9) fp32 -> dw_s[tap][channel], zero for channels C .. cpad - 1
        const int nq = (cpad * 9 + 3) >> 2;
        for (int q = tid; q < nq; q += 256) {
            const int i0 = 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i0 + 3 < C * 9 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0) {
                v = *reinterpret_cast<const float4*>(dw + i0);
            } else {
                if (i0 < C * 9) v.x = dw[i0];
                if (i0 + 1 < C * 9) v.y = dw[i0 + 1];
                if (i0 + 2 < C * 9) v.z = dw[i0 + 2];
                if (i0 + 3 < C * 9) v.w = dw[i0 + 3];
            }
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j, ch = i / 9;
                if (i < cpad * 9) dw_s[(i - 9 * ch) * kDwHoistC + ch] = e[j];
            }
        }
    };
    auto fetch_w = [&](int c0) {   // pointwise slice: F rows x 4 chunks of 16 B
        const int f = tid >> 2, q = tid & 3;
        const __half* wp = pw + (int64_t)f * cpad + c0 + 8 * q;
        if (F >= 64 || f < F) wreg0 = *reinterpret_cast<const uint4*>(wp);
        if (F >= 128) wreg1 = *reinterpret_cast<const uint4*>(wp + (int64_t)64 * cpad);
    };
    auto commit_in = [&](Stage& S) __attribute__((always_inline)) {
        auto& st = S.st;
        auto& st2 = S.st2;
        auto& dreg = S.dreg;
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int hp = sps + SPT * it;
            if (hp < kScNH) {
                constexpr int NV = WIDE ? 8 : 4;
                const __half2* h2 = WIDE ? reinterpret_cast<const __half2*>(&st[WIDE ? it : 0])
                                         : reinterpret_cast<const __half2*>(&st2[WIDE ? 0 : it]);
                float v[NV];
#pragma unroll
                for (int k = 0; k < NV / 2; ++k) {
                    const float2 f2 = __half22float2(h2[k]);
                    v[2 * k] = f2.x;
                    v[2 * k + 1] = f2.y;
                }
                if (ACT && goff[it] >= 0) {
#pragma unroll
                    for (int k = 0; k < NV; ++k) v[k] = mishf(v[k]);
                }
                float* d = in_s + hp * kScInPS + sch;
                *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
                if (WIDE) *reinterpret_cast<float4*>(d + 4) = make_float4(v[NV - 4], v[NV - 3], v[NV - 2], v[NV - 1]);
            }
        }
        if (!hoist) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + 256 * i;
                if (idx < kScKC * 9) dw_s[(idx % 9) * kScKC + idx / 9] = dreg[i];
            }
        }
    };
    auto commit_w = [&]() {
        const int f = tid >> 2, q = tid & 3;
        __half* wd = w_s + f * kScKC + ((q ^ (((f & 15) >> 2) & 2)) << 3);
        if (F >= 64 || f < F) *reinterpret_cast<uint4*>(wd) = wreg0;
        if (F >= 128) *reinterpret_cast<uint4*>(wd + 64 * kScKC) = wreg1;
    };

    const int cq = tid & 7, strip = tid >> 3;
    const int drow = strip >> 2, dxs = (strip & 3) * 4;
    const int coff = n * kScKC + ((g ^ ((n >> 2) & 2)) << 3);   // halves: row n, chunk g (8 channels)

    constexpr int kYh = kScTH * kScTW * kScKC;
    auto depthwise = [&](__half* yd, int c0) {   // c0: the step's first channel (tap column when the taps are staged whole)
        {   // depthwise in fp32: 4 pixels x 4 channels per thread
            float4 wq[9];
            const float* tp = dw_s + (hoist ? c0 : 0) + 4 * cq;
#pragma unroll
            for (int k = 0; k < 9; ++k) wq[k] = *reinterpret_cast<const float4*>(tp + k * dws);
            float4 a[4];
#pragma unroll
            for (int px = 0; px < 4; ++px) a[px] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                float4 r[6];
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    r[j] = *reinterpret_cast<const float4*>(in_s + ((drow + ky) * kScHW + dxs + j) * kScInPS + 4 * cq);
#pragma unroll
                for (int px = 0; px < 4; ++px)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float4 v = r[px + kx];
                        const float4 wk = wq[ky * 3 + kx];
                        a[px].x = fmaf(wk.x, v.x, a[px].x);
                        a[px].y = fmaf(wk.y, v.y, a[px].y);
                        a[px].z = fmaf(wk.z, v.z, a[px].z);
                        a[px].w = fmaf(wk.w, v.w, a[px].w);
                    }
            }
#pragma unroll
            for (int px = 0; px < 4; ++px) {   // 4 halves = half a 16-byte chunk of the pixel's row
                const int pix = drow * kScTW + dxs + px;
                const int chunk = cq >> 1;
                st4(yd + pix * kScKC + ((chunk ^ (((pix & 15) >> 2) & 2)) << 3) + 4 * (cq & 1), a[px]);
            }
        }
    };
    auto pointwise = [&](const __half* ys) {
        {   // pointwise: one 16x16x32 matrix instruction per accumulator
            f16x8v yv[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
                yv[m] = *reinterpret_cast<const f16x8v*>(ys + (32 * wave + 16 * m) * kScKC + coff);
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft) {
                const f16x8v wv = *reinterpret_cast<const f16x8v*>(w_s + 16 * ft * kScKC + coff);
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[m][ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, yv[m], acc[m][ft], 0, 0, 0);
            }
        }
    };
    // Software pipeline, as in the fp32 kernel: the matrix work of step k and the depthwise convolution of step
    // k + 1 share one barrier interval (y_s double-buffered), inputs are requested two steps ahead of their
    // matrix work; two barriers per step instead of three serial phases.
    //   A: y_s[k&1] complete, w_s and in_s free   -> commit weights(k), inputs(k+1)
    //   B: staged                                 -> prefetch, pointwise(k) || depthwise(k+1)
    const int nsteps = cpad / kScKC;
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < NFT; ++i) acc[m][i] = f32x4v{0.f, 0.f, 0.f, 0.f};
    };
    fetch_in(0, sa);
    if (DEEP && nsteps > 1) fetch_in(kScKC, sb);
    fetch_w(0);
    if (hoist) stage_all_taps();     // (visible after the barrier that follows the first commit_in)
    int v = blockIdx.x;
    do {
        const int eb = b, eX0 = X0, eY0 = Y0;     // the tile whose outputs this iteration produces
        const bool more = RES && v + (int)gridDim.x < n_work;
        auto next_tile_request = [&]() __attribute__((always_inline)) {
            locate(v + (int)gridDim.x);
            set_goff();
            fetch_in(0, sa);
        };
        commit_in(sa);
        __syncthreads();
        if (DEEP) {
            if (nsteps > 2) fetch_in(2 * kScKC, sa);
        } else {
            if (nsteps > 1) fetch_in(kScKC, sa);
            else if (more) next_tile_request();
        }
        depthwise(y_s, 0);
        // iteration k: weights of step k and inputs of step k + 1 into LDS, requests for what comes next, then the matrix
        // work of step k beside the depthwise convolution of step k + 1
        auto iteration = [&](int k, Stage& S) __attribute__((always_inline)) {
            __syncthreads();
            commit_w();
            commit_in(S);
            __syncthreads();
            fetch_w((k + 1) * kScKC);
            if (DEEP) {
                if (k + 3 < nsteps) fetch_in((k + 3) * kScKC, S);
            } else {
                if (k + 2 < nsteps) fetch_in((k + 2) * kScKC, S);
                else if (more) next_tile_request();
            }
            if (RES && k == 0) zero_acc();
            pointwise(y_s + (k & 1) * kYh);
            depthwise(y_s + ((k + 1) & 1) * kYh, (k + 1) * kScKC);
        };
        if (DEEP) {
            for (int k = 0; k + 1 < nsteps; k += 2) {
                iteration(k, sb);                             // step k + 1 (odd) lives in set b
                if (k + 2 < nsteps) iteration(k + 1, sa);     // step k + 2 (even) in set a
            }
        } else {
            for (int k = 0; k + 1 < nsteps; ++k) iteration(k, sa);
        }
        __syncthreads();
        commit_w();
        __syncthreads();
        if (RES && nsteps == 1) zero_acc();
        // Round 4: the lane's bias values are requested BEFORE the last matrix step.  Loaded per block inside the store loop
        // each load sat behind an `s_waitcnt vmcnt(0)` in front of its use, and vmcnt counts stores on gfx950: every block
        // waited for the previous block's output stores to be acknowledged (the fp32 kernel's last step: 11.5 k cycles
        // against 5 k for the others, optflow.hip).
        float4 bvs[QPWC_SC16_BIAS_EARLY ? NFT : 1];
        if (QPWC_SC16_BIAS_EARLY) {
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft) bvs[ft] = *reinterpret_cast<const float4*>(bias + 16 * ft + 4 * g);
        }
        pointwise(y_s + ((nsteps - 1) & 1) * kYh);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int pix = 32 * wave + 16 * m + n;
            const int gy = eY0 + pix / kScTW, gx = eX0 + pix % kScTW;
            const bool ok = gy < H && gx < W;
            __half* o = out + ((int64_t)(eb * H + gy) * W + gx) * F;
            auto finish = [&](int ft) __attribute__((always_inline)) {
                const float4 bv = QPWC_SC16_BIAS_EARLY ? bvs[QPWC_SC16_BIAS_EARLY ? ft : 0]
                                                       : *reinterpret_cast<const float4*>(bias + 16 * ft + 4 * g);
                float4 z = make_float4(acc[m][ft][0] + bv.x, acc[m][ft][1] + bv.y, acc[m][ft][2] + bv.z,
                                       acc[m][ft][3] + bv.w);
                if (ACT_OUT) z = make_float4(mishf(z.x), mishf(z.y), mishf(z.z), mishf(z.w));
                return z;
            };
            if (QPWC_SC16_WIDE_STORES && NFT >= 2) {
                // Round 4: 16-byte stores.  Lane (n, g) holds channels 16 ft + 4 g .. + 3 of blocks ft and ft + 1, 8 bytes
                // each; the lanes of rows g and g ^ 1 (16 lanes apart, same pixel) trade one of the two through
                // v_permlane16_swap (swaps the odd rows of its first operand with the even rows of its second) and
                // each then owns 8 consecutive channels: even g those of block ft, odd g those of block ft + 1 --
                // half the store instructions (16 -> 8 per lane at 128 outputs; the kernel is bound by their count).
#pragma unroll
                for (int fp = 0; fp < NFT / 2; ++fp) {
                    const float4 z0 = finish(2 * fp), z1 = finish(2 * fp + 1);
                    const __half2 a0 = __floats2half2_rn(z0.x, z0.y), a1 = __floats2half2_rn(z0.z, z0.w);
                    const __half2 b0 = __floats2half2_rn(z1.x, z1.y), b1 = __floats2half2_rn(z1.z, z1.w);
                    const auto r0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a0),
                                                                     __builtin_bit_cast(unsigned, b0), false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a1),
                                                                     __builtin_bit_cast(unsigned, b1), false, false);
                    // even g: r*[0] = own block-ft pair, r*[1] = row g + 1's block-ft pair (channels 4 g + 4 ..)
                    // odd g : r*[0] = row g - 1's block-(ft + 1) pair (channels 4 g - 4 ..), r*[1] = own
                    const uint4 pk = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                    const int ch = (g & 1) ? 16 * (2 * fp + 1) + 4 * (g - 1) : 16 * (2 * fp) + 4 * g;
                    if (ok) *reinterpret_cast<uint4*>(o + ch) = pk;
                }
            } else if (ok) {
#pragma unroll
                for (int ft = 0; ft < NFT; ++ft) st4(o + 16 * ft + 4 * g, finish(ft));
            }
        }
        if (more) fetch_w(0);
    } while (RES && (v += (int)gridDim.x) < n_work);
}

#ifndef QPWC_SC16_STREAM
#define QPWC_SC16_STREAM 0   // lab note only: make ab ABSRC=optflow ABFLAGS=-DQPWC_SC16_STREAM=1
#endif
#if QPWC_SC16_STREAM
#include "experimental/sepconv_f16_stream.inc"
#endif

#ifndef QPWC_SC16_RESIDENT
#define QPWC_SC16_RESIDENT 512   // resident workgroups of the fp16 fused SeparableConv2D (0 = one workgroup per tile)
#endif
#ifndef QPWC_SC16_RES_MAXF
#define QPWC_SC16_RES_MAXF 32    // widest layer that takes the resident form.  tools/sep16bench.py, B=32, us, resident vs
                                 // one workgroup per tile (A/B build, one call): L4 F=32 65.4 vs 75.2, F=16 32.1 vs 35.4; L3 18.6 vs
                                 // 21.6, 10.5 vs 11.0 -- but F=64 170.8 vs 155.2 (172 registers: 2 instead of 3 waves per SIMD)
                                 // and F=128 266.8 vs 247.8: the narrow layers only, as in fp32
#endif
#ifndef QPWC_SC16_DIRECT
#define QPWC_SC16_DIRECT 0   // lab note (experimental/sepconv_f16_direct.inc): parity-green, slower -- never in the product build
#endif
#if QPWC_SC16_DIRECT
#include "experimental/sepconv_f16_direct.inc"
#endif

template <int F>
static void sepconv_f16_dispatch(const DwSrc& d, bool wide, int act, const float* dw, const __half* pw,
                                 const float* bias, __half* out, int H, int W, int C, int cpad, int tiles_x,
                                 int tiles_y, dim3 grid, hipStream_t s) {
    // round 3: launches of more than QPWC_SC16_RESIDENT tiles run as that many resident workgroups (2 per CU)
    const int n_work = (int)grid.x;
    const int n_res = (F <= 32 ? 3 : 2) * (QPWC_SC16_RESIDENT / 2);   // 3 / 2 workgroups per CU fit (168 / 230 registers)
    const bool resident = QPWC_SC16_RESIDENT > 0 && F <= QPWC_SC16_RES_MAXF && n_work > n_res;
    if (resident) grid.x = n_res;
#define QPWC_SCH_LAUNCH(ACT, AO, WD)                                                                           \
    do {                                                                                                       \
        if (F <= QPWC_SC16_RES_MAXF && resident)                                                               \
            hipLaunchKernelGGL((sepconv3x3_fused_f16_kernel<F, ACT, AO, WD, (F <= QPWC_SC16_RES_MAXF)>), grid,  \
                               dim3(256), 0, s, d, dw, pw, bias, out, H, W, C, cpad, tiles_x, tiles_y, n_work); \
        else                                                                                                   \
            hipLaunchKernelGGL((sepconv3x3_fused_f16_kernel<F, ACT, AO, WD, false>), grid, dim3(256), 0, s, d,  \
                               dw, pw, bias, out, H, W, C, cpad, tiles_x, tiles_y, n_work);                    \
    } while (0)
    const bool in_act = (act & 1) != 0, out_act = (act & 2) != 0;
    if (wide) {
        if (in_act) { if (out_act) QPWC_SCH_LAUNCH(true, true, true); else QPWC_SCH_LAUNCH(true, false, true); }
        else        { if (out_act) QPWC_SCH_LAUNCH(false, true, true); else QPWC_SCH_LAUNCH(false, false, true); }
    } else {
        if (in_act) { if (out_act) QPWC_SCH_LAUNCH(true, true, false); else QPWC_SCH_LAUNCH(true, false, false); }
        else        { if (out_act) QPWC_SCH_LAUNCH(false, true, false); else QPWC_SCH_LAUNCH(false, false, false); }
    }
#undef QPWC_SCH_LAUNCH
}

// wide (16-byte loads): one source, C % 8 == 0, pixel stride % 8 == 0, 16-byte aligned base;
// otherwise 8-byte loads over up to three sources -- the capi checks what each form requires
int sepconv3x3_f16_launch(const void* const* srcs, const int* chans, const int64_t* strides, int n_src, int act,
                          const void* dw, const void* pw, const void* bias, void* out, int B, int H, int W,
                          int F, hipStream_t s) {
    DwSrc d;
    int C = 0;
    for (int i = 0; i < 3; ++i) {
        d.ptr[i] = i < n_src ? srcs[i] : nullptr;
        d.ch[i] = i < n_src ? chans[i] : 0;
        d.stride[i] = i < n_src ? strides[i] : 0;
        C += d.ch[i];
    }
    const bool wide = n_src == 1 && C % 8 == 0 && strides[0] % 8 == 0 && reinterpret_cast<uintptr_t>(srcs[0]) % 16 == 0;
    const int cpad = (C + kScKC - 1) / kScKC * kScKC;
    const int tiles_x = (W + kScTW - 1) / kScTW, tiles_y = (H + kScTH - 1) / kScTH;
    const int64_t nblk = (int64_t)tiles_x * tiles_y * B;
    if (nblk > INT32_MAX) {
        set_error("sepconv3x3_f16: too many tiles");
        return QPWC_E_SHAPE;
    }
    const dim3 grid((unsigned)nblk);
    const __half* hp = (const __half*)pw;
    const float *fdw = (const float*)dw, *fb = (const float*)bias;
#if QPWC_SC16_STREAM
    // lab note (experimental/sepconv_f16_stream.inc): parity-green, slower -- never in the product build
    if ((act & 1) == 0 && (int64_t)((W + kS2T - 1) / kS2T) * ((H + kS2T - 1) / kS2T) * B <= INT32_MAX) {
        const bool oa = (act & 2) != 0;
        switch (F) {
            case 128: sepconv_f16_stream_dispatch<128>(d, wide, oa, fdw, hp, fb, (__half*)out, H, W, C, cpad, B, s); break;
            case 64: sepconv_f16_stream_dispatch<64>(d, wide, oa, fdw, hp, fb, (__half*)out, H, W, C, cpad, B, s); break;
            case 32: sepconv_f16_stream_dispatch<32>(d, wide, oa, fdw, hp, fb, (__half*)out, H, W, C, cpad, B, s); break;
            case 16: sepconv_f16_stream_dispatch<16>(d, wide, oa, fdw, hp, fb, (__half*)out, H, W, C, cpad, B, s); break;
            default: set_error("sepconv3x3_f16: unsupported filter count %d (16/32/64/128)", F); return QPWC_E_SHAPE;
        }
        return check_launch("sepconv3x3_f16_stream_kernel");
    }
#endif
#if QPWC_SC16_DIRECT
    if (F >= QPWC_SC16_DIRECT_MINF) {
        switch (F) {
            case 128: sepconv_f16_direct_dispatch<128>(d, wide, act, fdw, hp, fb, (__half*)out, H, W, C, cpad, tiles_x, tiles_y, grid, s); break;
            case 64: sepconv_f16_direct_dispatch<64>(d, wide, act, fdw, hp, fb, (__half*)out, H, W, C, cpad, tiles_x, tiles_y, grid, s); break;
            case 32: sepconv_f16_direct_dispatch<32>(d, wide, act, fdw, hp, fb, (__half*)out, H, W, C, cpad, tiles_x, tiles_y, grid, s); break;
            case 16: sepconv_f16_direct_dispatch<16>(d, wide, act, fdw, hp, fb, (__half*)out, H, W, C, cpad, tiles_x, tiles_y, grid, s); break;
            default: set_error("sepconv3x3_f16: unsupported filter count %d (16/32/64/128)", F); return QPWC_E_SHAPE;
        }
        return check_launch("sepconv3x3_f16_direct_kernel");
    }
#endif
    switch (F) {
        case 128: sepconv_f16_dispatch<128>(d, wide, act, fdw, hp, fb, (__half*)out, H, W, C, cpad, tiles_x, tiles_y, grid, s); break;
        case 64: sepconv_f16_dispatch<64>(d, wide, act, fdw, hp, fb, (__half*)out, H, W, C, cpad, tiles_x, tiles_y, grid, s); break;
        case 32: sepconv_f16_dispatch<32>(d, wide, act, fdw, hp, fb, (__half*)out, H, W, C, cpad, tiles_x, tiles_y, grid, s); break;
        case 16: sepconv_f16_dispatch<16>(d, wide, act, fdw, hp, fb, (__half*)out, H, W, C, cpad, tiles_x, tiles_y, grid, s); break;
        default: set_error("sepconv3x3_f16: unsupported filter count %d (16/32/64/128)", F); return QPWC_E_SHAPE;
    }
    return check_launch("sepconv3x3_fused_f16_kernel");
}


}  // namespace qpwc
