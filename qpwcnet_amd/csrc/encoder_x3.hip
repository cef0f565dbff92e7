// The encoder's 3x3 stride-1 'same' convolution + bias + Mish (conv_aa / conv_b, qpwcnet/core/non_layers.py:410-449
// with use_normalizer=False, pwcnet.py:146), fp32 in and out, with the products on the bf16 matrix instructions as
// three-way splits of both operands (split_bf16.h: six partial products, about one fp32 rounding per product at worst, 2^-28 on average).
// The fp32-instruction kernels of encoder.hip run these layers at 70-82 TF (0.45-0.5 of the fp32 matrix peak); here
// the matrix work of a layer (2.4 GFLOP x 6 partial products) is 7-8 us of the bf16 pipe and the layer is bound by
// staging and its activations' bytes instead.
//
// The input tile (with its 3x3 halo) is split ONCE per workgroup while it is staged: LDS holds three bf16 images of
// the tile (pixels of C halves, 16-byte chunk q of halo pixel p at slot x3_slot(q, p) -- the layout of the fp16
// kernels); the weights are split once on the host side of the boundary (qpwc_split_bf16x3_fwd) and streamed as
// 16-byte operands.
#include "common.h"
#include "split_bf16.h"

namespace qpwc {

namespace {

__device__ __forceinline__ float x3_mishf(float x) {
    const float e = __builtin_amdgcn_exp2f(fminf(x, 20.0f) * 1.4426950408889634f);
    const float t = e * (e + 2.0f);
    const float m = x * (t * __builtin_amdgcn_rcpf(t + 2.0f));
#if QPWC_MISH_SELECT
    return x > 20.0f ? x : m;
#else
    return m;   // x > 20: e is clamped, t / (t + 2) rounds to 1 +- 1 ulp, m = x to 2 ulp -- no compare + select per value
#endif
}

constexpr int kX3TW = 16, kX3HW = kX3TW + 2;

// 16-byte chunk q of halo pixel hp within a pixel of C bf16 values (32 / 64 / 128 / 256 / 512 bytes): the sixteen
// pixels one matrix operand read covers must spread over the 64 banks
template <int C>
__device__ __forceinline__ int x3_slot(int q, int hp) {
    return C == 16 ? (q ^ ((hp >> 3) & 1)) : (C == 32 ? (q ^ ((0 - (hp >> 2)) & 3)) : (C == 64 ? (q ^ ((hp >> 1) & 7)) : (q ^ (hp & 15))));
}

// Stage the (TH + 2) x 18 halo tile of all C channels as three bf16 images (zero outside the image): all global
// loads first, then split + LDS writes.  s1 = image 1; images 2 and 3 follow at PL halves each.
template <int C, int TH>
__device__ __forceinline__ void x3_stage_tile(const float* __restrict__ xb, unsigned short* s1, int tid, int Y0, int X0,
                                              int H, int W) {
    constexpr int NQ = C / 8, NH = (TH + 2) * kX3HW, PL = NH * C;
    constexpr int NST = (NH * NQ + 255) / 256;
    float4 st[NST][2];
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int idx = tid + 256 * it;
        const int hp = idx / NQ, q = idx - hp * NQ;
        const int hy = hp / kX3HW, hx = hp - hy * kX3HW;
        const int gy = Y0 - 1 + hy, gx = X0 - 1 + hx;
        const bool ok = idx < NH * NQ && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const float* p = xb + ((int64_t)gy * W + gx) * C + 8 * q;
        st[it][0] = ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
        st[it][1] = ok ? *reinterpret_cast<const float4*>(p + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int idx = tid + 256 * it;
        const int hp = idx / NQ, q = idx - hp * NQ;
        if (idx < NH * NQ) {
            uint4 p1, p2, p3;
            split8_bf16x3(st[it][0], st[it][1], p1, p2, p3);
            unsigned short* d = s1 + hp * C + 8 * x3_slot<C>(q, hp);
            *reinterpret_cast<uint4*>(d) = p1;
            *reinterpret_cast<uint4*>(d + PL) = p2;
            *reinterpret_cast<uint4*>(d + 2 * PL) = p3;
        }
    }
}

// zero border of the padded output (columns W.., rows H..) for channels [c0, c0 + 4 NQF) -- the 'SAME' padding the
// following stride-2 convolution reads -- written by the edge tiles
template <int TH>
__device__ __forceinline__ void x3_zero_border(float* ob, int tid, int Y0, int X0, int H, int W, int pad_h, int pad_w,
                                               int C, int c0, int NQF) {
    const int Wo = W + pad_w;
    if (pad_w > 0 && X0 + kX3TW >= W) {
        for (int i = tid; i < TH * pad_w * NQF; i += 256) {
            const int q = i % NQF, r = i / NQF, col = r % pad_w, row = r / pad_w;
            const int gy = Y0 + row;
            if (gy < H) *reinterpret_cast<float4*>(ob + ((int64_t)gy * Wo + W + col) * C + c0 + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (pad_h > 0 && Y0 + TH >= H) {
        const int x_end = (X0 + kX3TW >= W) ? Wo : X0 + kX3TW;   // the corner belongs to the last tile
        for (int i = tid; i < pad_h * (x_end - X0) * NQF; i += 256) {
            const int q = i % NQF, r = i / NQF, col = r % (x_end - X0), row = r / (x_end - X0);
            *reinterpret_cast<float4*>(ob + ((int64_t)(H + row) * Wo + X0 + col) * C + c0 + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// Wide levels (C = 64 / 128 / 256): a wave owns ONE block of 16 outputs for all TH rows of a TH x 16 pixel tile;
// workgroup = 4 waves = 64 outputs, grid = tiles x C / 64.  One step = one tap of one 32-channel block: three
// 16-byte weight operands per lane (streamed from L1 / L2 two steps ahead, ring of three) and, per tile row, three
// ds_read_b128 that feed six matrix instructions.  TH = 8 / 4 / 2: 256 workgroups and 864 matrix instructions per
// wave at every level of the 256x512 pyramid.
// w3: [3 images][9 taps][C out][C in] bf16.
template <int C, int TH>
__global__ __launch_bounds__(256, (3 * (TH + 2) * kX3HW * C * 2 > 80 * 1024 || TH >= 8) ? 1 : 2) void conv3x3_mish_x3_wide_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ w3, const float* __restrict__ bias,
    float* __restrict__ out, int H, int W, int pad_h, int pad_w, int tiles_x, int tiles_y, int n_tiles) {
    constexpr int NH = (TH + 2) * kX3HW, PL = NH * C;
    constexpr int NKB = C / 32;
    __shared__ __attribute__((aligned(16))) unsigned short in_s[3 * PL];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int slice = blockIdx.x / n_tiles;                       // 64 outputs
    const int tile = xcd_swizzle(blockIdx.x % n_tiles, n_tiles);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int X0 = tx * kX3TW, Y0 = ty * TH;
    const int fo = 64 * slice + 16 * wave;                        // this wave's output block
    const float* xb = x + (int64_t)b * H * W * C;

    // weights of output row fo + n, input channels 32 kb + 8 g .. + 7, tap: one 16-byte operand per image; a ring of
    // nine steps, requested eight steps (one 32-channel block) ahead of their use
    const unsigned short* wl = w3 + (int64_t)(fo + n) * C + 8 * g;
    uint4 wr[9][3];
    auto load_w = [&](uint4 (&w)[3], int kb, int tap) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
            w[p] = *reinterpret_cast<const uint4*>(wl + (int64_t)(p * 9 + tap) * C * C + 32 * kb);
    };
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) load_w(wr[tap], 0, tap);
    x3_stage_tile<C, TH>(xb, in_s, tid, Y0, X0, H, W);
    f32x4s acc[TH];
#pragma unroll
    for (int m = 0; m < TH; ++m) acc[m] = f32x4s{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll 1
    for (int kb = 0; kb < NKB; ++kb) {
        // the operand addresses depend on kb through the swizzle: derived from opaque copies of the lane ids they are
        // recomputed per block instead of staying live (72 registers) across the loop
        int nn = n, gg = g;
        asm volatile("" : "+v"(nn), "+v"(gg));
        // one step = one tile row of one tap: three ds_read_b128 feed six matrix instructions; the reads run two
        // steps ahead (ring of three) -- with one wave per SIMD nothing else hides their latency
        uint4 bb[3][3];
        auto read_b = [&](uint4 (&bv)[3], int i) __attribute__((always_inline)) {
            const int tap = i / TH, m = i - tap * TH, ky = tap / 3, kx = tap - 3 * ky;
            const int hp = (m + ky) * kX3HW + nn + kx;
            const unsigned short* bp = in_s + hp * C + 8 * x3_slot<C>(4 * kb + gg, hp);
            bv[0] = *reinterpret_cast<const uint4*>(bp);
            bv[1] = *reinterpret_cast<const uint4*>(bp + PL);
            bv[2] = *reinterpret_cast<const uint4*>(bp + 2 * PL);
        };
        read_b(bb[0], 0);
        read_b(bb[1], 1);
#pragma unroll
        for (int i = 0; i < 9 * TH; ++i) {
            const int tap = i / TH, m = i - tap * TH;
            if (m == 0) {
                const int t2 = (tap + 8) % 9, kb2 = tap + 8 >= 9 ? kb + 1 : kb;
                if (kb2 < NKB) load_w(wr[t2], kb2, t2);
            }
            if (i + 2 < 9 * TH) read_b(bb[(i + 2) % 3], i + 2);
            const uint4 (&w)[3] = wr[tap];
            const uint4 (&bv)[3] = bb[i % 3];
            acc[m] = mfma_bf16x3(w[0], w[1], w[2], bv[0], bv[1], bv[2], acc[m]);
            __builtin_amdgcn_sched_barrier(0);   // keeps the reads two steps ahead and bounds the registers
        }
    }
    // ---- bias + Mish: lane = pixel n of tile row m, outputs fo + 4g .. + 3 ----
    const int Wo = W + pad_w;
    float* ob = out + (int64_t)b * (H + pad_h) * Wo * C;
    const float4 bq = *reinterpret_cast<const float4*>(bias + fo + 4 * g);
#pragma unroll
    for (int m = 0; m < TH; ++m) {
        const int gy = Y0 + m, gx = X0 + n;
        if (gy < H && gx < W)
            *reinterpret_cast<float4*>(ob + ((int64_t)gy * Wo + gx) * C + fo + 4 * g) =
                make_float4(x3_mishf(acc[m][0] + bq.x), x3_mishf(acc[m][1] + bq.y), x3_mishf(acc[m][2] + bq.z),
                            x3_mishf(acc[m][3] + bq.w));
    }
    x3_zero_border<TH>(ob, tid, Y0, X0, H, W, pad_h, pad_w, C, 64 * slice, 16);
}

// ---------------------------------------------------------------------------
// Narrow levels (C = 16 / 32): 16 x 16 pixel tile, a wave owns four tile rows and ALL outputs.
// C = 32: per output block the 9 taps x 3 images of weights sit in 108 registers for the tile.
// C = 16: one matrix instruction covers TWO taps (k-slots g = 0, 1: tap 2 j, channels 0-7 / 8-15; g = 2, 3: tap
// 2 j + 1), five tap pairs, the tenth half zero: 30 instructions per 16 pixels x 16 outputs.
template <int C>
__global__ __launch_bounds__(256, 2) void conv3x3_mish_x3_narrow_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ w3, const float* __restrict__ bias,
    float* __restrict__ out, int H, int W, int pad_h, int pad_w, int tiles_x, int tiles_y) {
    constexpr int TH = 16, RW = 4;
    constexpr int NH = (TH + 2) * kX3HW, PL = NH * C;
    constexpr int NFT = C / 16;
    __shared__ __attribute__((aligned(16))) unsigned short in_s[3 * PL];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int X0 = tx * kX3TW, Y0 = ty * TH;
    const float* xb = x + (int64_t)b * H * W * C;
    const int Wo = W + pad_w;
    float* ob = out + (int64_t)b * (H + pad_h) * Wo * C;

    if constexpr (C == 32) {
        uint4 wv[9][3];
        auto load_w = [&](int ft) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    wv[tap][p] = *reinterpret_cast<const uint4*>(w3 + ((int64_t)(p * 9 + tap) * C + 16 * ft + n) * C + 8 * g);
        };
        load_w(0);
        x3_stage_tile<C, TH>(xb, in_s, tid, Y0, X0, H, W);
        __syncthreads();
#pragma unroll 1
        for (int ft = 0; ft < NFT; ++ft) {
            f32x4s acc[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[r] = f32x4s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    const int hp = (RW * wave + r + ky) * kX3HW + n + kx;
                    const unsigned short* bp = in_s + hp * C + 8 * x3_slot<C>(g, hp);
                    const uint4 b1 = *reinterpret_cast<const uint4*>(bp);
                    const uint4 b2 = *reinterpret_cast<const uint4*>(bp + PL);
                    const uint4 b3 = *reinterpret_cast<const uint4*>(bp + 2 * PL);
                    acc[r] = mfma_bf16x3(wv[tap][0], wv[tap][1], wv[tap][2], b1, b2, b3, acc[r]);
                }
                __builtin_amdgcn_sched_barrier(0);   // operand reads are not hoisted across taps (registers)
            }
            if (ft + 1 < NFT) load_w(ft + 1);   // in flight behind the epilogue
            const float4 bq = *reinterpret_cast<const float4*>(bias + 16 * ft + 4 * g);
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int gy = Y0 + RW * wave + r, gx = X0 + n;
                if (gy < H && gx < W)
                    *reinterpret_cast<float4*>(ob + ((int64_t)gy * Wo + gx) * C + 16 * ft + 4 * g) =
                        make_float4(x3_mishf(acc[r][0] + bq.x), x3_mishf(acc[r][1] + bq.y),
                                    x3_mishf(acc[r][2] + bq.z), x3_mishf(acc[r][3] + bq.w));
            }
        }
    } else {
        // C = 16: this lane's tap of pair j is 2 j + (g >> 1), its channels 8 (g & 1) .. + 7
        const int gh = g >> 1, gq = g & 1;
        uint4 wv[5][3];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int tap = 2 * j + gh;
#pragma unroll
            for (int p = 0; p < 3; ++p)
                wv[j][p] = tap < 9 ? *reinterpret_cast<const uint4*>(w3 + ((int64_t)(p * 9 + tap) * C + n) * C + 8 * gq)
                                   : make_uint4(0u, 0u, 0u, 0u);
        }
        x3_stage_tile<C, TH>(xb, in_s, tid, Y0, X0, H, W);
        __syncthreads();
        f32x4s acc[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[r] = f32x4s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int tap = (2 * j + gh) < 9 ? 2 * j + gh : 8;   // the zero half reads a valid pixel
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int hp = (RW * wave + r + ky) * kX3HW + n + kx;
                const unsigned short* bp = in_s + hp * C + 8 * x3_slot<C>(gq, hp);
                const uint4 b1 = *reinterpret_cast<const uint4*>(bp);
                const uint4 b2 = *reinterpret_cast<const uint4*>(bp + PL);
                const uint4 b3 = *reinterpret_cast<const uint4*>(bp + 2 * PL);
                acc[r] = mfma_bf16x3(wv[j][0], wv[j][1], wv[j][2], b1, b2, b3, acc[r]);
            }
        }
        const float4 bq = *reinterpret_cast<const float4*>(bias + 4 * g);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int gy = Y0 + RW * wave + r, gx = X0 + n;
            if (gy < H && gx < W)
                *reinterpret_cast<float4*>(ob + ((int64_t)gy * Wo + gx) * C + 4 * g) =
                    make_float4(x3_mishf(acc[r][0] + bq.x), x3_mishf(acc[r][1] + bq.y), x3_mishf(acc[r][2] + bq.z),
                                x3_mishf(acc[r][3] + bq.w));
        }
    }
    x3_zero_border<TH>(ob, tid, Y0, X0, H, W, pad_h, pad_w, C, 0, C / 4);
}

template <int C, int TH>
static int conv3x3_mish_x3_wide_launch(const void* x, const void* w3, const void* bias, void* out, int B, int H, int W,
                                       int pad_h, int pad_w, hipStream_t s) {
    const int tiles_x = (W + kX3TW - 1) / kX3TW, tiles_y = (H + TH - 1) / TH;
    const int64_t n_tiles = (int64_t)tiles_x * tiles_y * B;
    if (n_tiles * (C / 64) > INT32_MAX) {
        set_error("conv3x3_mish_x3: too many tiles");
        return QPWC_E_SHAPE;
    }
    hipLaunchKernelGGL((conv3x3_mish_x3_wide_kernel<C, TH>), dim3((unsigned)(n_tiles * (C / 64))), dim3(256), 0, s,
                       (const float*)x, (const unsigned short*)w3, (const float*)bias, (float*)out, H, W, pad_h, pad_w,
                       tiles_x, tiles_y, (int)n_tiles);
    return check_launch("conv3x3_mish_x3_wide_kernel");
}

template <int C>
static int conv3x3_mish_x3_narrow_launch(const void* x, const void* w3, const void* bias, void* out, int B, int H, int W,
                                         int pad_h, int pad_w, hipStream_t s) {
    const int tiles_x = (W + kX3TW - 1) / kX3TW, tiles_y = (H + 15) / 16;
    const int64_t n_tiles = (int64_t)tiles_x * tiles_y * B;
    if (n_tiles > INT32_MAX) {
        set_error("conv3x3_mish_x3: too many tiles");
        return QPWC_E_SHAPE;
    }
    hipLaunchKernelGGL((conv3x3_mish_x3_narrow_kernel<C>), dim3((unsigned)n_tiles), dim3(256), 0, s, (const float*)x,
                       (const unsigned short*)w3, (const float*)bias, (float*)out, H, W, pad_h, pad_w, tiles_x, tiles_y);
    return check_launch("conv3x3_mish_x3_narrow_kernel");
}

#ifndef QPWC_X3_TH64
#define QPWC_X3_TH64 8
#endif
// TH of the wide kernel by image size: the smallest tile count that still gives every CU a workgroup
int conv3x3_mish_x3_launch(const void* x, const void* w3, const void* bias, void* out, int B, int H, int W, int C,
                           int pad_h, int pad_w, hipStream_t s) {
    switch (C) {
        case 16: return conv3x3_mish_x3_narrow_launch<16>(x, w3, bias, out, B, H, W, pad_h, pad_w, s);
        case 32: return conv3x3_mish_x3_narrow_launch<32>(x, w3, bias, out, B, H, W, pad_h, pad_w, s);
        case 64: return conv3x3_mish_x3_wide_launch<64, QPWC_X3_TH64>(x, w3, bias, out, B, H, W, pad_h, pad_w, s);
        case 128: return conv3x3_mish_x3_wide_launch<128, 4>(x, w3, bias, out, B, H, W, pad_h, pad_w, s);
        case 256: return conv3x3_mish_x3_wide_launch<256, 2>(x, w3, bias, out, B, H, W, pad_h, pad_w, s);
        default: set_error("conv3x3_mish_x3: C=%d not in {16,32,64,128,256}", C); return QPWC_E_SHAPE;
    }
}

// ---------------------------------------------------------------------------
// conv_a of encoder levels 3..5 (3x3, stride 2, TF 'SAME', bias, Mish; non_layers.py:402-409), C_in = 32 / 64 / 128 ->
// 2 C_in, on the zero-bordered (B, H+1, W+1, C_in) input the previous level's conv_b wrote -- the bf16x3 form of
// conv3x3s2_mish_wide_kernel: the (2 TH + 1) x 33 input patch is staged as three bf16 images, split once, in two
// column-parity planes (pixel = ((col & 1) * IH + row) * PW + col / 2), so that the 16 output pixels of a row read 16
// consecutive pixels for every tap; a wave = one block of 16 outputs x TH output rows, workgroup = 64 outputs,
// grid = tiles x 2 C_in / 64; weights (3, 9, 2 C_in, C_in) bf16 streamed through a ring of nine steps as in the
// stride-1 kernel.
template <int CI, int TH>
__global__ __launch_bounds__(256, (3 * 2 * (2 * TH + 1) * (kX3TW + 1) * CI * 2 > 80 * 1024) ? 1 : 2) void conv3x3s2_mish_x3_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ w3, const float* __restrict__ bias,
    float* __restrict__ out, int H, int W, int tiles_x, int tiles_y, int n_tiles) {
    constexpr int CO = 2 * CI, NQ = CI / 8, NKB = CI / 32;
    constexpr int IH = 2 * TH + 1, PW = kX3TW + 1;
    constexpr int NPX = IH * 33, PL = 2 * IH * PW * CI;      // staged pixels; halves per image
    constexpr int NST = (NPX * NQ + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned short in_s[3 * PL];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int slice = blockIdx.x / n_tiles;
    const int tile = xcd_swizzle(blockIdx.x % n_tiles, n_tiles);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int X0 = tx * kX3TW, Y0 = ty * TH;               // output coordinates
    const int Hp = H + 1, Wp = W + 1, Ho = H / 2, Wo = W / 2;
    const int fo = 64 * slice + 16 * wave;
    const float* xb = x + (int64_t)b * Hp * Wp * CI;

    const unsigned short* wl = w3 + (int64_t)(fo + n) * CI + 8 * g;
    uint4 wr[9][3];
    auto load_w = [&](uint4 (&w)[3], int kb, int tap) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
            w[p] = *reinterpret_cast<const uint4*>(wl + (int64_t)(p * 9 + tap) * CO * CI + 32 * kb);
    };
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) load_w(wr[tap], 0, tap);
    {
        float4 st[NST][2];
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = tid + 256 * it;
            const int pxl = idx / NQ, q = idx - pxl * NQ;
            const int row = pxl / 33, col = pxl - row * 33;
            const int gy = 2 * Y0 + row, gx = 2 * X0 + col;
            const bool ok = idx < NPX * NQ && gy < Hp && gx < Wp;
            const float* p = xb + ((int64_t)gy * Wp + gx) * CI + 8 * q;
            st[it][0] = ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
            st[it][1] = ok ? *reinterpret_cast<const float4*>(p + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = tid + 256 * it;
            const int pxl = idx / NQ, q = idx - pxl * NQ;
            const int row = pxl / 33, col = pxl - row * 33;
            const int pix = ((col & 1) * IH + row) * PW + (col >> 1);
            if (idx < NPX * NQ) {
                uint4 p1, p2, p3;
                split8_bf16x3(st[it][0], st[it][1], p1, p2, p3);
                unsigned short* d = in_s + pix * CI + 8 * x3_slot<CI>(q, pix);
                *reinterpret_cast<uint4*>(d) = p1;
                *reinterpret_cast<uint4*>(d + PL) = p2;
                *reinterpret_cast<uint4*>(d + 2 * PL) = p3;
            }
        }
    }
    f32x4s acc[TH];
#pragma unroll
    for (int m = 0; m < TH; ++m) acc[m] = f32x4s{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll 1
    for (int kb = 0; kb < NKB; ++kb) {
        int nn = n, gg = g;
        asm volatile("" : "+v"(nn), "+v"(gg));
        uint4 bb[3][3];
        auto read_b = [&](uint4 (&bv)[3], int i) __attribute__((always_inline)) {
            const int tap = i / TH, m = i - tap * TH, ky = tap / 3, kx = tap - 3 * ky;
            const int pix = ((kx & 1) * IH + 2 * m + ky) * PW + nn + (kx >> 1);
            const unsigned short* bp = in_s + pix * CI + 8 * x3_slot<CI>(4 * kb + gg, pix);
            bv[0] = *reinterpret_cast<const uint4*>(bp);
            bv[1] = *reinterpret_cast<const uint4*>(bp + PL);
            bv[2] = *reinterpret_cast<const uint4*>(bp + 2 * PL);
        };
        read_b(bb[0], 0);
        read_b(bb[1], 1);
#pragma unroll
        for (int i = 0; i < 9 * TH; ++i) {
            const int tap = i / TH, m = i - tap * TH;
            if (m == 0) {
                const int t2 = (tap + 8) % 9, kb2 = tap + 8 >= 9 ? kb + 1 : kb;
                if (kb2 < NKB) load_w(wr[t2], kb2, t2);
            }
            if (i + 2 < 9 * TH) read_b(bb[(i + 2) % 3], i + 2);
            const uint4 (&w)[3] = wr[tap];
            const uint4 (&bv)[3] = bb[i % 3];
            acc[m] = mfma_bf16x3(w[0], w[1], w[2], bv[0], bv[1], bv[2], acc[m]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float* ob = out + (int64_t)b * Ho * Wo * CO;
    const float4 bq = *reinterpret_cast<const float4*>(bias + fo + 4 * g);
#pragma unroll
    for (int m = 0; m < TH; ++m) {
        const int gy = Y0 + m, gx = X0 + n;
        if (gy < Ho && gx < Wo)
            *reinterpret_cast<float4*>(ob + ((int64_t)gy * Wo + gx) * CO + fo + 4 * g) =
                make_float4(x3_mishf(acc[m][0] + bq.x), x3_mishf(acc[m][1] + bq.y), x3_mishf(acc[m][2] + bq.z),
                            x3_mishf(acc[m][3] + bq.w));
    }
}

template <int CI, int TH>
static int conv3x3s2_mish_x3_launch_t(const void* x, const void* w3, const void* bias, void* out, int B, int H, int W,
                                      hipStream_t s) {
    const int Ho = H / 2, Wo = W / 2;
    const int tiles_x = (Wo + kX3TW - 1) / kX3TW, tiles_y = (Ho + TH - 1) / TH;
    const int64_t n_tiles = (int64_t)tiles_x * tiles_y * B;
    if (n_tiles * (2 * CI / 64) > INT32_MAX) {
        set_error("conv3x3s2_mish_x3: too many tiles");
        return QPWC_E_SHAPE;
    }
    hipLaunchKernelGGL((conv3x3s2_mish_x3_kernel<CI, TH>), dim3((unsigned)(n_tiles * (2 * CI / 64))), dim3(256), 0, s,
                       (const float*)x, (const unsigned short*)w3, (const float*)bias, (float*)out, H, W, tiles_x, tiles_y,
                       (int)n_tiles);
    return check_launch("conv3x3s2_mish_x3_kernel");
}

int conv3x3s2_mish_x3_launch(const void* x, const void* w3, const void* bias, void* out, int B, int H, int W, int CI,
                             hipStream_t s) {
    switch (CI) {
        case 32: return conv3x3s2_mish_x3_launch_t<32, 4>(x, w3, bias, out, B, H, W, s);
        case 64: return conv3x3s2_mish_x3_launch_t<64, 2>(x, w3, bias, out, B, H, W, s);
        case 128: return conv3x3s2_mish_x3_launch_t<128, 2>(x, w3, bias, out, B, H, W, s);
        default: set_error("conv3x3s2_mish_x3: C_in=%d not in {32,64,128}", CI); return QPWC_E_SHAPE;
    }
}

// ---------------------------------------------------------------------------
// The decoder's UpConv (Conv2DTranspose 4x4, stride 2, 'same', bias, Mish; non_layers.py:196-210) written into the
// `up` half of the concat([up, skip]) buffer -- the bf16x3 form of upconv4x4s2_mish_kernel: an output pixel of parity
// (py, px) sees 2 x 2 of the 4 x 4 taps, so a wave = one parity x one block of 16 outputs x TH input rows, workgroup = the
// four parities, grid = tiles x F / 16; the (TH + 2) x 18 halo tile of all C channels as three bf16 images; a wave's four
// taps x 32 channels of split weights in 48 registers, the next 32-channel block's prefetched.
// w3: (3, 16 taps, F, C) bf16.
template <int C, int TH>
__global__ __launch_bounds__(256, (3 * (TH + 2) * kX3HW * C * 2 > 80 * 1024) ? 1 : 2) void upconv4x4s2_mish_x3_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ w3, const float* __restrict__ bias,
    float* __restrict__ out, int H, int W, int F, int out_pixel_stride, int tiles_x, int tiles_y, int n_tiles) {
    constexpr int NH = (TH + 2) * kX3HW, PL = NH * C, NKB = C / 32;
    __shared__ __attribute__((aligned(16))) unsigned short in_s[3 * PL];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int py = wave >> 1, px = wave & 1;
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int fblk = blockIdx.x / n_tiles;
    const int tile = xcd_swizzle(blockIdx.x % n_tiles, n_tiles);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int X0 = tx * kX3TW, Y0 = ty * TH;
    const int fo = 16 * fblk;
    const float* xb = x + (int64_t)b * H * W * C;
    // the 2 x 2 taps of this wave's parity: input offset (dy, dx), kernel position (ky, kx)
    const int dy1 = py ? 1 : -1, dx1 = px ? 1 : -1;          // second tap; the first is offset 0
    const int ky0 = py ? 2 : 1, ky1 = py ? 0 : 3, kx0 = px ? 2 : 1, kx1 = px ? 0 : 3;
    const int kpos[4] = {ky0 * 4 + kx0, ky0 * 4 + kx1, ky1 * 4 + kx0, ky1 * 4 + kx1};
    const int offy[4] = {0, 0, dy1, dy1}, offx[4] = {0, dx1, 0, dx1};
    const int64_t wplane = (int64_t)16 * F * C;
    uint4 wv[4][3], wn[4][3];
    auto load_w = [&](uint4 (&w)[4][3], int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                w[t][p] = *reinterpret_cast<const uint4*>(w3 + p * wplane + ((int64_t)kpos[t] * F + fo + n) * C + 32 * kb + 8 * g);
    };
    load_w(wv, 0);
    x3_stage_tile<C, TH>(xb, in_s, tid, Y0, X0, H, W);
    f32x4s acc[TH];
#pragma unroll
    for (int m = 0; m < TH; ++m) acc[m] = f32x4s{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll 1
    for (int kb = 0; kb < NKB; ++kb) {
        if (kb + 1 < NKB) load_w(wn, kb + 1);
        int nn = n, gg = g;
        asm volatile("" : "+v"(nn), "+v"(gg));
        uint4 bb[3][3];
        auto read_b = [&](uint4 (&bv)[3], int i) __attribute__((always_inline)) {
            const int t = i / TH, m = i - t * TH;
            const int hp = (m + 1 + offy[t]) * kX3HW + nn + 1 + offx[t];
            const unsigned short* bp = in_s + hp * C + 8 * x3_slot<C>(4 * kb + gg, hp);
            bv[0] = *reinterpret_cast<const uint4*>(bp);
            bv[1] = *reinterpret_cast<const uint4*>(bp + PL);
            bv[2] = *reinterpret_cast<const uint4*>(bp + 2 * PL);
        };
        read_b(bb[0], 0);
        read_b(bb[1], 1);
#pragma unroll
        for (int i = 0; i < 4 * TH; ++i) {
            const int t = i / TH, m = i - t * TH;
            if (i + 2 < 4 * TH) read_b(bb[(i + 2) % 3], i + 2);
            const uint4 (&bv)[3] = bb[i % 3];
            acc[m] = mfma_bf16x3(wv[t][0], wv[t][1], wv[t][2], bv[0], bv[1], bv[2], acc[m]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) wv[t][p] = wn[t][p];
    }
    const int H2 = 2 * H, W2 = 2 * W;
    float* ob = out + (int64_t)b * H2 * W2 * out_pixel_stride;
    const float4 bq = *reinterpret_cast<const float4*>(bias + fo + 4 * g);
#pragma unroll
    for (int m = 0; m < TH; ++m) {
        const int gy = Y0 + m, gx = X0 + n;
        if (gy < H && gx < W)
            *reinterpret_cast<float4*>(ob + ((int64_t)(2 * gy + py) * W2 + 2 * gx + px) * out_pixel_stride + fo + 4 * g) =
                make_float4(x3_mishf(acc[m][0] + bq.x), x3_mishf(acc[m][1] + bq.y), x3_mishf(acc[m][2] + bq.z),
                            x3_mishf(acc[m][3] + bq.w));
    }
}

template <int C, int TH>
static int upconv_x3_launch_t(const void* x, const void* w3, const void* bias, void* out, int B, int H, int W, int F,
                              int out_pixel_stride, hipStream_t s) {
    const int tiles_x = (W + kX3TW - 1) / kX3TW, tiles_y = (H + TH - 1) / TH;
    const int64_t n_tiles = (int64_t)tiles_x * tiles_y * B;
    if (n_tiles * (F / 16) > INT32_MAX) {
        set_error("upconv4x4s2_mish_x3: too many tiles");
        return QPWC_E_SHAPE;
    }
    hipLaunchKernelGGL((upconv4x4s2_mish_x3_kernel<C, TH>), dim3((unsigned)(n_tiles * (F / 16))), dim3(256), 0, s,
                       (const float*)x, (const unsigned short*)w3, (const float*)bias, (float*)out, H, W, F,
                       out_pixel_stride, tiles_x, tiles_y, (int)n_tiles);
    return check_launch("upconv4x4s2_mish_x3_kernel");
}

int upconv4x4s2_mish_x3_launch(const void* x, const void* w3, const void* bias, void* out, int B, int H, int W, int C,
                               int F, int out_pixel_stride, hipStream_t s) {
    switch (C) {
        case 64: return upconv_x3_launch_t<64, 8>(x, w3, bias, out, B, H, W, F, out_pixel_stride, s);
        case 128: return upconv_x3_launch_t<128, 4>(x, w3, bias, out, B, H, W, F, out_pixel_stride, s);
        case 256: return upconv_x3_launch_t<256, 2>(x, w3, bias, out, B, H, W, F, out_pixel_stride, s);
        default: set_error("upconv4x4s2_mish_x3: C=%d not in {64,128,256}", C); return QPWC_E_SHAPE;
    }
}

// ---------------------------------------------------------------------------
// fp32 array -> its three bf16 images, out[p * n + i] (weights, once per model)
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ src, unsigned short* __restrict__ out,
                                                           int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned p1, p2, p3;
    split2_bf16x3(src[i], 0.f, p1, p2, p3);
    out[i] = (unsigned short)p1;
    out[n + i] = (unsigned short)p2;
    out[2 * n + i] = (unsigned short)p3;
}

int split_bf16x3_launch(const void* src, void* out, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)src,
                       (unsigned short*)out, n);
    return check_launch("split_bf16x3_kernel");
}

}  // namespace qpwc
