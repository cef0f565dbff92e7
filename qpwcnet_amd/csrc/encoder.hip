// 3x3 stride-1 'same' convolution + bias + Mish of the encoder's DownConv blocks
// (conv_aa / conv_b, qpwcnet/core/non_layers.py:410-449 with use_normalizer=False, pwcnet.py:146)
// for the narrow levels (C_in = C_out = 16 or 32), channels-last fp32, gfx950.
//
// At 16 / 32 channels the layer is a K = 144 / 288, N = 16 / 32 implicit GEMM over 0.5 M / 0.13 M
// pixels: 2.4 GFLOP against 67 / 34 MB of activations.  The library kernels run it at 45-58 TF and
// need a separate bias+Mish pass; here the tile's halo lives in LDS, the nine shifted views of it are
// the B operands of v_mfma_f32_16x16x4_f32 (rows = output channels, cols = 16 pixels of a tile row),
// the weights of a wave's output block sit in registers for the whole tile, and bias + Mish (and the
// zero border the next stride-2 convolution's 'SAME' padding needs) are the epilogue.
//
// Workgroup = 4 waves = an 8 x 16 pixel tile; a wave owns two tile rows.  k-slot g of a lane owns
// channels 4g..4g+3 of a 16-channel chunk, so every operand is one 16-byte access; LDS pixels are
// C floats with the 16-byte chunk c of halo pixel p stored at chunk c ^ ((p >> 1) & (C/4 - 1)) for
// C = 32, c ^ ((p >> 2) & 3) for C = 16 (ec_slot).
// Measured, 16 x 128 x 256 x 16 (36 us): without the matrix instructions 23 us, without Mish 31.5, without
// the halo loads 33.5 -- one tile per workgroup runs the chip in rounds (all resident workgroups load,
// then all compute).  Resident workgroups walking the tiles with the next halo prefetched into registers
// and a double-buffered LDS image need 144 / 196 registers (loop-invariant addresses stay live): 34.7 us
// at C = 16, 34.8 (from 31.8) at C = 32 -- not kept.
#include "common.h"

namespace qpwc {

// (same fast form as optflow.hip's mishf)
__device__ __forceinline__ float enc_mishf(float x) {
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

typedef float f32x4e __attribute__((ext_vector_type(4)));

// two consecutive elements as fp32 / four fp32 values stored as four elements (one 8- or 16-byte access)
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float2 ld2(const __half* p) { return __half22float2(*reinterpret_cast<const __half2*>(p)); }
__device__ __forceinline__ void st4q(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4q(__half* p, float4 v) {
    __half2 lo = __floats2half2_rn(v.x, v.y), hi = __floats2half2_rn(v.z, v.w);
    uint2 u;
    u.x = *reinterpret_cast<unsigned*>(&lo);
    u.y = *reinterpret_cast<unsigned*>(&hi);
    *reinterpret_cast<uint2*>(p) = u;
}

constexpr int kEcTH = 8, kEcTW = 16;
constexpr int kEcHW = kEcTW + 2, kEcNH = (kEcTH + 2) * kEcHW;   // 180 halo pixels

// weight: [9 taps][C out][C in] fp32; out: (B, H + pad_h, W + pad_w, C), border zero
// 16-byte chunk q of halo pixel hp sits at chunk ec_slot(q, hp) of the pixel's row: the matrix operand
// reads (16 lanes = 16 consecutive pixels, one chunk each) must cover all 64 banks.  128-B pixels
// (C = 32): q ^ (hp >> 1); 64-B pixels (C = 16): four pixels span the banks once, so the pixels
// hp, hp+4, hp+8, hp+12 of a read need four different chunks: q ^ (hp >> 2).
template <int C>
__device__ __forceinline__ int ec_slot(int q, int hp) {
    return C == 16 ? (q ^ ((hp >> 2) & 3)) : (q ^ ((hp >> 1) & (C / 4 - 1)));
}

// NT > 1 (A/B: make ab ABSRC=encoder ABFLAGS=-DQPWC_ENC_NT=2): the workgroup walks NT x-adjacent tiles with a
// double-buffered halo image: the loads of tile t + 1 are issued before the matrix work of tile t and written to
// the other LDS buffer after its epilogue (one barrier per tile), so that only the first load and the last store of
// a workgroup are exposed.  Measured in one call (round 2; 66 / 104 registers once the lane offsets are kept from
// being hoisted out of the tile loop): C = 16 34.0 (NT 1) / 36.9 (2) / 37.1 (4) us, C = 32 30.5 / 33.4 / 38.6 --
// slower, like round 1's register-heavier attempt: fewer, longer-lived workgroups lose more than the hidden loads
// win.  The product build is NT = 1.
#ifndef QPWC_ENC_NT
#define QPWC_ENC_NT 1
#endif
// Explicit operand prefetch in the wide fp32 kernels (round 3): a kernel's (tap, 16-channel chunk) steps as one flat
// unrolled loop, the TH ds_read_b128 of step i + 1 issued before the 4 TH matrix instructions of step i, a
// sched_barrier per step.  As the compiler places them the reads go out 1-4 matrix instructions (32-128 cycles) before
// their first use and every step ends in an `s_waitcnt lgkmcnt(0)` that one wave per SIMD (the 128- and 256-channel
// levels) has nothing to hide behind.  Measured in one call (tools/encbench.py, tools/step_time.py, three interleaved
// runs): stride-1 layers C = 64 / 128 / 256: 33.3 / 30.4 / 33.2 -> 29.7 / 29.7 / 30.1 us (two steps ahead: 32.0 / 29.4 /
// 30.2), step 1.2006 -> 1.1826 ms -- kept (QPWC_ENC_PIPE = 1).  The stride-2 layers do not move (17.4-20.6 us either
// way, step 1.1816 vs 1.1809) and the decoder's transposed convolution gets 7 us SLOWER per step with it (1.1886 vs
// 1.1816: on the second queue, beside the coarse flow levels, see QpwcNet.dec_chunks; on the finest decoder level only,
// QPWC_UPCONV_PIPE = 2: 1.1758 vs 1.1780, inside the noise) -- both stay as the compiler schedules them.
#ifndef QPWC_ENC16_PIPE
#define QPWC_ENC16_PIPE 1   // the fp16 wide kernel: a tap's operand reads one tap ahead (0 = as the compiler places them)
#endif
// (the same one / two taps ahead in the fp16 transposed and stride-2 kernels: config 5's step 1.709 vs 1.706 ms -- off)
#ifndef QPWC_UPCONV16_PIPE
#define QPWC_UPCONV16_PIPE 0
#endif
#ifndef QPWC_S2_16_PIPE
#define QPWC_S2_16_PIPE 0
#endif
#ifndef QPWC_S2_PIPE
#define QPWC_S2_PIPE 0
#endif
#ifndef QPWC_UPCONV_PIPE
#define QPWC_UPCONV_PIPE 0
#endif
// Round 4: the next block's weights are requested UNCONDITIONALLY (the last trip re-reads its own block, an L1 hit).  With
// `if (kb + 1 < NKB) load_w(...)` the compiler's s_waitcnt pass must be right for the trip that issued no request as well, so
// it waited with vmcnt(7), (6), .. (0) through the trip's eight groups of matrix instructions -- i.e. for the requests that
// were issued a moment ago at the top of THIS trip, one by one, instead of for the ones issued a whole trip earlier
// (vmcnt is a counter of outstanding requests, in order): an L2 round trip exposed at the top of every 32-channel block.
#ifndef QPWC_UPCONV16_NFB_MIN_WGS
#define QPWC_UPCONV16_NFB_MIN_WGS 512   // fp16 transposed convolution: see nfb in upconv_f16_launch_t (1 << 30 = one block per workgroup)
#endif
#ifndef QPWC_UPCONV256_TH
#define QPWC_UPCONV256_TH 2   // input rows per workgroup of the 256- / 128-channel transposed convolutions (A/B, round 4)
#endif
#ifndef QPWC_UPCONV128_TH
#define QPWC_UPCONV128_TH 4
#endif
#ifndef QPWC_UPCONV_LDS_TOTAL
#define QPWC_UPCONV_LDS_TOTAL 0
#endif
#ifndef QPWC_UPCONV64_LDS_TOTAL
#define QPWC_UPCONV64_LDS_TOTAL 82944   // the same for the finest decoder level only (it has ~120 us of slack before flow level 4
                                        // needs it): ONE of its workgroups per CU leaves the flow chain's 79-80 KiB workgroups room beside
                                        // it -- config 2 step 1.1124 vs 1.1158 ms (five interleaved pairs); 0 = off
#endif
#ifndef QPWC_W_NEXT_ALWAYS
#define QPWC_W_NEXT_ALWAYS 1
#endif
#ifndef QPWC_W_NEXT_ALWAYS_F16
#define QPWC_W_NEXT_ALWAYS_F16 0   // the fp16 kernels (16-cycle matrix instructions, two waves per SIMD) lose with it: config 5 1.627-1.629 vs 1.604-1.611 ms
#endif
#if QPWC_W_NEXT_ALWAYS
#define QPWC_LOAD_W_NEXT(END) load_w(wn, kb + 1 < (END) ? kb + 1 : kb)
#else
#define QPWC_LOAD_W_NEXT(END) do { if (kb + 1 < (END)) load_w(wn, kb + 1); } while (0)
#endif
#ifndef QPWC_ENC_NARROW_EARLY
#define QPWC_ENC_NARROW_EARLY 0   // A/B (round 4): narrow fp32 kernel, weights (C = 16) + bias requested with the tile's inputs: 32.4 vs 32.3, 30.2 vs 30.3 us, step +-0 -- off
#endif
#ifndef QPWC_ENC_WIDE_WAVES
#define QPWC_ENC_WIDE_WAVES 2   // waves per SIMD the wide fp32 kernel is compiled for (2: 256 registers; 1: the accumulators move to AGPRs, +-0)
#endif
#ifndef QPWC_ENC_PIPE_FENCE
#define QPWC_ENC_PIPE_FENCE 1
#endif
#ifndef QPWC_ENC_PIPE
#define QPWC_ENC_PIPE 1   // operand reads of the wide fp32 kernels issued this many steps ahead (0 = as the compiler places them)
#endif
#ifdef QPWC_ENC_STAMP
// diagnostic build only (make ab ABSRC=encoder ABFLAGS=-DQPWC_ENC_STAMP; tools/enc_census.py): per workgroup of
// the narrow kernel: start, inputs landed, staged (after the barrier), matrix work done, end, HW_ID | XCC_ID << 32
__device__ long long g_enc_census[4096 * 6];
#define ENC_CENSUS(i) do { if (census_on) census_p[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ENC_CENSUS(i) do { } while (0)
#endif
template <int C, int NT>
__global__ __launch_bounds__(256, 4) void conv3x3_mish_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ weight,
                                                              const float* __restrict__ bias,
                                                              float* __restrict__ out, int H, int W,
                                                              int pad_h, int pad_w, int tiles_x, int tiles_y,
                                                              int groups_x) {
    constexpr int NQ = C / 4;     // 16-byte chunks per pixel
    constexpr int NFT = C / 16;   // output blocks of 16 channels
    constexpr int NKC = C / 16;   // 16-channel k chunks
    constexpr int NLD = (kEcNH * NQ + 255) / 256;   // staging loads per thread and tile
    constexpr int NBUF = NT > 1 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float in_s[NBUF * kEcNH * C];
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int lane = tid0 & 63, n0 = lane & 15, g0 = lane >> 4;
    const int grp = xcd_swizzle(blockIdx.x, gridDim.x);
    const int gx0 = grp % groups_x, ty = (grp / groups_x) % tiles_y, b = grp / (groups_x * tiles_y);
    const int tx0 = gx0 * NT;
    const int Y0 = ty * kEcTH;
    const float* xb = x + (int64_t)b * H * W * C;
    const int Ho = H + pad_h, Wo = W + pad_w;
    float* ob = out + (int64_t)b * Ho * Wo * C;

    auto stage_load = [&](int tid, int tx, float4 (&v)[NLD]) {
        const int X0 = tx * kEcTW;
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int idx = tid + 256 * it;
            const int hp = idx / NQ, q = idx - hp * NQ;
            const int hy = hp / kEcHW, hx = hp - hy * kEcHW;
            const int gy = Y0 - 1 + hy, gx = X0 - 1 + hx;
            v[it] = (idx < kEcNH * NQ && gy >= 0 && gy < H && gx >= 0 && gx < W)
                        ? *reinterpret_cast<const float4*>(xb + ((int64_t)gy * W + gx) * C + 4 * q)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage_write = [&](int tid, int buf, const float4 (&v)[NLD]) {
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int idx = tid + 256 * it;
            const int hp = idx / NQ, q = idx - hp * NQ;
            if (idx < kEcNH * NQ)
                *reinterpret_cast<float4*>(in_s + buf * (kEcNH * C) + hp * C + 4 * ec_slot<C>(q, hp)) = v[it];
        }
    };
#ifdef QPWC_ENC_STAMP
    const bool census_on = tid0 == 0 && blockIdx.x < 4096;
    long long* census_p = g_enc_census + (blockIdx.x < 4096 ? blockIdx.x : 0) * 6;
    if (census_on)
        census_p[5] = (long long)(unsigned)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) |
                      ((long long)(unsigned)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) << 32);
#endif
    ENC_CENSUS(0);
    float4 st[NLD];
    stage_load(tid0, tx0, st);
    // Round 4 (one tile per workgroup): the first output block's weights and every bias value are requested HERE, with the
    // tile's inputs, and land under the same wait.  As the compiler placed them, each tap's weights were loaded a few matrix
    // instructions ahead of their use behind counted vmcnt waits, and the bias right before the epilogue behind
    // `s_waitcnt vmcnt(0)` -- L2-hit latencies exposed inside a 2.3 k-cycle matrix phase.
    constexpr bool EARLYB = QPWC_ENC_NARROW_EARLY && NT == 1;   // every bias value
    constexpr bool EARLY = EARLYB && C == 16;                   // ... and the weights (C = 32: 18 requests with 18 address pairs on top
                                                                // of the staged tile spill 29-33 registers under the 128 of four waves per SIMD)
    f32x4e wpre[EARLY ? 9 : 1][EARLY ? NKC : 1];
    float4 bpre[EARLYB ? NFT : 1];
    if (EARLYB) {
        if (EARLY) {
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc)
                    wpre[k][kc] = *reinterpret_cast<const f32x4e*>(weight + ((int64_t)k * C + n0) * C + 16 * kc + 4 * g0);
        }
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) bpre[ft] = *reinterpret_cast<const float4*>(bias + 16 * ft + 4 * g0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // also a compiler barrier: the loads above stay above
    }
#ifdef QPWC_ENC_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    ENC_CENSUS(1);
    stage_write(tid0, 0, st);
    __syncthreads();
    ENC_CENSUS(2);

#pragma unroll 1   // a real loop: unrolled, the compiler hoists the next tiles' weight loads (73 / 102 registers at NT = 2 / 4)
    for (int t = 0; t < NT; ++t) {
        const int tx = tx0 + t;
        if (tx >= tiles_x) break;                      // workgroup-uniform
        const bool more = t + 1 < NT && tx + 1 < tiles_x;

        const int X0 = tx * kEcTW;
        const float* tile_s = in_s + (t & (NBUF - 1)) * (kEcNH * C);
        // the weights are re-read per tile (L1 hits): hoisted out of the tile loop they would stay live across
        // it (36 / 144 registers) -- the opaque copy of the pointer keeps the loads inside
        const float* wt = weight;
        int n = n0, g = g0, tid = tid0;
        if (NT > 1) asm volatile("" : "+r"(wt), "+v"(n), "+v"(g), "+v"(tid));   // and the lane offsets derived from these
        if (more) stage_load(tid, tx + 1, st);         // in flight behind this tile's matrix work
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) {
            // weights of output block ft for this lane: row f = 16 ft + n, channels 16 kc + 4g .. + 3, 9 taps
            f32x4e wld[EARLY ? 1 : 9][EARLY ? 1 : NKC];
            if (!EARLY) {
#pragma unroll
                for (int k = 0; k < 9; ++k)
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc)
                        wld[k][kc] = *reinterpret_cast<const f32x4e*>(wt + ((int64_t)k * C + 16 * ft + n) * C + 16 * kc + 4 * g);
            }
            auto wv = [&](int k, int kc) __attribute__((always_inline)) -> f32x4e& {
                if constexpr (EARLY) return wpre[k][kc];
                else return wld[k][kc];
            };
            f32x4e acc[2];
            acc[0] = f32x4e{0.f, 0.f, 0.f, 0.f};
            acc[1] = f32x4e{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc) {
                        f32x4e bv[2];
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const int hp = (2 * wave + m + ky) * kEcHW + n + kx;
                            const int q = 4 * kc + g;
                            const int slot = ec_slot<C>(q, hp);
                            bv[m] = *reinterpret_cast<const f32x4e*>(tile_s + hp * C + 4 * slot);
                        }
                        // alternate the two accumulators: a dependent matrix instruction issued back to back
                        // waits 40 cycles instead of 32
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                            for (int m = 0; m < 2; ++m)
                                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv(ky * 3 + kx, kc)[tt], bv[m][tt], acc[m], 0, 0, 0);
                        // the next output block's weights of this (tap, chunk) are requested as soon as its registers are free:
                        // a whole block of matrix work lies between the request and the first use
                        if (EARLY && ft + 1 < NFT) {
                            __builtin_amdgcn_sched_barrier(0);   // (unfenced, the scheduler hoists these requests: two register sets, 33 spilled)
                            wpre[ky * 3 + kx][kc] = *reinterpret_cast<const f32x4e*>(
                                wt + ((int64_t)(ky * 3 + kx) * C + 16 * (ft + 1) + n) * C + 16 * kc + 4 * g);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            if (ft == NFT - 1) ENC_CENSUS(3);
            // ---- bias + Mish: lane = pixel n of tile row 2 wave + m, outputs 16 ft + 4g .. + 3 ----
            const float4 bq = EARLYB ? bpre[ft] : *reinterpret_cast<const float4*>(bias + 16 * ft + 4 * g);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int gy = Y0 + 2 * wave + m, gx = X0 + n;
                if (gy < H && gx < W)
                    *reinterpret_cast<float4*>(ob + ((int64_t)gy * Wo + gx) * C + 16 * ft + 4 * g) =
                        make_float4(enc_mishf(acc[m][0] + bq.x), enc_mishf(acc[m][1] + bq.y),
                                    enc_mishf(acc[m][2] + bq.z), enc_mishf(acc[m][3] + bq.w));
            }
        }
        // ---- zero border of the padded output (columns W.., rows H..), written by the edge tiles ----
        if (pad_w > 0 && X0 + kEcTW >= W) {
            for (int i = tid; i < kEcTH * pad_w * NQ; i += 256) {
                const int q = i % NQ, r = i / NQ, col = r % pad_w, row = r / pad_w;
                const int gy = Y0 + row;
                if (gy < H) *reinterpret_cast<float4*>(ob + ((int64_t)gy * Wo + W + col) * C + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (pad_h > 0 && Y0 + kEcTH >= H) {
            const int x_end = (X0 + kEcTW >= W) ? Wo : X0 + kEcTW;   // the corner belongs to the last tile
            for (int i = tid; i < pad_h * (x_end - X0) * NQ; i += 256) {
                const int q = i % NQ, r = i / NQ, col = r % (x_end - X0), row = r / (x_end - X0);
                *reinterpret_cast<float4*>(ob + ((int64_t)(H + row) * Wo + X0 + col) * C + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        ENC_CENSUS(4);
        if (more) {
            // buffer (t + 1) & 1 was last read by tile t - 1, whose waves all passed the previous barrier
            stage_write(tid, (t + 1) & (NBUF - 1), st);
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------
// The same layer at the wide encoder levels (C_in = C_out = 64 / 128 / 256; 32x64 .. 8x16 pixel images
// at 256x512): the weights no longer fit a wave's registers for all outputs, so a wave owns ONE block
// of 16 outputs for ALL pixels of the tile (TH x 16 pixels = TH accumulators) and keeps that block's
// weights for 32 input channels at a time in registers (9 taps x 2 k-chunks x 4 = 72, the next 32
// channels' block prefetched into a second set); a workgroup =
// 4 waves = 64 outputs, the grid = tiles x C/64 output slices.  TH = 8 / 4 / 2 keeps the grid at one
// workgroup per CU for the 256x512 pyramid (256 workgroups at every level) and the work per wave
// constant (1152 matrix instructions).  Every B operand (16 pixels x 4 channels per k-slot) is one
// ds_read_b128 from the halo tile and feeds 4 matrix instructions; LDS pixels are C floats with the
// 16-byte chunk q of halo pixel p at q ^ (p & 15) (16 consecutive pixels of one chunk cover all banks).
// Replaces library convolution (73 TF) + bias/Mish pass (+ a zeroing launch for its split-K variants).
// Round 4, KS = 2 -- measured +-0, not launched by the product build (QPWC_ENC_KSPLIT_MINC) -- for the 128- / 256-channel
// levels, whose 256 workgroups are ONE wave per SIMD (every LDS or weight wait of that wave idles the SIMD's matrix
// pipe, 0.55 busy at the 2.4 GHz the chip holds under this kernel, tools/clock_under.py): the workgroup has EIGHT waves -- waves 4-7 take the second half of
// the input-channel blocks for the same four output blocks, each wave loads only its half's weights (the weight
// traffic of the launch does not change), and the two halves meet in LDS before bias + Mish: two waves per SIMD with
// half the matrix instructions each, one hiding the other's waits.  The sum is (first half) + (second half) instead
// of one chain: another fp32 rounding order of the same products.
template <int C, int TH, int KS = 1>
__global__ __launch_bounds__(256 * KS, KS == 1 ? QPWC_ENC_WIDE_WAVES : 1) void conv3x3_mish_wide_kernel(
    const float* __restrict__ x, const float* __restrict__ weight, const float* __restrict__ bias,
    float* __restrict__ out, int H, int W, int pad_h, int pad_w, int tiles_x, int tiles_y, int n_tiles) {
    constexpr int NT = 256 * KS;                   // threads
    constexpr int NQ = C / 4;                      // 16-byte chunks per pixel
    constexpr int HH = TH + 2, NH = HH * kEcHW;    // halo rows / pixels
    constexpr int NKB = C / 32;                    // 32-channel blocks of the reduction
    constexpr int NST = (NH * NQ + NT - 1) / NT;   // staging loads per thread
    static_assert(NKB % KS == 0, "the K halves are whole 32-channel blocks");
    __shared__ __attribute__((aligned(16))) float in_s[NH * C];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ow = wave & 3, kh = wave >> 2;       // output block of the slice, half of the reduction
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int slice = blockIdx.x / n_tiles;                       // 64 outputs
    const int tile = xcd_swizzle(blockIdx.x % n_tiles, n_tiles);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int X0 = tx * kEcTW, Y0 = ty * TH;
    const int fo = 64 * slice + 16 * ow;                          // this wave's output block
    const float* xb = x + (int64_t)b * H * W * C;

    // ---- stage the halo tile (zero outside the image): all loads first, then the LDS writes ----
    {
        float4 st[NST];
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = tid + NT * it;
            const int hp = idx / NQ, q = idx - hp * NQ;
            const int hy = hp / kEcHW, hx = hp - hy * kEcHW;
            const int gy = Y0 - 1 + hy, gx = X0 - 1 + hx;
            st[it] = (idx < NH * NQ && gy >= 0 && gy < H && gx >= 0 && gx < W)
                         ? *reinterpret_cast<const float4*>(xb + ((int64_t)gy * W + gx) * C + 4 * q)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = tid + NT * it;
            const int hp = idx / NQ, q = idx - hp * NQ;
            if (idx < NH * NQ) *reinterpret_cast<float4*>(in_s + hp * C + 4 * (q ^ (hp & 15))) = st[it];
        }
    }
    f32x4e acc[TH];
#pragma unroll
    for (int m = 0; m < TH; ++m) acc[m] = f32x4e{0.f, 0.f, 0.f, 0.f};
    // weights of output row fo + n, input channels 32 kb + 16 kc + 4 g .. + 3, 9 taps: 72 registers per
    // block, the next block's loads are issued before the current block's matrix instructions
    f32x4e wv[9][2], wn[9][2];
    auto load_w = [&](f32x4e (&w)[9][2], int kb) {
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
                w[k][kc] = *reinterpret_cast<const f32x4e*>(weight + ((int64_t)k * C + fo + n) * C + 32 * kb + 16 * kc + 4 * g);
    };
    const int kb0 = kh * (NKB / KS), kb1 = kb0 + NKB / KS;
    load_w(wv, kb0);
    __syncthreads();   // the halo tile is complete (the first weight loads are in flight behind it)
    // one 32-channel block with the weights in `wv`; the two register sets alternate (QPWC_W_NEXT_ALWAYS)
    auto block = [&](f32x4e (&wv)[9][2], int kb) __attribute__((always_inline)) {
#if QPWC_ENC_PIPE
        // one step = (tap, 16-channel chunk): TH ds_read_b128 feed 4 TH matrix instructions; the reads of step i + 1
        // are issued before the matrix instructions of step i (the wide levels run one wave per SIMD: nothing else
        // covers the LDS latency)
        constexpr int RD = QPWC_ENC_PIPE;       // steps ahead
        f32x4e bb[RD + 1][TH];
        auto read_b = [&](f32x4e (&bv)[TH], int i) __attribute__((always_inline)) {
            const int tap = i >> 1, kc = i & 1, ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int m = 0; m < TH; ++m) {
                const int hp = (m + ky) * kEcHW + n + kx;
                const int q = 8 * kb + 4 * kc + g;
                bv[m] = *reinterpret_cast<const f32x4e*>(in_s + hp * C + 4 * (q ^ (hp & 15)));
            }
        };
#pragma unroll
        for (int i = 0; i < RD; ++i) read_b(bb[i], i);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            if (i + RD < 18) read_b(bb[(i + RD) % (RD + 1)], i + RD);
            // (round 4: without this fence the scheduler sinks the reads to the END of the step, into the registers the step
            // has just finished with: one buffer, the reads three matrix instructions ahead of their use.  Fenced, a true
            // step ahead: C = 64 27.9-28.6 vs 29.5-29.7 us, C = 128 +1 %, C = 256 29.0-29.7 vs 28.4-28.6 -- C = 64 only)
            if (QPWC_ENC_PIPE_FENCE && C == 64) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < TH; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i >> 1][i & 1][t], bb[i % (RD + 1)][m][t], acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#else
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    f32x4e bv[TH];
#pragma unroll
                    for (int m = 0; m < TH; ++m) {
                        const int hp = (m + ky) * kEcHW + n + kx;
                        const int q = 8 * kb + 4 * kc + g;
                        bv[m] = *reinterpret_cast<const f32x4e*>(in_s + hp * C + 4 * (q ^ (hp & 15)));
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int m = 0; m < TH; ++m)
                            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ky * 3 + kx][kc][t], bv[m][t], acc[m], 0, 0, 0);
                }
#endif
    };
#if QPWC_W_NEXT_ALWAYS
    {
        int kb = kb0;
#pragma unroll 1
        for (; kb + 1 < kb1; kb += 2) {
            load_w(wn, kb + 1);
            __builtin_amdgcn_sched_barrier(0);   // (unfenced, the scheduler sinks the requests to just in front of their use)
            block(wv, kb);
            load_w(wv, kb + 2 < kb1 ? kb + 2 : kb);   // (last trip: a harmless re-read)
            __builtin_amdgcn_sched_barrier(0);
            block(wn, kb + 1);
        }
        if (kb < kb1) block(wv, kb);   // odd number of blocks
    }
#else
#pragma unroll 1
    for (int kb = kb0; kb < kb1; ++kb) {
        if (kb + 1 < kb1) load_w(wn, kb + 1);
        block(wv, kb);
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) wv[k][kc] = wn[k][kc];
    }
#endif
    if (KS > 1) {
        // the second half's sums travel through LDS (the halo tile is dead once every wave has left the loop)
        __syncthreads();
        f32x4e* red = reinterpret_cast<f32x4e*>(in_s) + (ow * TH) * 64 + lane;
        if (kh == 1) {
#pragma unroll
            for (int m = 0; m < TH; ++m) red[m * 64] = acc[m];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int m = 0; m < TH; ++m) acc[m] += red[m * 64];
        }
    }
    // ---- bias + Mish: lane = pixel n of tile row m, outputs fo + 4g .. + 3 ----
    const int Ho = H + pad_h, Wo = W + pad_w;
    float* ob = out + (int64_t)b * Ho * Wo * C;
    const float4 bq = *reinterpret_cast<const float4*>(bias + fo + 4 * g);
#pragma unroll
    for (int m = 0; m < TH; ++m) {
        const int gy = Y0 + m, gx = X0 + n;
        if (kh == 0 && gy < H && gx < W)
            *reinterpret_cast<float4*>(ob + ((int64_t)gy * Wo + gx) * C + fo + 4 * g) =
                make_float4(enc_mishf(acc[m][0] + bq.x), enc_mishf(acc[m][1] + bq.y),
                            enc_mishf(acc[m][2] + bq.z), enc_mishf(acc[m][3] + bq.w));
    }
    // ---- zero border of the padded output, this slice's 64 channels, written by the edge tiles ----
    if (pad_w > 0 && X0 + kEcTW >= W) {
        for (int i = tid; i < TH * pad_w * 16; i += NT) {
            const int q = i & 15, r = i >> 4, col = r % pad_w, row = r / pad_w;
            const int gy = Y0 + row;
            if (gy < H) *reinterpret_cast<float4*>(ob + ((int64_t)gy * Wo + W + col) * C + 64 * slice + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (pad_h > 0 && Y0 + TH >= H) {
        const int x_end = (X0 + kEcTW >= W) ? Wo : X0 + kEcTW;   // the corner belongs to the last tile
        for (int i = tid; i < pad_h * (x_end - X0) * 16; i += NT) {
            const int q = i & 15, r = i >> 4, col = r % (x_end - X0), row = r / (x_end - X0);
            *reinterpret_cast<float4*>(ob + ((int64_t)(H + row) * Wo + X0 + col) * C + 64 * slice + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

#ifndef QPWC_ENC_KSPLIT_MINC
#define QPWC_ENC_KSPLIT_MINC (1 << 30)   // channel count from which the wide stride-1 kernel splits its reduction over 8 waves.
                                         // Round 4, one call (tools/encbench.py, us): never / from 128 / from 64 channels: C = 64 28.9 / 28.7 /
                                         // 31.4, C = 128 28.1 / 28.5 / 27.6, C = 256 29.2 / 31.3 / 30.7; step 1.1773 / 1.1778 / 1.1790 ms --
                                         // a second wave per SIMD does not fill the matrix pipe either: never (the form stays, tested)
#endif
template <int C, int TH>
static int conv3x3_mish_wide_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H,
                                    int W, int pad_h, int pad_w, hipStream_t s) {
    const int tiles_x = (W + kEcTW - 1) / kEcTW, tiles_y = (H + TH - 1) / TH;
    const int64_t n_tiles = (int64_t)tiles_x * tiles_y * B;
    if (n_tiles * (C / 64) > INT32_MAX) {
        set_error("conv3x3_mish: too many tiles");
        return QPWC_E_SHAPE;
    }
    if (C >= QPWC_ENC_KSPLIT_MINC)
        hipLaunchKernelGGL((conv3x3_mish_wide_kernel<C, TH, 2>), dim3((unsigned)(n_tiles * (C / 64))), dim3(512), 0, s,
                           (const float*)x, (const float*)weight, (const float*)bias, (float*)out, H, W, pad_h, pad_w,
                           tiles_x, tiles_y, (int)n_tiles);
    else
        hipLaunchKernelGGL((conv3x3_mish_wide_kernel<C, TH, 1>), dim3((unsigned)(n_tiles * (C / 64))), dim3(256), 0, s,
                           (const float*)x, (const float*)weight, (const float*)bias, (float*)out, H, W, pad_h, pad_w,
                           tiles_x, tiles_y, (int)n_tiles);
    return check_launch("conv3x3_mish_wide_kernel");
}

// ---------------------------------------------------------------------------
// fp16-storage twin of the two kernels above (BASELINE configs[4]; the reference itself has no fp16 path):
// x, weight ([9 taps][C out][C in]) and out fp16, bias fp32; products on v_mfma_f32_16x16x32_f16 with fp32
// accumulation, bias + Mish in fp32, ONE rounding to fp16 at the store (the library path it replaces rounds the
// convolution output and the activation).  One matrix instruction per tap and 32-channel block where fp32 needs
// eight, so the layer is bound by its activations' bytes; the structure is kept simple: 8 x 16 (TH x 16) pixel tile,
// the whole halo tile of all input channels in LDS (pixels of CP = max(C, 32) halves, 16-byte chunk q of halo pixel p
// at slot f16_slot(q, p)), a workgroup = 4 waves = TH / 4 tile rows each x min(C, 64) outputs (grid = tiles x C / 64),
// the weights streamed from L1 / L2 (16 bytes per lane, tap and output block).  C = 16: the upper half of every
// 32-channel pixel is zero in LDS and the weight lanes of k >= 16 are zero.
typedef _Float16 f16x8e __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4e __attribute__((ext_vector_type(4)));

template <int CP>
__device__ __forceinline__ int f16_slot(int q, int hp) {
    // pixels of 64 / 128 / 256 / 512 bytes: 16 consecutive pixels x one chunk must spread over the 16 sixteen-byte
    // slots of the 64 banks
    return CP == 32 ? (q ^ ((0 - (hp >> 2)) & 3)) : (CP == 64 ? (q ^ ((hp >> 1) & 7)) : (q ^ (hp & 15)));
}

template <int C, int TH>
__global__ __launch_bounds__(256) void conv3x3_mish_f16_kernel(const __half* __restrict__ x,
                                                              const __half* __restrict__ weight,
                                                              const float* __restrict__ bias,
                                                              __half* __restrict__ out, int H, int W, int pad_h,
                                                              int pad_w, int tiles_x, int tiles_y, int n_tiles) {
    constexpr int CP = C < 32 ? 32 : C;            // halves per pixel in LDS
    constexpr int NQ = CP / 8, NQG = C / 8;        // 16-byte chunks per pixel: LDS / global
    constexpr int HH = TH + 2, NH = HH * kEcHW;
    constexpr int NKB = CP / 32;
    constexpr int FW = C < 64 ? C : 64, NFT = FW / 16, NSL = C / FW;   // outputs per workgroup, blocks, slices
    constexpr int RW = TH / 4;                     // tile rows per wave
    constexpr int NST = (NH * NQ + 255) / 256;
    __shared__ __attribute__((aligned(16))) __half in_s[NH * CP];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int slice = blockIdx.x / n_tiles;
    const int tile = xcd_swizzle(blockIdx.x % n_tiles, n_tiles);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int X0 = tx * kEcTW, Y0 = ty * TH;
    const int f0 = FW * slice;
    const __half* xb = x + (int64_t)b * H * W * C;
    {
        uint4 st[NST];
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = tid + 256 * it;
            const int hp = idx / NQ, q = idx - hp * NQ;
            const int hy = hp / kEcHW, hx = hp - hy * kEcHW;
            const int gy = Y0 - 1 + hy, gx = X0 - 1 + hx;
            st[it] = (idx < NH * NQ && q < NQG && gy >= 0 && gy < H && gx >= 0 && gx < W)
                         ? *reinterpret_cast<const uint4*>(xb + ((int64_t)gy * W + gx) * C + 8 * q)
                         : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = tid + 256 * it;
            const int hp = idx / NQ, q = idx - hp * NQ;
            if (idx < NH * NQ) *reinterpret_cast<uint4*>(in_s + hp * CP + 8 * f16_slot<CP>(q, hp)) = st[it];
        }
    }
    const int Ho = H + pad_h, Wo = W + pad_w;
    __half* ob = out + (int64_t)b * Ho * Wo * C;
    if (C >= 64) {
        // wide levels: a wave owns ONE block of 16 outputs for all TH tile rows and keeps that block's weights for 32
        // input channels in registers (9 taps x 16 bytes, the next block's requested before this block's matrix work)
        // -- no weight is fetched twice by a workgroup, every B operand read feeds one matrix instruction
        const int fo = f0 + 16 * wave;
        f32x4e acw[TH];
#pragma unroll
        for (int r = 0; r < TH; ++r) acw[r] = f32x4e{0.f, 0.f, 0.f, 0.f};
        f16x8e wv[9], wn[9];
        auto load_w = [&](f16x8e (&w)[9], int kb) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                w[tap] = *reinterpret_cast<const f16x8e*>(weight + ((int64_t)tap * C + fo + n) * C + 32 * kb + 8 * g);
        };
        load_w(wv, 0);
        __syncthreads();
        // one 32-channel block with the weights in `wv`; the two register sets alternate (QPWC_W_NEXT_ALWAYS)
        auto block = [&](f16x8e (&wv)[9], int kb) __attribute__((always_inline)) {
#if QPWC_ENC16_PIPE
            // one step = one tap: TH ds_read_b128 feed TH matrix instructions of 16 cycles each; the reads of tap
            // t + 1 go out before the matrix instructions of tap t (round 3: as the compiler placed them every matrix
            // instruction waited for its own read -- 33 us for a layer whose bytes and products are 4 us each)
            f16x8e bb[2][TH];
            auto read_b = [&](f16x8e (&bv)[TH], int tap) __attribute__((always_inline)) {
                const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
                for (int r = 0; r < TH; ++r) {
                    const int hp = (r + ky) * kEcHW + n + kx;
                    bv[r] = *reinterpret_cast<const f16x8e*>(in_s + hp * CP + 8 * f16_slot<CP>(4 * kb + g, hp));
                }
            };
            read_b(bb[0], 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + 1 < 9) read_b(bb[(tap + 1) & 1], tap + 1);
#pragma unroll
                for (int r = 0; r < TH; ++r)
                    acw[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[tap], bb[tap & 1][r], acw[r], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#else
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
                for (int r = 0; r < TH; ++r) {
                    const int hp = (r + ky) * kEcHW + n + kx;
                    const f16x8e bv = *reinterpret_cast<const f16x8e*>(in_s + hp * CP + 8 * f16_slot<CP>(4 * kb + g, hp));
                    acw[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[tap], bv, acw[r], 0, 0, 0);
                }
            }
#endif
        };
#if QPWC_W_NEXT_ALWAYS_F16
        {
            int kb = 0;
#pragma unroll 1
            for (; kb + 1 < NKB; kb += 2) {
                load_w(wn, kb + 1);
                __builtin_amdgcn_sched_barrier(0);   // (unfenced, the scheduler sinks the requests to just in front of their use)
                block(wv, kb);
                load_w(wv, kb + 2 < NKB ? kb + 2 : kb);   // (last trip: a harmless re-read)
                __builtin_amdgcn_sched_barrier(0);
                block(wn, kb + 1);
            }
            if (kb < NKB) block(wv, kb);   // odd number of blocks
        }
#else
#pragma unroll 1
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb + 1 < NKB) load_w(wn, kb + 1);
            block(wv, kb);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) wv[tap] = wn[tap];
        }
#endif
        const float4 bq = *reinterpret_cast<const float4*>(bias + fo + 4 * g);
#pragma unroll
        for (int r = 0; r < TH; ++r) {
            const int gy = Y0 + r, gx = X0 + n;
            if (gy < H && gx < W) {
                f16x4e o;
                o[0] = (_Float16)enc_mishf(acw[r][0] + bq.x);
                o[1] = (_Float16)enc_mishf(acw[r][1] + bq.y);
                o[2] = (_Float16)enc_mishf(acw[r][2] + bq.z);
                o[3] = (_Float16)enc_mishf(acw[r][3] + bq.w);
                *reinterpret_cast<f16x4e*>(ob + ((int64_t)gy * Wo + gx) * C + fo + 4 * g) = o;
            }
        }
    } else {
    f32x4e acc[RW][NFT];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) acc[r][ft] = f32x4e{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const bool kvalid = 8 * g < C;   // C = 16: lanes of k >= 16 carry zeros
#pragma unroll 1
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            f16x8e wv[NFT];
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft) {
                wv[ft] = f16x8e{0, 0, 0, 0, 0, 0, 0, 0};
                if (kvalid)
                    wv[ft] = *reinterpret_cast<const f16x8e*>(weight + ((int64_t)tap * C + f0 + 16 * ft + n) * C + 32 * kb + 8 * g);
            }
            f16x8e bv[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int hp = (RW * wave + r + ky) * kEcHW + n + kx;
                bv[r] = *reinterpret_cast<const f16x8e*>(in_s + hp * CP + 8 * f16_slot<CP>(4 * kb + g, hp));
            }
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
                for (int r = 0; r < RW; ++r)
                    acc[r][ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[ft], bv[r], acc[r][ft], 0, 0, 0);
        }
    }
    // ---- bias + Mish, one rounding to fp16: lane = pixel n of tile row RW wave + r, outputs f0 + 16 ft + 4g .. + 3 ----
    // (all bias values requested before the first store: a load behind stores waits for them too, vmcnt counts both)
    float4 bqs[NFT];
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft) bqs[ft] = *reinterpret_cast<const float4*>(bias + f0 + 16 * ft + 4 * g);
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft) {
        const float4 bq = bqs[ft];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int gy = Y0 + RW * wave + r, gx = X0 + n;
            if (gy < H && gx < W) {
                f16x4e o;
                o[0] = (_Float16)enc_mishf(acc[r][ft][0] + bq.x);
                o[1] = (_Float16)enc_mishf(acc[r][ft][1] + bq.y);
                o[2] = (_Float16)enc_mishf(acc[r][ft][2] + bq.z);
                o[3] = (_Float16)enc_mishf(acc[r][ft][3] + bq.w);
                *reinterpret_cast<f16x4e*>(ob + ((int64_t)gy * Wo + gx) * C + f0 + 16 * ft + 4 * g) = o;
            }
        }
    }
    }
    // ---- zero border of the padded output, this slice's FW channels (FW / 8 sixteen-byte chunks per pixel) ----
    constexpr int NQF = FW / 8;
    if (pad_w > 0 && X0 + kEcTW >= W) {
        for (int i = tid; i < TH * pad_w * NQF; i += 256) {
            const int q = i % NQF, r = i / NQF, col = r % pad_w, row = r / pad_w;
            const int gy = Y0 + row;
            if (gy < H) *reinterpret_cast<uint4*>(ob + ((int64_t)gy * Wo + W + col) * C + f0 + 8 * q) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    if (pad_h > 0 && Y0 + TH >= H) {
        const int x_end = (X0 + kEcTW >= W) ? Wo : X0 + kEcTW;   // the corner belongs to the last tile
        for (int i = tid; i < pad_h * (x_end - X0) * NQF; i += 256) {
            const int q = i % NQF, r = i / NQF, col = r % (x_end - X0), row = r / (x_end - X0);
            *reinterpret_cast<uint4*>(ob + ((int64_t)(H + row) * Wo + X0 + col) * C + f0 + 8 * q) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    (void)NSL;
}

// ---------------------------------------------------------------------------
// The narrow levels of the fp16 layer (C = 16 / 32), round 3.  The kernel above runs them at 60 / 34 us for 64 frames
// (config 5) where their bytes allow 17 / 8: half of every C = 16 matrix instruction multiplies zeros, the weights are
// fetched again for every tap and every matrix instruction waits for its own operand read.  Here (the structure of
// conv3x3_mish_x3_narrow_kernel): 16 x 16 pixel tile, a wave owns four tile rows and ALL outputs, the weights of the
// whole layer sit in registers (C = 32: 9 taps x 2 output blocks; C = 16: 5 tap PAIRS -- k-slots g = 0, 1 carry tap 2j,
// g = 2, 3 tap 2j + 1, so no lane multiplies padding except in the tenth half), and a tap's four operand reads go out
// one tap ahead of its matrix instructions.
template <int C>
__device__ __forceinline__ int f16n_slot(int q, int hp) {
    return C == 16 ? (q ^ ((hp >> 3) & 1)) : (q ^ ((0 - (hp >> 2)) & 3));
}

template <int C>
__global__ __launch_bounds__(256, C == 16 ? 4 : 3) void conv3x3_mish_f16_narrow_kernel(
    const __half* __restrict__ x, const __half* __restrict__ weight, const float* __restrict__ bias,
    __half* __restrict__ out, int H, int W, int pad_h, int pad_w, int tiles_x, int tiles_y) {
    constexpr int TH = 16, RW = 4, HWD = kEcTW + 2;
    constexpr int NQ = C / 8, NH = (TH + 2) * HWD;
    constexpr int NST = (NH * NQ + 255) / 256;
    constexpr int NFT = C / 16;
    __shared__ __attribute__((aligned(16))) __half in_s[NH * C];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int X0 = tx * kEcTW, Y0 = ty * TH;
    const __half* xb = x + (int64_t)b * H * W * C;
    const int Ho = H + pad_h, Wo = W + pad_w;
    __half* ob = out + (int64_t)b * Ho * Wo * C;
    {
        uint4 st[NST];
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = tid + 256 * it;
            const int hp = idx / NQ, q = idx - hp * NQ;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int gy = Y0 - 1 + hy, gx = X0 - 1 + hx;
            st[it] = (idx < NH * NQ && gy >= 0 && gy < H && gx >= 0 && gx < W)
                         ? *reinterpret_cast<const uint4*>(xb + ((int64_t)gy * W + gx) * C + 8 * q)
                         : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = tid + 256 * it;
            const int hp = idx / NQ, q = idx - hp * NQ;
            if (idx < NH * NQ) *reinterpret_cast<uint4*>(in_s + hp * C + 8 * f16n_slot<C>(q, hp)) = st[it];
        }
    }
    // steps: C = 32: the 9 taps (lane's k-slot = channels 8g .. 8g+7); C = 16: 5 tap pairs (tap 2j + (g >> 1), channels
    // 8 (g & 1) .. + 7; the tenth half has zero weights and reads a valid pixel)
    constexpr int NS = C == 16 ? 5 : 9;
    const int gh = C == 16 ? (g >> 1) : 0, gq = C == 16 ? (g & 1) : g;
    f16x8e wv[NS][NFT];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int tap = C == 16 ? 2 * j + gh : j;
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) {
            wv[j][ft] = f16x8e{0, 0, 0, 0, 0, 0, 0, 0};
            if (tap < 9) wv[j][ft] = *reinterpret_cast<const f16x8e*>(weight + ((int64_t)tap * C + 16 * ft + n) * C + 8 * gq);
        }
    }
    f32x4e acc[RW][NFT];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) acc[r][ft] = f32x4e{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    f16x8e bb[2][RW];
    auto read_b = [&](f16x8e (&bv)[RW], int j) __attribute__((always_inline)) {
        const int t0 = C == 16 ? 2 * j + gh : j;
        const int tap = t0 < 9 ? t0 : 8;
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int hp = (RW * wave + r + ky) * HWD + n + kx;
            bv[r] = *reinterpret_cast<const f16x8e*>(in_s + hp * C + 8 * f16n_slot<C>(gq, hp));
        }
    };
    read_b(bb[0], 0);
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        if (j + 1 < NS) read_b(bb[(j + 1) & 1], j + 1);
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
            for (int r = 0; r < RW; ++r)
                acc[r][ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[j][ft], bb[j & 1][r], acc[r][ft], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- bias + Mish, one rounding to fp16: lane = pixel n of tile row RW wave + r, outputs 16 ft + 4g .. + 3 ----
    // (all bias values requested before the first store: a load behind stores waits for them too, vmcnt counts both)
    float4 bqs[NFT];
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft) bqs[ft] = *reinterpret_cast<const float4*>(bias + 16 * ft + 4 * g);
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft) {
        const float4 bq = bqs[ft];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int gy = Y0 + RW * wave + r, gx = X0 + n;
            if (gy < H && gx < W) {
                f16x4e o;
                o[0] = (_Float16)enc_mishf(acc[r][ft][0] + bq.x);
                o[1] = (_Float16)enc_mishf(acc[r][ft][1] + bq.y);
                o[2] = (_Float16)enc_mishf(acc[r][ft][2] + bq.z);
                o[3] = (_Float16)enc_mishf(acc[r][ft][3] + bq.w);
                *reinterpret_cast<f16x4e*>(ob + ((int64_t)gy * Wo + gx) * C + 16 * ft + 4 * g) = o;
            }
        }
    }
    // ---- zero border of the padded output (C / 8 sixteen-byte chunks per pixel) ----
    if (pad_w > 0 && X0 + kEcTW >= W) {
        for (int i = tid; i < TH * pad_w * NQ; i += 256) {
            const int q = i % NQ, r = i / NQ, col = r % pad_w, row = r / pad_w;
            const int gy = Y0 + row;
            if (gy < H) *reinterpret_cast<uint4*>(ob + ((int64_t)gy * Wo + W + col) * C + 8 * q) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    if (pad_h > 0 && Y0 + TH >= H) {
        const int x_end = (X0 + kEcTW >= W) ? Wo : X0 + kEcTW;   // the corner belongs to the last tile
        for (int i = tid; i < pad_h * (x_end - X0) * NQ; i += 256) {
            const int q = i % NQ, r = i / NQ, col = r % (x_end - X0), row = r / (x_end - X0);
            *reinterpret_cast<uint4*>(ob + ((int64_t)(H + row) * Wo + X0 + col) * C + 8 * q) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
}

#ifndef QPWC_ENC16_NARROW
#define QPWC_ENC16_NARROW 1   // 0: the narrow levels on conv3x3_mish_f16_kernel (A/B)
#endif
template <int C>
static int conv3x3_mish_f16_narrow_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H,
                                          int W, int pad_h, int pad_w, hipStream_t s) {
    const int tiles_x = (W + kEcTW - 1) / kEcTW, tiles_y = (H + 15) / 16;
    const int64_t n_tiles = (int64_t)tiles_x * tiles_y * B;
    if (n_tiles > INT32_MAX) {
        set_error("conv3x3_mish_f16: too many tiles");
        return QPWC_E_SHAPE;
    }
    hipLaunchKernelGGL((conv3x3_mish_f16_narrow_kernel<C>), dim3((unsigned)n_tiles), dim3(256), 0, s, (const __half*)x,
                       (const __half*)weight, (const float*)bias, (__half*)out, H, W, pad_h, pad_w, tiles_x, tiles_y);
    return check_launch("conv3x3_mish_f16_narrow_kernel");
}

template <int C, int TH>
static int conv3x3_mish_f16_launch_t(const void* x, const void* weight, const void* bias, void* out, int B, int H,
                                     int W, int pad_h, int pad_w, hipStream_t s) {
    constexpr int NSL = C < 64 ? 1 : C / 64;
    const int tiles_x = (W + kEcTW - 1) / kEcTW, tiles_y = (H + TH - 1) / TH;
    const int64_t n_tiles = (int64_t)tiles_x * tiles_y * B;
    if (n_tiles * NSL > INT32_MAX) {
        set_error("conv3x3_mish_f16: too many tiles");
        return QPWC_E_SHAPE;
    }
    hipLaunchKernelGGL((conv3x3_mish_f16_kernel<C, TH>), dim3((unsigned)(n_tiles * NSL)), dim3(256), 0, s,
                       (const __half*)x, (const __half*)weight, (const float*)bias, (__half*)out, H, W, pad_h, pad_w,
                       tiles_x, tiles_y, (int)n_tiles);
    return check_launch("conv3x3_mish_f16_kernel");
}

int conv3x3_mish_f16_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                            int C, int pad_h, int pad_w, hipStream_t s) {
    switch (C) {
        case 16: return QPWC_ENC16_NARROW ? conv3x3_mish_f16_narrow_launch<16>(x, weight, bias, out, B, H, W, pad_h, pad_w, s)
                                          : conv3x3_mish_f16_launch_t<16, 8>(x, weight, bias, out, B, H, W, pad_h, pad_w, s);
        case 32: return QPWC_ENC16_NARROW ? conv3x3_mish_f16_narrow_launch<32>(x, weight, bias, out, B, H, W, pad_h, pad_w, s)
                                          : conv3x3_mish_f16_launch_t<32, 8>(x, weight, bias, out, B, H, W, pad_h, pad_w, s);
        case 64: return conv3x3_mish_f16_launch_t<64, 8>(x, weight, bias, out, B, H, W, pad_h, pad_w, s);
        case 128: return conv3x3_mish_f16_launch_t<128, 8>(x, weight, bias, out, B, H, W, pad_h, pad_w, s);
        case 256: return conv3x3_mish_f16_launch_t<256, 4>(x, weight, bias, out, B, H, W, pad_h, pad_w, s);
        default: set_error("conv3x3_mish_f16: C=%d not in {16,32,64,128,256}", C); return QPWC_E_SHAPE;
    }
}

// ---------------------------------------------------------------------------
// UpConv of the decoder (non_layers.py:196-210): Conv2DTranspose(F, 4x4, stride 2, 'same') + bias + Mish,
// written straight into channels [0, F) of the concat([up, skip]) buffer (pwcnet.py:186-195).
// out[2y+py, 2x+px] only sees the 2 x 2 taps of the 4 x 4 kernel that match its parity:
//   py = 0: (input row y, ky = 1), (y-1, ky = 3);   py = 1: (y, ky = 2), (y+1, ky = 0);   same in x,
// i.e. four independent 2x2 convolutions (K = 4 C) over the same input tile.  Workgroup = 4 waves = the 4
// parities of ONE block of 16 outputs for a TH x 16 input tile (grid = tiles x F/16); a wave keeps its 4
// taps x 32 input channels of weights in 32 registers (next block prefetched), every B operand is one
// ds_read_b128 of the (TH+2) x 18 halo tile (pixels = C floats, chunk q at q ^ (p & 15)).
// No zero-fill launch, no separate bias/Mish pass, short workgroups (512 matrix instructions per wave).
// (Resident workgroups walking the tiles behind a capped grid: 1.294 ms/step at 256, 1.262 at 384 workgroups,
// against 1.257 for four plain launches over quarters of the batch -- see UpConv.cat_skip.)
// weight: [ky*4+kx][F][C] fp32.
// Round 4: the SKIP half of the decoder's concat([up, skip]) copied by the same launch (qpwc_upconv4x4s2_mish_cat_fwd):
// lane (pixel n, quad g) of output block fo also moves channels fo + 4g .. + 3 of its output pixels from `skip` to channels
// [F, 2F) of `out` -- F skip channels, the decoder's case at every level -- instead of a separate qpwc_copy_pixels launch
// beside the flow levels (four launches, 41 us of the second queue per step).  p == nullptr: no copy.
struct UpSkip {
    const void* p;
    long long bs, rs, ps;   // batch / row / pixel stride of the skip tensor, elements
};

template <int C, int TH>
__global__ __launch_bounds__(256, 2) void upconv4x4s2_mish_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ weight,
                                                                  const float* __restrict__ bias,
                                                                  float* __restrict__ out, int H, int W, int F,
                                                                  int out_pixel_stride, int tiles_x, int tiles_y,
                                                                  int n_tiles, UpSkip skip) {
    constexpr int NQ = C / 4, NKB = C / 32;
    constexpr int HH = TH + 2, NH = HH * kEcHW;
    constexpr int NST = (NH * NQ + 255) / 256;
    __shared__ __attribute__((aligned(16))) float in_s[NH * C];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int py = wave >> 1, px = wave & 1;
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int fblk = blockIdx.x / n_tiles;
    const int tile = xcd_swizzle(blockIdx.x % n_tiles, n_tiles);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int X0 = tx * kEcTW, Y0 = ty * TH;
    const int fo = 16 * fblk;
    const float* xb = x + (int64_t)b * H * W * C;
    {
        float4 st[NST];
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = tid + 256 * it;
            const int hp = idx / NQ, q = idx - hp * NQ;
            const int hy = hp / kEcHW, hx = hp - hy * kEcHW;
            const int gy = Y0 - 1 + hy, gx = X0 - 1 + hx;
            st[it] = (idx < NH * NQ && gy >= 0 && gy < H && gx >= 0 && gx < W)
                         ? *reinterpret_cast<const float4*>(xb + ((int64_t)gy * W + gx) * C + 4 * q)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = tid + 256 * it;
            const int hp = idx / NQ, q = idx - hp * NQ;
            if (idx < NH * NQ) *reinterpret_cast<float4*>(in_s + hp * C + 4 * (q ^ (hp & 15))) = st[it];
        }
    }
    // the 2 x 2 taps of this wave's parity: input offset (dy, dx), kernel position (ky, kx)
    const int dy1 = py ? 1 : -1, dx1 = px ? 1 : -1;          // second tap; the first is offset 0
    const int ky0 = py ? 2 : 1, ky1 = py ? 0 : 3, kx0 = px ? 2 : 1, kx1 = px ? 0 : 3;
    const int kpos[4] = {ky0 * 4 + kx0, ky0 * 4 + kx1, ky1 * 4 + kx0, ky1 * 4 + kx1};
    const int offy[4] = {0, 0, dy1, dy1}, offx[4] = {0, dx1, 0, dx1};
    f32x4e acc[TH];
#pragma unroll
    for (int m = 0; m < TH; ++m) acc[m] = f32x4e{0.f, 0.f, 0.f, 0.f};
    f32x4e wv[4][2], wn[4][2];
    auto load_w = [&](f32x4e (&w)[4][2], int kb) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
                w[t][kc] = *reinterpret_cast<const f32x4e*>(weight + ((int64_t)kpos[t] * F + fo + n) * C + 32 * kb + 16 * kc + 4 * g);
    };
    load_w(wv, 0);
    __syncthreads();
    // one 32-channel block of the reduction with the weights in `wv` (the two register sets alternate: no copy, and
    // every trip issues the same requests, so the compiler's vmcnt counts are exact -- see QPWC_W_NEXT_ALWAYS)
    auto block = [&](f32x4e (&wv)[4][2], int kb) __attribute__((always_inline)) {
        // QPWC_UPCONV_PIPE: 1 = every level, 2 = the finest decoder level (C = 64) only
        if constexpr (QPWC_UPCONV_PIPE == 1 || (QPWC_UPCONV_PIPE == 2 && C == 64)) {
        // one step = (tap, 16-channel chunk); the operand reads of step i + 1 go out before the matrix instructions of
        // step i (as in conv3x3_mish_wide_kernel)
        f32x4e bb[2][TH];
        auto read_b = [&](f32x4e (&bv)[TH], int i) __attribute__((always_inline)) {
            const int t = i >> 1, kc = i & 1;
#pragma unroll
            for (int m = 0; m < TH; ++m) {
                const int hp = (m + 1 + offy[t]) * kEcHW + n + 1 + offx[t];
                bv[m] = *reinterpret_cast<const f32x4e*>(in_s + hp * C + 4 * ((8 * kb + 4 * kc + g) ^ (hp & 15)));
            }
        };
        read_b(bb[0], 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i + 1 < 8) read_b(bb[(i + 1) & 1], i + 1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < TH; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i >> 1][i & 1][j], bb[i & 1][m][j], acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                f32x4e bv[TH];
#pragma unroll
                for (int m = 0; m < TH; ++m) {
                    const int hp = (m + 1 + offy[t]) * kEcHW + n + 1 + offx[t];
                    bv[m] = *reinterpret_cast<const f32x4e*>(in_s + hp * C + 4 * ((8 * kb + 4 * kc + g) ^ (hp & 15)));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < TH; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t][kc][j], bv[m][j], acc[m], 0, 0, 0);
            }
        }
    };
    static_assert(NKB % 2 == 0, "two blocks per trip");
    if constexpr (QPWC_W_NEXT_ALWAYS && NKB > 2) {   // (C = 64, two blocks: +-0, and 5 spilled registers at TH = 8)
#pragma unroll 1
        for (int kb = 0; kb < NKB; kb += 2) {
            load_w(wn, kb + 1);
            __builtin_amdgcn_sched_barrier(0);        // (unfenced, the scheduler sinks the requests to just in front of their use)
            block(wv, kb);
            load_w(wv, kb + 2 < NKB ? kb + 2 : kb);   // (last trip: a harmless re-read)
            __builtin_amdgcn_sched_barrier(0);
            block(wn, kb + 1);
        }
    } else {
#pragma unroll 1
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb + 1 < NKB) load_w(wn, kb + 1);
            block(wv, kb);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) wv[t][kc] = wn[t][kc];
        }
    }
    const int H2 = 2 * H, W2 = 2 * W;
    float* ob = out + (int64_t)b * H2 * W2 * out_pixel_stride;
    // the skip values of this lane's output pixels: requested before the bias, stored behind the outputs
    const float* sk = reinterpret_cast<const float*>(skip.p);
    float4 sv[TH];
    if (sk != nullptr) {
#pragma unroll
        for (int m = 0; m < TH; ++m) {
            const int gy = Y0 + m, gx = X0 + n;
            sv[m] = (gy < H && gx < W)
                        ? *reinterpret_cast<const float4*>(sk + b * skip.bs + (int64_t)(2 * gy + py) * skip.rs +
                                                           (int64_t)(2 * gx + px) * skip.ps + fo + 4 * g)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float4 bq = *reinterpret_cast<const float4*>(bias + fo + 4 * g);
#pragma unroll
    for (int m = 0; m < TH; ++m) {
        const int gy = Y0 + m, gx = X0 + n;
        if (gy < H && gx < W) {
            float* op = ob + ((int64_t)(2 * gy + py) * W2 + 2 * gx + px) * out_pixel_stride + fo + 4 * g;
            *reinterpret_cast<float4*>(op) =
                make_float4(enc_mishf(acc[m][0] + bq.x), enc_mishf(acc[m][1] + bq.y),
                            enc_mishf(acc[m][2] + bq.z), enc_mishf(acc[m][3] + bq.w));
            if (sk != nullptr) *reinterpret_cast<float4*>(op + F) = sv[m];
        }
    }
}

template <int C, int TH>
static int upconv_launch_t(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W, int F,
                           int out_pixel_stride, hipStream_t s, UpSkip skip) {
    const int tiles_x = (W + kEcTW - 1) / kEcTW, tiles_y = (H + TH - 1) / TH;
    const int64_t n_tiles = (int64_t)tiles_x * tiles_y * B;
    if (n_tiles * (F / 16) > INT32_MAX) {
        set_error("upconv4x4s2_mish: too many tiles");
        return QPWC_E_SHAPE;
    }
    // QPWC_UPCONV_LDS_TOTAL (A/B, round 4): pad the workgroup's LDS footprint to this many bytes with dynamic LDS the kernel never
    // touches -- 81 KiB leaves ONE decoder workgroup per CU and room for a 79 KiB SeparableConv2D workgroup of the flow chain
    constexpr size_t kStatic = (size_t)(TH + 2) * kEcHW * C * sizeof(float);
    // (the finest decoder level of a SMALL step only: at config 4's 32 k workgroups one per CU costs +0.7 %)
    const size_t want = (C == 64 && QPWC_UPCONV64_LDS_TOTAL && n_tiles * (F / 16) <= 2048) ? QPWC_UPCONV64_LDS_TOTAL
                                                                                         : QPWC_UPCONV_LDS_TOTAL;
    const size_t extra = want > kStatic ? want - kStatic : 0;
    hipLaunchKernelGGL((upconv4x4s2_mish_kernel<C, TH>), dim3((unsigned)(n_tiles * (F / 16))), dim3(256), extra, s,
                       (const float*)x, (const float*)weight, (const float*)bias, (float*)out, H, W, F,
                       out_pixel_stride, tiles_x, tiles_y, (int)n_tiles, skip);
    return check_launch("upconv4x4s2_mish_kernel");
}

int upconv4x4s2_mish_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W, int C,
                            int F, int out_pixel_stride, hipStream_t s, const void* skip, int64_t skip_bs, int64_t skip_rs,
                            int64_t skip_ps) {
    const UpSkip sk{skip, (long long)skip_bs, (long long)skip_rs, (long long)skip_ps};
    switch (C) {
        case 64: return upconv_launch_t<64, 8>(x, weight, bias, out, B, H, W, F, out_pixel_stride, s, sk);
        case 128: return upconv_launch_t<128, QPWC_UPCONV128_TH>(x, weight, bias, out, B, H, W, F, out_pixel_stride, s, sk);
        case 256: return upconv_launch_t<256, QPWC_UPCONV256_TH>(x, weight, bias, out, B, H, W, F, out_pixel_stride, s, sk);
        default: set_error("upconv4x4s2_mish: C=%d not in {64,128,256}", C); return QPWC_E_SHAPE;
    }
}

// fp16-storage twin of upconv4x4s2_mish_kernel (BASELINE configs[4]): x, weight ([16 taps][F][C]) and out fp16, bias
// fp32; the same work split (wave = output parity, workgroup = one block of 16 outputs for a TH x 16 input tile), one
// v_mfma_f32_16x16x32_f16 per tap, tile row and 32-channel block, weights of the current block in 16 registers with
// the next block's requested ahead, halo tile in LDS as in conv3x3_mish_f16_kernel.  out_pixel_stride in halves.
template <int C, int TH>
__global__ __launch_bounds__(256, 2) void upconv4x4s2_mish_f16_kernel(const __half* __restrict__ x,
                                                                      const __half* __restrict__ weight,
                                                                      const float* __restrict__ bias,
                                                                      __half* __restrict__ out, int H, int W, int F,
                                                                      int out_pixel_stride, int tiles_x, int tiles_y,
                                                                      int n_tiles, UpSkip skip, int nfb) {
    constexpr int NQ = C / 8, NKB = C / 32;
    constexpr int HH = TH + 2, NH = HH * kEcHW;
    constexpr int NST = (NH * NQ + 255) / 256;
    __shared__ __attribute__((aligned(16))) __half in_s[NH * C];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int py = wave >> 1, px = wave & 1;
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    // Round 4: a workgroup stages its halo tile ONCE and walks nfb blocks of 16 outputs with it (fp16: the 16-cycle matrix
    // instructions no longer hide F / 16 workgroups re-staging the same tile; grid = tiles x F / 16 / nfb)
    const int fblk0 = (blockIdx.x / n_tiles) * nfb;
    const int tile = xcd_swizzle(blockIdx.x % n_tiles, n_tiles);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int X0 = tx * kEcTW, Y0 = ty * TH;
    int fo = 16 * fblk0;
    const __half* xb = x + (int64_t)b * H * W * C;
    {
        uint4 st[NST];
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = tid + 256 * it;
            const int hp = idx / NQ, q = idx - hp * NQ;
            const int hy = hp / kEcHW, hx = hp - hy * kEcHW;
            const int gy = Y0 - 1 + hy, gx = X0 - 1 + hx;
            st[it] = (idx < NH * NQ && gy >= 0 && gy < H && gx >= 0 && gx < W)
                         ? *reinterpret_cast<const uint4*>(xb + ((int64_t)gy * W + gx) * C + 8 * q)
                         : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = tid + 256 * it;
            const int hp = idx / NQ, q = idx - hp * NQ;
            if (idx < NH * NQ) *reinterpret_cast<uint4*>(in_s + hp * C + 8 * f16_slot<C>(q, hp)) = st[it];
        }
    }
    const int dy1 = py ? 1 : -1, dx1 = px ? 1 : -1;
    const int ky0 = py ? 2 : 1, ky1 = py ? 0 : 3, kx0 = px ? 2 : 1, kx1 = px ? 0 : 3;
    const int kpos[4] = {ky0 * 4 + kx0, ky0 * 4 + kx1, ky1 * 4 + kx0, ky1 * 4 + kx1};
    const int offy[4] = {0, 0, dy1, dy1}, offx[4] = {0, dx1, 0, dx1};
    f32x4e acc[TH];
    f16x8e wv[4], wn[4];
    auto load_w = [&](f16x8e (&w)[4], int kb) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            w[t] = *reinterpret_cast<const f16x8e*>(weight + ((int64_t)kpos[t] * F + fo + n) * C + 32 * kb + 8 * g);
    };
    load_w(wv, 0);
    __syncthreads();
    // one 32-channel block with the weights in `wv`; the two register sets alternate (QPWC_W_NEXT_ALWAYS)
    auto block = [&](f16x8e (&wv)[4], int kb) __attribute__((always_inline)) {
#if QPWC_UPCONV16_PIPE
        // a tap's TH operand reads go out one tap ahead of its matrix instructions (as in conv3x3_mish_f16_kernel)
        f16x8e bb[2][TH];
        auto read_b = [&](f16x8e (&bv)[TH], int t) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < TH; ++m) {
                const int hp = (m + 1 + offy[t]) * kEcHW + n + 1 + offx[t];
                bv[m] = *reinterpret_cast<const f16x8e*>(in_s + hp * C + 8 * f16_slot<C>(4 * kb + g, hp));
            }
        };
        read_b(bb[0], 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t + 1 < 4) read_b(bb[(t + 1) & 1], t + 1);
#pragma unroll
            for (int m = 0; m < TH; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[t], bb[t & 1][m], acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#else
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < TH; ++m) {
                const int hp = (m + 1 + offy[t]) * kEcHW + n + 1 + offx[t];
                const f16x8e bv = *reinterpret_cast<const f16x8e*>(in_s + hp * C + 8 * f16_slot<C>(4 * kb + g, hp));
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[t], bv, acc[m], 0, 0, 0);
            }
#endif
    };
#pragma unroll 1
    for (int fb = 0; fb < nfb; ++fb) {
    fo = 16 * (fblk0 + fb);
    if (fb > 0) load_w(wv, 0);     // (the first block's weights went out before the barrier)
#pragma unroll
    for (int m = 0; m < TH; ++m) acc[m] = f32x4e{0.f, 0.f, 0.f, 0.f};
#if QPWC_W_NEXT_ALWAYS_F16
    {
        int kb = 0;
#pragma unroll 1
        for (; kb + 1 < NKB; kb += 2) {
            load_w(wn, kb + 1);
            __builtin_amdgcn_sched_barrier(0);   // (unfenced, the scheduler sinks the requests to just in front of their use)
            block(wv, kb);
            load_w(wv, kb + 2 < NKB ? kb + 2 : kb);   // (last trip: a harmless re-read)
            __builtin_amdgcn_sched_barrier(0);
            block(wn, kb + 1);
        }
        if (kb < NKB) block(wv, kb);   // odd number of blocks
    }
#else
#pragma unroll 1
    for (int kb = 0; kb < NKB; ++kb) {
        if (kb + 1 < NKB) load_w(wn, kb + 1);
        block(wv, kb);
#pragma unroll
        for (int t = 0; t < 4; ++t) wv[t] = wn[t];
    }
#endif
    const int H2 = 2 * H, W2 = 2 * W;
    __half* ob = out + (int64_t)b * H2 * W2 * out_pixel_stride;
    const __half* sk = reinterpret_cast<const __half*>(skip.p);   // (see upconv4x4s2_mish_kernel)
    f16x4e sv[TH];
    if (sk != nullptr) {
#pragma unroll
        for (int m = 0; m < TH; ++m) {
            const int gy = Y0 + m, gx = X0 + n;
            sv[m] = (gy < H && gx < W)
                        ? *reinterpret_cast<const f16x4e*>(sk + b * skip.bs + (int64_t)(2 * gy + py) * skip.rs +
                                                           (int64_t)(2 * gx + px) * skip.ps + fo + 4 * g)
                        : f16x4e{0, 0, 0, 0};
        }
    }
    const float4 bq = *reinterpret_cast<const float4*>(bias + fo + 4 * g);
#pragma unroll
    for (int m = 0; m < TH; ++m) {
        const int gy = Y0 + m, gx = X0 + n;
        if (gy < H && gx < W) {
            f16x4e o;
            o[0] = (_Float16)enc_mishf(acc[m][0] + bq.x);
            o[1] = (_Float16)enc_mishf(acc[m][1] + bq.y);
            o[2] = (_Float16)enc_mishf(acc[m][2] + bq.z);
            o[3] = (_Float16)enc_mishf(acc[m][3] + bq.w);
            __half* op = ob + ((int64_t)(2 * gy + py) * W2 + 2 * gx + px) * out_pixel_stride + fo + 4 * g;
            *reinterpret_cast<f16x4e*>(op) = o;
            if (sk != nullptr) *reinterpret_cast<f16x4e*>(op + F) = sv[m];
        }
    }
    }   // fb
}

template <int C, int TH>
static int upconv_f16_launch_t(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W, int F,
                               int out_pixel_stride, hipStream_t s, UpSkip skip) {
    const int tiles_x = (W + kEcTW - 1) / kEcTW, tiles_y = (H + TH - 1) / TH;
    const int64_t n_tiles = (int64_t)tiles_x * tiles_y * B;
    if (n_tiles * (F / 16) > INT32_MAX) {
        set_error("upconv4x4s2_mish_f16: too many tiles");
        return QPWC_E_SHAPE;
    }
    // output blocks per workgroup: as many as leave the launch QPWC_UPCONV16_NFB_MIN_WGS workgroups (two per CU)
    int nfb = 1;
    while (nfb * 2 <= F / 16 && (F / 16) % (nfb * 2) == 0 && n_tiles * (F / 16) / (nfb * 2) >= QPWC_UPCONV16_NFB_MIN_WGS) nfb *= 2;
    hipLaunchKernelGGL((upconv4x4s2_mish_f16_kernel<C, TH>), dim3((unsigned)(n_tiles * (F / 16) / nfb)), dim3(256), 0, s,
                       (const __half*)x, (const __half*)weight, (const float*)bias, (__half*)out, H, W, F,
                       out_pixel_stride, tiles_x, tiles_y, (int)n_tiles, skip, nfb);
    return check_launch("upconv4x4s2_mish_f16_kernel");
}

int upconv4x4s2_mish_f16_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                                int C, int F, int out_pixel_stride, hipStream_t s, const void* skip, int64_t skip_bs,
                                int64_t skip_rs, int64_t skip_ps) {
    const UpSkip sk{skip, (long long)skip_bs, (long long)skip_rs, (long long)skip_ps};
    switch (C) {
        case 64: return upconv_f16_launch_t<64, 8>(x, weight, bias, out, B, H, W, F, out_pixel_stride, s, sk);
        case 128: return upconv_f16_launch_t<128, 8>(x, weight, bias, out, B, H, W, F, out_pixel_stride, s, sk);
        case 256: return upconv_f16_launch_t<256, 4>(x, weight, bias, out, B, H, W, F, out_pixel_stride, s, sk);
        default: set_error("upconv4x4s2_mish_f16: C=%d not in {64,128,256}", C); return QPWC_E_SHAPE;
    }
}

// ---------------------------------------------------------------------------
// First encoder layer, enc.0.conv_a (Conv2D 3 -> 16, 3x3, stride 2, 'same', Mish; non_layers.py:402-409)
// straight from the (B,H,W,6) input pair: Split(2) (pwcnet.py:229), the stacking of both frames on the
// batch axis, TensorFlow's 'SAME' padding for even H, W (0 before, 1 after), the convolution, bias and
// Mish in one launch -- the 25 MB input is read once and nothing but the 16-channel output is written.
// K = 27: one v_mfma_f32_16x16x4_f32 per tap with k-slot g = input channel (slot 3 = 0).
// Workgroup = 8 x 16 output pixels of BOTH frames (they share the staged 17 x 33 x 6 input patch);
// wave w -> frame w >> 1, output rows 4 (w & 1) .. + 3.   weight: [9 taps][16 out][4] fp32 (slot 3 = 0).
#ifndef QPWC_FC_ROWS
#define QPWC_FC_ROWS 4     // output rows per wave of the first encoder layer: the workgroup's tile is 2 QPWC_FC_ROWS x 16 (A/B, round 4)
#endif
constexpr int kFcTR = QPWC_FC_ROWS, kFcTH = 2 * kFcTR;
constexpr int kFcIH = 2 * kFcTH + 1, kFcIW = 2 * kEcTW + 1;   // 17 x 33 input pixels at four rows per wave

// T = __half: the fp16-storage form (BASELINE configs[4]) -- pairs and out fp16, the patch converted to fp32 on its
// way into LDS (exact), weights / bias / arithmetic as for fp32, one rounding at the store.
template <typename T>
__global__ __launch_bounds__(256, 4) void first_conv_mish_kernel(const T* __restrict__ x,
                                                                 const float* __restrict__ weight,
                                                                 const float* __restrict__ bias,
                                                                 T* __restrict__ out, int B, int H, int W,
                                                                 int tiles_x, int tiles_y, int in_nchw) {
    __shared__ __attribute__((aligned(16))) float in_s[kFcIH * kFcIW * 6];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int X0 = tx * kEcTW, Y0 = ty * kFcTH;           // output coordinates
    const int Ho = H / 2, Wo = W / 2;
    const T* xb = x + (int64_t)b * H * W * 6;
    // ---- stage the input patch: rows 2 Y0 .. + 16, columns 2 X0 .. + 32, 6 channels (pieces of 2 channels) ----
    if (in_nchw) {   // (B,6,H,W) 'channels_first' input: six planes, lanes walk a row of a plane
        for (int idx = tid; idx < 6 * kFcIH * kFcIW; idx += 256) {
            const int c = idx / (kFcIH * kFcIW), r = idx - c * (kFcIH * kFcIW);
            const int row = r / kFcIW, col = r - row * kFcIW;
            const int gy = 2 * Y0 + row, gx = 2 * X0 + col;
            float v = 0.f;
            if (gy < H && gx < W) v = ld<T>(xb + ((int64_t)c * H + gy) * W + gx);
            in_s[r * 6 + c] = v;
        }
    } else {
        // round 4: every request of the patch goes out before the first LDS write (as a rolled loop each of a thread's
        // seven pieces was loaded, waited for with vmcnt(0) and written in turn: seven memory round trips per workgroup)
        constexpr int NP = (kFcIH * kFcIW * 3 + 255) / 256;
        float2 pv[NP];
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int idx = tid + 256 * it;
            const int row = idx / (kFcIW * 3), e = idx - row * (kFcIW * 3);   // e = float2 index inside the row
            const int gy = 2 * Y0 + row, gx = 2 * X0 + e / 3;
            pv[it] = make_float2(0.f, 0.f);
            if (idx < kFcIH * kFcIW * 3 && gy < H && gx < W) pv[it] = ld2(xb + ((int64_t)gy * W + 2 * X0) * 6 + 2 * e);
        }
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int idx = tid + 256 * it;
            const int row = idx / (kFcIW * 3), e = idx - row * (kFcIW * 3);
            if (idx < kFcIH * kFcIW * 3) *reinterpret_cast<float2*>(in_s + row * (kFcIW * 6) + 2 * e) = pv[it];
        }
    }
    float wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = weight[(k * 16 + n) * 4 + g];   // W[tap][out n][channel g]
    const float4 bq = *reinterpret_cast<const float4*>(bias + 4 * g);
    __syncthreads();

    const int f = wave >> 1;
    T* ob = out + (int64_t)(f * B + b) * Ho * Wo * 16;
    f32x4e acc[kFcTR];
#pragma unroll
    for (int r = 0; r < kFcTR; ++r) acc[r] = f32x4e{0.f, 0.f, 0.f, 0.f};
    // (round 4) a tap's four operand reads are issued together, one tap ahead of the matrix instructions that use them: as
    // the compiler placed them every matrix instruction waited for its own ds_read_b32
    float vb[2][kFcTR];
    auto read_tap = [&](float (&v)[kFcTR], int k) __attribute__((always_inline)) {
        const int ky = k / 3, kx = k - 3 * ky;
#pragma unroll
        for (int r = 0; r < kFcTR; ++r) {
            const int oy = kFcTR * (wave & 1) + r;
            v[r] = in_s[((2 * oy + ky) * kFcIW + 2 * n + kx) * 6 + 3 * f + (g < 3 ? g : 2)];   // (no branch around the read; k-slot 3 is zeroed at its use)
        }
    };
    read_tap(vb[0], 0);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (k + 1 < 9) read_tap(vb[(k + 1) & 1], k + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < kFcTR; ++r)   // independent accumulators (output rows) per tap
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[k], g < 3 ? vb[k & 1][r] : 0.0f, acc[r], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < kFcTR; ++r) {
        const int gy = Y0 + kFcTR * (wave & 1) + r, gx = X0 + n;
        if (gy < Ho && gx < Wo)
            st4q(ob + ((int64_t)gy * Wo + gx) * 16 + 4 * g,
                 make_float4(enc_mishf(acc[r][0] + bq.x), enc_mishf(acc[r][1] + bq.y), enc_mishf(acc[r][2] + bq.z),
                             enc_mishf(acc[r][3] + bq.w)));
    }
}

int first_conv_mish_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                           int layout, int dtype, hipStream_t s) {
    const int tiles_x = (W / 2 + kEcTW - 1) / kEcTW, tiles_y = (H / 2 + kFcTH - 1) / kFcTH;
    const int64_t nblk = (int64_t)tiles_x * tiles_y * B;
    if (nblk > INT32_MAX) {
        set_error("first_conv_mish: too many tiles");
        return QPWC_E_SHAPE;
    }
    if (dtype == QPWC_F32)
        hipLaunchKernelGGL(first_conv_mish_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, s, (const float*)x,
                           (const float*)weight, (const float*)bias, (float*)out, B, H, W, tiles_x, tiles_y,
                           layout == QPWC_NCHW ? 1 : 0);
    else
        hipLaunchKernelGGL(first_conv_mish_kernel<__half>, dim3((unsigned)nblk), dim3(256), 0, s, (const __half*)x,
                           (const float*)weight, (const float*)bias, (__half*)out, B, H, W, tiles_x, tiles_y,
                           layout == QPWC_NCHW ? 1 : 0);
    return check_launch("first_conv_mish_kernel");
}

// ---------------------------------------------------------------------------
// conv_a of the second encoder level: Conv2D(16 -> 32, 3x3, stride 2, 'same', Mish) (non_layers.py:402-409)
// reading the zero-bordered (B, H+1, W+1, 16) output of the level before (the border IS TensorFlow's
// 'SAME' padding for even H, W).  8 x 16 output pixels per workgroup; the 17 x 33 input patch is staged
// with even and odd columns in separate planes, so that the three column taps of 16 neighbouring outputs
// are 16 consecutive pixels of a plane (1 KB contiguous ds_read_b128).  Weights (9 taps x 2 output blocks)
// stay in registers.   weight: [9][32 out][16 in] fp32; x pixel rows are Wp = W + 1 pixels long.
constexpr int kS2IH = 2 * kEcTH + 1;      // 17 input rows
constexpr int kS2PW = kEcTW + 1;          // 17 pixels per parity plane row (even: 17 used, odd: 16 used)

__global__ __launch_bounds__(256, 4) void conv3x3s2_mish_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ weight,
                                                                const float* __restrict__ bias,
                                                                float* __restrict__ out, int H, int W,
                                                                int tiles_x, int tiles_y) {
    constexpr int CI = 16, CO = 32;
    __shared__ __attribute__((aligned(16))) float in_s[2 * kS2IH * kS2PW * CI];   // [parity][row][col/2][16]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int X0 = tx * kEcTW, Y0 = ty * kEcTH;   // output coordinates
    const int Hp = H + 1, Wp = W + 1, Ho = H / 2, Wo = W / 2;
    const float* xb = x + (int64_t)b * Hp * Wp * CI;
    // ---- stage rows 2 Y0 .. + 16, columns 2 X0 .. + 32 (inside the padded input or zero) ----
    for (int idx = tid; idx < kS2IH * 33 * 4; idx += 256) {
        const int q = idx & 3, pxl = idx >> 2;
        const int row = pxl / 33, col = pxl - row * 33;
        const int gy = 2 * Y0 + row, gx = 2 * X0 + col;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy < Hp && gx < Wp) v = *reinterpret_cast<const float4*>(xb + ((int64_t)gy * Wp + gx) * CI + 4 * q);
        const int pix = ((col & 1) * kS2IH + row) * kS2PW + (col >> 1);
        *reinterpret_cast<float4*>(in_s + pix * CI + 4 * ec_slot<CI>(q, pix)) = v;
    }
    f32x4e wv[9][2];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
            wv[k][ft] = *reinterpret_cast<const f32x4e*>(weight + ((int64_t)k * CO + 16 * ft + n) * CI + 4 * g);
    __syncthreads();

    f32x4e acc[2][2];   // [tile row of this wave][output block]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) acc[m][ft] = f32x4e{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            f32x4e bv[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int row = 2 * (2 * wave + m) + ky;            // input row of output row 2 wave + m
                const int par = kx & 1, pc = n + (kx >> 1);          // column 2 n + kx -> plane, index
                const int pix = (par * kS2IH + row) * kS2PW + pc;
                bv[m] = *reinterpret_cast<const f32x4e*>(in_s + pix * CI + 4 * ec_slot<CI>(g, pix));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft)
                        acc[m][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ky * 3 + kx][ft][t], bv[m][t], acc[m][ft], 0, 0, 0);
        }
    float* ob = out + (int64_t)b * Ho * Wo * CO;
    const float4 bq2[2] = {*reinterpret_cast<const float4*>(bias + 4 * g), *reinterpret_cast<const float4*>(bias + 16 + 4 * g)};
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
        const float4 bq = bq2[ft];     // (both requested before the first store: vmcnt counts stores too)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int gy = Y0 + 2 * wave + m, gx = X0 + n;
            if (gy < Ho && gx < Wo)
                *reinterpret_cast<float4*>(ob + ((int64_t)gy * Wo + gx) * CO + 16 * ft + 4 * g) =
                    make_float4(enc_mishf(acc[m][ft][0] + bq.x), enc_mishf(acc[m][ft][1] + bq.y),
                                enc_mishf(acc[m][ft][2] + bq.z), enc_mishf(acc[m][ft][3] + bq.w));
        }
    }
}

// ---------------------------------------------------------------------------
// conv_a of the wide encoder levels: Conv2D(CI -> 2 CI, 3x3, stride 2, 'same', Mish) for CI = 32 / 64 / 128
// on the zero-bordered (B, H+1, W+1, CI) output of the level before.  Work split as in
// conv3x3_mish_wide_kernel (a wave = one block of 16 outputs for the TH x 16 output pixels of the tile,
// 32 input channels of weights in registers at a time, next block prefetched; workgroup = 64 outputs,
// grid = tiles x 2 CI / 64), input patch as in conv3x3s2_mish_kernel (even and odd columns in separate
// LDS planes, so the column taps of 16 neighbouring outputs are 16 consecutive plane pixels).
template <int CI, int TH>
__global__ __launch_bounds__(256, (2 * (2 * TH + 1) * (kEcTW + 1) * CI * 4 > 80 * 1024) ? 1 : 2) void conv3x3s2_mish_wide_kernel(const float* __restrict__ x,
                                                                     const float* __restrict__ weight,
                                                                     const float* __restrict__ bias,
                                                                     float* __restrict__ out, int H, int W,
                                                                     int tiles_x, int tiles_y, int n_tiles) {
    constexpr int CO = 2 * CI, NQ = CI / 4, NKB = CI / 32;
    constexpr int IH = 2 * TH + 1, PW = kEcTW + 1;         // input rows of a tile; pixels per plane row
    constexpr int NPX = IH * 33;                           // staged input pixels (33 columns)
    constexpr int NST = (NPX * NQ + 255) / 256;
    __shared__ __attribute__((aligned(16))) float in_s[2 * IH * PW * CI];   // [parity][row][col/2][CI]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int slice = blockIdx.x / n_tiles;
    const int tile = xcd_swizzle(blockIdx.x % n_tiles, n_tiles);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int X0 = tx * kEcTW, Y0 = ty * TH;               // output coordinates
    const int Hp = H + 1, Wp = W + 1, Ho = H / 2, Wo = W / 2;
    const int fo = 64 * slice + 16 * wave;
    const float* xb = x + (int64_t)b * Hp * Wp * CI;
    auto slot = [](int q, int pix) { return CI == 32 ? (q ^ ((pix >> 1) & 7)) : (q ^ (pix & 15)); };
    // rows 2 Y0 .. + 2 TH, columns 2 X0 .. + 32 (inside the padded input or zero); all loads of a thread in
    // flight at once (chunks of 8 cost one exposed memory latency per chunk)
    constexpr int kChunk = NST;
#pragma unroll
    for (int base = 0; base < NST; base += kChunk) {
        float4 st[kChunk];
#pragma unroll
        for (int i = 0; i < kChunk; ++i) {
            const int idx = tid + 256 * (base + i);
            const int pxl = idx / NQ, q = idx - pxl * NQ;
            const int row = pxl / 33, col = pxl - row * 33;
            const int gy = 2 * Y0 + row, gx = 2 * X0 + col;
            st[i] = (base + i < NST && idx < NPX * NQ && gy < Hp && gx < Wp)
                        ? *reinterpret_cast<const float4*>(xb + ((int64_t)gy * Wp + gx) * CI + 4 * q)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < kChunk; ++i) {
            const int idx = tid + 256 * (base + i);
            const int pxl = idx / NQ, q = idx - pxl * NQ;
            const int row = pxl / 33, col = pxl - row * 33;
            const int pix = ((col & 1) * IH + row) * PW + (col >> 1);
            if (base + i < NST && idx < NPX * NQ) *reinterpret_cast<float4*>(in_s + pix * CI + 4 * slot(q, pix)) = st[i];
        }
    }
    f32x4e acc[TH];
#pragma unroll
    for (int m = 0; m < TH; ++m) acc[m] = f32x4e{0.f, 0.f, 0.f, 0.f};
    f32x4e wv[9][2], wn[9][2];
    auto load_w = [&](f32x4e (&w)[9][2], int kb) {
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
                w[k][kc] = *reinterpret_cast<const f32x4e*>(weight + ((int64_t)k * CO + fo + n) * CI + 32 * kb + 16 * kc + 4 * g);
    };
    load_w(wv, 0);
    __syncthreads();
    // one 32-channel block with the weights in `wv`; the two register sets alternate (QPWC_W_NEXT_ALWAYS)
    auto block = [&](f32x4e (&wv)[9][2], int kb) __attribute__((always_inline)) {
#if QPWC_S2_PIPE
        // one step = (tap, 16-channel chunk); the operand reads of step i + 1 go out before the matrix instructions of
        // step i (as in conv3x3_mish_wide_kernel)
        f32x4e bb[2][TH];
        auto read_b = [&](f32x4e (&bv)[TH], int i) __attribute__((always_inline)) {
            const int tap = i >> 1, kc = i & 1, ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int m = 0; m < TH; ++m) {
                const int pix = ((kx & 1) * IH + 2 * m + ky) * PW + n + (kx >> 1);
                bv[m] = *reinterpret_cast<const f32x4e*>(in_s + pix * CI + 4 * slot(8 * kb + 4 * kc + g, pix));
            }
        };
        read_b(bb[0], 0);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            if (i + 1 < 18) read_b(bb[(i + 1) & 1], i + 1);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < TH; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i >> 1][i & 1][t], bb[i & 1][m][t], acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#else
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    f32x4e bv[TH];
#pragma unroll
                    for (int m = 0; m < TH; ++m) {
                        const int pix = ((kx & 1) * IH + 2 * m + ky) * PW + n + (kx >> 1);
                        bv[m] = *reinterpret_cast<const f32x4e*>(in_s + pix * CI + 4 * slot(8 * kb + 4 * kc + g, pix));
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int m = 0; m < TH; ++m)
                            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ky * 3 + kx][kc][t], bv[m][t], acc[m], 0, 0, 0);
                }
#endif
    };
#if QPWC_W_NEXT_ALWAYS
    {
        int kb = 0;
#pragma unroll 1
        for (; kb + 1 < NKB; kb += 2) {
            load_w(wn, kb + 1);
            __builtin_amdgcn_sched_barrier(0);   // (unfenced, the scheduler sinks the requests to just in front of their use)
            block(wv, kb);
            load_w(wv, kb + 2 < NKB ? kb + 2 : kb);   // (last trip: a harmless re-read)
            __builtin_amdgcn_sched_barrier(0);
            block(wn, kb + 1);
        }
        if (kb < NKB) block(wv, kb);   // odd number of blocks
    }
#else
#pragma unroll 1
    for (int kb = 0; kb < NKB; ++kb) {
        if (kb + 1 < NKB) load_w(wn, kb + 1);
        block(wv, kb);
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) wv[k][kc] = wn[k][kc];
    }
#endif
    float* ob = out + (int64_t)b * Ho * Wo * CO;
    const float4 bq = *reinterpret_cast<const float4*>(bias + fo + 4 * g);
#pragma unroll
    for (int m = 0; m < TH; ++m) {
        const int gy = Y0 + m, gx = X0 + n;
        if (gy < Ho && gx < Wo)
            *reinterpret_cast<float4*>(ob + ((int64_t)gy * Wo + gx) * CO + fo + 4 * g) =
                make_float4(enc_mishf(acc[m][0] + bq.x), enc_mishf(acc[m][1] + bq.y),
                            enc_mishf(acc[m][2] + bq.z), enc_mishf(acc[m][3] + bq.w));
    }
}

template <int CI, int TH>
static int conv3x3s2_mish_wide_launch(const void* x, const void* weight, const void* bias, void* out, int B,
                                      int H, int W, hipStream_t s) {
    const int tiles_x = (W / 2 + kEcTW - 1) / kEcTW, tiles_y = (H / 2 + TH - 1) / TH;
    const int64_t n_tiles = (int64_t)tiles_x * tiles_y * B;
    const int slices = 2 * CI / 64;
    if (n_tiles * slices > INT32_MAX) {
        set_error("conv3x3s2_mish: too many tiles");
        return QPWC_E_SHAPE;
    }
    hipLaunchKernelGGL((conv3x3s2_mish_wide_kernel<CI, TH>), dim3((unsigned)(n_tiles * slices)), dim3(256), 0, s,
                       (const float*)x, (const float*)weight, (const float*)bias, (float*)out, H, W, tiles_x, tiles_y,
                       (int)n_tiles);
    return check_launch("conv3x3s2_mish_wide_kernel");
}

// CI = 16 -> conv3x3s2_mish_kernel (both output blocks in one wave); 32 / 64 / 128 -> the wide kernel
int conv3x3s2_mish_any_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                              int CI, hipStream_t s);

int conv3x3s2_mish_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                          hipStream_t s) {
    const int tiles_x = (W / 2 + kEcTW - 1) / kEcTW, tiles_y = (H / 2 + kEcTH - 1) / kEcTH;
    const int64_t nblk = (int64_t)tiles_x * tiles_y * B;
    if (nblk > INT32_MAX) {
        set_error("conv3x3s2_mish: too many tiles");
        return QPWC_E_SHAPE;
    }
    hipLaunchKernelGGL(conv3x3s2_mish_kernel, dim3((unsigned)nblk), dim3(256), 0, s, (const float*)x,
                       (const float*)weight, (const float*)bias, (float*)out, H, W, tiles_x, tiles_y);
    return check_launch("conv3x3s2_mish_kernel");
}

int conv3x3s2_mish_any_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                              int CI, hipStream_t s) {
    switch (CI) {
        case 16: return conv3x3s2_mish_launch(x, weight, bias, out, B, H, W, s);
        case 32: return conv3x3s2_mish_wide_launch<32, 4>(x, weight, bias, out, B, H, W, s);
        case 64: return conv3x3s2_mish_wide_launch<64, 2>(x, weight, bias, out, B, H, W, s);
        case 128: return conv3x3s2_mish_wide_launch<128, 2>(x, weight, bias, out, B, H, W, s);
        default: set_error("conv3x3s2_mish: C_in=%d not in {16,32,64,128}", CI); return QPWC_E_SHAPE;
    }
}

// fp16-storage twin of the stride-2 kernels (conv_a of encoder levels 2..5, BASELINE configs[4]): x_padded
// (B, H+1, W+1, CI), weight ([9][2 CI][CI]) and out (B, H/2, W/2, 2 CI) fp16, bias fp32.  TH x 16 output pixels per
// workgroup; the (2 TH + 1) x 33 input patch sits in LDS with even and odd columns in separate planes, so the taps of
// 16 neighbouring outputs are 16 consecutive pixels of a plane; pixels of max(CI, 32) halves, chunk swizzle as in
// conv3x3_mish_f16_kernel.  A wave owns one block of 16 outputs (2 CI = 32: two waves per block, half the rows each)
// with its weights for 32 input channels in registers; one v_mfma_f32_16x16x32_f16 per tap, row and 32-channel block.
template <int CI, int TH>
__global__ __launch_bounds__(256) void conv3x3s2_mish_f16_kernel(const __half* __restrict__ x,
                                                                const __half* __restrict__ weight,
                                                                const float* __restrict__ bias,
                                                                __half* __restrict__ out, int H, int W, int tiles_x,
                                                                int tiles_y, int n_tiles) {
    constexpr int CO = 2 * CI;
    constexpr int CP = CI < 32 ? 32 : CI;
    constexpr int NQ = CP / 8, NQG = CI / 8, NKB = CP / 32;
    constexpr int IH = 2 * TH + 1, PW = kEcTW + 1;            // patch rows, pixels per parity-plane row
    constexpr int NB = CO / 16;                               // output blocks
    constexpr int WB = NB < 4 ? NB : 4, WR = 4 / WB;          // waves across blocks / across row groups
    constexpr int RW = TH / WR;                               // rows per wave
    constexpr int NCH = IH * (2 * kEcTW + 1) * NQ;            // staged chunks
    constexpr int NST = (NCH + 255) / 256;
    __shared__ __attribute__((aligned(16))) __half in_s[2 * IH * PW * CP];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int slice = blockIdx.x / n_tiles;
    const int tile = xcd_swizzle(blockIdx.x % n_tiles, n_tiles);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int X0 = tx * kEcTW, Y0 = ty * TH;                  // output coordinates
    const int Hp = H + 1, Wp = W + 1, Ho = H / 2, Wo = W / 2;
    const __half* xb = x + (int64_t)b * Hp * Wp * CI;
    {
        uint4 st[NST];
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = tid + 256 * it;
            const int q = idx % NQ, pc = idx / NQ;
            const int row = pc / (2 * kEcTW + 1), col = pc - row * (2 * kEcTW + 1);
            const int gy = 2 * Y0 + row, gx = 2 * X0 + col;
            st[it] = (idx < NCH && q < NQG && gy < Hp && gx < Wp)
                         ? *reinterpret_cast<const uint4*>(xb + ((int64_t)gy * Wp + gx) * CI + 8 * q)
                         : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = tid + 256 * it;
            const int q = idx % NQ, pc = idx / NQ;
            const int row = pc / (2 * kEcTW + 1), col = pc - row * (2 * kEcTW + 1);
            const int hp = ((col & 1) * IH + row) * PW + (col >> 1);
            if (idx < NCH) *reinterpret_cast<uint4*>(in_s + hp * CP + 8 * f16_slot<CP>(q, hp)) = st[it];
        }
    }
    const int blk = WB == 4 ? 4 * slice + wave : (wave % WB);
    const int r0 = WB == 4 ? 0 : (wave / WB) * RW;
    const int fo = 16 * blk;
    f32x4e acc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = f32x4e{0.f, 0.f, 0.f, 0.f};
    const bool kvalid = 8 * g < CI;
    f16x8e wv[9], wn[9];
    auto load_w = [&](f16x8e (&w)[9], int kb) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            w[tap] = f16x8e{0, 0, 0, 0, 0, 0, 0, 0};
            if (kvalid) w[tap] = *reinterpret_cast<const f16x8e*>(weight + ((int64_t)tap * CO + fo + n) * CI + 32 * kb + 8 * g);
        }
    };
    load_w(wv, 0);
    __syncthreads();
    // one 32-channel block with the weights in `wv`; the two register sets alternate (QPWC_W_NEXT_ALWAYS)
    auto block = [&](f16x8e (&wv)[9], int kb) __attribute__((always_inline)) {
#if QPWC_S2_16_PIPE
        // operand reads two taps ahead of their matrix instructions (RW of 16 cycles per tap: one tap would not cover
        // the LDS latency)
        f16x8e bb[3][RW];
        auto read_b = [&](f16x8e (&bv)[RW], int tap) __attribute__((always_inline)) {
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int hp = ((kx & 1) * IH + 2 * (r0 + r) + ky) * PW + n + (kx >> 1);
                bv[r] = *reinterpret_cast<const f16x8e*>(in_s + hp * CP + 8 * f16_slot<CP>(4 * kb + g, hp));
            }
        };
        read_b(bb[0], 0);
        read_b(bb[1], 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 2 < 9) read_b(bb[(tap + 2) % 3], tap + 2);
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[tap], bb[tap % 3][r], acc[r], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#else
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int hp = ((kx & 1) * IH + 2 * (r0 + r) + ky) * PW + n + (kx >> 1);
                const f16x8e bv = *reinterpret_cast<const f16x8e*>(in_s + hp * CP + 8 * f16_slot<CP>(4 * kb + g, hp));
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[tap], bv, acc[r], 0, 0, 0);
            }
        }
#endif
    };
#if QPWC_W_NEXT_ALWAYS_F16
    {
        int kb = 0;
#pragma unroll 1
        for (; kb + 1 < NKB; kb += 2) {
            load_w(wn, kb + 1);
            __builtin_amdgcn_sched_barrier(0);   // (unfenced, the scheduler sinks the requests to just in front of their use)
            block(wv, kb);
            load_w(wv, kb + 2 < NKB ? kb + 2 : kb);   // (last trip: a harmless re-read)
            __builtin_amdgcn_sched_barrier(0);
            block(wn, kb + 1);
        }
        if (kb < NKB) block(wv, kb);   // odd number of blocks
    }
#else
#pragma unroll 1
    for (int kb = 0; kb < NKB; ++kb) {
        if (kb + 1 < NKB) load_w(wn, kb + 1);
        block(wv, kb);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) wv[tap] = wn[tap];
    }
#endif
    __half* ob = out + (int64_t)b * Ho * Wo * CO;
    const float4 bq = *reinterpret_cast<const float4*>(bias + fo + 4 * g);
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int gy = Y0 + r0 + r, gx = X0 + n;
        if (gy < Ho && gx < Wo)
            st4q(ob + ((int64_t)gy * Wo + gx) * CO + fo + 4 * g,
                 make_float4(enc_mishf(acc[r][0] + bq.x), enc_mishf(acc[r][1] + bq.y), enc_mishf(acc[r][2] + bq.z),
                             enc_mishf(acc[r][3] + bq.w)));
    }
}

template <int CI, int TH>
static int conv3x3s2_mish_f16_launch_t(const void* x, const void* weight, const void* bias, void* out, int B, int H,
                                       int W, hipStream_t s) {
    constexpr int NSL = (2 * CI / 16) < 4 ? 1 : (2 * CI / 16) / 4;
    const int tiles_x = (W / 2 + kEcTW - 1) / kEcTW, tiles_y = (H / 2 + TH - 1) / TH;
    const int64_t n_tiles = (int64_t)tiles_x * tiles_y * B;
    if (n_tiles * NSL > INT32_MAX) {
        set_error("conv3x3s2_mish_f16: too many tiles");
        return QPWC_E_SHAPE;
    }
    hipLaunchKernelGGL((conv3x3s2_mish_f16_kernel<CI, TH>), dim3((unsigned)(n_tiles * NSL)), dim3(256), 0, s,
                       (const __half*)x, (const __half*)weight, (const float*)bias, (__half*)out, H, W, tiles_x,
                       tiles_y, (int)n_tiles);
    return check_launch("conv3x3s2_mish_f16_kernel");
}

int conv3x3s2_mish_f16_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                              int CI, hipStream_t s) {
    switch (CI) {
        case 16: return conv3x3s2_mish_f16_launch_t<16, 8>(x, weight, bias, out, B, H, W, s);
        case 32: return conv3x3s2_mish_f16_launch_t<32, 8>(x, weight, bias, out, B, H, W, s);
        case 64: return conv3x3s2_mish_f16_launch_t<64, 4>(x, weight, bias, out, B, H, W, s);
        case 128: return conv3x3s2_mish_f16_launch_t<128, 2>(x, weight, bias, out, B, H, W, s);
        default: set_error("conv3x3s2_mish_f16: C_in=%d not in {16,32,64,128}", CI); return QPWC_E_SHAPE;
    }
}

int conv3x3_mish_launch(const void* x, const void* weight, const void* bias, void* out, int B, int H, int W,
                        int C, int pad_h, int pad_w, hipStream_t s) {
    const int tiles_x = (W + kEcTW - 1) / kEcTW, tiles_y = (H + kEcTH - 1) / kEcTH;
    constexpr int NT = QPWC_ENC_NT;
    const int groups_x = (tiles_x + NT - 1) / NT;
    const int64_t nblk = (int64_t)groups_x * tiles_y * B;
    if (nblk > INT32_MAX) {
        set_error("conv3x3_mish: too many tiles");
        return QPWC_E_SHAPE;
    }
    const dim3 grid((unsigned)nblk);
    if (C == 16)
        hipLaunchKernelGGL((conv3x3_mish_kernel<16, NT>), grid, dim3(256), 0, s, (const float*)x, (const float*)weight,
                           (const float*)bias, (float*)out, H, W, pad_h, pad_w, tiles_x, tiles_y, groups_x);
    else if (C == 32)
        hipLaunchKernelGGL((conv3x3_mish_kernel<32, NT>), grid, dim3(256), 0, s, (const float*)x, (const float*)weight,
                           (const float*)bias, (float*)out, H, W, pad_h, pad_w, tiles_x, tiles_y, groups_x);
    else if (C == 64)
        return conv3x3_mish_wide_launch<64, 4>(x, weight, bias, out, B, H, W, pad_h, pad_w, s);
    else if (C == 128)
        return conv3x3_mish_wide_launch<128, 4>(x, weight, bias, out, B, H, W, pad_h, pad_w, s);
    else if (C == 256)
        return conv3x3_mish_wide_launch<256, 2>(x, weight, bias, out, B, H, W, pad_h, pad_w, s);
    else {
        set_error("conv3x3_mish: C=%d not in {16,32,64,128,256}", C);
        return QPWC_E_SHAPE;
    }
    return check_launch("conv3x3_mish_kernel");
}

}  // namespace qpwc

#ifdef QPWC_ENC_STAMP
extern "C" int qpwc_debug_enc_census(long long* out, int n) {
    if (n > 4096 * 6) n = 4096 * 6;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(qpwc::g_enc_census), n * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
#endif
