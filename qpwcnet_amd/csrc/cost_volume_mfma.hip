// CostVolume forward on the fp32 matrix cores of gfx950 (MI355X, CDNA4, wave64).
//
// Reference semantics: qpwcnet/core/layers.py:72-100 == layers.py:128-132:
//   out[b,y,x,(dy+4)*9+(dx+4)] = lrelu( mean_c prv[b,y,x,c] * nxt0[b,y+dy,x+dx,c] )
//
// The correlation of a 4x4 pixel tile with its 12x12 neighbourhood is nine
// 16x16 dot-product blocks over C: exactly v_mfma_f32_16x16x4_f32 (exact fp32
// FMA chains, same rate as the vector ALU but 1 VGPR per operand per 1024 FMAs).
//   rows  (A operand) = the 16 pixels of one 4x4 block of nxt   (9 blocks / tile)
//   cols  (B operand) = the 16 pixels of the prv tile
//   k                 = channels; lane (pixel n = lane&15, slot g = lane>>4) loads
//                       CPL consecutive channels [step*4*CPL + g*CPL, +CPL) with 16-byte
//                       loads straight from global/L2 (the 4 lanes of a pixel cover a
//                       contiguous 16*CPL-byte run: one full 128-B line for CPL = 8),
//                       and feeds them to CPL successive k-steps.  No LDS staging:
//                       operands go HBM/L2 -> VGPR -> matrix core.
// 81 of the 144 products per pixel are wanted (56 %); the other 63 are the price of
// the dense 16x16 shape and never leave the CU.
//
// Result lane (col = pixel p, rows 4g..4g+3) holds 4 x-consecutive neighbours, so
// each accumulator leaves as ONE ds_write_b128 into a per-wave [16 px][12x12] frame
// (pixel stride 148 floats: conflict free).  Reading the frame back with the
// window offset (py,px) yields the 81 channels in output order, and the wave
// stores whole 4-pixel rows: 324 contiguous floats, 256 B per store instruction.
//
// Coarse levels have few pixels and many channels (8x16x256): KS waves of a
// workgroup split the channel range of one tile (split-K) and their partial
// frames are summed during the read-back, so every level fills the chip.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace qpwc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFramePS = 148;                 // floats per pixel frame (144 + 4 pad)
constexpr int kFrameFloats = 16 * kFramePS;   // per wave
constexpr int kRowStep = 4 * kFramePS + 12;   // frame offset of the next tile row (py+1, same px)

// Frame window -> output rows of one 4x4 tile, for ONE wave that owns the whole frame.
// The CU's vector-memory pipeline is what this kernel family saturates (s_memtime
// stamps: a third of a wave's life was spent waiting to ISSUE its 24 dword stores), so
// the dense case stores 4 consecutive output floats per lane: a tile row of 324 floats
// is 81 x 16 B, rows are 16-byte aligned when W % 4 == 0, and a wave needs 8 store
// instructions instead of 24.  All LDS reads are issued before the first use.
// Frame byte offsets of the four consecutive output floats 4q..4q+3 of a tile row (q = lane and
// lane + 64): element e = 81*px + 9*ky + kx of the 4 px x 81 row sits at frame float
// px*(kFramePS+1) + 12*ky + kx.  A per-lane table (one 16-byte load) replaces ~100 vector-ALU
// instructions of constant divisions per wave -- the SIMD's issue slots are the contended resource
// here (matrix cores 53 % + vector ALU ~35 % busy on the same SIMD).
struct FoffTable {
    unsigned short v[64][8];
};
// S = floats per output pixel: 81 (dense) or 84 (81 channels + 3 zero pads: 16-byte aligned pixels)
constexpr FoffTable make_foff_table(int S) {
    FoffTable t{};
    for (int lane = 0; lane < 64; ++lane)
        for (int it = 0; it < 2; ++it) {
            const int q = lane + 64 * it;
            const int qq = q < S ? q : 0;
            for (int c = 0; c < 4; ++c) {
                const int e = 4 * qq + c;
                const int epx = e / S, k = e - S * epx;
                const int ky = k / 9, kx = k - 9 * ky;
                // pads (k = 81..83: elements 1..3 of the float4 that starts at channel 80) read channel
                // 80's frame entry and are zeroed after the read
                const int kk = k < 81 ? k : 80;
                const int kky = kk / 9, kkx = kk - 9 * kky;
                (void)ky; (void)kx;
                t.v[lane][4 * it + c] = (unsigned short)(4 * (epx * kFramePS + epx + kky * 12 + kkx));
            }
        }
    return t;
}
__device__ const FoffTable kFoffTable = make_foff_table(81);
__device__ const FoffTable kFoffTable84 = make_foff_table(84);

__device__ __forceinline__ uint4 load_foff(int lane, int out_pix_stride = 81) {
    return out_pix_stride == 84 ? *reinterpret_cast<const uint4*>(&kFoffTable84.v[lane][0])
                                : *reinterpret_cast<const uint4*>(&kFoffTable.v[lane][0]);
}

// `tab` = load_foff(lane), fetched by the caller long before the epilogue.
// Returns true when the dense path ran (exactly 8 store instructions per wave).
template <typename T, bool PAD84>
__device__ __forceinline__ bool store_tile_impl(const float* fr, T* ob, int lane, int x0, int y0, int H,
                                           int W, int out_pix_stride, float slope, float inv_c,
                                           float cf, uint4 tab) {
    const int row_stride = W * out_pix_stride;
    // dense rows: 4 px x 81 floats, or 4 px x 84 with the 3 pads of every pixel written as zeros (`tab`
    // then comes from the 84-wide table; the caller asked for a zero-padded 84-channel volume)
    if (out_pix_stride == (PAD84 ? 84 : 81) && (W & 3) == 0 && x0 + 4 <= W && y0 + 4 <= H &&
        (reinterpret_cast<uintptr_t>(ob) & (4 * sizeof(T) - 1)) == 0) {  // wave-uniform
        constexpr bool padded = PAD84;
        // lane owns elements 4*q .. 4*q+3 of the 324-float row, q = lane (+64 for lanes 0..16)
        const unsigned tw[4] = {tab.x, tab.y, tab.z, tab.w};
        const char* frb = reinterpret_cast<const char*>(fr);
        float v[4][2][4];
#pragma unroll
        for (int row = 0; row < 4; ++row)
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned w = tw[2 * it + (c >> 1)];
                    const unsigned off = (c & 1) ? (w >> 16) : (w & 0xffffu);
                    v[row][it][c] = *reinterpret_cast<const float*>(frb + off + row * (kRowStep * 4));
                }
        if (padded) {
            // float4 q of a row holds channels 80..83 of a pixel when q % 21 == 20: elements 1..3 are pads
            const bool pad0 = lane % 21 == 20, pad1 = (lane + 64) % 21 == 20;
#pragma unroll
            for (int row = 0; row < 4; ++row)
#pragma unroll
                for (int c = 1; c < 4; ++c) {
                    v[row][0][c] = pad0 ? 0.0f : v[row][0][c];
                    v[row][1][c] = pad1 ? 0.0f : v[row][1][c];
                }
        }
        const bool second = lane < (padded ? 20 : 17);  // float4 64..80 (..83)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        auto emit = [&](auto act) {
#pragma unroll
            for (int row = 0; row < 4; ++row) {
                T* orow = ob + (int64_t)row * row_stride;
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const float4 o = act(v[row][it]);
                    if (it == 0 || second) st4(orow + 4 * (lane + 64 * it), o);
                }
            }
        };
        // mean then LeakyReLU.  For 0 <= slope <= 1 and an exact 1/C (power of two),
        // lrelu(v/C) == max(v*(1/C), v*(slope/C)) bit for bit: two packed multiplies and a max
        // instead of multiply, compare, multiply, select.
        if (inv_c > 0.f && slope >= 0.f && slope <= 1.f) {  // wave-uniform
            const float k1 = inv_c, k2 = inv_c * slope;
            emit([&](const float* x) {
                const f32x2 lo = {x[0], x[1]}, hi = {x[2], x[3]};
                const f32x2 a0 = lo * k1, b0 = lo * k2, a1 = hi * k1, b1 = hi * k2;
                return make_float4(fmaxf(a0.x, b0.x), fmaxf(a0.y, b0.y), fmaxf(a1.x, b1.x), fmaxf(a1.y, b1.y));
            });
        } else if (inv_c > 0.f) {
            emit([&](const float* x) {
                return make_float4(lrelu(x[0] * inv_c, slope), lrelu(x[1] * inv_c, slope),
                                   lrelu(x[2] * inv_c, slope), lrelu(x[3] * inv_c, slope));
            });
        } else {
            emit([&](const float* x) {
                return make_float4(lrelu(x[0] / cf, slope), lrelu(x[1] / cf, slope), lrelu(x[2] / cf, slope),
                                   lrelu(x[3] / cf, slope));
            });
        }
        return true;
    }
    // general case: strided output (concat buffer), ragged image edge
    int foff[6];
    unsigned goff[6];
    int epx[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int e = lane + 64 * s;     // element of the 4 px * 81 row
        const int ee = e < 324 ? e : 0;  // s == 5 covers elements 320..323 only
        epx[s] = ee / 81;
        const int k = ee - 81 * epx[s];
        const int ky = k / 9, kx = k - 9 * ky;
        foff[s] = epx[s] * kFramePS + epx[s] + ky * 12 + kx;  // window origin (py, px) + (ky, kx)
        goff[s] = (unsigned)(epx[s] * out_pix_stride + k);
    }
    float v[4][6];
#pragma unroll
    for (int row = 0; row < 4; ++row)
#pragma unroll
        for (int s = 0; s < 6; ++s) v[row][s] = fr[foff[s] + row * kRowStep];
    const bool tail = lane < 4;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        T* orow = ob + (int64_t)row * row_stride;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const float m = inv_c > 0.f ? v[row][s] * inv_c : v[row][s] / cf;
            if ((s < 5 || tail) && y0 + row < H && x0 + epx[s] < W) st(orow + goff[s], lrelu(m, slope));
        }
    }
    return false;
}

template <typename T>
__device__ __forceinline__ bool store_tile(const float* fr, T* ob, int lane, int x0, int y0, int H, int W,
                                           int out_pix_stride, float slope, float inv_c, float cf, uint4 tab,
                                           bool pad84 = false) {
    if (pad84)  // wave-uniform
        return store_tile_impl<T, true>(fr, ob, lane, x0, y0, H, W, out_pix_stride, slope, inv_c, cf, tab);
    return store_tile_impl<T, false>(fr, ob, lane, x0, y0, H, W, out_pix_stride, slope, inv_c, cf, tab);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned kOob = 0x80000000u;  // >= any descriptor size we accept: the load returns 0

// WarpV2's clamp-mode blend (common.h: blend<QPWC_WARP_CLAMP>) on two channels at a time: the same three lerps, every
// subtract / multiply / add rounded separately (contraction off), as packed fp32 instructions (v_pk_add_f32 /
// v_pk_mul_f32) -- bit-identical to the scalar form, half the vector-ALU issue slots of the gather.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 blend_clamp2(float ax, float ay, f32x2 tl, f32x2 tr, f32x2 bl, f32x2 br) {
#pragma clang fp contract(off)
    const f32x2 top = ax * (tr - tl) + tl;
    const f32x2 bot = ax * (br - bl) + bl;
    return ay * (bot - top) + top;
}
#ifndef QPWC_PK_BLEND
#define QPWC_PK_BLEND 0   // 1 = packed fp32 blend: bit-identical, measured +-0 (B=8 L4 47.5 vs 47.5-47.9 us), so the scalar code of common.h stays
#endif
__device__ __forceinline__ u32x4 blend_clamp_chunk(float ax, float ay, u32x4 tl, u32x4 tr, u32x4 bl, u32x4 br) {
    u32x4 v;
    if (!QPWC_PK_BLEND) {
        Taps t;
        t.ax = ax;
        t.ay = ay;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            v[e] = __float_as_uint(blend<QPWC_WARP_CLAMP>(t, __uint_as_float(tl[e]), __uint_as_float(tr[e]),
                                                          __uint_as_float(bl[e]), __uint_as_float(br[e])));
        return v;
    }
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
        const f32x2 r = blend_clamp2(ax, ay, f32x2{__uint_as_float(tl[e]), __uint_as_float(tl[e + 1])},
                                     f32x2{__uint_as_float(tr[e]), __uint_as_float(tr[e + 1])},
                                     f32x2{__uint_as_float(bl[e]), __uint_as_float(bl[e + 1])},
                                     f32x2{__uint_as_float(br[e]), __uint_as_float(br[e + 1])});
        v[e] = __float_as_uint(r.x);
        v[e + 1] = __float_as_uint(r.y);
    }
    return v;
}

// Region order of the workgroup-shared kernels: column STRIPS of 128 pixels, row-major inside a strip.  An XCD takes a
// contiguous run of this order (xcd_swizzle) and has ~100 regions in flight: in plain row-major order over a 1024-pixel
// wide level (BASELINE config 4) that is one whole row of regions, whose 16-row neighbourhood plus the 84-float
// outputs do not survive in the 4 MB L2 until the next row of regions asks for the shared half again -- measured HBM
// reads 1.44 x (plain) / 1.9 x (fused) the algorithmic bytes (profiles/r03_pmc_*_c4.txt).  In a 128-pixel strip the
// regions in flight form a compact patch and the next row of the strip follows within a few workgroups.
// Levels up to 128 pixels wide keep the old order (one strip).  All scalar arithmetic.
template <int SW>   // strip width in regions (a power of two)
__device__ __forceinline__ void region_coords(int region, int regs_x, int regs_y, int& rx, int& ry, int& b) {
    const int per_img = regs_x * regs_y;
    b = region / per_img;
    int i = region - b * per_img;
    if (regs_x <= SW) {
        ry = i / regs_x;
        rx = i - ry * regs_x;
        return;
    }
    const int strip = SW * regs_y, full = regs_x / SW;
    const int s = i / strip;
    if (s < full) {
        i -= s * strip;
        ry = i / SW;
        rx = s * SW + (i - ry * SW);
    } else {
        const int rem = regs_x - full * SW;
        i -= full * strip;
        ry = i / rem;
        rx = full * SW + (i - ry * rem);
    }
}

// Operand fetch of the per-wave kernel, two stages:
//  (1) COALESCED raw buffer loads (descriptor in SGPRs + 32-bit lane byte offset + scalar
//      step offset; offsets beyond the descriptor return zero = ZeroPadding2D for free).
//      Load layout: lane l -> pixel l>>2 of the 4x4 block, quarter l&3, so the 4 adjacent
//      lanes of a pixel read 64 contiguous bytes per instruction (the texture addresser
//      merges adjacent lanes only: in matrix-core layout, where adjacent lanes are adjacent
//      PIXELS, every lane was its own L1 access and the kernel was TA-bound);
//  (2) one ds_bpermute per register moves the data to the matrix-core layout
//      (lane l -> pixel l&15, k-slot l>>4): source lane 4*(l&15) + (l>>4).
// prv and nxt use the same channel-to-k map, and the sum over channels does not care.
template <int CPL, typename T>
struct OperandLoad;
template <>
struct OperandLoad<4, float> {
    static constexpr int NREG = 4, QBYTES = 16;
    static __device__ __forceinline__ void run(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff,
                                               unsigned (&v)[NREG]) {
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = a[i];
    }
    static __device__ __forceinline__ void unpack(const unsigned (&v)[NREG], float (&f)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
    }
};
template <>
struct OperandLoad<8, float> {
    static constexpr int NREG = 8, QBYTES = 16;
    static __device__ __forceinline__ void run(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff,
                                               unsigned (&v)[NREG]) {
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 64, soff, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = a[i];
            v[4 + i] = b[i];
        }
    }
    static __device__ __forceinline__ void unpack(const unsigned (&v)[NREG], float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = __uint_as_float(v[i]);
    }
};
__device__ __forceinline__ void unpack_half2(unsigned w, float& lo, float& hi) {
    const float2 f = __half22float2(*reinterpret_cast<const __half2*>(&w));
    lo = f.x;
    hi = f.y;
}
template <>
struct OperandLoad<4, __half> {
    static constexpr int NREG = 2, QBYTES = 8;
    static __device__ __forceinline__ void run(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff,
                                               unsigned (&v)[NREG]) {
        const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        v[0] = a[0];
        v[1] = a[1];
    }
    static __device__ __forceinline__ void unpack(const unsigned (&v)[NREG], float (&f)[4]) {
        unpack_half2(v[0], f[0], f[1]);
        unpack_half2(v[1], f[2], f[3]);
    }
};
template <>
struct OperandLoad<8, __half> {
    static constexpr int NREG = 4, QBYTES = 16;
    static __device__ __forceinline__ void run(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff,
                                               unsigned (&v)[NREG]) {
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = a[i];
    }
    static __device__ __forceinline__ void unpack(const unsigned (&v)[NREG], float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) unpack_half2(v[i], f[2 * i], f[2 * i + 1]);
    }
};

// KS  = waves that split the channels of one tile; WPB = waves per workgroup.
// `inv_c` > 0 means C is a power of two and mean = sum * (1/C) is exact; otherwise
// the mean is an IEEE division like the reference's reduce_mean.
template <typename T, int CPL, int KS, int WPB>
__global__ __launch_bounds__(64 * WPB, 4) void cost_volume_mfma_kernel(
    const T* __restrict__ prv, const T* __restrict__ nxt, T* __restrict__ out, int H, int W, int C,
    int tiles_x, int tiles_y, int n_tiles, int out_pix_stride, float slope, float inv_c, int pad84) {
    QPWC_FLOW_CHAIN_PRIO();
    constexpr int G = WPB / KS;       // tiles per workgroup
    constexpr int CSTEP = 4 * CPL;    // channels consumed per load step
    constexpr int ES = sizeof(T);
    static_assert(WPB % KS == 0, "split-K must divide the workgroup");
    __shared__ __attribute__((aligned(16))) float frames[WPB * kFrameFloats];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar
    const int lane = threadIdx.x & 63;
    const int grp = wave / KS, ks = wave % KS;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x) * G + grp;
    const bool active = tile < n_tiles;  // wave-uniform
    const int n = lane & 15, g = lane >> 4;  // matrix-core layout: pixel n, k-slot g

    int tx = 0, ty = 0, b = 0;
    if (active) {
        tx = tile % tiles_x;
        ty = (tile / tiles_x) % tiles_y;
        b = tile / (tiles_x * tiles_y);
    }
    const int x0 = tx * 4, y0 = ty * 4;

    f32x4 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (active) {
        // One descriptor per operand covering exactly image b: rows above / below the
        // image fall outside it (negative offsets wrap to >= 2^31) and read as zero;
        // columns left / right of the image are sent out of range explicitly.
        const int img_bytes = H * W * C * ES;  // < 2^31, checked on the host
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T*>(prv) + (int64_t)b * H * W * C, 0, img_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T*>(nxt) + (int64_t)b * H * W * C, 0, img_bytes, 0x00020000);
        using OL = OperandLoad<CPL, T>;
        constexpr int NREG = OL::NREG;
        const int tile_off = ((y0 - 4) * W + (x0 - 4)) * C * ES;  // scalar, may be negative
        // load layout: pixel (ly, lx) = lane>>2 of the block, quarter lane&3
        const int lp = lane >> 2, ly = lp >> 2, lx = lp & 3;
        const int lane_off = (ly * W + lx) * C * ES + (lane & 3) * OL::QBYTES;
        const int perm = (4 * n + g) * 4;  // ds_bpermute byte address of the source lane
        unsigned off_p = (unsigned)(tile_off + lane_off + (4 * W + 4) * C * ES);
        if (x0 + lx >= W) off_p = kOob;
        unsigned off_n[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int xx = x0 - 4 + 4 * j + lx;
            const bool col_ok = xx >= 0 && xx < W;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const unsigned o = (unsigned)(tile_off + lane_off + (4 * i * W + 4 * j) * C * ES);
                off_n[i][j] = col_ok ? o : kOob;
            }
        }
        const int nsteps = C / CSTEP;
        const int s_lo = ks * nsteps / KS, s_hi = (ks + 1) * nsteps / KS;
        for (int s = s_lo; s < s_hi; ++s) {
            const int soff = s * CSTEP * ES;  // scalar
            unsigned pr[NREG], nr[3][3][NREG];
            OL::run(rp, off_p, soff, pr);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) OL::run(rn, off_n[i][j], soff, nr[i][j]);
            // load layout -> matrix-core layout
            float pv[CPL], nv[3][3][CPL];
#pragma unroll
            for (int q = 0; q < NREG; ++q) pr[q] = (unsigned)__builtin_amdgcn_ds_bpermute(perm, (int)pr[q]);
            OL::unpack(pr, pv);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
#pragma unroll
                    for (int q = 0; q < NREG; ++q)
                        nr[i][j][q] = (unsigned)__builtin_amdgcn_ds_bpermute(perm, (int)nr[i][j][q]);
                    OL::unpack(nr[i][j], nv[i][j]);
                }
#pragma unroll
            for (int t = 0; t < CPL; ++t)
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(nv[i][j][t], pv[t], acc[i][j],
                                                                         0, 0, 0);
        }
    }

    // ---- accumulators -> frame: lane (pixel p = n, neighbour row g) ----------
    float* fr = frames + wave * kFrameFloats;
    {
        float* dst = fr + n * kFramePS + g * 12;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4*>(dst + 48 * i + 4 * j) = acc[i][j];
    }
    if (KS > 1)
        __syncthreads();
    else
        __builtin_amdgcn_wave_barrier();  // same-wave LDS traffic is ordered; keep the compiler in line

    // ---- frame window -> 81 channels, whole tile rows of 4 px * 81 floats ----
    if (!active) return;
    const float cf = (float)C;
    const float* f0 = frames + grp * KS * kFrameFloats;
    T* ob = out + ((int64_t)(b * H + y0) * W + x0) * out_pix_stride;  // scalar
    if (KS == 1) {  // the wave owns the whole frame: batched reads, 16-byte stores where possible
        store_tile<T>(fr, ob, lane, x0, y0, H, W, out_pix_stride, slope, inv_c, cf,
                      load_foff(lane, pad84 ? 84 : 81), pad84 != 0);
        return;
    }
    const int row_stride = W * out_pix_stride;
    const bool use_mul = inv_c > 0.f;  // wave-uniform
    const float scale = use_mul ? inv_c : cf;
    auto readback = [&](auto use_mul_c) {
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int e = lane + 64 * s;  // element of the 4 px * 81 row
        const int epx = e / 81, k = e - 81 * epx;
        const int ky = k / 9, kx = k - 9 * ky;
        const bool e_ok = e < 324 && x0 + epx < W;
        const int foff = epx * kFramePS + epx + ky * 12 + kx;  // window origin (py, px) + (ky, kx)
        const unsigned goff = (unsigned)(epx * out_pix_stride + k);
#pragma unroll
        for (int row = 0; row < 4; ++row) {
            if (KS > 1 && (row * 6 + s) % KS != ks) continue;  // the KS waves share the 24 items
            if (e_ok && y0 + row < H) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < KS; ++w) v += f0[w * kFrameFloats + foff + row * kRowStep];
                v = decltype(use_mul_c)::value ? v * scale : v / scale;
                st(ob + (int64_t)row * row_stride + goff, lrelu(v, slope));
            }
        }
    }
    };
    if (use_mul)
        readback(std::true_type{});
    else
        readback(std::false_type{});
}

// ---------------------------------------------------------------------------
// Workgroup-shared variant for levels with many tiles (fp32, C % 32 == 0):
// 4 waves own an 8x8 pixel region (2x2 tiles).  The 16 nxt blocks of its 16x16
// neighbourhood and the 4 prv blocks are staged ONCE per 32-channel step into
// LDS (5 block loads per tile instead of 10, every load a full 128-B line per 8
// lanes), and each wave reads its 10 blocks back directly in matrix-core layout
// with ds_read_b128 -- no ds_bpermute.  LDS image: block-major, pixel = 128 B,
// 16-byte chunk c of pixel n stored at chunk c ^ (n >> 1): conflict-free for the
// staging writes (8 lanes = one pixel) and for the operand reads (16 lanes = 16
// pixels of one chunk).  The per-wave output frames alias the staging area.
constexpr int kRegBlocks = 20;                       // 16 nxt + 4 prv
constexpr int kRegStageBytes = kRegBlocks * 2048;    // 32 channels fp32 per step
constexpr int kRegLdsBytes = kRegStageBytes > 4 * kFrameFloats * 4 ? kRegStageBytes : 4 * kFrameFloats * 4;

//
// WARP = true is the fused UpFlow front end (qpwcnet/core/non_layers.py:377-380: nxt_w = WarpV2(nxt, flo);
// cost = cv(prv, nxt_w)): the staging step of a nxt piece gathers the four corner chunks of its pixel's
// bilinear sample (tfa dense_image_warp, clamp-to-border: taps_clamp / blend of common.h, the very code of
// warp.hip, every multiply and add rounded separately) and writes the blended chunk into the same swizzled
// LDS slot -- from there on the kernel is the unfused one, so the result equals warp -> cost volume bit
// for bit while nxt_w never exists in memory.  Pixels of the 16 x 16 neighbourhood outside the image are the
// ZeroPadding2D of the WARPED image: zero, not sampled.  A lane computes the taps of ONE of its eight
// pieces (the eight lanes of a pixel would all compute the same ones) and the group shares them through
// the 1 KB of the staging area that only its own wave writes later.
#ifdef QPWC_CV_STAMP
__device__ long long g_cv_stamps[8 * 16];
#define CV_STAMP() do { if (cv_stamp_on && cv_si < 16) cv_sp[cv_si++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CV_STAMP() do { } while (0)
#endif
#ifndef QPWC_WARP_OCC
#define QPWC_WARP_OCC 3   // waves per SIMD of the fused kernel (A/B: make ab ABFLAGS=-DQPWC_WARP_OCC=4)
#endif
template <bool WARP>
__global__ __launch_bounds__(256, WARP ? QPWC_WARP_OCC : 4) void cost_volume_mfma_lds_kernel(
    const float* __restrict__ prv, const float* __restrict__ nxt, const float* __restrict__ flo,
    float* __restrict__ out, int H, int W, int C, int regs_x, int regs_y, int out_pix_stride, float slope,
    float inv_c, int pad84) {
    QPWC_FLOW_CHAIN_PRIO();
    __shared__ __attribute__((aligned(16))) char smem[kRegLdsBytes];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int region = xcd_swizzle(blockIdx.x, gridDim.x);
    int rx, ry, b;
    region_coords<16>(region, regs_x, regs_y, rx, ry, b);
    const int X0 = rx * 8, Y0 = ry * 8;
#ifdef QPWC_CV_STAMP
    // diagnostic build only (make ab ABFLAGS=-DQPWC_CV_STAMP): shader-clock stamps of wave 0 of 8 workgroups
    const bool cv_stamp_on = lane == 0 && wave == 0 && (blockIdx.x % 509) == 3 && blockIdx.x / 509 < 8;
    long long* cv_sp = g_cv_stamps + (blockIdx.x / 509) * 16;
    int cv_si = 0;
#endif
    CV_STAMP();

    const int img_bytes = H * W * C * 4;  // < 2^31, checked on the host
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(prv) + (int64_t)b * H * W * C, 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(nxt) + (int64_t)b * H * W * C, 0, img_bytes, 0x00020000);

    // ---- staging map: piece idx = it*256 + tid -> block 2*it + (tid>>7), pixel, chunk
    const int hi = tid >> 7, within = tid & 127, sn = within >> 3, sc = within & 7;
    const int spy = sn >> 2, spx = sn & 3;
    const int lds_w = hi * 2048 + sn * 128 + ((sc ^ (sn >> 1)) << 4);  // + it*4096
    // byte offsets inside image b, 32-bit modular arithmetic: a row above the image gives a
    // "negative" = huge unsigned offset, which the descriptor turns into zeros like a row below
    unsigned goff[10];
    {
        const unsigned pix = (unsigned)C * 4u;                       // bytes per pixel (scalar)
        const unsigned rowb = (unsigned)W * pix;                     // bytes per image row (scalar)
        const int xn = X0 - 4 + 4 * hi + spx;                        // column for even `it` (bj = hi)
        const unsigned base_n = (unsigned)(Y0 - 4 + spy) * rowb + (unsigned)xn * pix + (unsigned)sc * 16u;
        const bool ok0 = xn >= 0 && xn < W, ok1 = xn + 8 >= 0 && xn + 8 < W;   // bj = hi, hi + 2
#pragma unroll
        for (int it = 0; it < 8; ++it) {  // nxt block (bi, bj) = (it >> 1, 2 * (it & 1) + hi)
            const unsigned o = base_n + (unsigned)(it >> 1) * 4u * rowb + (unsigned)(it & 1) * 8u * pix;
            goff[it] = ((it & 1) ? ok1 : ok0) ? o : kOob;
        }
        const int xp = X0 + 4 * hi + spx;                            // prv tile (ti, tj) = (it - 8, hi)
        const unsigned base_p = (unsigned)(Y0 + spy) * rowb + (unsigned)xp * pix + (unsigned)sc * 16u;
        const bool okp = xp < W;
        goff[8] = okp ? base_p : kOob;
        goff[9] = okp ? base_p + 4u * rowb : kOob;
    }
    // ---- WARP: corner (y0, x0) byte offset + chunk and the two lerp factors of the eight nxt pieces ----
    float wax[WARP ? 8 : 1], way[WARP ? 8 : 1];
    const unsigned pixb = (unsigned)C * 4u, rowb2 = (unsigned)W * pixb;
    if (WARP) {
        // this lane's share: piece it = sc, i.e. nxt block (sc >> 1, 2 (sc & 1) + hi), pixel (spy, spx)
        const int yy = Y0 - 4 + 4 * (sc >> 1) + spy, xx = X0 - 4 + 4 * (2 * (sc & 1) + hi) + spx;
        const bool inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
        float2 f = make_float2(0.f, 0.f);
        if (inside) f = *reinterpret_cast<const float2*>(flo + ((int64_t)(b * H + yy) * W + xx) * 2);
        const Taps t = taps_clamp(yy, xx, f.x, f.y, H, W);
        uint4 rec;
        rec.x = inside ? (unsigned)(t.y0 * W + t.x0) * pixb : kOob;   // outside: every corner reads zero
        rec.y = __float_as_uint(t.ax);
        rec.z = __float_as_uint(t.ay);
        rec.w = 0u;
        char* xch = smem + wave * 1024 + (lane >> 3) * 128;   // 8 records of the lane group
        *reinterpret_cast<uint4*>(xch + sc * 16) = rec;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const uint4 r = *reinterpret_cast<const uint4*>(xch + it * 16);
            goff[it] = r.x + (unsigned)sc * 16u;
            wax[it] = __uint_as_float(r.y);
            way[it] = __uint_as_float(r.z);
        }
        __builtin_amdgcn_wave_barrier();   // the records are in registers before this wave's staging writes
    }

    // ---- operand map (matrix-core layout): lane = pixel n, k-slot g ------------
    const int n = lane & 15, g = lane >> 4;
    const int ti = wave >> 1, tj = wave & 1;
    const int lds_r = n * 128;  // + block*2048 + ((4u+g) ^ (n>>1))*16
    const int sw = n >> 1;

    uint4 tab = load_foff(lane, pad84 ? 84 : 81);  // epilogue offsets: in flight behind the first staging loads
    f32x4 acc[3][3];
    const int nsteps = C / 32;
    // one 32-channel step; FIRST: the accumulators start from the instruction's zero operand
    // (no 36 register clears) and the offset table is pinned once the staging loads have landed
    auto step = [&](int s, auto first) {
        constexpr bool FIRST = decltype(first)::value;
        const int soff = s * 128;
        if (WARP) {
            // PR pieces per round: 4 PR corner loads in flight, blended and written to LDS at once.  Holding all
            // ten staged pieces in registers as the unfused form does is out of reach (32 corner chunks); the
            // fused kernel runs 3 waves per SIMD (<= 168 registers) so that a round can be 4 pieces.
            constexpr int PR = (FIRST && QPWC_WARP_OCC < 4) ? 4 : 2;   // later steps carry the 36 accumulators as well
            u32x4 sp[2];
#pragma unroll
            for (int it = 8; it < 10; ++it) sp[it - 8] = __builtin_amdgcn_raw_buffer_load_b128(rp, goff[it], soff, 0);
            u32x4 c[PR][4];
            auto issue = [&](int it) {
#pragma unroll
                for (int q = 0; q < PR; ++q) {
                    // the corner deltas ride in the scalar offset (clamp mode: all four corners of a pixel inside
                    // the image are inside the image; a pixel outside has voffset >= 2^31 = out of range)
                    const unsigned o = goff[it + q];
                    c[q][0] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff, 0);
                    c[q][1] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff + (int)pixb, 0);
                    c[q][2] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff + (int)rowb2, 0);
                    c[q][3] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff + (int)(rowb2 + pixb), 0);
                }
            };
            issue(0);
            if (!FIRST) __syncthreads();  // previous step's operand reads are done
#pragma unroll
            for (int it = 0; it < 8; it += PR) {
                u32x4 v[PR];
#pragma unroll
                for (int q = 0; q < PR; ++q) {
                    Taps t;
                    t.ax = wax[it + q];
                    t.ay = way[it + q];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[q][e] = __float_as_uint(blend<QPWC_WARP_CLAMP>(
                            t, __uint_as_float(c[q][0][e]), __uint_as_float(c[q][1][e]),
                            __uint_as_float(c[q][2][e]), __uint_as_float(c[q][3][e])));
                }
                if (it + PR < 8) issue(it + PR);
#pragma unroll
                for (int q = 0; q < PR; ++q) *reinterpret_cast<u32x4*>(smem + lds_w + (it + q) * 4096) = v[q];
            }
#pragma unroll
            for (int it = 8; it < 10; ++it) *reinterpret_cast<u32x4*>(smem + lds_w + it * 4096) = sp[it - 8];
        } else {
            u32x4 st[10];
#pragma unroll
            for (int it = 0; it < 8; ++it) st[it] = __builtin_amdgcn_raw_buffer_load_b128(rn, goff[it], soff, 0);
#pragma unroll
            for (int it = 8; it < 10; ++it) st[it] = __builtin_amdgcn_raw_buffer_load_b128(rp, goff[it], soff, 0);
            if (FIRST) CV_STAMP();   // loads issued
            if (!FIRST) __syncthreads();  // previous step's operand reads are done
#pragma unroll
            for (int it = 0; it < 10; ++it) *reinterpret_cast<u32x4*>(smem + lds_w + it * 4096) = st[it];
            if (FIRST) CV_STAMP();   // loads landed, written to LDS
        }
        if (FIRST) asm volatile("" : "+v"(tab.x), "+v"(tab.y), "+v"(tab.z), "+v"(tab.w));
        __syncthreads();
        if (FIRST) CV_STAMP();       // barrier: image complete
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int coff = (((4 * u + g) ^ sw) << 4) + lds_r;
            const f32x4 pv = *reinterpret_cast<const f32x4*>(smem + (16 + wave) * 2048 + coff);
            f32x4 nv[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    nv[i][j] = *reinterpret_cast<const f32x4*>(smem + ((ti + i) * 4 + tj + j) * 2048 + coff);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const f32x4 c0 = (FIRST && u == 0 && t == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(nv[i][j][t], pv[t], c0, 0, 0, 0);
                    }
        }
    };
    step(0, std::true_type{});
    for (int s = 1; s < nsteps; ++s) step(s, std::false_type{});
    CV_STAMP();       // matrix work issued
    __syncthreads();  // staging area becomes the four output frames
    CV_STAMP();

    float* fr = reinterpret_cast<float*>(smem) + wave * kFrameFloats;
    {
        float* dst = fr + n * kFramePS + g * 12;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4*>(dst + 48 * i + 4 * j) = acc[i][j];
    }
    __builtin_amdgcn_wave_barrier();
    CV_STAMP();       // frame written (includes the wait for the last matrix results)

    const int x0 = X0 + 4 * tj, y0 = Y0 + 4 * ti;
    if (x0 >= W || y0 >= H) return;
    float* ob = out + ((int64_t)(b * H + y0) * W + x0) * out_pix_stride;
    store_tile<float>(fr, ob, lane, x0, y0, H, W, out_pix_stride, slope, inv_c, (float)C, tab, pad84 != 0);
    CV_STAMP();       // read back, activated, stores issued
#ifdef QPWC_CV_STAMP
    asm volatile("s_waitcnt vmcnt(0)");
    CV_STAMP();       // stores acknowledged
#endif
}

// ---------------------------------------------------------------------------
// 16 x 16 pixel regions (round 3): the same scheme with SIXTEEN waves per workgroup (4 x 4 tiles), one workgroup
// per CU.  Why: a region's 16 + 8 wide neighbourhood is staged once for 16 tiles, so every nxt pixel is staged by
// 2.25 workgroups instead of 4 (36 + 16 = 52 block loads per 16 tiles = 3.25 per tile instead of 5) -- for the
// fused front end (WARP), whose staging step is a 4-corner gather per piece and is bound by the L1's 64 B/clk
// (round 2's counters: 3.3 x the vector-memory instructions of the unfused kernel, texture addresser the busiest
// unit), that is 44 % fewer gather instructions and 44 % less L2 -> L1 traffic; at BASELINE config 4's size, where
// a level's nxt no longer sits in the L2s, it is HBM traffic as well.
//   pieces : block B = it*8 + (wave >> 1) of the 52-block image, pixel (tid & 127) >> 3, chunk tid & 7;
//            blocks 0..35 = nxt (bi, bj) = (B / 6, B % 6) of the 24 x 24 neighbourhood, 36..51 = prv tile B - 36.
//            Waves 0..7 stage 5 nxt + 2 prv pieces per lane and step, waves 8..15 4 + 2 (wave-uniform).
//   LDS    : 52 x 2 KB staging image (swizzled as above), aliased by the sixteen 9.25 KB output frames (148 KB).
//   the rest (operand reads, 72 matrix instructions per wave and 32-channel step, frame, store_tile) is the
//   8 x 8 kernel's code: the result is bit-identical to it.
constexpr int kR16NxtBlocks = 36, kR16Blocks = 52;
constexpr int kR16StageBytes = kR16Blocks * 2048;
constexpr int kR16FrameBytes = 16 * kFrameFloats * 4;
constexpr int kR16LdsBytes = kR16StageBytes > kR16FrameBytes ? kR16StageBytes : kR16FrameBytes;

template <bool WARP>
__global__ __launch_bounds__(1024) void cost_volume_mfma_lds16_kernel(
    const float* __restrict__ prv, const float* __restrict__ nxt, const float* __restrict__ flo,
    float* __restrict__ out, int H, int W, int C, int regs_x, int regs_y, int out_pix_stride, float slope,
    float inv_c, int pad84) {
    QPWC_FLOW_CHAIN_PRIO();
    __shared__ __attribute__((aligned(16))) char smem[kR16LdsBytes];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int region = xcd_swizzle(blockIdx.x, gridDim.x);
    int rx, ry, b;
    region_coords<8>(region, regs_x, regs_y, rx, ry, b);
    const int X0 = rx * 16, Y0 = ry * 16;

    const int img_bytes = H * W * C * 4;  // (H + 24) * (W + 24) * C * 4 < 2^31, checked on the host
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(prv) + (int64_t)b * H * W * C, 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(nxt) + (int64_t)b * H * W * C, 0, img_bytes, 0x00020000);

    // ---- staging map ------------------------------------------------------------------------------------
    const int h = wave >> 1;                       // scalar: block column of a round
    const bool low = wave < 8;                     // scalar: this wave has a fifth nxt piece
    const int sn = (tid & 127) >> 3, sc = tid & 7;
    const int spy = sn >> 2, spx = sn & 3;
    const int lds_w = h * 2048 + sn * 128 + ((sc ^ (sn >> 1)) << 4);   // + it * 16384
    const unsigned pixb = (unsigned)C * 4u, rowb = (unsigned)W * pixb;
    unsigned goffn[5], goffp[2];
    const int itp = low ? 5 : 4;                   // rounds of this wave's two prv pieces: itp, itp + 1
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        const int B = it * 8 + h, bi = B / 6, bj = B - 6 * bi;       // scalar
        const int x = X0 - 4 + 4 * bj + spx;
        // rows above / below the image leave the descriptor's range by themselves (32-bit modular offsets)
        const unsigned o = (unsigned)(Y0 - 4 + 4 * bi + spy) * rowb + (unsigned)x * pixb + (unsigned)sc * 16u;
        goffn[it] = (x >= 0 && x < W) ? o : kOob;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int P = (itp + k) * 8 + h - kR16NxtBlocks;              // prv tile, scalar, 0..15
        const int x = X0 + 4 * (P & 3) + spx;
        const unsigned o = (unsigned)(Y0 + 4 * (P >> 2) + spy) * rowb + (unsigned)x * pixb + (unsigned)sc * 16u;
        goffp[k] = x < W ? o : kOob;
    }
    // ---- WARP: lane sc < 5 of a pixel's eight computes the taps of piece sc.  The records (corner offset, two lerp
    // factors) live in the 45 KB of the frame area that the staging image does not cover, 1 KB per wave, for the whole
    // step loop: a step reads a piece's record when it issues the piece's gathers, so nothing per piece stays in
    // registers across the matrix phases (15 registers that spilled at 128 per wave).
    char* const rec = smem + kR16StageBytes + wave * 1024 + (lane >> 3) * 128;   // 8 records of the lane group
    if (WARP) {
        const int B = sc * 8 + h, bi = B / 6, bj = B - 6 * bi;
        const int yy = Y0 - 4 + 4 * bi + spy, xx = X0 - 4 + 4 * bj + spx;
        const bool inside = B < kR16NxtBlocks && sc < 5 && yy >= 0 && yy < H && xx >= 0 && xx < W;
        float2 f = make_float2(0.f, 0.f);
        if (inside) f = *reinterpret_cast<const float2*>(flo + ((int64_t)(b * H + yy) * W + xx) * 2);
        const Taps t = taps_clamp(yy, xx, f.x, f.y, H, W);
        uint4 r;
        r.x = inside ? (unsigned)(t.y0 * W + t.x0) * pixb : kOob;   // outside: every corner reads zero
        r.y = __float_as_uint(t.ax);
        r.z = __float_as_uint(t.ay);
        r.w = 0u;
        *reinterpret_cast<uint4*>(rec + sc * 16) = r;
        __builtin_amdgcn_wave_barrier();   // same-wave LDS traffic is ordered
    }

    // ---- operand map (matrix-core layout): lane = pixel n, k-slot g; wave = tile (ti, tj) ----------------
    const int n = lane & 15, g = lane >> 4;
    const int ti = wave >> 2, tj = wave & 3;
    const int lds_r = n * 128;
    const int sw = n >> 1;

    uint4 tab = load_foff(lane, pad84 ? 84 : 81);
    f32x4 acc[3][3];
    const int nsteps = C / 32;
    auto step = [&](int s, auto first) {
        constexpr bool FIRST = decltype(first)::value;
        const int soff = s * 128;
        if (WARP) {
            // three rounds of at most eight loads: pieces (0, 1), (2, 3), then (4 | the two prv pieces) -- with the 36
            // accumulators live in the later steps that is what 128 registers hold without spilling
            u32x4 c[2][4];
            float ax[2], ay[2];
            auto issue = [&](int it, int q) {
                const uint4 r = *reinterpret_cast<const uint4*>(rec + it * 16);
                const unsigned o = r.x + (unsigned)sc * 16u;
                ax[q] = __uint_as_float(r.y);
                ay[q] = __uint_as_float(r.z);
                c[q][0] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff, 0);
                c[q][1] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff + (int)pixb, 0);
                c[q][2] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff + (int)rowb, 0);
                c[q][3] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff + (int)(rowb + pixb), 0);
            };
            auto mix = [&](int q) {
                Taps t;
                t.ax = ax[q];
                t.ay = ay[q];
                u32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = __float_as_uint(blend<QPWC_WARP_CLAMP>(
                        t, __uint_as_float(c[q][0][e]), __uint_as_float(c[q][1][e]), __uint_as_float(c[q][2][e]),
                        __uint_as_float(c[q][3][e])));
                return v;
            };
            if (FIRST) {
                // no accumulator is live yet: every gather of the step is issued before the first one is used (one
                // memory round trip instead of three -- with one workgroup per CU nobody else hides them)
                u32x4 c4[3][4], p0, p1;
                float ax4[3], ay4[3];
                auto issue4 = [&](int it, int q) {
                    const uint4 r = *reinterpret_cast<const uint4*>(rec + it * 16);
                    const unsigned o = r.x + (unsigned)sc * 16u;
                    ax4[q] = __uint_as_float(r.y);
                    ay4[q] = __uint_as_float(r.z);
                    c4[q][0] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff, 0);
                    c4[q][1] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff + (int)pixb, 0);
                    c4[q][2] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff + (int)rowb, 0);
                    c4[q][3] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff + (int)(rowb + pixb), 0);
                };
                auto mix4 = [&](int q) {
                    Taps t;
                    t.ax = ax4[q];
                    t.ay = ay4[q];
                    u32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = __float_as_uint(blend<QPWC_WARP_CLAMP>(
                            t, __uint_as_float(c4[q][0][e]), __uint_as_float(c4[q][1][e]),
                            __uint_as_float(c4[q][2][e]), __uint_as_float(c4[q][3][e])));
                    return v;
                };
                issue(0, 0);
                issue(1, 1);
                issue4(2, 0);
                issue4(3, 1);
                if (low) issue4(4, 2);
                p0 = __builtin_amdgcn_raw_buffer_load_b128(rp, goffp[0], soff, 0);
                p1 = __builtin_amdgcn_raw_buffer_load_b128(rp, goffp[1], soff, 0);
                *reinterpret_cast<u32x4*>(smem + lds_w + 0 * 16384) = mix(0);
                *reinterpret_cast<u32x4*>(smem + lds_w + 1 * 16384) = mix(1);
                *reinterpret_cast<u32x4*>(smem + lds_w + 2 * 16384) = mix4(0);
                *reinterpret_cast<u32x4*>(smem + lds_w + 3 * 16384) = mix4(1);
                if (low) *reinterpret_cast<u32x4*>(smem + lds_w + 4 * 16384) = mix4(2);
                *reinterpret_cast<u32x4*>(smem + lds_w + itp * 16384) = p0;
                *reinterpret_cast<u32x4*>(smem + lds_w + (itp + 1) * 16384) = p1;
            } else {
                issue(0, 0);
                issue(1, 1);
                __syncthreads();  // previous step's operand reads are done
                {
                    const u32x4 v0 = mix(0), v1 = mix(1);
                    issue(2, 0);
                    issue(3, 1);
                    *reinterpret_cast<u32x4*>(smem + lds_w + 0 * 16384) = v0;
                    *reinterpret_cast<u32x4*>(smem + lds_w + 1 * 16384) = v1;
                }
                {
                    const u32x4 v0 = mix(0), v1 = mix(1);
                    if (low) issue(4, 0);
                    c[1][0] = __builtin_amdgcn_raw_buffer_load_b128(rp, goffp[0], soff, 0);
                    c[1][1] = __builtin_amdgcn_raw_buffer_load_b128(rp, goffp[1], soff, 0);
                    *reinterpret_cast<u32x4*>(smem + lds_w + 2 * 16384) = v0;
                    *reinterpret_cast<u32x4*>(smem + lds_w + 3 * 16384) = v1;
                }
                if (low) *reinterpret_cast<u32x4*>(smem + lds_w + 4 * 16384) = mix(0);
                *reinterpret_cast<u32x4*>(smem + lds_w + itp * 16384) = c[1][0];
                *reinterpret_cast<u32x4*>(smem + lds_w + (itp + 1) * 16384) = c[1][1];
            }
        } else {
            u32x4 st[5], sp[2];
#pragma unroll
            for (int it = 0; it < 4; ++it) st[it] = __builtin_amdgcn_raw_buffer_load_b128(rn, goffn[it], soff, 0);
            if (low) st[4] = __builtin_amdgcn_raw_buffer_load_b128(rn, goffn[4], soff, 0);
#pragma unroll
            for (int k = 0; k < 2; ++k) sp[k] = __builtin_amdgcn_raw_buffer_load_b128(rp, goffp[k], soff, 0);
            if (!FIRST) __syncthreads();
#pragma unroll
            for (int it = 0; it < 4; ++it) *reinterpret_cast<u32x4*>(smem + lds_w + it * 16384) = st[it];
            if (low) *reinterpret_cast<u32x4*>(smem + lds_w + 4 * 16384) = st[4];
#pragma unroll
            for (int k = 0; k < 2; ++k) *reinterpret_cast<u32x4*>(smem + lds_w + (itp + k) * 16384) = sp[k];
        }
        if (FIRST) asm volatile("" : "+v"(tab.x), "+v"(tab.y), "+v"(tab.z), "+v"(tab.w));
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int coff = (((4 * u + g) ^ sw) << 4) + lds_r;
            const f32x4 pv = *reinterpret_cast<const f32x4*>(smem + (kR16NxtBlocks + wave) * 2048 + coff);
            f32x4 nv[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    nv[i][j] = *reinterpret_cast<const f32x4*>(smem + ((ti + i) * 6 + tj + j) * 2048 + coff);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const f32x4 c0 = (FIRST && u == 0 && t == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(nv[i][j][t], pv[t], c0, 0, 0, 0);
                    }
        }
    };
    step(0, std::true_type{});
    for (int s = 1; s < nsteps; ++s) step(s, std::false_type{});
    __syncthreads();  // staging area becomes the sixteen output frames

    float* fr = reinterpret_cast<float*>(smem) + wave * kFrameFloats;
    {
        float* dst = fr + n * kFramePS + g * 12;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4*>(dst + 48 * i + 4 * j) = acc[i][j];
    }
    __builtin_amdgcn_wave_barrier();

    const int x0 = X0 + 4 * tj, y0 = Y0 + 4 * ti;
    if (x0 >= W || y0 >= H) return;
    float* ob = out + ((int64_t)(b * H + y0) * W + x0) * out_pix_stride;
    store_tile<float>(fr, ob, lane, x0, y0, H, W, out_pix_stride, slope, inv_c, (float)C, tab, pad84 != 0);
}

// ---------------------------------------------------------------------------
// 8 x 16 pixel regions (round 3), fused front end only: EIGHT waves (2 x 4 tiles) per workgroup, two workgroups per
// CU.  The 16 x 16 form above needs the whole LDS of a CU, so nothing overlaps a region's gather, matrix and store
// phases -- fine from two 32-channel steps on, where the gather dominates, but a loss at the finest level (C = 32, one
// step: 52.7 vs 48.7 us at B=8).  This form keeps two independent workgroups on a CU (64 KB staging / 74 KB frames
// each) and still stages every nxt pixel 3 x instead of 4 x (24 + 8 = 32 block loads per 8 tiles = 4 per tile).
//   pieces : block B = it*4 + (wave >> 1), it = 0..7: blocks 0..23 = nxt (bi, bj) = (B / 6, B % 6) of the 16-row x
//            24-column neighbourhood, 24..31 = prv tile B - 24; every lane stages 6 nxt + 2 prv pieces per step.
#ifndef QPWC_R8_PRV_FIRST
#define QPWC_R8_PRV_FIRST 0   // A/B (round 4): the two prv pieces requested before the first round of gathers -- 50.5-51.4 vs 49.1-49.5 us in one call: off
#endif
#ifndef QPWC_R8_PR
#define QPWC_R8_PR 3   // A/B: 6 = every gather of the first step in one round: 48.0-48.8 vs 47.3-47.8 us, not better
#endif
constexpr int kR8NxtBlocks = 24, kR8Blocks = 32;
constexpr int kR8StageBytes = kR8Blocks * 2048;              // 65536
constexpr int kR8FrameBytes = 8 * kFrameFloats * 4;          // 75776: the 10 KB past the staging image hold the records
static_assert(kR8FrameBytes - kR8StageBytes >= 8 * 1024, "records need 1 KB per wave past the staging image");

__global__ __launch_bounds__(512, 2) void cost_volume_mfma_lds8x16_warp_kernel(
    const float* __restrict__ prv, const float* __restrict__ nxt, const float* __restrict__ flo,
    float* __restrict__ out, int H, int W, int C, int regs_x, int regs_y, int out_pix_stride, float slope,
    float inv_c, int pad84) {
    QPWC_FLOW_CHAIN_PRIO();
    __shared__ __attribute__((aligned(16))) char smem[kR8FrameBytes];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int region = xcd_swizzle(blockIdx.x, gridDim.x);
    int rx, ry, b;
    region_coords<8>(region, regs_x, regs_y, rx, ry, b);
    const int X0 = rx * 16, Y0 = ry * 8;

    const int img_bytes = H * W * C * 4;  // (H + 16) * (W + 24) * C * 4 < 2^31, checked on the host
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(prv) + (int64_t)b * H * W * C, 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(nxt) + (int64_t)b * H * W * C, 0, img_bytes, 0x00020000);

    const int h = wave >> 1;                       // scalar: block of a round
    const int sn = (tid & 127) >> 3, sc = tid & 7;
    const int spy = sn >> 2, spx = sn & 3;
    const int lds_w = h * 2048 + sn * 128 + ((sc ^ (sn >> 1)) << 4);   // + it * 8192
    const unsigned pixb = (unsigned)C * 4u, rowb = (unsigned)W * pixb;
    unsigned goffp[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int P = (6 + k) * 4 + h - kR8NxtBlocks;                 // prv tile, scalar, 0..7
        const int x = X0 + 4 * (P & 3) + spx;
        const unsigned o = (unsigned)(Y0 + 4 * (P >> 2) + spy) * rowb + (unsigned)x * pixb + (unsigned)sc * 16u;
        goffp[k] = x < W ? o : kOob;
    }
    // lane sc < 6 of a pixel's eight computes the taps of piece sc; the records stay in LDS for the whole step loop
    char* const rec = smem + kR8StageBytes + wave * 1024 + (lane >> 3) * 128;
    {
        const int B = sc * 4 + h, bi = B / 6, bj = B - 6 * bi;
        const int yy = Y0 - 4 + 4 * bi + spy, xx = X0 - 4 + 4 * bj + spx;
        const bool inside = sc < 6 && yy >= 0 && yy < H && xx >= 0 && xx < W;
        float2 f = make_float2(0.f, 0.f);
        if (inside) f = *reinterpret_cast<const float2*>(flo + ((int64_t)(b * H + yy) * W + xx) * 2);
        const Taps t = taps_clamp(yy, xx, f.x, f.y, H, W);
        uint4 r;
        r.x = inside ? (unsigned)(t.y0 * W + t.x0) * pixb : kOob;   // outside: every corner reads zero
        r.y = __float_as_uint(t.ax);
        r.z = __float_as_uint(t.ay);
        r.w = 0u;
        *reinterpret_cast<uint4*>(rec + sc * 16) = r;
        __builtin_amdgcn_wave_barrier();
    }

    const int n = lane & 15, g = lane >> 4;
    const int ti = wave >> 2, tj = wave & 3;
    const int lds_r = n * 128;
    const int sw = n >> 1;

    uint4 tab = load_foff(lane, pad84 ? 84 : 81);
    f32x4 acc[3][3];
    const int nsteps = C / 32;
    auto step = [&](int s, auto first) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first)::value;
        constexpr int PR = FIRST ? QPWC_R8_PR : 2;   // pieces per round: no accumulator is live in the first step
        const int soff = s * 128;
        u32x4 c[PR][4];
        float ax[PR], ay[PR];
        auto issue = [&](int it, int q) __attribute__((always_inline)) {
            const uint4 r = *reinterpret_cast<const uint4*>(rec + it * 16);
            const unsigned o = r.x + (unsigned)sc * 16u;
            ax[q] = __uint_as_float(r.y);
            ay[q] = __uint_as_float(r.z);
            c[q][0] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff, 0);
            c[q][1] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff + (int)pixb, 0);
            c[q][2] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff + (int)rowb, 0);
            c[q][3] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff + (int)(rowb + pixb), 0);
        };
        auto mix = [&](int q) __attribute__((always_inline)) {
            return blend_clamp_chunk(ax[q], ay[q], c[q][0], c[q][1], c[q][2], c[q][3]);
        };
        // Round 4 (QPWC_R8_PRV_FIRST): in the first step -- no accumulator is live -- the two prv pieces, which depend on
        // nothing, are requested BEFORE the first round of gathers instead of after the last one, where their round trip
        // stood between the last blend and the barrier in front of the matrix phase.
        constexpr bool PRV_FIRST = FIRST && QPWC_R8_PRV_FIRST;
        u32x4 pp0, pp1;
        if (PRV_FIRST) {
            pp0 = __builtin_amdgcn_raw_buffer_load_b128(rp, goffp[0], soff, 0);
            pp1 = __builtin_amdgcn_raw_buffer_load_b128(rp, goffp[1], soff, 0);
        }
#pragma unroll
        for (int q = 0; q < PR; ++q) issue(q, q);
        if (!FIRST) __syncthreads();  // previous step's operand reads are done
        if (PRV_FIRST) {
            *reinterpret_cast<u32x4*>(smem + lds_w + 6 * 8192) = pp0;
            *reinterpret_cast<u32x4*>(smem + lds_w + 7 * 8192) = pp1;
        }
#pragma unroll
        for (int it = 0; it < 6; it += PR) {
            u32x4 v[PR];
#pragma unroll
            for (int q = 0; q < PR; ++q) v[q] = mix(q);
            if (it + PR < 6) {
#pragma unroll
                for (int q = 0; q < PR; ++q) issue(it + PR + q, q);
            } else if (!PRV_FIRST) {   // last round: the two prv pieces ride in the freed registers
                c[0][0] = __builtin_amdgcn_raw_buffer_load_b128(rp, goffp[0], soff, 0);
                c[0][1] = __builtin_amdgcn_raw_buffer_load_b128(rp, goffp[1], soff, 0);
            }
#pragma unroll
            for (int q = 0; q < PR; ++q) *reinterpret_cast<u32x4*>(smem + lds_w + (it + q) * 8192) = v[q];
        }
        if (!PRV_FIRST) {
            *reinterpret_cast<u32x4*>(smem + lds_w + 6 * 8192) = c[0][0];
            *reinterpret_cast<u32x4*>(smem + lds_w + 7 * 8192) = c[0][1];
        }
        if (FIRST) asm volatile("" : "+v"(tab.x), "+v"(tab.y), "+v"(tab.z), "+v"(tab.w));
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int coff = (((4 * u + g) ^ sw) << 4) + lds_r;
            const f32x4 pv = *reinterpret_cast<const f32x4*>(smem + (kR8NxtBlocks + wave) * 2048 + coff);
            f32x4 nv[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    nv[i][j] = *reinterpret_cast<const f32x4*>(smem + ((ti + i) * 6 + tj + j) * 2048 + coff);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const f32x4 c0 = (FIRST && u == 0 && t == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(nv[i][j][t], pv[t], c0, 0, 0, 0);
                    }
        }
    };
    step(0, std::true_type{});
    for (int s = 1; s < nsteps; ++s) step(s, std::false_type{});
    __syncthreads();  // staging area (and the records) become the eight output frames

    float* fr = reinterpret_cast<float*>(smem) + wave * kFrameFloats;
    {
        float* dst = fr + n * kFramePS + g * 12;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4*>(dst + 48 * i + 4 * j) = acc[i][j];
    }
    __builtin_amdgcn_wave_barrier();

    const int x0 = X0 + 4 * tj, y0 = Y0 + 4 * ti;
    if (x0 >= W || y0 >= H) return;
    float* ob = out + ((int64_t)(b * H + y0) * W + x0) * out_pix_stride;
    store_tile<float>(fr, ob, lane, x0, y0, H, W, out_pix_stride, slope, inv_c, (float)C, tab, pad84 != 0);
}

// ---------------------------------------------------------------------------
// fp16 storage (BASELINE configs[4]): same workgroup-shared scheme on the fp16 matrix
// cores, v_mfma_f32_16x16x32_f16 (fp32 accumulate): one instruction per block per
// 32-channel step instead of eight, so the kernel is purely bandwidth-bound.
// LDS image: block = 16 px x 64 B; 16-byte chunk c (8 channels) of pixel n at chunk
// c ^ ((n >> 2) & 2): conflict-free staging writes and ds_read_b128 operand reads.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int kRegStageBytesH = kRegBlocks * 1024;
constexpr int kRegLdsBytesH = kRegStageBytesH > 4 * kFrameFloats * 4 ? kRegStageBytesH : 4 * kFrameFloats * 4;

// WARP: the fused UpFlow front end as in the fp32 kernel -- a nxt piece is the bilinear blend (fp32 arithmetic on
// the fp16 corner chunks, ONE rounding to fp16: exactly what the fp16-storage WarpV2 kernel stores) of four
// corner chunks; taps computed once per pixel by one of its four lanes and exchanged through the wave's own 1 KB.
template <bool WARP>
__global__ __launch_bounds__(256, WARP ? 3 : 4) void cost_volume_mfma_lds_f16_kernel(
    const __half* __restrict__ prv, const __half* __restrict__ nxt, const float* __restrict__ flo,
    __half* __restrict__ out, int H, int W, int C, int regs_x, int regs_y, int out_pix_stride, float slope,
    float inv_c, int pad84) {
    __shared__ __attribute__((aligned(16))) char smem[kRegLdsBytesH];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int region = xcd_swizzle(blockIdx.x, gridDim.x);
    int rx, ry, b;
    region_coords<16>(region, regs_x, regs_y, rx, ry, b);
    const int X0 = rx * 8, Y0 = ry * 8;

    const int img_bytes = H * W * C * 2;
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__half*>(prv) + (int64_t)b * H * W * C, 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__half*>(nxt) + (int64_t)b * H * W * C, 0, img_bytes, 0x00020000);

    // staging map: piece idx = it*256 + tid -> block 4*it + wave, pixel lane>>2, chunk lane&3
    const int sn = lane >> 2, sc = lane & 3;
    const int spy = sn >> 2, spx = sn & 3;
    const int lds_w = wave * 1024 + sn * 64 + ((sc ^ ((sn >> 2) & 2)) << 4);  // + it*4096
    unsigned goff[5];
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        const int blk = 4 * it + wave;  // wave-uniform
        int y, x;
        if (blk < 16) {
            y = Y0 - 4 + 4 * (blk >> 2) + spy;
            x = X0 - 4 + 4 * (blk & 3) + spx;
        } else {
            y = Y0 + 4 * ((blk - 16) >> 1) + spy;
            x = X0 + 4 * ((blk - 16) & 1) + spx;
        }
        goff[it] = (x >= 0 && x < W) ? (unsigned)((y * W + x) * C * 2 + sc * 16) : kOob;
    }

    // ---- WARP: corner (y0, x0) byte offset + chunk and the lerp factors of the four nxt pieces ----
    float wax[WARP ? 4 : 1], way[WARP ? 4 : 1];
    const unsigned pixb = (unsigned)C * 2u, rowb2 = (unsigned)W * pixb;
    if (WARP) {
        const int blk = 4 * sc + wave;   // this lane's share: piece it = sc
        const int yy = Y0 - 4 + 4 * (blk >> 2) + spy, xx = X0 - 4 + 4 * (blk & 3) + spx;
        const bool inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
        float2 f = make_float2(0.f, 0.f);
        if (inside) f = *reinterpret_cast<const float2*>(flo + ((int64_t)(b * H + yy) * W + xx) * 2);
        const Taps t = taps_clamp(yy, xx, f.x, f.y, H, W);
        uint4 rec;
        rec.x = inside ? (unsigned)(t.y0 * W + t.x0) * pixb : kOob;
        rec.y = __float_as_uint(t.ax);
        rec.z = __float_as_uint(t.ay);
        rec.w = 0u;
        char* xch = smem + wave * 1024 + (lane >> 2) * 64;   // 4 records of the lane group
        *reinterpret_cast<uint4*>(xch + sc * 16) = rec;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint4 r = *reinterpret_cast<const uint4*>(xch + it * 16);
            goff[it] = r.x + (unsigned)sc * 16u;
            wax[it] = __uint_as_float(r.y);
            way[it] = __uint_as_float(r.z);
        }
        __builtin_amdgcn_wave_barrier();
    }

    const int n = lane & 15, g = lane >> 4;
    const int ti = wave >> 1, tj = wave & 1;
    const int coff = n * 64 + ((g ^ ((n >> 2) & 2)) << 4);

    f32x4 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 tab = load_foff(lane, pad84 ? 84 : 81);  // epilogue offsets, fetched early
    const int nsteps = C / 32;
    for (int s = 0; s < nsteps; ++s) {
        const int soff = s * 64;
        u32x4 st[5];
        st[4] = __builtin_amdgcn_raw_buffer_load_b128(rp, goff[4], soff, 0);
        if (WARP) {
            // two pieces per round: 8 corner chunks (8 halves each) in flight
#pragma unroll
            for (int it = 0; it < 4; it += 2) {
                u32x4 c[2][4];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const unsigned o = goff[it + q];
                    c[q][0] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff, 0);
                    c[q][1] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff + (int)pixb, 0);
                    c[q][2] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff + (int)rowb2, 0);
                    c[q][3] = __builtin_amdgcn_raw_buffer_load_b128(rn, o, soff + (int)(rowb2 + pixb), 0);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    Taps t;
                    t.ax = wax[it + q];
                    t.ay = way[it + q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {   // one 32-bit word = two halves
                        float tl0, tl1, tr0, tr1, bl0, bl1, br0, br1;
                        unpack_half2(c[q][0][e], tl0, tl1);
                        unpack_half2(c[q][1][e], tr0, tr1);
                        unpack_half2(c[q][2][e], bl0, bl1);
                        unpack_half2(c[q][3][e], br0, br1);
                        const __half2 h = __floats2half2_rn(blend<QPWC_WARP_CLAMP>(t, tl0, tr0, bl0, br0),
                                                            blend<QPWC_WARP_CLAMP>(t, tl1, tr1, bl1, br1));
                        st[it + q][e] = *reinterpret_cast<const unsigned*>(&h);
                    }
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < 4; ++it) st[it] = __builtin_amdgcn_raw_buffer_load_b128(rn, goff[it], soff, 0);
        }
        if (s > 0) __syncthreads();
#pragma unroll
        for (int it = 0; it < 5; ++it) *reinterpret_cast<u32x4*>(smem + lds_w + it * 4096) = st[it];
        __syncthreads();
        const f16x8 pv = *reinterpret_cast<const f16x8*>(smem + (16 + wave) * 1024 + coff);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const f16x8 nv = *reinterpret_cast<const f16x8*>(smem + ((ti + i) * 4 + tj + j) * 1024 + coff);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(nv, pv, acc[i][j], 0, 0, 0);
            }
    }
    __syncthreads();

    float* fr = reinterpret_cast<float*>(smem) + wave * kFrameFloats;
    {
        float* dst = fr + n * kFramePS + g * 12;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4*>(dst + 48 * i + 4 * j) = acc[i][j];
    }
    __builtin_amdgcn_wave_barrier();

    const int x0 = X0 + 4 * tj, y0 = Y0 + 4 * ti;
    if (x0 >= W || y0 >= H) return;
    __half* ob = out + ((int64_t)(b * H + y0) * W + x0) * out_pix_stride;
    store_tile<__half>(fr, ob, lane, x0, y0, H, W, out_pix_stride, slope, inv_c, (float)C, tab, pad84 != 0);
}

static int launch_lds_f16(const __half* prv, const __half* nxt, const float* flo, __half* out, int B, int H, int W,
                          int C, int64_t ops, float slope, int pad84, hipStream_t s) {
    const int regs_x = (W + 7) / 8, regs_y = (H + 7) / 8;
    const int64_t nblk = (int64_t)regs_x * regs_y * B;
    if (nblk > INT32_MAX || (int64_t)(H + 8) * (W + 8) * C * 2 >= 0x7fffffff ||
        (int64_t)H * W * ops > INT32_MAX) {
        set_error("image too large for 32-bit tile indexing");
        return QPWC_E_SHAPE;
    }
    const float inv_c = (C & (C - 1)) == 0 ? 1.0f / (float)C : 0.0f;
    if (flo) {
        if (dry_run("cost_volume_mfma_lds_f16_kernel<true>")) return QPWC_OK;
        hipLaunchKernelGGL(cost_volume_mfma_lds_f16_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, s, prv, nxt, flo,
                           out, H, W, C, regs_x, regs_y, (int)ops, slope, inv_c, pad84);
        return check_launch("cost_volume_mfma_lds_f16_kernel<warp>");
    }
    if (dry_run("cost_volume_mfma_lds_f16_kernel")) return QPWC_OK;
    hipLaunchKernelGGL(cost_volume_mfma_lds_f16_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, s, prv, nxt,
                       (const float*)nullptr, out, H, W, C, regs_x, regs_y, (int)ops, slope, inv_c, pad84);
    return check_launch("cost_volume_mfma_lds_f16_kernel");
}


#ifdef QPWC_EXPERIMENTAL
#include "experimental/cost_volume_pipe.inc"
#endif

// The product build reads NO environment variable: every rank of a multi-GPU job runs the same kernels.
// Only `make experimental` (libqpwc_exp.so, reported by qpwc_build_info()) keeps the A/B switch
// QPWC_CV_LDS=0 = every shape on the per-wave split-K kernel.
#ifdef QPWC_EXPERIMENTAL
static int lds_mode() {
    static const int v = [] {
        const char* e = getenv("QPWC_CV_LDS");
        return e ? atoi(e) : 1;
    }();
    return v;
}
#else
static constexpr int lds_mode() { return 1; }
#endif

// Region size policy of the workgroup-shared fp32 kernels (measured, tools/kbench.py --regions):
// 16 x 16 regions need one workgroup of 16 waves per CU to be worth it.
#ifndef QPWC_FUSED_MIN_REGIONS
#define QPWC_FUSED_MIN_REGIONS 256   // 8 x 8 regions per launch from which the fused front end runs on the workgroup-shared kernel (fp32)
#endif
#ifndef QPWC_R16_MIN_WARP
#define QPWC_R16_MIN_WARP 512    // 16 x 16 regions per launch from which the fused front end takes them (two rounds of
                                 // one-per-CU workgroups; see use_regions16)
#endif
#ifndef QPWC_R16_MIN_PLAIN
#define QPWC_R16_MIN_PLAIN (1 << 30)   // the plain cost volume: never (B=8 L3 19.4 vs 17.6 us, config 4 L1-L3 +5 ... +14 %:
                                       // its staging is 10 plain loads per lane, not the limiter -- DESIGN.md 4.5)
#endif
#ifndef QPWC_CV_R16
#define QPWC_CV_R16 1   // 0 = never, 1 = by the rule below, 2 = wherever the kernel applies (A/B builds)
#endif
static bool use_regions16(int B, int H, int W, int C, bool warp) {
    const int64_t regs16 = (int64_t)((W + 15) / 16) * ((H + 15) / 16) * B;
    if (QPWC_CV_R16 == 0 || (int64_t)(H + 24) * (W + 24) * C * 4 >= 0x7fffffff) return false;
    if (QPWC_CV_R16 == 2) return true;
    // One workgroup per CU means nothing overlaps a region's staging, matrix and store phases: with a single
    // 32-channel step (C = 32, the finest level) the three phases are of comparable length and the 8 x 8 kernel's
    // three workgroups per CU win (B=8 L4: 48.7 vs 52.7 us); from two steps on the staging share grows and the
    // 16 x 16 regions do (L3, C = 64: 22.3 vs 26.3-27.5 us; config 4's L1-L3: -8 ... -10 %).
    // A launch of ONE round (256 regions: config 2's L3) is left to the 8 x 16 regions since round 3: alone it is 1 us
    // slower there (23.5 vs 22.4 us), but inside the two-queue forward the finest decoder level runs beside this launch
    // (QpwcNet.dec_chunks) and a 148 KB workgroup cannot share a CU with that kernel's 46 KB ones, a 75 KB one can:
    // 50.7 -> ~45 us in the step's kernel trace, step 1.2027 -> 1.1988 ms (three interleaved pairs of one call).
    return C >= 64 && regs16 >= (warp ? QPWC_R16_MIN_WARP : QPWC_R16_MIN_PLAIN);
}

static int launch_lds16(const float* prv, const float* nxt, const float* flo, float* out, int B, int H, int W, int C,
                        int64_t ops, float slope, int pad84, hipStream_t s) {
    const int regs_x = (W + 15) / 16, regs_y = (H + 15) / 16;
    const int64_t nblk = (int64_t)regs_x * regs_y * B;
    if (nblk > INT32_MAX || (int64_t)H * W * ops > INT32_MAX) {
        set_error("image too large for 32-bit tile indexing");
        return QPWC_E_SHAPE;
    }
    const float inv_c = (C & (C - 1)) == 0 ? 1.0f / (float)C : 0.0f;
    if (flo) {
        if (dry_run("cost_volume_mfma_lds16_kernel<true>")) return QPWC_OK;
        hipLaunchKernelGGL(cost_volume_mfma_lds16_kernel<true>, dim3((unsigned)nblk), dim3(1024), 0, s, prv, nxt,
                           flo, out, H, W, C, regs_x, regs_y, (int)ops, slope, inv_c, pad84);
        return check_launch("cost_volume_mfma_lds16_kernel<warp>");
    }
#if QPWC_R16_MIN_PLAIN < (1 << 30)   // A/B builds only: the plain cost volume measured slower on 16 x 16 regions
    hipLaunchKernelGGL(cost_volume_mfma_lds16_kernel<false>, dim3((unsigned)nblk), dim3(1024), 0, s, prv, nxt,
                       (const float*)nullptr, out, H, W, C, regs_x, regs_y, (int)ops, slope, inv_c, pad84);
    return check_launch("cost_volume_mfma_lds16_kernel");
#else
    set_error("cost volume on 16 x 16 regions: fused front end only in this build");
    return QPWC_E_SHAPE;
#endif
}

#ifndef QPWC_R8X16_ANYC
#define QPWC_R8X16_ANYC 1     // 8 x 16 regions for every channel count the 16 x 16 form does not take (0: C = 32 only)
#endif
#ifndef QPWC_R8X16_MIN
#define QPWC_R8X16_MIN 512    // 8 x 16 regions per launch from which the single-step fused front end takes them (2 per CU)
#endif
static int launch_lds8x16_warp(const float* prv, const float* nxt, const float* flo, float* out, int B, int H, int W,
                               int C, int64_t ops, float slope, int pad84, hipStream_t s) {
    const int regs_x = (W + 15) / 16, regs_y = (H + 7) / 8;
    const int64_t nblk = (int64_t)regs_x * regs_y * B;
    const float inv_c = (C & (C - 1)) == 0 ? 1.0f / (float)C : 0.0f;
    if (dry_run("cost_volume_mfma_lds8x16_warp_kernel")) return QPWC_OK;
    hipLaunchKernelGGL(cost_volume_mfma_lds8x16_warp_kernel, dim3((unsigned)nblk), dim3(512), 0, s, prv, nxt, flo, out,
                       H, W, C, regs_x, regs_y, (int)ops, slope, inv_c, pad84);
    return check_launch("cost_volume_mfma_lds8x16_warp_kernel");
}

static int launch_lds(const float* prv, const float* nxt, const float* flo, float* out, int B, int H, int W, int C,
                      int64_t ops, float slope, int pad84, hipStream_t s) {
    if (use_regions16(B, H, W, C, flo != nullptr))
        return launch_lds16(prv, nxt, flo, out, B, H, W, C, ops, slope, pad84, s);
    if (flo && (C == 32 || QPWC_R8X16_ANYC) && QPWC_R8X16_MIN > 0 && (int64_t)((W + 15) / 16) * ((H + 7) / 8) * B >= QPWC_R8X16_MIN &&
        (int64_t)(H + 16) * (W + 24) * C * 4 < 0x7fffffff && (int64_t)H * W * ops <= INT32_MAX)
        return launch_lds8x16_warp(prv, nxt, flo, out, B, H, W, C, ops, slope, pad84, s);
    const int regs_x = (W + 7) / 8, regs_y = (H + 7) / 8;
    const int64_t nblk = (int64_t)regs_x * regs_y * B;
    if (nblk > INT32_MAX || (int64_t)(H + 8) * (W + 8) * C * 4 >= 0x7fffffff ||
        (int64_t)H * W * ops > INT32_MAX) {
        set_error("image too large for 32-bit tile indexing");
        return QPWC_E_SHAPE;
    }
    const float inv_c = (C & (C - 1)) == 0 ? 1.0f / (float)C : 0.0f;
    if (flo) {
        if (dry_run("cost_volume_mfma_lds_kernel<true>")) return QPWC_OK;
        hipLaunchKernelGGL(cost_volume_mfma_lds_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, s, prv, nxt,
                           flo, out, H, W, C, regs_x, regs_y, (int)ops, slope, inv_c, pad84);
        return check_launch("cost_volume_mfma_lds_kernel<warp>");
    }
#ifdef QPWC_EXPERIMENTAL
    {
        const int rc = launch_experimental(prv, nxt, out, H, W, C, regs_x, regs_y, nblk, ops, slope, inv_c, s);
        if (rc <= 0) return rc;
    }
#endif
    if (dry_run("cost_volume_mfma_lds_kernel")) return QPWC_OK;
    hipLaunchKernelGGL(cost_volume_mfma_lds_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, s, prv, nxt,
                       (const float*)nullptr, out, H, W, C, regs_x, regs_y, (int)ops, slope, inv_c, pad84);
    return check_launch("cost_volume_mfma_lds_kernel");
}

template <typename T, int CPL, int KS>
static int launch_mfma(const T* prv, const T* nxt, T* out, int B, int H, int W, int C, int64_t ops,
                       float slope, int pad84, hipStream_t s) {
    constexpr int WPB = KS <= 4 ? 4 : KS;
    constexpr int G = WPB / KS;
    const int tiles_x = (W + 3) / 4, tiles_y = (H + 3) / 4;
    const int64_t n_tiles = (int64_t)tiles_x * tiles_y * B;
    const int64_t nblk = (n_tiles + G - 1) / G;
    if (nblk > INT32_MAX || (int64_t)(H + 8) * (W + 8) * C * (int64_t)sizeof(T) >= 0x7fffffff ||
        (int64_t)H * W * ops > INT32_MAX) {
        set_error("image too large for 32-bit tile indexing");
        return QPWC_E_SHAPE;
    }
    const float inv_c = (C & (C - 1)) == 0 ? 1.0f / (float)C : 0.0f;  // exact for powers of two
    if (dry_run("cost_volume_mfma_kernel")) return QPWC_OK;
    hipLaunchKernelGGL((cost_volume_mfma_kernel<T, CPL, KS, WPB>), dim3((unsigned)nblk),
                       dim3(64 * WPB), 0, s, prv, nxt, out, H, W, C, tiles_x, tiles_y, (int)n_tiles,
                       (int)ops, slope, inv_c, pad84);
    return check_launch("cost_volume_mfma_kernel");
}

template <typename T, int CPL>
static int dispatch_mfma(const T* prv, const T* nxt, T* out, int B, int H, int W, int C, int64_t ops,
                         float slope, int pad84, hipStream_t s) {
    const int64_t n_tiles = (int64_t)((W + 3) / 4) * ((H + 3) / 4) * B;
    const int nsteps = C / (4 * CPL);
    // enough waves for ~4 per SIMD on 1024 SIMDs; split channels when tiles are few
    int ks = 1;
    while (ks < 8 && n_tiles * ks < 4096 && nsteps % (ks * 2) == 0) ks *= 2;
    switch (ks) {
        case 1: return launch_mfma<T, CPL, 1>(prv, nxt, out, B, H, W, C, ops, slope, pad84, s);
        case 2: return launch_mfma<T, CPL, 2>(prv, nxt, out, B, H, W, C, ops, slope, pad84, s);
        case 4: return launch_mfma<T, CPL, 4>(prv, nxt, out, B, H, W, C, ops, slope, pad84, s);
        default: return launch_mfma<T, CPL, 8>(prv, nxt, out, B, H, W, C, ops, slope, pad84, s);
    }
}

// NHWC, search range 4, C % 16 == 0, 16-byte aligned operands.  Returns 1 if the
// shape is not eligible (caller falls back to the LDS-tiled vector kernel).
// pad84: `out` pixels are 84 floats apart and channels 81..83 are to be zero.  The dense epilogue
// writes them itself; *pads_written says whether every tile of this launch takes that path (if not, the
// caller zeroes the pads with its own small launch).
// flo != nullptr: the fused WarpV2 + cost volume (fp32, workgroup-shared kernel only; returns 1 for shapes
// that kernel does not take, and the caller uses the LDS-tiled vector kernel's fused form).
int cost_volume_mfma_launch(const void* prv, const void* nxt, const void* flo, void* out, int B, int H, int W,
                            int C, int dtype, int64_t ops, float slope, int pad84, bool* pads_written,
                            hipStream_t s) {
    if (pads_written) *pads_written = false;
    if (C % 16 != 0 || (reinterpret_cast<uintptr_t>(prv) | reinterpret_cast<uintptr_t>(nxt)) % 16)
        return 1;
    const int64_t regions8 = (int64_t)((W + 7) / 8) * ((H + 7) / 8) * B;
    if (flo && (C % 32 != 0 || reinterpret_cast<uintptr_t>(flo) % 8 || H < 2 || W < 2 || regions8 < QPWC_FUSED_MIN_REGIONS))
        return 1;
    if (flo && lds_mode() == 0) return 1;   // (experimental build, QPWC_CV_LDS=0: no fused form on the split-K kernel)
    // 32-bit byte offsets inside one image (buffer descriptors) and 32-bit element offsets
    // inside one output image: larger problems take the 64-bit vector kernel instead
    const int64_t es = dtype == QPWC_F16 ? 2 : 4;
    if ((int64_t)(H + 8) * (W + 8) * C * es >= 0x7fffffff || (int64_t)H * W * ops > INT32_MAX ||
        (int64_t)((W + 3) / 4) * ((H + 3) / 4) * B > INT32_MAX)
        return 1;
    if (dtype == QPWC_F32) {
        // >= one region per CU: share the staged neighbourhood across a workgroup (L2 of the 256x512
        // pyramid, 256 regions x 4 steps: 10.2 us vs 11.8 us on the per-wave split-K kernel)
        if (C % 32 == 0 && regions8 >= (flo ? QPWC_FUSED_MIN_REGIONS : 256) && lds_mode() != 0)
        {
            // every tile dense: full 4x4 tiles, 16-byte aligned rows
            if (pads_written)
                *pads_written = pad84 && W % 4 == 0 && H % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0;
            return launch_lds((const float*)prv, (const float*)nxt, (const float*)flo, (float*)out, B, H, W, C, ops,
                              slope, pad84, s);
        }
        if (C % 32 == 0)
            return dispatch_mfma<float, 8>((const float*)prv, (const float*)nxt, (float*)out, B, H, W,
                                           C, ops, slope, pad84, s);
        return dispatch_mfma<float, 4>((const float*)prv, (const float*)nxt, (float*)out, B, H, W, C,
                                       ops, slope, pad84, s);
    }
    if (C % 32 == 0 && (int64_t)((W + 7) / 8) * ((H + 7) / 8) * B >= 256 && lds_mode() != 0) {
        if (pads_written)   // dense 8-byte rows of 4 px x 84 halves
            *pads_written = pad84 && W % 4 == 0 && H % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 8 == 0;
        return launch_lds_f16((const __half*)prv, (const __half*)nxt, (const float*)flo, (__half*)out, B, H, W, C, ops,
                              slope, pad84, s);
    }
    if (C % 32 == 0)
        return dispatch_mfma<__half, 8>((const __half*)prv, (const __half*)nxt, (__half*)out, B, H,
                                        W, C, ops, slope, pad84, s);
    return dispatch_mfma<__half, 4>((const __half*)prv, (const __half*)nxt, (__half*)out, B, H, W, C,
                                    ops, slope, pad84, s);
}

}  // namespace qpwc

// Which build this is: the product library has no run-time kernel switches; the experimental one does
// and says so, so that a bench line can never silently come from it.
extern "C" const char* qpwc_build_info(void) {
#ifdef QPWC_EXPERIMENTAL
    return "libqpwc_hip gfx950 EXPERIMENTAL (env switches QPWC_CV_LDS / QPWC_CV_RING / QPWC_CV_PIPE active)";
#else
    return "libqpwc_hip gfx950 product (no environment switches)";
#endif
}

#ifdef QPWC_CV_STAMP
extern "C" int qpwc_debug_cv_stamps(long long* out, int n) {
    if (n > 8 * 16) n = 8 * 16;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(qpwc::g_cv_stamps), n * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
#endif

#if defined(QPWC_EXPERIMENTAL) && defined(QPWC_STAMP)
#include "experimental/debug_exports.inc"
#endif
