// Pointwise half of a split SeparableConv2D (non_layers.py:223-231) for the FEW-PIXEL levels: out[M, F] = y[M, C] . W[F, C]^T + b
// on v_mfma_f32_16x16x4_f32 (round 4).  The wide first OptFlow layer of the two coarsest levels (M = 1 k / 4 k pixels,
// C = 596 / 342, F = 128) stays split into depthwise kernel + GEMM because the fused kernel walks its 19 / 11 steps of 32
// channels one barrier interval after the other; the GEMM was the last library launch of the step (hipBLASLt, 12-14 us for
// 0.15-0.36 GFLOP: start-up bound).  Here a workgroup owns 16 pixels: their C values are staged once (the tile is one
// contiguous, 16-byte aligned run of 16 C floats), wave w multiplies them with rows 16 w .. 16 w + 15 of W streamed from
// L2 six 16-channel chunks ahead, two accumulators (even / odd chunks) so that consecutive matrix instructions are
// independent.  No atomics, no split of the reduction across workgroups: the result is a pure function of the inputs.
// MEASURED SLOWER than the library GEMM it was meant to replace (15.4 / 12.3 / 23.1 us against 7.5 / 10.6 / 13.1 us at the
// L0 / L1 / L2 shapes, tools/pwbench.py): 16 pixels per workgroup make every workgroup stream the whole weight matrix
// (311 KB at L0) from L2.  OptFlow.own_pointwise is off; the entry point stays, tested, for callers without a GEMM library.
// W: (F, cpad) row-major, zero padded to cpad = ceil(C / 32) * 32 (ops.pad_pointwise); F % 16 == 0, F <= 256.
#include "common.h"

namespace qpwc {

typedef float f32x4p __attribute__((ext_vector_type(4)));
constexpr int kPwRows = 16;

template <int NB, int kPwAhead>   // F / 16 = waves per workgroup; chunks of 16 channels of W in flight per wave and set
__global__ __launch_bounds__(NB * 64) void pointwise_bias_kernel(const float* __restrict__ y, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, float* __restrict__ out,
                                                                 int M, int C, int cpad) {
    extern __shared__ __attribute__((aligned(16))) float ys[];        // 16 rows x (cpad rounded up to six chunks + 4)
    constexpr int NT = NB * 64, F = NB * 16;
    const int cpr = (cpad / 16 + 2 * kPwAhead - 1) / (2 * kPwAhead) * (2 * kPwAhead) * 16;   // columns incl. the zero chunks of the last trip
    const int S = cpr + 4;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * kPwRows;
    const int rows = min(kPwRows, M - row0);
    // ---- this wave's first chunks of W: in flight behind the staging ----
    const float* wr = w + (int64_t)(16 * wave + n) * cpad + 4 * g;
    const int nch = cpad / 16;
    f32x4p wv[kPwAhead];
#pragma unroll
    for (int j = 0; j < kPwAhead; ++j) wv[j] = *reinterpret_cast<const f32x4p*>(wr + 16 * (j < nch ? j : nch - 1));
    // ---- stage the 16 x C tile (flat 16-byte loads, scattered into padded rows), zero the pad columns ----
    const float* yt = y + (int64_t)row0 * C;
    const int nflo = rows * C;                    // floats of the tile that exist
    auto fetch = [&](int i) __attribute__((always_inline)) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (4 * i + 3 < nflo) {
            v = *reinterpret_cast<const float4*>(yt + 4 * i);
        } else {                               // the last rows of the last tile / the tile's last float4
            if (4 * i + 0 < nflo) v.x = yt[4 * i + 0];
            if (4 * i + 1 < nflo) v.y = yt[4 * i + 1];
            if (4 * i + 2 < nflo) v.z = yt[4 * i + 2];
        }
        return v;
    };
    auto scatter = [&](int i, float4 v) __attribute__((always_inline)) {
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = 4 * i + k;
            const int r = idx / C, c = idx - r * C;
            if (idx < kPwRows * C) ys[r * S + c] = e[k];
        }
    };
    // the first kPwUnroll pieces of a thread are requested together (one memory round trip), the rest -- only for
    // C > 128 NB -- in a plain loop
    constexpr int kPwUnroll = 6;
    float4 sv[kPwUnroll];
#pragma unroll
    for (int u = 0; u < kPwUnroll; ++u) sv[u] = fetch(tid + NT * u);
#pragma unroll
    for (int u = 0; u < kPwUnroll; ++u) scatter(tid + NT * u, sv[u]);
    for (int i = tid + NT * kPwUnroll; 4 * i < kPwRows * C; i += NT) scatter(i, fetch(i));
    const int npad = cpr - C;
    for (int i = tid; i < kPwRows * npad; i += NT) {
        const int r = i / npad, c = C + (i - r * npad);
        ys[r * S + c] = 0.0f;
    }
    __syncthreads();
    // ---- D[f][px] += W[f][k] y[px][k]: rows = this wave's 16 outputs, columns = the 16 pixels ----
    // Two register sets of kPwAhead chunks alternate: while one feeds the matrix instructions the other is refilled, every
    // trip issues the same requests and reads the same chunks (the tile is zero padded to a multiple of 2 kPwAhead chunks;
    // a chunk past the end multiplies W's last chunk, re-read, with zeros), so that the compiler's vmcnt counts are exact
    // and no end-of-trip copy waits for the refill.  (With a `j < nch` branch around the requests every chunk waited for
    // the request issued a moment earlier: 15.9 us for the L0 shape; DESIGN.md 7.0a, finding 3.)
    f32x4p acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float* yr = ys + n * S + 4 * g;
    const int nchr = cpr / 16;
    f32x4p wn[kPwAhead];
    auto half_trip = [&](f32x4p (&use)[kPwAhead], f32x4p (&fill)[kPwAhead], int j0) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < kPwAhead; ++jj) {
            const int jn = j0 + jj + kPwAhead;
            fill[jj] = *reinterpret_cast<const f32x4p*>(wr + 16 * (jn < nch ? jn : nch - 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < kPwAhead; ++jj) {
            const f32x4p a = use[jj];
            const f32x4p b = *reinterpret_cast<const f32x4p*>(yr + 16 * (j0 + jj));
            if (jj & 1) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc1, 0, 0, 0);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc0, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int j0 = 0; j0 < nchr; j0 += 2 * kPwAhead) {
        half_trip(wv, wn, j0);
        half_trip(wn, wv, j0 + kPwAhead);
    }
    // ---- + bias, store: lane (pixel n, quad g) holds outputs 16 wave + 4g .. + 3 of pixel n ----
    const float4 bq = *reinterpret_cast<const float4*>(bias + 16 * wave + 4 * g);
    if (n < rows)
        *reinterpret_cast<float4*>(out + (int64_t)(row0 + n) * F + 16 * wave + 4 * g) =
            make_float4(acc0[0] + acc1[0] + bq.x, acc0[1] + acc1[1] + bq.y, acc0[2] + acc1[2] + bq.z,
                        acc0[3] + acc1[3] + bq.w);
}

int pointwise_bias_launch(const void* y, const void* w, const void* bias, void* out, int64_t M, int C, int cpad, int F,
                          hipStream_t s) {
    const int64_t nwg = (M + kPwRows - 1) / kPwRows;
    if (nwg > INT32_MAX) {
        set_error("pointwise_bias: too many rows");
        return QPWC_E_SHAPE;
    }
    // chunks per register set: the one of {6, 5, 4} that pads the reduction least (cpad 608 -> 38 chunks -> 40 with 5)
    const int nch = cpad / 16;
    int pf = 6, best = INT32_MAX;
    for (int c = 6; c >= 4; --c) {
        const int r = (nch + 2 * c - 1) / (2 * c) * (2 * c);
        if (r < best) { best = r; pf = c; }
    }
    const size_t lds = (size_t)kPwRows * (best * 16 + 4) * sizeof(float);
    if (lds > 64 * 1024) {
        set_error("pointwise_bias: C=%d needs %zu bytes of LDS per workgroup (limit 64 KiB)", C, lds);
        return QPWC_E_SHAPE;
    }
#define QPWC_PW_LAUNCH2(NB, PF)                                                                                         \
    hipLaunchKernelGGL((pointwise_bias_kernel<NB, PF>), dim3((unsigned)nwg), dim3(NB * 64), lds, s, (const float*)y,      \
                       (const float*)w, (const float*)bias, (float*)out, (int)M, C, cpad)
#define QPWC_PW_LAUNCH(NB) do { if (pf == 6) QPWC_PW_LAUNCH2(NB, 6); else if (pf == 5) QPWC_PW_LAUNCH2(NB, 5); else QPWC_PW_LAUNCH2(NB, 4); } while (0)
    switch (F / 16) {
        case 1: QPWC_PW_LAUNCH(1); break;
        case 2: QPWC_PW_LAUNCH(2); break;
        case 4: QPWC_PW_LAUNCH(4); break;
        case 8: QPWC_PW_LAUNCH(8); break;
        case 16: QPWC_PW_LAUNCH(16); break;
        default: set_error("pointwise_bias: F=%d not in {16,32,64,128,256}", F); return QPWC_E_SHAPE;
    }
#undef QPWC_PW_LAUNCH
#undef QPWC_PW_LAUNCH2
    return check_launch("pointwise_bias_kernel");
}

}  // namespace qpwc
