// Store-path microbenchmark 2 (diagnostic): the cost-volume epilogue's store shapes.
// A tile = 4 rows of 324 floats (row stride = W*81 floats, W=256); per tile either
//   mode 0: 8 instructions (per row: 64 lanes x 16 B, then 17 lanes x 16 B)   [current epilogue]
//   mode 1: 6 instructions (the 324 float4 of the tile in lane order, crossing rows)
//   mode 2: mode 0 with non-temporal stores
// waves/CU = 4, 8, 16.   hipcc --offload-arch=gfx950 -O3 storebench2.hip -o storebench2
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long tiles_per_wave, long n_tiles_x, long row_stride) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const float4 v = make_float4(lane, 1.f, 2.f, 3.f);
    for (long t = 0; t < tiles_per_wave; ++t) {
        const long tile = wave * tiles_per_wave + t;
        const long ty = tile / n_tiles_x, tx = tile % n_tiles_x;
        float* ob = out + ty * 4 * row_stride + tx * 324;
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float4* p = reinterpret_cast<float4*>(ob + r * row_stride) + lane;
                if (MODE == 2) { typedef float f4 __attribute__((ext_vector_type(4))); const f4 vv = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(vv, reinterpret_cast<f4*>(p)); if (lane < 17) __builtin_nontemporal_store(vv, reinterpret_cast<f4*>(p + 64)); }
                else { *p = v; if (lane < 17) p[64] = v; }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int idx = lane + 64 * i;
                if (idx < 324) {
                    const int r = idx / 81, q = idx - 81 * r;
                    reinterpret_cast<float4*>(ob + r * row_stride)[q] = v;
                }
            }
        }
    }
}

int main() {
    const long W = 256, H = 128 * 16, row_stride = W * 81;       // floats; 16 images of 128 rows
    const long n_tiles_x = W / 4, n_tiles = n_tiles_x * (H / 4);  // 32768 tiles, 170 MB
    float* buf;
    hipMalloc(&buf, H * row_stride * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (int wpc : {4, 8, 16}) {
            const int blocks = 256 * wpc / 4;
            const long tiles_per_wave = n_tiles / (blocks * 4);
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, buf, tiles_per_wave, n_tiles_x, row_stride);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, buf, tiles_per_wave, n_tiles_x, row_stride);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, buf, tiles_per_wave, n_tiles_x, row_stride);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double bytes = (double)blocks * 4 * tiles_per_wave * 5184.0;
            printf("mode %d waves/CU %2d: %.1f us  %.2f TB/s\n", mode, wpc, best * 1e3, bytes / best * 1e-9);
        }
    return 0;
}
