// Store-path microbenchmark (diagnostic): W waves per CU stream 1 KiB-per-instruction stores
// (16 B per lane) the way the cost-volume epilogue does -- rows of 324 floats per 4 pixels, 4 rows
// per tile -- with the row start either 16-byte or 128-byte aligned.  Prints B/clk/CU and TB/s.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/storebench.hip -o /tmp/storebench && /tmp/storebench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void store_kernel(float* out, long tiles_per_wave, long row_floats,
                                                     int stores_per_tile, long total_floats) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const float4 v = make_float4(lane, 1.f, 2.f, 3.f);
    for (long t = 0; t < tiles_per_wave; ++t) {
        const long tile = wave * tiles_per_wave + t;
        float* base = out + (tile * stores_per_tile) * row_floats;
#pragma unroll 8
        for (int r = 0; r < stores_per_tile; ++r) {
            float* p = base + r * row_floats + 4 * lane;
            if (p + 4 <= out + total_floats) *reinterpret_cast<float4*>(p) = v;
        }
    }
}

__global__ __launch_bounds__(256) void load_kernel(const float* in, float* sink, long tiles_per_wave,
                                                    long row_floats, int loads_per_tile, long total_floats) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    float4 acc = make_float4(0, 0, 0, 0);
    for (long t = 0; t < tiles_per_wave; ++t) {
        const long tile = wave * tiles_per_wave + t;
        const float* base = in + (tile * loads_per_tile) * row_floats;
#pragma unroll 8
        for (int r = 0; r < loads_per_tile; ++r) {
            const float* p = base + r * row_floats + 4 * lane;
            if (p + 4 <= in + total_floats) {
                const float4 x = *reinterpret_cast<const float4*>(p);
                acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
            }
        }
    }
    if (acc.x == 12345.678f) sink[0] = acc.y + acc.z + acc.w;
}

int main() {
    const long total = 96l << 20;  // floats: 384 MiB
    float* buf; float* sink;
    hipMalloc(&buf, total * 4 + 4096);
    hipMalloc(&sink, 64);
    hipMemset(buf, 0, total * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int cus = 256;
    for (int is_load = 0; is_load < 2; ++is_load)
    for (int align = 0; align < 2; ++align) {
        const long row = align ? 256 : 324;  // floats per store row; 324 -> 16-B aligned only (first row offset 0)
        for (int wpc : {4, 8, 16, 32}) {
            const int blocks = cus * wpc / 4;
            const long waves = (long)blocks * 4;
            const int per_tile = 8;
            const long tiles_per_wave = total / (waves * per_tile * row);
            float ms_best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (is_load)
                    hipLaunchKernelGGL(load_kernel, dim3(blocks), dim3(256), 0, 0, buf + (align ? 0 : 4), sink, tiles_per_wave, row, per_tile, total);
                else
                    hipLaunchKernelGGL(store_kernel, dim3(blocks), dim3(256), 0, 0, buf + (align ? 0 : 4), tiles_per_wave, row, per_tile, total);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < ms_best) ms_best = ms;
            }
            const double bytes = (double)waves * tiles_per_wave * per_tile * 1024.0;
            printf("%s %s waves/CU %2d: %.3f ms  %.2f TB/s  (%.1f B/clk/CU at 2.0 GHz)\n", is_load ? "load " : "store",
                   align ? "128B-aligned rows" : "16B-aligned rows ", wpc, ms_best, bytes / ms_best * 1e-9,
                   bytes / ms_best * 1e-3 / 256 / 2.0e9 * 1e3 * 1e-3 * 1e3);
        }
    }
    return 0;
}
