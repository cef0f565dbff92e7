// Device-copy variants for the roofline's "achievable HBM ceiling" (read + write bytes / time).
// hipcc --offload-arch=gfx950 -O3 -o copybench copybench.hip && ./copybench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_strided(const u4* __restrict__ s, u4* __restrict__ d, long n) {
    const long stride = (long)gridDim.x * 256 * U;
    for (long i = (long)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + 256 * u < n) v[u] = NT ? __builtin_nontemporal_load(s + i + 256 * u) : s[i + 256 * u];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + 256 * u < n) { if (NT) __builtin_nontemporal_store(v[u], d + i + 256 * u); else d[i + 256 * u] = v[u]; }
    }
}

template <int U, bool NT>
static float run(const u4* s, u4* d, long n, int blocks, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copy_strided<U, NT>), dim3(blocks), dim3(256), 0, 0, s, d, n);
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((copy_strided<U, NT>), dim3(blocks), dim3(256), 0, 0, s, d, n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 2.0f * n * 16 * reps / (ms * 1e-3f) / 1e9f;
}

int main() {
    const long bytes = 512L << 20, n = bytes / 16;
    u4 *s, *d;
    hipMalloc(&s, bytes); hipMalloc(&d, bytes);
    hipMemset(s, 1, bytes);
    const int full1 = (int)((n + 255) / 256);
    printf("U=1 plain one-shot   : %.0f GB/s\n", run<1, false>(s, d, n, full1, 10));
    printf("U=1 nt    one-shot   : %.0f GB/s\n", run<1, true>(s, d, n, full1, 10));
    printf("U=4 plain one-shot   : %.0f GB/s\n", run<4, false>(s, d, n, full1 / 4, 10));
    printf("U=4 nt    one-shot   : %.0f GB/s\n", run<4, true>(s, d, n, full1 / 4, 10));
    printf("U=8 plain one-shot   : %.0f GB/s\n", run<8, false>(s, d, n, full1 / 8, 10));
    for (int per_cu : {4, 8, 16, 32}) {
        printf("U=4 plain %2d/CU grid : %.0f GB/s\n", per_cu, run<4, false>(s, d, n, 256 * per_cu, 10));
        printf("U=4 nt    %2d/CU grid : %.0f GB/s\n", per_cu, run<4, true>(s, d, n, 256 * per_cu, 10));
        printf("U=2 plain %2d/CU grid : %.0f GB/s\n", per_cu, run<2, false>(s, d, n, 256 * per_cu, 10));
    }
    hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("hipMemcpyAsync D2D   : %.0f GB/s\n", 2.0 * bytes * 10 / (ms * 1e-3) / 1e9);
    return 0;
}
