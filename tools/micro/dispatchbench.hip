// How much of a CU's workgroup slots stay empty between short-lived workgroups?  4096 workgroups of 256 threads that
// only spin for `life` shader cycles, with an LDS allocation that admits 8 / 3 / 2 of them per CU; the launch should
// take rounds x life if a finished workgroup's slot were refilled at once.
//   hipcc --offload-arch=gfx950 -O3 -o dispatchbench tools/micro/dispatchbench.hip && ./dispatchbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int LDS_BYTES>
__global__ __launch_bounds__(256) void spin_kernel(long long life, int* sink) {
    __shared__ char lds[LDS_BYTES];
    const long long t0 = __builtin_amdgcn_s_memtime();
    lds[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    while (__builtin_amdgcn_s_memtime() - t0 < life) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x + 1) & 255] == 77 && life < 0) sink[0] = 1;
}

template <int LDS_BYTES>
static void run(const char* name, int per_cu, int grid, long long life, int* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(spin_kernel<LDS_BYTES>, dim3(grid), dim3(256), 0, 0, life, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(spin_kernel<LDS_BYTES>, dim3(grid), dim3(256), 0, 0, life, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double rounds = (double)grid / (256.0 * per_cu);
    printf("%-10s grid %5d  life %6lld cycles  %4.1f rounds  launch %7.1f us  = %7.0f cycles/round at 2.1 GHz (life x %.2f)\n",
           name, grid, life, rounds, us, us * 2100.0 / rounds, us * 2100.0 / rounds / (double)life);
}

int main() {
    int* sink;
    hipMalloc(&sink, 4);
    for (long long life : {4000LL, 8000LL, 16000LL, 32000LL}) {
        run<8 * 1024>("8 per CU", 8, 4096, life, sink);     // LDS 8 KB: the wave limit (8 per SIMD) decides
        run<40 * 1024>("3 per CU", 3, 3072, life, sink);    // 40 KB -> 3 (gfx950 LDS allocation granularity permitting)
        run<64 * 1024>("2 per CU", 2, 4096, life, sink);
    }
    return 0;
}
