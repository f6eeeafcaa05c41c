// dispatch-rate probe: kernels whose blocks do a fixed small amount of work; time vs number of blocks
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void spin(float *out, int iters) {
    float v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (v == 123.456f) out[0] = v;
}
int main() {
    float *d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {64, 256, 512}) for (int iters : {0, 200, 2000}) for (int blocks : {1000, 4000, 16000, 64000}) {
        hipLaunchKernelGGL(spin, dim3(blocks), dim3(threads), 0, 0, d, iters);
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(spin, dim3(blocks), dim3(threads), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("threads %4d iters %5d blocks %6d: %8.1f us per launch = %6.2f ns per block\n", threads, iters, blocks, ms * 100, ms * 1e5 / blocks);
    }
    return 0;
}
