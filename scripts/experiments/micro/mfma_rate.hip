// Issue rate of the fp32 MFMA shapes on gfx950: cycles per instruction for chains on 1 / 2 / 4 independent accumulators,
// one wave per SIMD (256 threads per block, one block per CU) and with 2 / 3 blocks per CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate scripts/experiments/micro/mfma_rate.hip && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, long long *cycles) {
    const float a = 1.f + threadIdx.x * 1e-3f, b = 0.5f;
    f32x16 c32[NACC];
    f32x4 c16[NACC];
    for (int i = 0; i < NACC; ++i) {
        for (int r = 0; r < 16; ++r) c32[i][r] = 0.f;
        for (int r = 0; r < 4; ++r) c16[i][r] = 0.f;
    }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (SHAPE == 32) c32[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c32[i], 0, 0, 0);
                else c16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c16[i], 0, 0, 0);
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) {
        for (int r = 0; r < 16; ++r) s += c32[i][r];
        for (int r = 0; r < 4; ++r) s += c16[i][r];
    }
    if (s == 123.456f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int SHAPE, int NACC>
void run(int blocks, const char *what) {
    float *out;
    long long *cyc;
    hipMalloc(&out, 1024);
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((rate_kernel<SHAPE, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((rate_kernel<SHAPE, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * NACC;                      // MFMAs per wave
    const double flop = n * (SHAPE == 32 ? 4096.0 : 2048.0) * 4 * blocks;
    printf("%-28s blocks %4d  acc %d: %7.1f us  %6.1f TFLOP/s  s_memtime ticks per MFMA (wave 0) %.2f\n", what, blocks, NACC, ms * 1e3,
           flop / (ms * 1e-3) * 1e-12, (double)h / n);
}

int main() {
    for (int blocks : {256, 512, 768}) {
        run<32, 1>(blocks, "v_mfma_f32_32x32x2_f32");
        run<32, 2>(blocks, "v_mfma_f32_32x32x2_f32");
        run<32, 4>(blocks, "v_mfma_f32_32x32x2_f32");
        run<16, 1>(blocks, "v_mfma_f32_16x16x4_f32");
        run<16, 2>(blocks, "v_mfma_f32_16x16x4_f32");
        run<16, 4>(blocks, "v_mfma_f32_16x16x4_f32");
    }
    return 0;
}
