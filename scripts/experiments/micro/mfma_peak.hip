// Micro-benchmark (GPU box): sustained issue rate of v_mfma_f32_16x16x4_f32 (the float32 WaveNet kernels' instruction) and
// v_mfma_f32_16x16x32_f16 over a whole-chip launch of several milliseconds: what fraction of the nominal 2.4 GHz peak
// (157.3 / 2 516 TFLOP/s) a kernel that does nothing but MFMAs reaches.  Independent accumulator chains per wave; 4, 8, 12
// and 16 waves per CU.  Build: hipcc --offload-arch=gfx950 -O3 -o scripts/experiments/micro/mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <bool F16>
__global__ __launch_bounds__(1024) void mfma_kernel(float *out, int iters) {
    extern __shared__ char lds[];
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a32 = 1.0f + threadIdx.x * 1e-6f, b32 = 0.5f;
    f16x8 a16, b16;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a16[i] = (_Float16)(1.0f + 0.001f * i);
        b16[i] = (_Float16)0.5f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (F16) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16, b16, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a32, b32, acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <bool F16>
static void run(int waves, float *out) {
    const int iters = F16 ? 40000 : 20000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mfma_kernel<F16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    mfma_kernel<F16><<<blocks, waves * 64, 160 * 1024 - 512>>>(out, 10);          // 160 KB of LDS: one block per CU
    (void)hipEventRecord(e0);
    mfma_kernel<F16><<<blocks, waves * 64, 160 * 1024 - 512>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * waves * iters * 32;
    const double flop = mfma * (F16 ? 16384.0 : 2048.0);
    const double peak = F16 ? 2516.6 : 157.3;
    printf("%s  waves/CU %2d: %8.3f ms, %8.1f TFLOP/s = %.3f of the nominal peak (%.1f)\n", F16 ? "v_mfma_f32_16x16x32_f16" : "v_mfma_f32_16x16x4_f32 ",
           waves, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / peak, peak);
}

int main() {
    float *out;
    (void)hipMalloc(&out, 4096);
    for (int w : {4, 8, 12, 16}) run<false>(w, out);
    for (int w : {4, 8, 12, 16}) run<true>(w, out);
    return 0;
}
