// Marginal cost of vector instructions beside fp32 MFMAs on gfx950 (2 waves per SIMD unless noted).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// SHAPE 0: 16x16x4 (8 accumulators), SHAPE 1: 32x32x2 (4 accumulators); NV = extra VALU per MFMA; KIND 0 v_fma, 1 v_exp, 2 v_pk_fma
// ROLE 0: every wave does MFMA + VALU; ROLE 1: waves of odd blocks do VALU only (same count), even blocks MFMA only
template <int SHAPE, int NV, int KIND, int ROLE>
__global__ __launch_bounds__(256, 2) void probe(float *out, int iters) {
    const int tid = threadIdx.x, lane = tid & 63;
    float a = lane * 0.001f, b = 0.5f + lane * 0.002f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    f32x4 acc4[8];
    f32x16 acc16[4];
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) acc4[j][r] = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc16[j][r] = 0.f;
    const bool do_mfma = ROLE == 0 || (blockIdx.x & 1) == 0;
    const bool do_valu = ROLE == 0 || (blockIdx.x & 1) == 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (do_mfma) {
                if (SHAPE == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc4[j]) : "v"(a), "v"(b));
                else if (j < 4) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc16[j]) : "v"(a), "v"(b));
            }
            if (do_valu && (SHAPE == 0 || j < 4)) {
#pragma unroll
                for (int n = 0; n < NV; ++n) {
                    const int i = (j * NV + n) & 7;
                    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a));
                    else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                    else asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<float2 *>(&v[i & 6])) : "v"(*reinterpret_cast<float2 *>(&v[(i + 2) & 6])));
                }
            }
        }
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) s += acc4[j][r];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc16[j][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int SHAPE, int NV, int KIND, int ROLE>
int run(const char *name, float *out) {
    const int blocks = 2048, iters = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<SHAPE, NV, KIND, ROLE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<SHAPE, NV, KIND, ROLE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    const double nmfma = (double)(ROLE ? blocks / 2 : blocks) * 4 * iters * (SHAPE == 0 ? 8 : 4);
    const double flop = nmfma * (SHAPE == 0 ? 2048.0 : 4096.0);
    // cycles per MFMA per SIMD at 2.4 GHz nominal: each SIMD hosts blocks*4/1024 waves in sequence pairs
    printf("%-44s %8.3f ms  %6.1f TF (%.3f)\n", name, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3);
    return 0;
}

int main() {
    float *out; CHECK(hipMalloc(&out, 4096 * 256 * 4));
    run<0, 0, 0, 0>("16x16x4 + 0 valu", out);
    run<0, 1, 0, 0>("16x16x4 + 1 v_fma / mfma", out);
    run<0, 2, 0, 0>("16x16x4 + 2 v_fma / mfma", out);
    run<0, 4, 0, 0>("16x16x4 + 4 v_fma / mfma", out);
    run<0, 6, 0, 0>("16x16x4 + 6 v_fma / mfma", out);
    run<0, 1, 1, 0>("16x16x4 + 1 v_exp / mfma", out);
    run<0, 2, 1, 0>("16x16x4 + 2 v_exp / mfma", out);
    run<0, 1, 2, 0>("16x16x4 + 1 v_pk_fma / mfma", out);
    run<0, 2, 2, 0>("16x16x4 + 2 v_pk_fma / mfma", out);
    run<1, 0, 0, 0>("32x32x2 + 0 valu", out);
    run<1, 2, 0, 0>("32x32x2 + 2 v_fma / mfma", out);
    run<1, 4, 0, 0>("32x32x2 + 4 v_fma / mfma", out);
    run<1, 8, 0, 0>("32x32x2 + 8 v_fma / mfma", out);
    run<0, 0, 0, 1>("split roles: 16x16x4 half the waves, no valu", out);
    run<0, 2, 0, 1>("split roles: + 2 v_fma/mfma-slot in partner", out);
    run<0, 4, 0, 1>("split roles: + 4 v_fma in partner", out);
    run<0, 6, 0, 1>("split roles: + 6 v_fma in partner", out);
    run<0, 2, 1, 1>("split roles: + 2 v_exp in partner", out);
    return 0;
}
