// Micro-probe: what limits an LDS-read + VALU + fp32-MFMA loop on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: bare MFMA, 6 accumulators round robin, operands in registers
// MODE 1: pairs alternate c0/c1 in groups of 8 (like the product kernel), operands in registers
// MODE 2: MODE 1 + operands via ds_read_b128 (12 per 24 MFMAs), prefetched one group ahead
// MODE 3: MODE 2 + VALU input combination (approx 3 VALU per MFMA)
template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[3 * 5632];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 3 * 5632; i += 256) lds[i] = (float)((i * 2654435761u) >> 20) * 1e-4f;
    __syncthreads();
    f32x16 acc[6];
    for (int j = 0; j < 6; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = lane * 0.001f, b = 0.5f + lane * 0.002f;
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        }
    } else if (MODE == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[2 * g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[2 * g], 0, 0, 0);
                    acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[2 * g + 1], 0, 0, 0);
                }
        }
    } else {
        int stage = 0;
        const float *xb = lds + lane * 4;
        float4 x[6], bw[2], bn[2], u0, u1;
#pragma unroll
        for (int q = 0; q < 6; ++q) x[q] = *reinterpret_cast<const float4 *>(xb + q * 320);
        bw[0] = *reinterpret_cast<const float4 *>(xb + 2560);
        bw[1] = *reinterpret_cast<const float4 *>(xb + 2560 + 256);
        for (int it = 0; it < iters; ++it) {
            const float *sb = lds + stage * 5632 + lane * 4;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                bn[0] = *reinterpret_cast<const float4 *>(sb + 2560 + ((2 * g + 2) % 6) * 256);
                bn[1] = *reinterpret_cast<const float4 *>(sb + 2560 + ((2 * g + 3) % 6) * 256);
                if (MODE >= 3) {
                    u0 = make_float4(fmaf(4.f, x[0].x, fmaf(-5.f, x[2].x, x[4].x)), fmaf(4.f, x[0].y, fmaf(-5.f, x[2].y, x[4].y)),
                                     fmaf(4.f, x[0].z, fmaf(-5.f, x[2].z, x[4].z)), fmaf(4.f, x[0].w, fmaf(-5.f, x[2].w, x[4].w)));
                    u1 = make_float4(fmaf(-4.f, x[1].x, x[3].x) + fmaf(-4.f, x[2].x, x[5].x), fmaf(-4.f, x[1].y, x[3].y) + fmaf(-4.f, x[2].y, x[5].y),
                                     fmaf(-4.f, x[1].z, x[3].z) + fmaf(-4.f, x[2].z, x[5].z), fmaf(-4.f, x[1].w, x[3].w) + fmaf(-4.f, x[2].w, x[5].w));
                } else { u0 = x[g]; u1 = x[g + 3]; }
                if (g == 1) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) x[q] = *reinterpret_cast<const float4 *>(sb + q * 320 + 64);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[2 * g] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.x, bw[0].x, acc[2 * g], 0, 0, 0);
                acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.x, bw[1].x, acc[2 * g + 1], 0, 0, 0);
                acc[2 * g] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.y, bw[0].y, acc[2 * g], 0, 0, 0);
                acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.y, bw[1].y, acc[2 * g + 1], 0, 0, 0);
                acc[2 * g] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.z, bw[0].z, acc[2 * g], 0, 0, 0);
                acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.z, bw[1].z, acc[2 * g + 1], 0, 0, 0);
                acc[2 * g] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.w, bw[0].w, acc[2 * g], 0, 0, 0);
                acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.w, bw[1].w, acc[2 * g + 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                bw[0] = bn[0]; bw[1] = bn[1];
            }
            stage = stage == 2 ? 0 : stage + 1;
        }
    }
    float s = 0.f;
    for (int j = 0; j < 6; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
int run(const char *name, int blocks, int iters, float *out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    CHECK(hipDeviceSynchronize());
    const int reps = 5;
    CHECK(hipEventRecord(e0, 0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double flop = (double)blocks * 4 * iters * 24 * 4096.0;
    printf("%-28s blocks %5d iters %5d  %8.3f ms  %7.1f TFLOP/s  (%.3f of 157.3)\n", name, blocks, iters, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3);
    return 0;
}

int main() {
    float *out; CHECK(hipMalloc(&out, 4096 * 256 * 4));
    const int iters = 400;
    for (int blocks : {256, 512, 1024, 2048}) {
        run<0>("bare 6 acc round robin", blocks, iters, out);
        run<1>("pairs c0/c1 x4", blocks, iters, out);
        run<2>("pairs + ds_read_b128", blocks, iters, out);
        run<3>("pairs + ds_read + valu", blocks, iters, out);
    }
    return 0;
}
