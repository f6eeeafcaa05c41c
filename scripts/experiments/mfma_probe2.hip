// Micro-probe 2: 16x16x4 fp32 MFMA, wave tile 16 groups x 64 columns x 6 products (24 accumulators of 4 regs),
// operands via ds_read_b64 (X) / ds_read_b128 (B), input combination on float2 (half the VALU per MFMA cycle).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: bare 16x16x4 MFMAs (24 accumulators)   MODE 1: + LDS operand reads   MODE 2: + VALU input combination
template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[3 * 5632];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 3 * 5632; i += 256) lds[i] = (float)((i * 2654435761u) >> 20) * 1e-4f;
    __syncthreads();
    f32x4 acc[6][4];
    for (int j = 0; j < 6; ++j) for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) acc[j][c][r] = 0.f;
    float a = lane * 0.001f, b = 0.5f + lane * 0.002f;
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[j][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j][c], 0, 0, 0);
        }
    } else {
        int stage = 0;
        float2 x[6], xn[6];
        float4 bw[2][2], bn[2][2];
        const float *xb0 = lds + lane * 2;
#pragma unroll
        for (int q = 0; q < 6; ++q) x[q] = *reinterpret_cast<const float2 *>(xb0 + q * 160);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 2; ++e) bw[h][e] = *reinterpret_cast<const float4 *>(lds + 2560 + (h * 2 + e) * 256 + lane * 4);
        for (int it = 0; it < iters; ++it) {
            const float *sb = lds + stage * 5632;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                // weights of the next pair: [product][column tile pair] -> float4 = (step0 tile0, step1 tile0, step0 tile1, step1 tile1)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        bn[h][e] = *reinterpret_cast<const float4 *>(sb + 2560 + ((((2 * g + 2) % 6) + h) * 2 + e) * 256 + lane * 4);
                float2 u0, u1;
                if (MODE >= 2) {
                    u0 = make_float2(fmaf(4.f, x[0].x, fmaf(-5.f, x[2].x, x[4].x)), fmaf(4.f, x[0].y, fmaf(-5.f, x[2].y, x[4].y)));
                    u1 = make_float2(fmaf(-4.f, x[1].x, x[3].x) + fmaf(-4.f, x[2].x, x[5].x), fmaf(-4.f, x[1].y, x[3].y) + fmaf(-4.f, x[2].y, x[5].y));
                } else { u0 = x[g]; u1 = x[g + 3]; }
                if (g == 1) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) xn[q] = *reinterpret_cast<const float2 *>(sb + q * 160 + 32 + lane * 2);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float2 u = h ? u1 : u0;
                    f32x4 *ac = acc[2 * g + h];
                    ac[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.x, bw[h][0].x, ac[0], 0, 0, 0);
                    ac[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.x, bw[h][0].z, ac[1], 0, 0, 0);
                    ac[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.x, bw[h][1].x, ac[2], 0, 0, 0);
                    ac[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.x, bw[h][1].z, ac[3], 0, 0, 0);
                    ac[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.y, bw[h][0].y, ac[0], 0, 0, 0);
                    ac[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.y, bw[h][0].w, ac[1], 0, 0, 0);
                    ac[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.y, bw[h][1].y, ac[2], 0, 0, 0);
                    ac[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.y, bw[h][1].w, ac[3], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h) { bw[h][0] = bn[h][0]; bw[h][1] = bn[h][1]; }
                if (g == 2) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) x[q] = xn[q];
                }
            }
            stage = stage == 2 ? 0 : stage + 1;
        }
    }
    float s = 0.f;
    for (int j = 0; j < 6; ++j) for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) s += acc[j][c][r];
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
    const double flop = (double)blocks * 4 * iters * 48 * 2048.0;
    printf("%-28s blocks %5d iters %5d  %8.3f ms  %7.1f TFLOP/s  (%.3f of 157.3)\n", name, blocks, iters, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3);
    return 0;
}

int main() {
    float *out; CHECK(hipMalloc(&out, 4096 * 256 * 4));
    const int iters = 400;
    for (int blocks : {512, 2048}) {
        run<0>("16x16x4 bare 24 acc", blocks, iters, out);
        run<1>("16x16x4 + ds_read", blocks, iters, out);
        run<2>("16x16x4 + ds_read + valu/2", blocks, iters, out);
    }
    return 0;
}
