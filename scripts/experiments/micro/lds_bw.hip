// Micro-benchmark (GPU box): LDS read bandwidth per CU for ds_read_b64 / ds_read_b128 with lane-contiguous (conflict-free)
// addresses, 1 .. 16 waves per CU, and the same with v_mfma_f32_16x16x32_f16 issued between the reads (do LDS reads and
// 16-bit MFMAs overlap?).  Build here: hipcc --offload-arch=gfx950 -O3 -o scripts/experiments/micro/lds_bw lds_bw.hip
// (the binary travels with the snapshot); run: gpurun -- scripts/experiments/micro/lds_bw
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int BYTES, int MFMA_PER_READ>
__global__ __launch_bounds__(1024) void lds_read_kernel(float *out, long long *cycles, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<float *>(lds)[i] = (float)i;
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 macc[4] = {acc, acc, acc, acc};
    const char *base = lds + (tid & 63) * BYTES + (tid >> 6) * 4096;      // a wave reads 64 x BYTES contiguous bytes
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (BYTES == 16) {
                f32x4 v;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(uintptr_t)(base + ((it * 8 + u) & 3) * 1024)));
                acc += v;
                if (MFMA_PER_READ) {
                    f16x8 a, b;
                    __builtin_memcpy(&a, &v, 16);
                    __builtin_memcpy(&b, &v, 16);
#pragma unroll
                    for (int m = 0; m < MFMA_PER_READ; ++m) macc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, macc[m & 3], 0, 0, 0);
                }
            } else {
                float2 v;
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(uintptr_t)(base + ((it * 8 + u) & 7) * 512)));
                acc[0] += v.x;
                acc[1] += v.y;
            }
        }
    }
    const long long t1 = clock64();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int m = 0; m < 4; ++m) s += macc[m][0];
    if (s == 123.456f) out[tid] = s;
}

template <int BYTES, int MPR>
static void run(const char *name, int waves, float *out, long long *cyc_d) {
    const int iters = 4000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void *>(lds_read_kernel<BYTES, MPR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // 160 KB of dynamic LDS: one block per CU, so `waves` is the number of waves per CU
    lds_read_kernel<BYTES, MPR><<<blocks, waves * 64, 160 * 1024 - 512>>>(out, cyc_d, 10);
    hipEventRecord(e0);
    lds_read_kernel<BYTES, MPR><<<blocks, waves * 64, 160 * 1024 - 512>>>(out, cyc_d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    long long cyc = 0;
    hipMemcpy(&cyc, cyc_d, sizeof(cyc), hipMemcpyDeviceToHost);
    const double bytes_per_block = (double)iters * 8 * waves * 64 * BYTES;
    // clock64 ticks at a fixed 100 MHz on this part; bytes per shader clock from the launch time at 2.4 GHz nominal
    const double rounds = blocks / 256.0;
    const double us_per_block = ms * 1e3 / rounds;
    printf("%-34s waves/CU %2d: %7.1f us per block, %6.1f bytes per ns and CU = %5.1f B/clk at 2.4 GHz", name, waves, us_per_block,
           bytes_per_block / (us_per_block * 1e3), bytes_per_block / (us_per_block * 1e3) / 2.4);
    if (MPR) printf(", %5.1f MFMA cycles (16 each) per ns and SIMD", (double)iters * 8 * MPR * waves / 4 * 16 / (us_per_block * 1e3));
    printf("  [clock64 ticks %lld]\n", cyc);
}

int main() {
    float *out;
    long long *cyc;
    hipMalloc(&out, 4096);
    hipMalloc(&cyc, 8 * 4096);
    for (int w : {1, 2, 4, 8, 16}) run<8, 0>("ds_read_b64", w, out, cyc);
    for (int w : {1, 2, 4, 8, 16}) run<16, 0>("ds_read_b128", w, out, cyc);
    for (int w : {4, 8, 16}) run<16, 1>("ds_read_b128 + 1 MFMA per read", w, out, cyc);
    for (int w : {4, 8, 16}) run<16, 2>("ds_read_b128 + 2 MFMA per read", w, out, cyc);
    for (int w : {4, 8, 16}) run<16, 4>("ds_read_b128 + 4 MFMA per read", w, out, cyc);
    return 0;
}
