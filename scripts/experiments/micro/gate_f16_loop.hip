// Micro-benchmark (GPU box): the inner loop of wn_gate_f16_kernel alone -- per tap 12 ds_read_b128 (8 weight + 4 activation
// operands, conflict-free addresses) and 24 v_mfma_f32_16x16x32_f16 on 16 accumulator tiles -- with no LDS-DMA, no barrier and
// no epilogue, 8 waves per CU (one block of 512 threads per CU).  What fraction of the 16-bit matrix peak does the loop reach
//   (a) reads, wait, MFMAs (program order of the kernel's source)     (b) the reads of tap t + 1 in front of the MFMAs of tap t
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/experiments/micro/gate_f16_loop gate_f16_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)

template <bool PIPE, int WAVES, int RT>
__global__ __launch_bounds__(WAVES * 64) void loop_kernel(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 32768; i += WAVES * 64) reinterpret_cast<float *>(lds)[i] = 1e-3f * (float)(i & 255);
    __syncthreads();
    f32x4 accm[RT][4], accx[RT][4];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) accm[r][c] = accx[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f16x8 *bb = reinterpret_cast<const f16x8 *>(lds) + lane;                     // weights: the same 24 KB for every wave
    const f16x8 *ab = reinterpret_cast<const f16x8 *>(lds + 65536) + wave * 512 + lane;  // activations: 8 KB per wave
    f16x8 bh[2][4], bl[2][4], ah[2][RT], al[2][RT];
    auto load_tap = [&](int tap, int set) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bh[set][c] = bb[((tap * 4 + c) * 2 + 0) * 64];
            bl[set][c] = bb[((tap * 4 + c) * 2 + 1) * 64];
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            ah[set][rt] = ab[(rt * 2 + 0) * 64];
            al[set][rt] = ab[(rt * 2 + 1) * 64];
        }
    };
    auto mma_tap = [&](int set) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                accm[rt][c] = MFMA(ah[set][rt], bh[set][c], accm[rt][c]);
                accx[rt][c] = MFMA(ah[set][rt], bl[set][c], accx[rt][c]);
                accx[rt][c] = MFMA(al[set][rt], bh[set][c], accx[rt][c]);
            }
    };
    for (int it = 0; it < iters; ++it) {
        if (!PIPE) {
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                load_tap(tap, 0);
                __builtin_amdgcn_sched_barrier(0);
                mma_tap(0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            load_tap(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_tap(1, 1);
            mma_tap(0);
#pragma unroll
            for (int i = 0; i < 8 + 2 * RT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 12 * RT / (8 + 2 * RT), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            load_tap(2, 0);
            mma_tap(1);
#pragma unroll
            for (int i = 0; i < 8 + 2 * RT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 12 * RT / (8 + 2 * RT), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_tap(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" ::: "memory");          // the LDS reads of the next iteration are not hoisted
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) s += accm[r][c][0] + accx[r][c][1];
    if (s == 123.456f) out[tid] = s;
}

template <bool PIPE, int WAVES, int RT>
static void run(float *out) {
    const int iters = 3000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(loop_kernel<PIPE, WAVES, RT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    loop_kernel<PIPE, WAVES, RT><<<blocks, WAVES * 64, 160 * 1024 - 512>>>(out, 10);
    (void)hipEventRecord(e0);
    loop_kernel<PIPE, WAVES, RT><<<blocks, WAVES * 64, 160 * 1024 - 512>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * WAVES * iters * (36 * RT) * 16384.0;
    const double lds_bytes = (double)WAVES * iters * 3 * (8 + 2 * RT) * 1024.0;                     // per block = per CU and round
    const double us_block = ms * 1e3 / (blocks / 256.0);
    printf("%d x 4 tiles per wave, %s, %2d waves per CU: %7.3f ms, %7.1f TFLOP/s = %.3f of the nominal 16-bit peak; LDS %5.1f B/clk at 2.4 GHz\n",
           RT, PIPE ? "reads of tap t+1 before the MFMAs of tap t" : "reads, wait, MFMAs                        ", WAVES, ms,
           flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 2516.6, lds_bytes / (us_block * 1e3) / 2.4);
}

int main() {
    float *out;
    (void)hipMalloc(&out, 4096);
    run<false, 4, 2>(out);
    run<false, 8, 2>(out);
    run<false, 16, 2>(out);
    run<true, 4, 2>(out);
    run<true, 8, 2>(out);
    run<false, 4, 4>(out);
    run<false, 8, 4>(out);
    run<true, 4, 4>(out);
    run<true, 8, 4>(out);
    return 0;
}
