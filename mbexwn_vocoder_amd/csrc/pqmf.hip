// PQMF synthesis filterbank, polyphase form.
//
// restates TFPQMF.synthesis   reference MBExWN_NVoc/vocoder/model/tf_preprocess.py:204-226
//   u_k[m*M] = M * x[m, k] (zero-stuffing with gain M), zero pad taps/2, cross-correlation with g_k:
//   y[n] = sum_k sum_j u_k[n + j - taps/2] * g_k[j]
// Only taps j == (taps/2 - n) mod M meet a non-zero sample, so with n = q*M + p
//   y[q*M + p] = sum_{i < n_dm} sum_k (M * x[q + dm_min + i, k]) * G[p, i, k],   G[p,i,k] = g_k[(dm_min+i)*M + taps/2 - p]
// (table built on the host: mbexwn_vocoder_amd/tables.py::pqmf_polyphase).
//
// As a matrix product: with the M-wide rows of x laid out flat, the inputs of step q are the K = n_dm * M consecutive
// values starting at row q + dm_min, so y[q*M + p] = sum_i xs[q*M + i] G[p][i]: rows = steps, K = 135, 15 columns.
// Block = 4 waves x 16 steps; a wave runs ceil(K / 4) v_mfma_f32_16x16x4_f32 steps on one 16 x 16 tile.  The polyphase
// weights are the B operand: lane (column p = l & 15, k = l >> 4) keeps G[p][4 s + k] for all steps s in registers (34 for
// the canonical bank, loaded once per wave from a transposed, zero padded copy of the table: 16 consecutive floats per k).  The A operand -- lane (row l & 15, k) reads xs[(q + row) M + 4 s + k] -- comes
// from the block's LDS stage of QB + n_dm - 1 rows (M * x, zero outside the item): one 4-byte LDS read per MFMA and lane,
// the rows M = 15 words apart, so conflict-free.  (Round 1 computed every output sample in its own thread: 2 LDS reads per
// multiply-add, LDS-bound at 70 us per 16 x 10 s.)
#include "mbx_kernels.h"

namespace mbx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PQMF_QB = 64;
constexpr int PQMF_MAX_KSTEPS = 48;        // K = n_dm * M <= 192

__global__ __launch_bounds__(256) void pqmf_kernel(const float *x, long long x_bstride, const int *n_frames,
                                                   int steps_per_frame, int max_steps, int M, const float *poly_t,
                                                   int n_dm, int dm_min, float *y, long long y_bstride) {
    extern __shared__ float smem[];
    const int b = blockIdx.y;
    const int steps = item_rows(n_frames, b, steps_per_frame, max_steps);
    const int q0 = blockIdx.x * PQMF_QB;
    if (q0 >= steps) return;
    const int n_rows = PQMF_QB + n_dm - 1;
    float *xs = smem;                      // (n_rows, M) + 4: M * x, zero outside the item
    const float *xb = x + (long long)b * x_bstride;
    const float gain = (float)M;
    // (n_rows * M + 4 <= 64 * 16 + 192 + 4 elements = at most five per thread.  All requests first, from clamped addresses,
    // selected afterwards: as `cond ? load : 0` inside the loop every element was a branch with its own s_waitcnt vmcnt(0),
    // five serial round trips per block)
    constexpr int XS_ITERS = 5;
    float xv[XS_ITERS];
#pragma unroll
    for (int it = 0; it < XS_ITERS; ++it) {
        const int i = threadIdx.x + 256 * it;
        const int r = i / M, k = i - r * M;
        const int m = min(max(q0 + dm_min + r, 0), steps - 1);
        xv[it] = xb[(long long)m * M + k];
    }
#pragma unroll
    for (int it = 0; it < XS_ITERS; ++it) {
        const int i = threadIdx.x + 256 * it;
        const int r = i / M;
        const int m = q0 + dm_min + r;
        if (i < n_rows * M + 4) xs[i] = (r < n_rows && m >= 0 && m < steps) ? gain * xv[it] : 0.f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int K = n_dm * M, ksteps = (K + 3) >> 2;
    // B operand: G[p = l16][4 s + kq] from the transposed, zero padded table (16 consecutive floats per k)
    float bw[PQMF_MAX_KSTEPS];
#pragma unroll
    for (int s2 = 0; s2 < PQMF_MAX_KSTEPS; ++s2) bw[s2] = s2 < ksteps ? poly_t[(4 * s2 + kq) * 16 + l16] : 0.f;
    __syncthreads();
    const float *xa = xs + (16 * wave + l16) * M + kq;      // A operand of step s: xa[4 s]
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < PQMF_MAX_KSTEPS; ++s2)
        if (s2 < ksteps) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[4 * s2], bw[s2], acc, 0, 0, 0);
    // register v = step 16 wave + 4 kq + v of the block, phase p = l16
    float *yb = y + (long long)b * y_bstride;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int q = q0 + 16 * wave + 4 * kq + v;
        if (q < steps && l16 < M) yb[(long long)q * M + l16] = acc[v];
    }
}

// any bank the matrix form does not fit (more than 16 bands or K > 192): one output sample per thread
__global__ __launch_bounds__(256) void pqmf_generic_kernel(const float *x, long long x_bstride, const int *n_frames,
                                                           int steps_per_frame, int max_steps, int M, const float *poly,
                                                           int n_dm, int dm_min, float *y, long long y_bstride) {
    extern __shared__ float smem[];
    const int b = blockIdx.y;
    const int steps = item_rows(n_frames, b, steps_per_frame, max_steps);
    const int q0 = blockIdx.x * PQMF_QB;
    if (q0 >= steps) return;
    const int n_rows = PQMF_QB + n_dm - 1;
    float *xs = smem;                      // (n_rows, M)   M * x, zero outside the item
    float *gs = smem + n_rows * M;         // (M, n_dm, M)
    const float *xb = x + (long long)b * x_bstride;
    const float gain = (float)M;
    for (int i = threadIdx.x; i < n_rows * M; i += blockDim.x) {
        const int r = i / M, k = i - r * M;
        const int m = q0 + dm_min + r;
        xs[i] = (m >= 0 && m < steps) ? gain * xb[(long long)m * M + k] : 0.f;
    }
    for (int i = threadIdx.x; i < M * n_dm * M; i += blockDim.x) gs[i] = poly[i];
    __syncthreads();
    float *yb = y + (long long)b * y_bstride;
    const int n_out = min(PQMF_QB, steps - q0) * M;
    for (int o = threadIdx.x; o < n_out; o += blockDim.x) {
        const int ql = o / M, p = o - ql * M;
        const float *xr = xs + ql * M;
        const float *gp = gs + p * n_dm * M;
        float acc = 0.f;
        for (int i = 0; i < n_dm * M; ++i) acc += xr[i] * gp[i];   // rows ql..ql+n_dm-1 are contiguous in xs
        yb[(long long)q0 * M + o] = acc;
    }
}

void launch_pqmf(const float *x, long long x_bstride, const int *n_frames, int steps_per_frame, int max_steps,
                 int batch, int subbands, const float *poly, const float *poly_t, int n_dm, int dm_min, float *y,
                 long long y_bstride, hipStream_t stream) {
    if (max_steps <= 0 || batch <= 0) return;
    const int n_rows = PQMF_QB + n_dm - 1;
    const dim3 grid((max_steps + PQMF_QB - 1) / PQMF_QB, batch);
    if (!poly_t || subbands > 16 || n_dm * subbands > 4 * PQMF_MAX_KSTEPS) {
        const size_t smem_g = sizeof(float) * (size_t)(n_rows * subbands + subbands * n_dm * subbands);
        hipLaunchKernelGGL(pqmf_generic_kernel, grid, dim3(256), smem_g, stream, x, x_bstride, n_frames, steps_per_frame,
                           max_steps, subbands, poly, n_dm, dm_min, y, y_bstride);
        return;
    }
    const size_t smem = sizeof(float) * (size_t)(n_rows * subbands + 4);
    hipLaunchKernelGGL(pqmf_kernel, grid, dim3(256), smem, stream, x,
                       x_bstride, n_frames, steps_per_frame, max_steps, subbands, poly_t, n_dm, dm_min, y, y_bstride);
}

}  // namespace mbx
