// PQMF synthesis filterbank, polyphase form.
//
// restates TFPQMF.synthesis   reference MBExWN_NVoc/vocoder/model/tf_preprocess.py:204-226
//   u_k[m*M] = M * x[m, k] (zero-stuffing with gain M), zero pad taps/2, cross-correlation with g_k:
//   y[n] = sum_k sum_j u_k[n + j - taps/2] * g_k[j]
// Only taps j == (taps/2 - n) mod M meet a non-zero sample, so with n = q*M + p
//   y[q*M + p] = sum_{i < n_dm} sum_k (M * x[q + dm_min + i, k]) * G[p, i, k],   G[p,i,k] = g_k[(dm_min+i)*M + taps/2 - p]
// (table built on the host: mbexwn_vocoder_amd/tables.py::pqmf_polyphase).
//
// Bandwidth-type stage: a block stages QB sub-band steps (+ halo) and the 8 KB polyphase table in LDS,
// reads x coalesced once, writes QB*M output samples coalesced.
#include "mbx_kernels.h"

namespace mbx {

constexpr int PQMF_QB = 64;

__global__ __launch_bounds__(256) void pqmf_kernel(const float *x, long long x_bstride, const int *n_frames,
                                                   int steps_per_frame, int max_steps, int M, const float *poly,
                                                   int n_dm, int dm_min, float *y, long long y_bstride) {
    extern __shared__ float smem[];
    const int b = blockIdx.y;
    const int steps = n_frames ? n_frames[b] * steps_per_frame : max_steps;
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
                 int batch, int subbands, const float *poly, int n_dm, int dm_min, float *y, long long y_bstride,
                 hipStream_t stream) {
    if (max_steps <= 0 || batch <= 0) return;
    const int n_rows = PQMF_QB + n_dm - 1;
    const size_t smem = sizeof(float) * (size_t)(n_rows * subbands + subbands * n_dm * subbands);
    hipLaunchKernelGGL(pqmf_kernel, dim3((max_steps + PQMF_QB - 1) / PQMF_QB, batch), dim3(256), smem, stream, x,
                       x_bstride, n_frames, steps_per_frame, max_steps, subbands, poly, n_dm, dm_min, y, y_bstride);
}

}  // namespace mbx
