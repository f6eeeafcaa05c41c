// WaveNet output stage: y = (y_acc +) x W + b (1x1, C -> n_out <= 32) followed by the post-net s = y W_post + b_post
// (1x1, n_out -> M <= 16), one kernel, one pass over x.  Two uses:
//   x = skip sum, W = W_end                        (end convolution after un-folded res/skip layers)
//   x = gate output of the last layer, W = W_skip W_end, y_acc = contributions of the earlier layers
//                                                  (skip path folded into the end convolution, engine.fold_skip_weights)
//
// Same arithmetic as the two EPI_LINEAR launches of conv1d_mfma_kernel it replaces (reference
// MBExWN_NVoc/vocoder/model/custom_AE_layers.py:338-341 `end` convolution of the WaveNet,
// custom_pulsed_generator.py:490-493,913-914 post-net): both are linear with no activation in between; y is still
// written because it is a stage output ("wn_out").
//
// Block = 32 rows, 4 waves that split the C input channels (wave w takes the 8-channel groups w, w+4, ...): the skip
// rows go straight from global memory into the MFMA A operand (lane = row, 16 bytes = four k steps of one lane half),
// W_end comes pre-packed from the host in the matching operand order [group][lane half][column][k step]
// (engine.pack_end_weights, zero padded to 32 columns), one coalesced 16-byte load per lane.  The four partial 32 x 32
// results are summed through LDS, which also puts whole rows in front of single lanes for the post-net.
// HBM-bound: rows x C x 4 bytes.
#include "mbx_kernels.h"

namespace mbx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void wn_tail_kernel(const float *skip, long long skip_bstride, const int *n_frames,
                                                      int rows_per_frame, int max_rows, int C, const float *w_end_packed,
                                                      const float *b_end, int n_out, const float *w_post,
                                                      const float *b_post, int M, const float *y_acc, float *y,
                                                      long long y_bstride, float *sub, long long sub_bstride) {
    __shared__ float tile[4 * 32 * 33];
    __shared__ float wp[32 * 16 + 16];            // n_out * M post weights, then M post biases
    const int b = blockIdx.y;
    const int rows = item_rows(n_frames, b, rows_per_frame, max_rows);
    const int m0 = blockIdx.x * 32;
    if (m0 >= rows) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lk = lane >> 5;

    const int nc8 = (C + 7) / 8;                  // groups of 8 input channels (4 k steps x 2 lane halves)
    const float *xr = skip + (long long)b * skip_bstride + (long long)min(m0 + lrow, rows - 1) * C + 4 * lk;
    const float4 *wv = reinterpret_cast<const float4 *>(w_end_packed) + lane;
    // what the epilogue adds to this thread's four outputs (bias, contributions of the earlier layers): requested here, with
    // everything else, from clamped addresses and masked afterwards.  (As `cond ? load : 0` the compiler puts every load into a
    // branch of its own and waits for it alone: the K loop was "one load, s_waitcnt vmcnt(0), four MFMAs" per group, and the
    // epilogue eight serial round trips behind the block's barrier -- read off the ISA in round 4.)
    const int e_rr = tid >> 3, e_nb = (tid & 7) * 4;
    const float *ya = y_acc ? y_acc + (long long)b * y_bstride + (long long)min(m0 + e_rr, rows - 1) * n_out : nullptr;
    // ... and the post-net's weights (n_out * M <= 512: two per thread) and biases, stored to LDS behind the K loop
    const int n_wp = n_out * M;
    const float wp0 = w_post[min(tid, n_wp - 1)], wp1 = w_post[min(tid + 256, n_wp - 1)];
    const float bp = b_post ? b_post[min(tid, M - 1)] : 0.f;
    float e_b[4], e_y[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = min(e_nb + q, n_out - 1);
        const float bb = b_end ? b_end[n] : 0.f, yy = ya ? ya[n] : 0.f;      // (wave-uniform conditions)
        e_b[q] = e_nb + q < n_out ? bb : 0.f;
        e_y[q] = e_nb + q < n_out ? yy : 0.f;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // groups of this wave in batches of TL_BATCH: all loads of a batch in flight, then its MFMAs
    constexpr int TL_BATCH = 6;
    for (int c0 = wave; c0 < nc8; c0 += 4 * TL_BATCH) {
        float4 a[TL_BATCH], w[TL_BATCH];
#pragma unroll
        for (int i = 0; i < TL_BATCH; ++i) {
            const int c = min(c0 + 4 * i, nc8 - 1);
            // C % 8 == 4: the upper lane half of the last group has no channels: its weights are zero, the row's last four
            // channels are read in their place (finite values x 0)
            a[i] = *reinterpret_cast<const float4 *>(xr + min(8 * c, C - 4 - 4 * lk));
            w[i] = wv[c * 64];
        }
#pragma unroll
        for (int i = 0; i < TL_BATCH; ++i) {
            if (c0 + 4 * i < nc8) {                          // wave-uniform
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, w[i].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, w[i].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, w[i].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, w[i].w, acc, 0, 0, 0);
            }
        }
    }
    // the post-net's weights (read behind the second barrier): staged here, behind the K loop's requests
    if (tid < n_wp) wp[tid] = wp0;
    if (tid + 256 < n_wp) wp[tid + 256] = wp1;
    if (tid < M) wp[32 * 16 + tid] = bp;
    // lane (column lrow) holds rows (r & 3) + 8 (r >> 2) + 4 lk of this wave's partial 32 x 32 result
    float *tw = tile + wave * 32 * 33;
#pragma unroll
    for (int r = 0; r < 16; ++r) tw[((r & 3) + 8 * (r >> 2) + 4 * lk) * 33 + lrow] = acc[r];
    __syncthreads();
    // sum of the four partial results + bias -> y (stage output) and tile 0
    {
        const int rr = e_rr, nb = e_nb;
        float *yb = y + (long long)b * y_bstride + (long long)(m0 + rr) * n_out;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = nb + q;
            const int o = rr * 33 + n;
            const float v = ((tile[o] + tile[32 * 33 + o]) + (tile[2 * 32 * 33 + o] + tile[3 * 32 * 33 + o])) + e_b[q] + e_y[q];
            tile[o] = v;
            if (n < n_out && m0 + rr < rows) yb[n] = v;
        }
    }
    __syncthreads();
    // post-net: thread (row, pair of outputs)
    {
        const int rr = tid & 31, mg = tid >> 5;
        if (m0 + rr < rows) {
            float *sb = sub + (long long)b * sub_bstride + (long long)(m0 + rr) * M;
            const float *tr = tile + rr * 33;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int m = 2 * mg + q;
                if (m < M) {
                    float s = 0.f;
                    for (int n = 0; n < n_out; ++n) s = fmaf(tr[n], wp[n * M + m], s);
                    sb[m] = s + wp[32 * 16 + m];
                }
            }
        }
    }
}

// returns false if the shapes do not fit (caller falls back to two generic convolutions)
bool launch_wn_tail(const float *skip, long long skip_bstride, const int *n_frames, int rows_per_frame, int max_rows,
                    int batch, int C, const float *w_end_packed, const float *b_end, int n_out, const float *w_post,
                    const float *b_post, int M, const float *y_acc, float *y, long long y_bstride, float *sub,
                    long long sub_bstride, hipStream_t stream) {
    if (n_out > 32 || M > 16 || C % 4 != 0 || skip_bstride % 4 != 0 || (uintptr_t)skip % 16 != 0 ||
        (uintptr_t)w_end_packed % 16 != 0)
        return false;
    if (max_rows <= 0 || batch <= 0) return true;
    hipLaunchKernelGGL(wn_tail_kernel, dim3((max_rows + 31) / 32, batch), dim3(256), 0, stream, skip, skip_bstride,
                       n_frames, rows_per_frame, max_rows, C, w_end_packed, b_end, n_out, w_post, b_post, M, y_acc, y,
                       y_bstride, sub, sub_bstride);
    return true;
}

}  // namespace mbx
