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
#include <algorithm>
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

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the same stage with every wave owning its rows.  The kernel above splits the channels over the four waves of a
// block: four partial 32 x 32 results meet through LDS behind a barrier, and the post-net is a scalar loop -- 76 us of its
// 129 us at 16 x 10 s remain when every global load is taken out (round-3 ablation), twice its 34 us of MFMA time.  Here:
// block = 4 waves x 16 rows; a wave contracts ALL channels of its 16 rows (v_mfma_f32_16x16x4_f32, two 16-column tiles),
// the activations straight from global memory (lane (row r16, kq) loads the 16 bytes of channels 16 j + 4 kq .. + 3: 64
// contiguous bytes per row and instruction), the packed weight image of wn_tail_kernel -- read as [16-channel block j][kq]
// [column][4 steps] -- staged once per block by LDS-DMA; no cross-wave sum.  The post-net is a second, 8-step MFMA chain on
// the wave's own 16 x 32 result (transposed through a 2 KB LDS tile of the wave's own).  One kernel at every launch size:
// results do not depend on the batch.
__device__ __forceinline__ void tl_lds_dma16_s(const float *sbase, unsigned voff_bytes, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_addr), "v"(voff_bytes), "s"(sbase)
                 : "memory", "m0");
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int T2_ROWS = 64;
constexpr int T2_YT = 16 * 33;                       // a wave's transposition tile (16 rows, pitch 33)

template <int NJ>   // blocks of 16 channels a wave loads and multiplies: ceil(8 ceil(C / 8) / 16) <= NJ
__global__ __launch_bounds__(256) void wn_tail2_kernel(const float *skip, long long skip_bstride, const int *n_frames,
                                                       int rows_per_frame, int max_rows, int C, const float *w_end_packed,
                                                       const float *b_end, int n_out, const float *w_post,
                                                       const float *b_post, int M, const float *y_acc, float *y,
                                                       long long y_bstride, float *sub, long long sub_bstride) {
    typedef __attribute__((address_space(3))) float lds_float;
    extern __shared__ __attribute__((aligned(16))) float t2lds[];
    const int nc8 = (C + 7) / 8;                      // 8-channel groups of the image = kilobytes to stage
    float *wimg = t2lds;                              // nc8 * 256 floats; after the K loop (behind a barrier) the same bytes hold:
    float *ytile = t2lds;                             // 4 waves x T2_YT
    float *wp = ytile + 4 * T2_YT;                    // 32 x 16 post weights (zero padded), then 16 post biases
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float *)t2lds);
    const int b = blockIdx.y;
    const int rows = item_rows(n_frames, b, rows_per_frame, max_rows);
    const int m0 = blockIdx.x * T2_ROWS;
    if (m0 >= rows) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;

    // ---- the weight image: kilobyte c of the image -> kilobyte c of LDS, dealt round-robin over the waves
    for (int c = wave; c < nc8; c += 4) tl_lds_dma16_s(w_end_packed + c * 256, 16u * (unsigned)lane, lds_base + 1024u * (unsigned)c);
    const int nj = (8 * nc8 + 15) / 16;               // blocks of 16 channels
    const int row_a = min(m0 + 16 * wave + r16, rows - 1);
    const float *xr = skip + (long long)b * skip_bstride + (long long)row_a * C;
    // what the epilogue adds to the lane's outputs: column n = r16 (+ 16 t), rows 4 kq + v of the wave's tile
    float e_add[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = r16 + 16 * t;
        const float bb = (b_end && n < n_out) ? b_end[n] : 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = min(m0 + 16 * wave + 4 * kq + v, rows - 1);
            const float yy = (y_acc && n < n_out) ? y_acc[(long long)b * y_bstride + (long long)row * n_out + n] : 0.f;
            e_add[t][v] = bb + yy;
        }
    }
    f32x4 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[t][v] = 0.f;

    // every activation of the wave's rows requested at once (NJ x 16 bytes a lane); the image's DMA is older than all of them,
    // so "at most NJ loads outstanding" = the image has landed
    float4 a[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int ch = 16 * min(j, nj - 1) + 4 * kq;
        // pieces behind the row's last channel meet zero weights (the image is zero padded to a multiple of 8 channels;
        // pieces behind the image are masked below): read the row's last four channels in their place (finite values)
        a[j] = *reinterpret_cast<const float4 *>(xr + min(ch, C - 4));
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NJ) : "memory");
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (j < nj) {                                  // (block-uniform)
            const bool in_img = 16 * j + 4 * kq < 8 * nc8;
            const float4 av = in_img ? a[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            const int g = min(4 * j + kq, 2 * nc8 - 1);              // [group c = g >> 1][lane half g & 1] of the image
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float4 bv = *reinterpret_cast<const float4 *>(wimg + (g * 32 + 16 * t + r16) * 4);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[t], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                   // the image has been read by every wave: its bytes are free
    // ---- post-net weights (n_out x M -> 32 x 16, zero padded) and biases
    for (int i = tid; i < 32 * 16; i += 256) {
        const int n = i >> 4, m = i & 15;
        wp[i] = (n < n_out && m < M) ? w_post[n * M + m] : 0.f;
    }
    if (tid < 16) wp[512 + tid] = (b_post && tid < M) ? b_post[tid] : 0.f;
    // ---- y = acc + bias + earlier layers (stage output), and the wave's 16 x 32 result row-major in its LDS tile
    float *yt = ytile + wave * T2_YT;
    float *yb = y + (long long)b * y_bstride;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = r16 + 16 * t;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int lr = 4 * kq + v;
            const int row = m0 + 16 * wave + lr;
            const float val = acc[t][v] + e_add[t][v];
            yt[lr * 33 + n] = n < n_out ? val : 0.f;
            if (n < n_out && row < rows) yb[(long long)row * n_out + n] = val;
        }
    }
    // ---- post-net: (16 x 32) x (32 x 16) on the matrix cores; lane (row r16, kq) supplies y[r16][4 s + kq], (kq, column r16)
    // supplies Wp[4 s + kq][r16]
    __syncthreads();                                   // the post-net's weights are in place
    f32x4 sacc;
#pragma unroll
    for (int v = 0; v < 4; ++v) sacc[v] = 0.f;
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8)
        sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(yt[r16 * 33 + 4 * s8 + kq], wp[(4 * s8 + kq) * 16 + r16], sacc, 0, 0, 0);
    if (r16 < M) {
        const float bp = wp[512 + r16];
        float *sb = sub + (long long)b * sub_bstride;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = m0 + 16 * wave + 4 * kq + v;
            if (row < rows) sb[(long long)row * M + r16] = sacc[v] + bp;
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
    // rows owned by waves (wn_tail2_kernel): up to 352 channels (22 blocks of 16 in a lane's registers)
    const int nc8 = (C + 7) / 8, nj = (8 * nc8 + 15) / 16;
    const size_t lds2 = std::max((size_t)nc8 * 256, (size_t)(4 * T2_YT + 32 * 16 + 16)) * sizeof(float);
    if (nj <= 22 && C >= 16) {
        // the smallest instantiation that holds the row: a lane issues NJ loads and keeps 4 NJ registers whatever C is
        // (same arithmetic in the same order for every NJ >= nj; ADVICE round 5: C = 64 used to issue 20 loads for 4)
        auto kern = nj <= 4 ? wn_tail2_kernel<4> : nj <= 8 ? wn_tail2_kernel<8> : nj <= 12 ? wn_tail2_kernel<12> :
                    nj <= 20 ? wn_tail2_kernel<20> : wn_tail2_kernel<22>;
        hipLaunchKernelGGL(kern, dim3((max_rows + T2_ROWS - 1) / T2_ROWS, batch), dim3(256), lds2, stream, skip, skip_bstride,
                           n_frames, rows_per_frame, max_rows, C, w_end_packed, b_end, n_out, w_post, b_post, M, y_acc, y,
                           y_bstride, sub, sub_bstride);
        return true;
    }
    hipLaunchKernelGGL(wn_tail_kernel, dim3((max_rows + 31) / 32, batch), dim3(256), 0, stream, skip, skip_bstride,
                       n_frames, rows_per_frame, max_rows, C, w_end_packed, b_end, n_out, w_post, b_post, M, y_acc, y,
                       y_bstride, sub, sub_bstride);
    return true;
}

}  // namespace mbx
