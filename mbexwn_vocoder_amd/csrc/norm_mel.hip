// Optional RMS normalisation of the mel input and de-normalisation of the output.
//
// restates NormMelComponents.normalize_inputs_by_rms   reference MBExWN_NVoc/vocoder/model/wavegen_1d.py:638-769
// as PaNWaveNet.infer applies it (wavegen_1d.py:493-495, 506-507), smoothing variant
// (normalize_rms_num_smooth_iters > 0):
//   rms[t]   = sqrt(sum_c (exp(mell[t,c]) * inv_enorm[c])^2 / rms_norm_fact)            (:689)
//              or, normalize_use_pinv: sqrt(sum_k (sum_c exp(mell[t,c]) pinv[c,k] / win_norm)^2 / rms_norm_fact)  (:684-685)
//              floored at 1/max_norm_fact (:690-691), compressed by pow(., exp) (:692-693)
//   per smoothing iteration (:697-726):
//     ext    = [rms[0], rms[0], rms[0..T-1], rms[T-1], rms[T-1]]                         (T + 4 frames)
//     gain[n] = sum_k ext[k] sw[n + cut - k hop] / max(eps, sum_k sw[n + cut - k hop])   overlap-add of the smoothing
//               window sw (length S), cut = S/2 + 2 hop - win/2
//     rms[t] = sum_j gain[t hop + j] gwin[j]                                            (strided VALID conv with the
//               unit-sum analysis window)
//   mell'[t,c] = mel_amp_scale * log(exp(mell[t,c]) / max(eps, rms[t]) * lin_amp_scale + lin_amp_off)   (:731-736)
//   audio[n]  *= max(gain[win/2 + n], eps)    with the gain of the LAST iteration        (:741-743, wavegen_1d.py:506-507)
// Mel-rate work plus one multiply per output sample: four tiny kernels, no host round trip.  Every length is the
// item's own (n_frames[b]); float32 with expf / logf / powf (not the fast intrinsics: the log-mel values feed the
// whole network).
#include "mbx_kernels.h"

namespace mbx {

constexpr float NM_EPS = 1e-7f;   // tf.keras.backend.epsilon()

__global__ __launch_bounds__(64) void nm_rms_kernel(NormMelConsts c, const float *mell, long long mel_bstride,
                                                    const int *n_frames, int max_frames, float *rms) {
    const int b = blockIdx.y, t = blockIdx.x;
    const int T = item_rows(n_frames, b, 1, max_frames);
    if (t >= T) return;
    const float *row = mell + (long long)b * mel_bstride + (long long)t * c.mel_channels;
    float s = 0.f;
    if (c.pinv) {
        // normalize_use_pinv (:684-685): spectrum = exp(mell) . pinv(mel filters)^T / win_norm, energy over all bins; a lane
        // walks the bins k = lane, lane + 64, ... (rows of the table are read coalesced)
        __shared__ float lin[256];
        for (int ch = threadIdx.x; ch < c.mel_channels; ch += 64) lin[ch] = expf(row[ch]);
        __syncthreads();
        const float inv_norm = 1.0f / c.win_norm;
        for (int k = threadIdx.x; k < c.n_bins; k += 64) {
            float v = 0.f;
            for (int ch = 0; ch < c.mel_channels; ++ch) v = fmaf(lin[ch], c.pinv[(long long)ch * c.n_bins + k], v);
            v *= inv_norm;
            s += v * v;
        }
    } else {
        for (int ch = threadIdx.x; ch < c.mel_channels; ch += 64) {
            const float v = expf(row[ch]) * c.inv_enorm[ch];
            s += v * v;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) {
        float r = sqrtf(s / c.rms_norm_fact);
        if (c.rms_floor > 0.f) r = fmaxf(r, c.rms_floor);
        if (c.use_compressor) r = powf(r, c.compressor_exp);
        rms[(long long)b * max_frames + t] = r;
    }
}

// gain[n] of one smoothing pass over the frame values r (T of them), n counted behind the cut
__device__ __forceinline__ float nm_gain_at(const NormMelConsts &c, const float *r, int T, int n) {
    const int pos = n + c.cut;
    int k0 = (pos - c.smooth_win + c.hop) / c.hop;       // ceil((pos - S + 1) / hop) for pos - S + 1 >= 0
    if (pos - c.smooth_win + 1 <= 0) k0 = 0;
    const int k1 = min(pos / c.hop, T + 3);
    float num = 0.f, den = 0.f;
    for (int k = k0; k <= k1; ++k) {
        const float wv = c.smooth_win_table[pos - k * c.hop];
        num += r[min(max(k - 2, 0), T - 1)] * wv;
        den += wv;
    }
    return num / fmaxf(NM_EPS, den);
}

__global__ __launch_bounds__(256) void nm_smooth_kernel(NormMelConsts c, const float *r_in, const int *n_frames,
                                                        int max_frames, float *r_out) {
    __shared__ float red[4];
    const int b = blockIdx.y, t = blockIdx.x;
    const int T = item_rows(n_frames, b, 1, max_frames);
    if (t >= T) return;
    const float *r = r_in + (long long)b * max_frames;
    float acc = 0.f;
    for (int j = threadIdx.x; j < c.win; j += 256) acc += nm_gain_at(c, r, T, t * c.hop + j) * c.gwin[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) r_out[(long long)b * max_frames + t] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void nm_scale_kernel(NormMelConsts c, const float *mell, long long mel_bstride, const float *rms,
                                const int *n_frames, int max_frames, float *out) {
    const int b = blockIdx.y;
    const int T = item_rows(n_frames, b, 1, max_frames);
    const long long total = (long long)T * c.mel_channels;
    const float *mb = mell + (long long)b * mel_bstride;
    float *ob = out + (long long)b * mel_bstride;
    const float *r = rms + (long long)b * max_frames;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i / c.mel_channels);
        const float m = expf(mb[i]) / fmaxf(NM_EPS, r[t]) * c.lin_amp_scale;
        ob[i] = c.mel_amp_scale * (c.use_max_limit ? logf(fmaxf(m, c.lin_amp_off)) : logf(m + c.lin_amp_off));
    }
}

__global__ void nm_apply_gain_kernel(NormMelConsts c, const float *r_last, const int *n_frames, int max_frames,
                                     float *audio, long long audio_bstride, int overwrite) {
    const int b = blockIdx.y;
    const int T = item_rows(n_frames, b, 1, max_frames);
    const long long total = (long long)T * c.hop;
    const float *r = r_last + (long long)b * max_frames;
    float *ab = audio + (long long)b * audio_bstride;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x)
    {
        const float g = fmaxf(nm_gain_at(c, r, T, c.win / 2 + (int)i), NM_EPS);
        ab[i] = overwrite ? g : ab[i] * g;
    }
}

// mell (B, Tmax, mel) -> mell' (same layout); rms_a / rms_b: scratch (B, Tmax) each.  Returns the buffer that holds the
// frame values the output gain is built from (input of the last smoothing pass), for launch_norm_mel_gain.
const float *launch_norm_mel(const NormMelConsts &c, const float *mell, long long mel_bstride, const int *n_frames,
                             int max_frames, int batch, float *rms_a, float *rms_b, float *mell_out,
                             hipStream_t stream) {
    if (max_frames <= 0 || batch <= 0) return rms_a;
    hipLaunchKernelGGL(nm_rms_kernel, dim3(max_frames, batch), dim3(64), 0, stream, c, mell, mel_bstride, n_frames,
                       max_frames, rms_a);
    float *cur = rms_a, *nxt = rms_b;
    for (int it = 0; it < c.iters; ++it) {
        hipLaunchKernelGGL(nm_smooth_kernel, dim3(max_frames, batch), dim3(256), 0, stream, c, cur, n_frames, max_frames,
                           nxt);
        float *t = cur;
        cur = nxt;
        nxt = t;
    }
    const long long total = (long long)max_frames * c.mel_channels;
    const int blocks = (int)min((total + 255) / 256, (long long)1024);
    hipLaunchKernelGGL(nm_scale_kernel, dim3(blocks, batch), dim3(256), 0, stream, c, mell, mel_bstride, cur, n_frames,
                       max_frames, mell_out);
    return nxt;   // input of the last pass
}

void launch_norm_mel_gain(const NormMelConsts &c, const float *r_last, const int *n_frames, int max_frames, int batch,
                          float *audio, long long audio_bstride, bool overwrite, hipStream_t stream) {
    if (max_frames <= 0 || batch <= 0) return;
    const long long total = (long long)max_frames * c.hop;
    const int blocks = (int)min((total + 255) / 256, (long long)2048);
    hipLaunchKernelGGL(nm_apply_gain_kernel, dim3(blocks, batch), dim3(256), 0, stream, c, r_last, n_frames, max_frames,
                       audio, audio_bstride, overwrite ? 1 : 0);
}

}  // namespace mbx
