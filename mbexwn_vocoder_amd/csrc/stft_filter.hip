// STFT-domain vocal-tract filter: excitation STFT x exp(cepstral envelope) -> inverse STFT.
//
// restates (float32, complex64 in the frequency domain)
//   _get_cepstral_windows      reference MBExWN_NVoc/vocoder/model/custom_pulsed_generator.py:507-525
//   generate_specenv           reference custom_pulsed_generator.py:793-855
//   pad + tf.signal.stft       reference custom_pulsed_generator.py:681-694
//   multiply + inverse_stft    reference custom_pulsed_generator.py:715-724
// tf.signal.stft / inverse_stft / inverse_stft_window_fn / hann_window are TensorFlow (third party,
// not in the reference tree); their published semantics are what is implemented here:
//   frames of `win` samples every `hop`, periodic Hann, zero-extended to fft_size, rFFT;
//   irFFT, first `win` samples x (hann / sum_of_4_shifted_hann^2), overlap-add.
// The reference keeps only frames 0..T-1 of T+2 and slices [win/2, win/2 + T*hop) of the overlap-add,
// so the head and the tail of the output are under-normalised; that taper is reproduced because the
// same frames are summed.
//
// One 256-thread block per (item, frame): three radix-2 Stockham FFTs of fft_size points in LDS
// (twiddles in LDS).  The three FFTs are ~0.1 % of the forward FLOPs; the stage is latency-type.
#include "mbx_kernels.h"

namespace mbx {

constexpr int FFT_THREADS = 256;

// Stockham autosort radix-2; n complex points ping-pong between a and b, returns the buffer holding the result
template <bool INVERSE>
__device__ float2 *fft_lds(float2 *a, float2 *b, const float2 *tw, int n, int tid) {
    float2 *in = a, *out = b;
    const int half = n >> 1;
    for (int ns = 1; ns < n; ns <<= 1) {
        const int tstep = half / ns;
        for (int j = tid; j < half; j += FFT_THREADS) {
            const int k = j & (ns - 1);
            float2 w = tw[k * tstep];
            if (INVERSE) w.y = -w.y;
            const float2 v0 = in[j];
            const float2 v1 = in[j + half];
            float2 t;
            t.x = v1.x * w.x - v1.y * w.y;
            t.y = v1.x * w.y + v1.y * w.x;
            const int j0 = ((j - k) << 1) + k;
            out[j0] = make_float2(v0.x + t.x, v0.y + t.y);
            out[j0 + ns] = make_float2(v0.x - t.x, v0.y - t.y);
        }
        __syncthreads();
        float2 *tmp = in;
        in = out;
        out = tmp;
    }
    return in;
}

constexpr int MAX_BINS_PER_THREAD = 5;   // fft_size/2 + 1 <= 4*256 + 1

__global__ __launch_bounds__(FFT_THREADS) void stft_filter_kernel(StftConsts c, const float *exc,
                                                                  long long exc_bstride, const float *ceps,
                                                                  long long ceps_bstride, const int *index,
                                                                  const int *n_frames, int max_frames,
                                                                  float *frames) {
    extern __shared__ float2 smem2[];
    const int n = c.fft_size, half = n >> 1;
    float2 *bufa = smem2;
    float2 *bufb = smem2 + n;
    float2 *tw = smem2 + 2 * n;            // n/2 twiddles
    const int b = blockIdx.y, t = blockIdx.x;
    const int T = n_frames ? n_frames[b] : max_frames;
    if (t >= T) return;
    const int tid = threadIdx.x;
    for (int i = tid; i < half; i += FFT_THREADS) tw[i] = reinterpret_cast<const float2 *>(c.twiddle)[i];

    // ---- 1. windowed excitation frame (zero padded signal: win/2 in front, win/2+hop+1 behind)
    const float *eb = exc + (long long)b * exc_bstride;
    const int n_sig = T * c.hop;
    for (int i = tid; i < n; i += FFT_THREADS) {
        float v = 0.f;
        if (i < c.win) {
            const int s = t * c.hop + i - c.win / 2;
            if (s >= 0 && s < n_sig) v = eb[s] * c.hann[i];
        }
        bufa[i] = make_float2(v, 0.f);
    }
    __syncthreads();
    float2 *res = fft_lds<false>(bufa, bufb, tw, n, tid);
    float2 xk[MAX_BINS_PER_THREAD];
#pragma unroll
    for (int i = 0; i < MAX_BINS_PER_THREAD; ++i) {
        const int k = tid + i * FFT_THREADS;
        xk[i] = (k <= half) ? res[k] : make_float2(0.f, 0.f);
    }
    __syncthreads();

    // ---- 2. one-sided cepstrum [0, c1*l1 .. c_{n_ceps-1}*l_{n_ceps-1}, 0 ...] -> log spectrum
    const float *cb = ceps + (long long)b * ceps_bstride + (long long)t * c.n_ceps;
    const float *lw = nullptr;
    if (c.n_ceps_windows > 0 && index) lw = c.ceps_windows + (long long)index[(long long)b * max_frames + t] * c.n_ceps;
    for (int i = tid; i < n; i += FFT_THREADS) {
        float v = 0.f;
        if (i >= 1 && i < c.n_ceps) v = lw ? cb[i] * lw[i] : cb[i];
        bufa[i] = make_float2(v, 0.f);
    }
    __syncthreads();
    res = fft_lds<false>(bufa, bufb, tw, n, tid);
    float2 *dst = (res == bufa) ? bufb : bufa;
    // ---- 3. H = exp(R * tanh(Re S) + j Im S) ; Y = X * H ; Hermitian extension for the inverse transform
#pragma unroll
    for (int i = 0; i < MAX_BINS_PER_THREAD; ++i) {
        const int k = tid + i * FFT_THREADS;
        if (k <= half) {
            const float2 s = res[k];
            const float re = (c.max_log_range > 0.f) ? c.max_log_range * tanhf(s.x) : s.x;
            const float mag = expf(re);
            float sn, cs;
            sincosf(s.y, &sn, &cs);
            const float2 h = make_float2(mag * cs, mag * sn);
            float2 yv;
            yv.x = xk[i].x * h.x - xk[i].y * h.y;
            yv.y = xk[i].x * h.y + xk[i].y * h.x;
            if (k == 0 || k == half) {
                dst[k] = make_float2(yv.x, 0.f);          // a real inverse transform ignores these imaginary parts
            } else {
                dst[k] = yv;
                dst[n - k] = make_float2(yv.x, -yv.y);
            }
        }
    }
    __syncthreads();
    float2 *other = (dst == bufa) ? bufb : bufa;
    res = fft_lds<true>(dst, other, tw, n, tid);
    // ---- 4. first win samples x synthesis window
    float *fb = frames + ((long long)b * max_frames + t) * c.win;
    const float scale = 1.0f / (float)n;
    for (int i = tid; i < c.win; i += FFT_THREADS) fb[i] = (res[i].x * scale) * c.inv_win[i];
}

void launch_stft_filter(const StftConsts &c, const float *exc, long long exc_bstride, const float *ceps,
                        long long ceps_bstride, const int *index, const int *n_frames, int max_frames, int batch,
                        float *frames, hipStream_t stream) {
    if (max_frames <= 0 || batch <= 0) return;
    const size_t smem = sizeof(float2) * (size_t)(2 * c.fft_size + c.fft_size / 2);
    hipLaunchKernelGGL(stft_filter_kernel, dim3(max_frames, batch), dim3(FFT_THREADS), smem, stream, c, exc,
                       exc_bstride, ceps, ceps_bstride, index, n_frames, max_frames, frames);
}

// Cepstral lifter selection from the smoothed F0 contour; one wavefront per (item, frame).
__global__ __launch_bounds__(64) void ceps_index_kernel(StftConsts c, const float *f0, long long f0_bstride,
                                                        const int *n_frames, int max_frames, int *index) {
    const int b = blockIdx.y, t = blockIdx.x;
    const int T = n_frames ? n_frames[b] : max_frames;
    if (t >= T) return;
    const int n = T * c.pulse_per_frame;
    const int taps = 2 * c.hop + 1, halfk = taps / 2;
    const float *fb = f0 + (long long)b * f0_bstride;
    float acc = 0.f;
    for (int j = threadIdx.x; j < taps; j += 64) {
        int s = t * c.pulse_per_frame + j - halfk;          // edge-replicated contour
        s = min(max(s, 0), n - 1);
        acc += fb[s] * c.f0_smooth[j];
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (threadIdx.x == 0) {
        const float lo = c.ceps_log10f0[0], hi = c.ceps_log10f0[c.n_ceps_windows - 1];
        float lg = (float)(1.0 / 2.302585092994046) * logf(acc);
        lg = fminf(fmaxf(lg, lo), hi);
        const float ratio = (lg - lo) / (hi - lo);
        index[(long long)b * max_frames + t] = (int)rintf(ratio * (float)(c.n_ceps_windows - 1));
    }
}

void launch_ceps_index(const StftConsts &c, const float *f0, long long f0_bstride, const int *n_frames,
                       int max_frames, int batch, int *index, hipStream_t stream) {
    if (max_frames <= 0 || batch <= 0) return;
    hipLaunchKernelGGL(ceps_index_kernel, dim3(max_frames, batch), dim3(64), 0, stream, c, f0, f0_bstride, n_frames,
                       max_frames, index);
}

// overlap-add in frame order + slice [win/2, win/2 + T*hop)
__global__ void overlap_add_kernel(StftConsts c, const float *frames, const int *n_frames, int max_frames,
                                   float *audio, long long audio_bstride) {
    const int b = blockIdx.y;
    const int T = n_frames ? n_frames[b] : max_frames;
    const int n_valid = T * c.hop, n_all = max_frames * c.hop;
    const float *fb = frames + (long long)b * max_frames * c.win;
    float *ab = audio + (long long)b * audio_bstride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_all; i += gridDim.x * blockDim.x) {
        float acc = 0.f;
        if (i < n_valid) {
            const int p = i + c.win / 2;
            const int t_hi = min(T - 1, p / c.hop);
            const int t_lo = p < c.win ? 0 : (p - c.win) / c.hop + 1;
            for (int t = t_lo; t <= t_hi; ++t) acc += fb[(long long)t * c.win + (p - t * c.hop)];
        }
        ab[i] = acc;
    }
}

void launch_overlap_add(const StftConsts &c, const float *frames, const int *n_frames, int max_frames, int batch,
                        float *audio, long long audio_bstride, hipStream_t stream) {
    if (max_frames <= 0 || batch <= 0) return;
    const int n_all = max_frames * c.hop;
    hipLaunchKernelGGL(overlap_add_kernel, dim3(min((n_all + 255) / 256, 2048), batch), dim3(256), 0, stream, c,
                       frames, n_frames, max_frames, audio, audio_bstride);
}

}  // namespace mbx
