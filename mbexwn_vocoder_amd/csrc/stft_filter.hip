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
// One 256-thread block per (item, frame).  The three transforms are real, so each runs as a complex transform of
// fft_size/2 points (even samples in the real part, odd samples in the imaginary part, one butterfly pass to
// separate / merge the halves): Stockham autosort in LDS, radix-4 passes (one butterfly per thread and pass for
// fft_size = 2048) plus one radix-2 pass when log2(fft_size/2) is odd; twiddles exp(-2 pi i m / fft_size) in LDS.
#include "fft_lds.h"
#include "mbx_kernels.h"

namespace mbx {

// Cepstral lifter selection from the smoothed F0 contour, by one wavefront (lane = threadIdx.x & 63); every lane
// returns the index.  reference custom_pulsed_generator.py:507-525 (601-tap Bartlett smoother on the edge-replicated
// contour, log10, clip, nearest of the n_ceps_windows rows)
__device__ __forceinline__ int ceps_index_of_frame(const StftConsts &c, const float *fb, int T, int t, int lane) {
    const int n = T * c.pulse_per_frame;
    const int taps = 2 * c.hop + 1, halfk = taps / 2;
    float acc = 0.f;
    for (int j = lane; j < taps; j += 64) {
        int s = t * c.pulse_per_frame + j - halfk;          // edge-replicated contour
        s = min(max(s, 0), n - 1);
        acc += fb[s] * c.f0_smooth[j];
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    const float lo = c.ceps_log10f0[0], hi = c.ceps_log10f0[c.n_ceps_windows - 1];
    float lg = (float)(1.0 / 2.302585092994046) * logf(acc);
    lg = fminf(fmaxf(lg, lo), hi);
    const float ratio = (lg - lo) / (hi - lo);
    return (int)rintf(ratio * (float)(c.n_ceps_windows - 1));
}

constexpr int MAX_BINS_PER_THREAD = 5;   // fft_size/2 + 1 <= 4*256 + 1

__global__ __launch_bounds__(FFT_THREADS) void stft_filter_kernel(StftConsts c, const float *exc,
                                                                  long long exc_bstride, const float *ceps,
                                                                  long long ceps_bstride, const int *index,
                                                                  const float *f0, long long f0_bstride, int *index_out,
                                                                  const int *n_frames, int max_frames,
                                                                  float *frames) {
    extern __shared__ float2 smem2[];
    const int n = c.fft_size, nc = n >> 1;
    float2 *bufa = smem2;
    float2 *bufb = smem2 + nc;
    float2 *tw = smem2 + 2 * nc;           // nc twiddles exp(-2 pi i m / n)
    const int b = blockIdx.y, t = blockIdx.x;
    const int T = item_rows(n_frames, b, 1, max_frames);
    if (t >= T) return;
    const int tid = threadIdx.x;
    for (int i = tid; i < nc; i += FFT_THREADS) tw[i] = reinterpret_cast<const float2 *>(c.twiddle)[i];
    // lifter row of this frame: given, or selected here from the F0 contour by the first wavefront while the others
    // stage the frame
    __shared__ int s_index;
    if (c.n_ceps_windows > 0 && !index && f0 && tid < 64) {
        const int idx = ceps_index_of_frame(c, f0 + (long long)b * f0_bstride, T, t, tid);
        if (tid == 0) {
            s_index = idx;
            if (index_out) index_out[(long long)b * max_frames + t] = idx;
        }
    }

    // ---- 1. windowed excitation frame (zero padded signal: win/2 in front, win/2+hop+1 behind)
    const float *eb = exc + (long long)b * exc_bstride;
    const int n_sig = T * c.hop;
    for (int m = tid; m < nc; m += FFT_THREADS) {
        float v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = 2 * m + q;
            v[q] = 0.f;
            if (i < c.win) {
                const int s = t * c.hop + i - c.win / 2;
                if (s >= 0 && s < n_sig) v[q] = eb[s] * c.hann[i];
            }
        }
        bufa[m] = make_float2(v[0], v[1]);
    }
    __syncthreads();
    float2 *res = fft_lds<false>(bufa, bufb, tw, nc, tid);
    float2 xk[MAX_BINS_PER_THREAD];
#pragma unroll
    for (int i = 0; i < MAX_BINS_PER_THREAD; ++i) {
        const int k = tid + i * FFT_THREADS;
        xk[i] = (k <= nc) ? real_bin(res, tw, k, nc) : make_float2(0.f, 0.f);
    }
    __syncthreads();

    // ---- 2. one-sided cepstrum [0, c1*l1 .. c_{n_ceps-1}*l_{n_ceps-1}, 0 ...] -> log spectrum (coefficient 0 is the
    // noise gain's business unless the filters preserve the energy, reference custom_pulsed_generator.py:817-826)
    const int first_ceps = c.preserve_energy ? 0 : 1;
    const float *cb = ceps + (long long)b * ceps_bstride + (long long)t * c.n_ceps;
    const float *lw = nullptr;
    if (c.n_ceps_windows > 0 && index) lw = c.ceps_windows + (long long)index[(long long)b * max_frames + t] * c.n_ceps;
    else if (c.n_ceps_windows > 0 && f0) lw = c.ceps_windows + (long long)s_index * c.n_ceps;   // written before the first barrier
    for (int m = tid; m < nc; m += FFT_THREADS) {
        float v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = 2 * m + q;
            v[q] = 0.f;
            if (i >= first_ceps && i < c.n_ceps) v[q] = lw ? cb[i] * lw[i] : cb[i];
        }
        bufa[m] = make_float2(v[0], v[1]);
    }
    __syncthreads();
    res = fft_lds<false>(bufa, bufb, tw, nc, tid);
    // ---- 3. H = exp(R * tanh(Re S) + j Im S) ; Y = X * H
    float2 hk[MAX_BINS_PER_THREAD];
    float h2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_BINS_PER_THREAD; ++i) {
        const int k = tid + i * FFT_THREADS;
        hk[i] = make_float2(0.f, 0.f);
        if (k <= nc) {
            const float2 s = real_bin(res, tw, k, nc);
            const float re = (c.max_log_range > 0.f) ? c.max_log_range * tanhf(s.x) : s.x;
            const float mag = expf(re);
            float sn, cs;
            sincosf(s.y, &sn, &cs);
            hk[i] = make_float2(mag * cs, mag * sn);
            if (c.preserve_energy) h2 += hk[i].x * hk[i].x + hk[i].y * hk[i].y;
        }
    }
    float inv_gain = 1.f;
    if (c.preserve_energy) {
        // filter_gain = sqrt(mean_k |H_k|^2) over the fft_size / 2 + 1 bins of the frame (reference :838-839); H /= gain
        __shared__ float s_h2[FFT_THREADS / 64];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) h2 += __shfl_xor(h2, o);
        if ((tid & 63) == 0) s_h2[tid >> 6] = h2;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < FFT_THREADS / 64; ++q) tot += s_h2[q];
        inv_gain = 1.f / sqrtf(tot / (float)(nc + 1));
    }
#pragma unroll
    for (int i = 0; i < MAX_BINS_PER_THREAD; ++i) {
        const int k = tid + i * FFT_THREADS;
        if (k <= nc) xk[i] = cmul(xk[i], make_float2(hk[i].x * inv_gain, hk[i].y * inv_gain));
    }
    __syncthreads();
    // Y[0..nc) to LDS; a real inverse transform ignores Im Y[0] and Im Y[nc], so Re Y[nc] travels in Im of entry 0
    float2 *ybuf = res, *zbuf = (res == bufa) ? bufb : bufa;
#pragma unroll
    for (int i = 0; i < MAX_BINS_PER_THREAD; ++i) {
        const int k = tid + i * FFT_THREADS;
        if (k == 0) ybuf[0].x = xk[i].x;
        else if (k < nc) ybuf[k] = xk[i];
        else if (k == nc) ybuf[0].y = xk[i].x;
    }
    __syncthreads();
    // ---- 4. merge: Z[k] = Ye[k] + i Yo[k], Ye = (Y[k] + conj Y[nc-k]) / 2, Yo = (Y[k] - conj Y[nc-k]) / 2 * conj W^k
    for (int k = tid; k < nc; k += FFT_THREADS) {
        const float2 y0 = ybuf[0];
        const float2 yk = k == 0 ? make_float2(y0.x, 0.f) : ybuf[k];
        const float2 yr = k == 0 ? make_float2(y0.y, 0.f) : ybuf[nc - k];
        const float2 ye = make_float2(0.5f * (yk.x + yr.x), 0.5f * (yk.y - yr.y));
        const float2 d = make_float2(0.5f * (yk.x - yr.x), 0.5f * (yk.y + yr.y));
        const float2 w = tw[k];
        const float2 yo = cmul(d, make_float2(w.x, -w.y));
        zbuf[k] = make_float2(ye.x - yo.y, ye.y + yo.x);
    }
    __syncthreads();
    res = fft_lds<true>(zbuf, ybuf, tw, nc, tid);
    // ---- 5. first win samples x synthesis window
    float *fb = frames + ((long long)b * max_frames + t) * c.win;
    const float scale = 1.0f / (float)nc;
    for (int i = tid; i < c.win; i += FFT_THREADS) {
        const float2 z = res[i >> 1];
        fb[i] = (((i & 1) ? z.y : z.x) * scale) * c.inv_win[i];
    }
}

// index: lifter rows (B, max_frames) or null; with null and f0 != null the rows are selected inside the kernel from
// the F0 contour (B, max_frames * pulse_per_frame) and written to index_out (may be null)
void launch_stft_filter(const StftConsts &c, const float *exc, long long exc_bstride, const float *ceps,
                        long long ceps_bstride, const int *index, const float *f0, long long f0_bstride,
                        int *index_out, const int *n_frames, int max_frames, int batch, float *frames,
                        hipStream_t stream) {
    if (max_frames <= 0 || batch <= 0) return;
    const size_t smem = sizeof(float2) * (size_t)(3 * (c.fft_size / 2));
    hipLaunchKernelGGL(stft_filter_kernel, dim3(max_frames, batch), dim3(FFT_THREADS), smem, stream, c, exc,
                       exc_bstride, ceps, ceps_bstride, index, f0, f0_bstride, index_out, n_frames, max_frames, frames);
}

// overlap-add in frame order + slice [win/2, win/2 + T*hop)
__global__ void overlap_add_kernel(StftConsts c, const float *frames, const int *n_frames, int max_frames,
                                   int out_frames, float *audio, long long audio_bstride) {
    const int b = blockIdx.y;
    const int T = item_rows(n_frames, b, 1, max_frames);
    const int n_valid = T * c.hop, n_all = out_frames * c.hop;
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

void launch_overlap_add(const StftConsts &c, const float *frames, const int *n_frames, int max_frames, int out_frames,
                        int batch, float *audio, long long audio_bstride, hipStream_t stream) {
    if (max_frames <= 0 || out_frames <= 0 || batch <= 0) return;
    const int n_all = out_frames * c.hop;
    hipLaunchKernelGGL(overlap_add_kernel, dim3(min((n_all + 255) / 256, 2048), batch), dim3(256), 0, stream, c,
                       frames, n_frames, max_frames, out_frames, audio, audio_bstride);
}

}  // namespace mbx
