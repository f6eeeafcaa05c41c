// Audio -> log-mel analysis: the step in front of the hot path.
//
// restates, for the configuration the reference's CLI uses (no band limiting, do_post=False):
//   compute_mel_spectrogram_internal   reference MBExWN_NVoc/vocoder/model/preprocess.py:417-572
//   calc_stft (magnitude, centred)     reference MBExWN_NVoc/sig_proc/spec/stft.py:14-96
//     frame t = win * x_reflect[t*hop - win/2 .. + win), zero-extended to fft_size, |rFFT|
//   mel = |X| . basis^T, log(max(mel, eps))
// The window (symmetric Hann, reference Mwindows.py:60-67) and the Slaney mel basis are tables of the host module
// (analysis.py), which also is the float64-transform oracle this kernel is tested against.
//
// One 256-thread block per (item, frame): windowed frame -> real FFT (complex Stockham FFT of fft_size/2 points in LDS,
// fft_lds.h) -> magnitudes in LDS -> one wavefront per mel channel sums its triangle (bins [lo, hi] of the dense basis
// row) -> log.  Bandwidth-type: reads hop samples and writes mel_channels floats per frame.
#include "fft_lds.h"
#include "mbx_kernels.h"

namespace mbx {

__global__ __launch_bounds__(FFT_THREADS) void mel_analysis_kernel(MelAnalysisArgs p) {
    extern __shared__ float2 smem[];
    const int nc = p.fft_size / 2;
    float2 *a = smem, *bq = smem + nc, *tw = smem + 2 * nc;
    float *mag = reinterpret_cast<float *>(smem + 3 * nc);            // nc + 1 magnitudes
    const int t = blockIdx.x, b = blockIdx.y;
    // item length from the device array, clamped to the item's row: a wrong entry must not address outside the buffer
    const int n = p.n_samples ? min(max(p.n_samples[b], 0), p.max_samples) : p.max_samples;
    const int frames = n / p.hop + 1;
    if (t >= frames) return;
    const int tid = threadIdx.x;
    const float *xb = p.audio + (long long)b * p.audio_bstride;
    for (int i = tid; i < nc; i += FFT_THREADS) tw[i] = reinterpret_cast<const float2 *>(p.twiddle)[i];
    // frame samples j = 2m, 2m+1 of the reflect-padded signal (numpy "reflect": no repeated edge sample)
    for (int m = tid; m < nc; m += FFT_THREADS) {
        float v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = 2 * m + q;
            float val = 0.f;
            if (j < p.win) {
                int s = t * p.hop + j - p.win / 2;
                if (s < 0) s = -s;
                if (s >= n) s = 2 * (n - 1) - s;
                s = min(max(s, 0), n - 1);
                if (n >= 1) val = p.window[j] * xb[s];      // an empty item is one frame of silence: log(eps) rows
            }
            v[q] = val;
        }
        a[m] = make_float2(v[0], v[1]);
    }
    __syncthreads();
    const float2 *z = fft_lds<false>(a, bq, tw, nc, tid);
    for (int k = tid; k <= nc; k += FFT_THREADS) {
        const float2 x = real_bin(z, tw, k, nc);
        mag[k] = sqrtf(x.x * x.x + x.y * x.y);
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    float *ob = p.out + ((long long)b * p.max_frames + t) * p.n_mels;
    for (int m = wave; m < p.n_mels; m += FFT_THREADS / 64) {
        const float *row = p.basis + (long long)m * (nc + 1);
        float acc = 0.f;
        for (int k = p.bin_lo[m] + lane; k <= p.bin_hi[m]; k += 64) acc = fmaf(mag[k], row[k], acc);
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) ob[m] = logf(fmaxf(acc, p.eps));
    }
}

bool launch_mel_analysis(const MelAnalysisArgs &a, hipStream_t stream) {
    const bool ok = a.fft_size >= 8 && a.fft_size <= 2048 && (a.fft_size & (a.fft_size - 1)) == 0 && a.win >= 2 &&
                    a.win <= a.fft_size && a.hop >= 1 && a.n_mels >= 1 && a.max_samples >= a.win / 2 + 1 && a.audio &&
                    a.window && a.twiddle && a.basis && a.bin_lo && a.bin_hi && a.out && a.max_frames >= a.max_samples / a.hop + 1;
    if (!ok) return false;
    if (a.batch <= 0) return true;
    const int nc = a.fft_size / 2;
    const size_t smem = sizeof(float2) * (size_t)(3 * nc) + sizeof(float) * (size_t)(nc + 1);
    hipLaunchKernelGGL(mel_analysis_kernel, dim3(a.max_samples / a.hop + 1, a.batch), dim3(FFT_THREADS), smem, stream, a);
    return true;
}

}  // namespace mbx
