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


// ---------------------------------------------------------------------------------------------------------------------
// fft_size = 2048: ONE WAVE PER FRAME.  The block-per-frame kernel above is bound by its vector instructions (1 785 per wave
// and frame, PMC round 2): five radix-4 passes per transform, each with its index arithmetic, twiddle loads and a block
// barrier.  Here a lane holds 16 of the 1 024 complex points in registers and a transform is three passes -- radix 16,
// radix 16, radix 4 (Stockham autosort: butterfly j of a pass with radix R over N/R butterflies, k = j mod ns, reads
// in[j + m N/R] w^(m k), w = exp(-+ 2 pi i / (ns R)), writes out[(j - k) R + k + m ns]; ns = 1, 16, 256) -- with two exchanges
// through an 8.5 KB LDS buffer of the wave's own: no block barrier (a wave's LDS instructions execute in order), the 16-point
// butterflies have compile-time twiddles, and the layouts are padded so that every exchange is bank-conflict free:
//   exchange 1: element e at e + (e >> 4)      (a lane writes out[16 j .. 16 j + 15]: lane stride 17 complex)
//   exchange 2: element e at e + 16 (e >> 8)   (a lane writes out[256 (j >> 4) + (j & 15) + 16 m]: the four 16-lane groups 32 banks apart)
// After the last pass lane l holds the elements l + 64 i (i = 0 .. 15) -- the input pattern of the first pass, so the
// spectrum product feeds the inverse transform from registers; only the mirrored element (nc - k) of the real-transform
// split / merge steps travels through LDS.  Block = 4 waves = 4 consecutive frames of one item.
typedef float cf __attribute__((ext_vector_type(2)));

__device__ __forceinline__ cf cf_mul(cf a, cf b) { return cf{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
template <bool INV>
__device__ __forceinline__ cf cf_rot(cf a) { return INV ? cf{-a.y, a.x} : cf{a.y, -a.x}; }       // a * (-i) (forward) | a * (+i)
template <bool INV>
__device__ __forceinline__ void cf_radix4(cf &a0, cf &a1, cf &a2, cf &a3) {
    const cf v0 = a0 + a2, v1 = a0 - a2, v2 = a1 + a3, v3 = cf_rot<INV>(a1 - a3);
    a0 = v0 + v2;
    a1 = v1 + v3;
    a2 = v0 - v2;
    a3 = v1 - v3;
}
// 16-point transform in place; X[n] ends up in a[4 (n & 3) + (n >> 2)]
template <bool INV>
__device__ __forceinline__ void cf_radix16(cf (&a)[16]) {
#pragma unroll
    for (int m0 = 0; m0 < 4; ++m0) cf_radix4<INV>(a[m0], a[m0 + 4], a[m0 + 8], a[m0 + 12]);       // a[m0 + 4 k0] = B[m0][k0]
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
    // a[m0 + 4 k0] *= w16^(m0 k0), w16 = exp(-+ 2 pi i / 16)
    auto tw = [&](cf &v, float cr, float ci) { v = cf_mul(v, cf{cr, INV ? ci : -ci}); };
    tw(a[1 + 4], c1, s1);                                      // w^1
    tw(a[2 + 4], h, h);                                        // w^2
    tw(a[3 + 4], s1, c1);                                      // w^3
    tw(a[1 + 8], h, h);                                        // w^2
    a[2 + 8] = cf_rot<INV>(a[2 + 8]);                          // w^4 = -+ i
    tw(a[3 + 8], -h, h);                                       // w^6
    tw(a[1 + 12], s1, c1);                                     // w^3
    tw(a[2 + 12], -h, h);                                      // w^6
    tw(a[3 + 12], -c1, -s1);                                   // w^9
#pragma unroll
    for (int k0 = 0; k0 < 4; ++k0) cf_radix4<INV>(a[4 * k0], a[4 * k0 + 1], a[4 * k0 + 2], a[4 * k0 + 3]);   // a[4 k0 + k1] = X[k0 + 4 k1]
}

// tanh, sin and cos of the spectral envelope without the library functions' special-case paths (the argument-reduction
// path of sincosf for huge arguments alone took the kernel to 256 registers): what matters for H = exp(R tanh(Re S) + j Im S)
// is the ABSOLUTE error of the exponent.  tanh(x) = 1 - 2 / (exp(2 x) + 1): absolute error ~1e-7 everywhere (exp2 overflow
// gives 1 exactly).  sin / cos: x = n pi/2 + r by a three-term Cody-Waite reduction (exact products for |n| < 2^11, i.e.
// |x| < 3 200, far above a log spectrum's phase; larger arguments lose accuracy gradually), minimax polynomials on
// [-pi/4, pi/4] (errors < 1e-7), quadrant by n.
__device__ __forceinline__ float sw_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);          // exp(2 x)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ void sw_sincos(float x, float &sn, float &cs) {
    const float n = rintf(x * 0.6366197723675814f);
    float r = fmaf(n, -1.5703125f, x);                                        // pi/2 = 1.5703125 + 4.837512969970703e-4 + 7.549789954e-8
    r = fmaf(n, -4.837512969970703125e-4f, r);
    r = fmaf(n, -7.54978995489188e-8f, r);
    const float r2 = r * r;
    const float ps = r * fmaf(r2, fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), 1.0f);
    const float pc = fmaf(r2, fmaf(r2, fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), -0.5f), 1.0f);
    const int q = (int)n;
    const float s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
    sn = (q & 2) ? -s0 : s0;
    cs = ((q + 1) & 2) ? -c0 : c0;
}

constexpr int SW_NC = 1024;                        // complex points of a transform
constexpr int SW_BUF = SW_NC + 64;                 // exchange buffer of a wave (padded layouts)
#define SW_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// a[m] = element lane + 64 m in, a[i] = element lane + 64 i out (un-normalised); t256[n] = exp(-2 pi i n / 256),
// tw[n] = exp(-2 pi i n / 2048), n < 1024
template <bool INV>
__device__ __forceinline__ void fft1024_wave(cf (&a)[16], cf *buf, const cf *t256, const cf *tw, int lane) {
    cf_radix16<INV>(a);
#pragma unroll
    for (int n = 0; n < 16; ++n) buf[17 * lane + n] = a[4 * (n & 3) + (n >> 2)];
    SW_SYNC();
#pragma unroll
    for (int m = 0; m < 16; ++m) a[m] = buf[lane + (lane >> 4) + 68 * m];
    SW_SYNC();
    const int k = lane & 15;
#pragma unroll
    for (int m = 1; m < 16; ++m) {
        cf w = t256[m * k];
        if (INV) w.y = -w.y;
        a[m] = cf_mul(a[m], w);
    }
    cf_radix16<INV>(a);
#pragma unroll
    for (int n = 0; n < 16; ++n) buf[272 * (lane >> 4) + k + 16 * n] = a[4 * (n & 3) + (n >> 2)];
    SW_SYNC();
    cf r[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int jj = lane + 64 * q;
        cf u[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) u[m] = buf[jj + 272 * m];
#pragma unroll
        for (int m = 1; m < 4; ++m) {
            const int idx = 2 * m * jj;                         // on the 2048-circle; < 2048
            cf w = tw[idx & (SW_NC - 1)];
            if (idx >= SW_NC) w = -w;
            if (INV) w.y = -w.y;
            u[m] = cf_mul(u[m], w);
        }
        cf_radix4<INV>(u[0], u[1], u[2], u[3]);
#pragma unroll
        for (int m = 0; m < 4; ++m) r[q + 4 * m] = u[m];        // element jj + 256 m = lane + 64 (q + 4 m)
    }
    SW_SYNC();
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = r[i];
}

// bins k = lane + 64 i of the real transform from Z (a[i] = Z[lane + 64 i]); the Nyquist bin (real) in ny (lane 0)
__device__ __forceinline__ void real_bins_wave(cf (&a)[16], float &ny, cf *buf, const cf *tw, int lane) {
#pragma unroll
    for (int i = 0; i < 16; ++i) buf[lane + 64 * i] = a[i];
    SW_SYNC();
    const cf z0 = a[0];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = lane + 64 * i;
        const cf zk = a[i], zr = buf[(SW_NC - k) & (SW_NC - 1)];
        const cf xe = cf{0.5f * (zk.x + zr.x), 0.5f * (zk.y - zr.y)};
        const cf xo = cf{0.5f * (zk.y + zr.y), -0.5f * (zk.x - zr.x)};
        a[i] = xe + cf_mul(tw[k], xo);
    }
    ny = z0.x - z0.y;                                           // (meaningful on lane 0: Z[0] = a[0] there)
    SW_SYNC();
}

// NMX / NMC: register slots (of 128 samples each) that can hold non-zero samples of the excitation frame (win <= 128 NMX) /
// cepstrum (n_ceps <= 128 NMC): the others are compile-time zeros, which prunes the first pass of the two forward
// transforms (canonical model: 10 and 2 of 16).  (Two or four frames per wave, to amortise the table setup of a block over
// more work: 135 against 128 us at 16 x 10 s -- dropped.)
template <int NMX, int NMC>
__global__ __launch_bounds__(256, 2) void stft_filter_wave_kernel(StftConsts c, const float *exc, long long exc_bstride,
                                                                   const float *ceps, long long ceps_bstride, const int *index,
                                                                   const float *f0, long long f0_bstride, int *index_out,
                                                                   const int *n_frames, int max_frames, float *frames) {
    __shared__ __attribute__((aligned(16))) cf s_tw[SW_NC];
    __shared__ __attribute__((aligned(16))) cf s_t256[256];
    __shared__ __attribute__((aligned(16))) cf s_buf[4][SW_BUF];
    __shared__ __attribute__((aligned(16))) float s_hann[128 * NMX], s_inv_win[128 * NMX];      // zero behind win
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int T = item_rows(n_frames, b, 1, max_frames);
    if (blockIdx.x * 4 >= T) return;
    {
        // the block's tables: all requests first (clamped addresses), then the stores -- written as plain copy loops every
        // iteration waits for its own load (nine serial round trips in front of a block's first transform)
        constexpr int NW = (128 * NMX + 255) / 256;
        const cf *twg = reinterpret_cast<const cf *>(c.twiddle);
        cf twv[SW_NC / 256];
        float hv[NW], iv[NW];
#pragma unroll
        for (int k = 0; k < SW_NC / 256; ++k) twv[k] = twg[tid + 256 * k];
        // exp(-2 pi i n / 256) = entry 8 n of the 2048-circle (second half circle: negated)
        cf w = twg[(8 * tid) & (SW_NC - 1)];
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int i = min(tid + 256 * k, c.win - 1);
            hv[k] = c.hann[i];
            iv[k] = c.inv_win[i];
        }
#pragma unroll
        for (int k = 0; k < SW_NC / 256; ++k) s_tw[tid + 256 * k] = twv[k];
        if (8 * tid >= SW_NC) w = -w;
        s_t256[tid] = w;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int i = tid + 256 * k;
            if (i < 128 * NMX) {
                s_hann[i] = i < c.win ? hv[k] : 0.f;
                s_inv_win[i] = i < c.win ? iv[k] * (1.0f / (float)SW_NC) : 0.f;   // with the inverse transform's 1 / nc (a power of two)
            }
        }
    }
    __syncthreads();                                            // (no block-wide synchronisation below this line)
    cf *buf = s_buf[wave];
    const cf *tw = s_tw, *t256 = s_t256;
    const float *eb = exc + (long long)b * exc_bstride;
    const int n_sig = T * c.hop;
    const int first_ceps = c.preserve_energy ? 0 : 1;
    const float one = 1.0f;

    const int t = blockIdx.x * 4 + wave;
    if (t < T) {
        // lifter row of this frame: given, or selected here from the F0 contour
        const float *lw = nullptr;
        if (c.n_ceps_windows > 0 && index) lw = c.ceps_windows + (long long)index[(long long)b * max_frames + t] * c.n_ceps;
        else if (c.n_ceps_windows > 0 && f0) {
            const int idx = ceps_index_of_frame(c, f0 + (long long)b * f0_bstride, T, t, lane);
            if (lane == 0 && index_out) index_out[(long long)b * max_frames + t] = idx;
            lw = c.ceps_windows + (long long)idx * c.n_ceps;
        }

        // ---- loads of both forward transforms, from clamped addresses and selected afterwards: all requests of a lane are in
        // flight at once.  Excitation frame: zero padded signal (win/2 in front, win/2+hop+1 behind) x Hann; cepstrum:
        // [0, c1*l1 .. c_{n_ceps-1}*l_{n_ceps-1}, 0 ...] (coefficient 0 is the noise gain's business unless the filters
        // preserve the energy, reference custom_pulsed_generator.py:817-826)
        cf a[16], cz[NMC];
        {
            const float *cb = ceps + (long long)b * ceps_bstride + (long long)t * c.n_ceps;
            const float *lwp = lw ? lw : &one;                  // (no lifter: times 1.0f, exact)
#pragma unroll
            for (int m = 0; m < NMC; ++m) {
                const int i0 = 2 * (lane + 64 * m);
                float v[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    // (times a 0 / 1 mask instead of a select: the compiler would sink the loads into a branch per value and
                    // wait for each on its own)
                    const int i = i0 + q, ic = min(i, c.n_ceps - 1);
                    v[q] = (cb[ic] * lwp[lw ? ic : 0]) * ((i >= first_ceps && i < c.n_ceps) ? 1.0f : 0.0f);
                }
                cz[m] = cf{v[0], v[1]};
            }
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                a[m] = cf{0.f, 0.f};
                if (m < NMX) {
                    const int i0 = 2 * (lane + 64 * m);
                    float v[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int i = i0 + q;
                        const int s = t * c.hop + i - c.win / 2;
                        v[q] = eb[min(max(s, 0), n_sig - 1)] * ((s >= 0 && s < n_sig) ? s_hann[i] : 0.0f);
                    }
                    a[m] = cf{v[0], v[1]};
                }
            }
        }
        // ---- 1. X = real transform of the excitation frame
        fft1024_wave<false>(a, buf, t256, tw, lane);
        float x_ny;
        real_bins_wave(a, x_ny, buf, tw, lane);
        cf xk[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) xk[i] = a[i];

        // ---- 2. S = real transform of the liftered cepstrum (log spectrum)
#pragma unroll
        for (int m = 0; m < 16; ++m) a[m] = m < NMC ? cz[m < NMC ? m : 0] : cf{0.f, 0.f};
        fft1024_wave<false>(a, buf, t256, tw, lane);
        float s_ny;
        real_bins_wave(a, s_ny, buf, tw, lane);

        // ---- 3. H = exp(R * tanh(Re S) + j Im S) ; Y = X * H (/ rms_k |H| when the filters preserve the energy)
        auto envelope = [&](cf sv) {
            const float re = (c.max_log_range > 0.f) ? c.max_log_range * sw_tanh(sv.x) : sv.x;
            const float mag = __builtin_amdgcn_exp2f(re * 1.4426950408889634f);
            float sn, cs;
            sw_sincos(sv.y, sn, cs);
            return cf{mag * cs, mag * sn};
        };
        float h2 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            a[i] = envelope(a[i]);
            h2 += a[i].x * a[i].x + a[i].y * a[i].y;
        }
        const cf h_ny = envelope(cf{s_ny, 0.f});                // lane 0: the Nyquist bin
        float inv_gain = 1.f;
        if (c.preserve_energy) {
            // filter_gain = sqrt(mean_k |H_k|^2) over the fft_size / 2 + 1 bins of the frame (reference :838-839)
            if (lane == 0) h2 += h_ny.x * h_ny.x + h_ny.y * h_ny.y;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) h2 += __shfl_xor(h2, o);
            inv_gain = 1.f / sqrtf(h2 / (float)(SW_NC + 1));
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) xk[i] = cf_mul(xk[i], a[i] * inv_gain);
        const float y_ny = x_ny * h_ny.x * inv_gain;            // Re Y[nc] (a real inverse transform ignores Im Y[0], Im Y[nc])

        // ---- 4. merge: Z[k] = Ye[k] + i Yo[k], Ye = (Y[k] + conj Y[nc-k]) / 2, Yo = (Y[k] - conj Y[nc-k]) / 2 * conj W^k
#pragma unroll
        for (int i = 0; i < 16; ++i) buf[lane + 64 * i] = xk[i];
        SW_SYNC();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = lane + 64 * i;
            cf yk = xk[i], yr = buf[(SW_NC - k) & (SW_NC - 1)];
            if (k == 0) {
                yk = cf{xk[0].x, 0.f};
                yr = cf{y_ny, 0.f};
            }
            const cf ye = cf{0.5f * (yk.x + yr.x), 0.5f * (yk.y - yr.y)};
            const cf d = cf{0.5f * (yk.x - yr.x), 0.5f * (yk.y + yr.y)};
            const cf w = tw[k];
            const cf yo = cf_mul(d, cf{w.x, -w.y});
            a[i] = cf{ye.x - yo.y, ye.y + yo.x};
        }
        SW_SYNC();
        fft1024_wave<true>(a, buf, t256, tw, lane);

        // ---- 5. first win samples x synthesis window
        float *fb = frames + ((long long)b * max_frames + t) * c.win;
#pragma unroll
        for (int i = 0; i < NMX; ++i) {
            const int i0 = 2 * (lane + 64 * i);                     // win is even
            const float2 iw = *reinterpret_cast<const float2 *>(s_inv_win + i0);
            if (i0 < c.win) *reinterpret_cast<float2 *>(fb + i0) = make_float2(a[i].x * iw.x, a[i].y * iw.y);
        }
    }
}

// index: lifter rows (B, max_frames) or null; with null and f0 != null the rows are selected inside the kernel from
// the F0 contour (B, max_frames * pulse_per_frame) and written to index_out (may be null)
void launch_stft_filter(const StftConsts &c, const float *exc, long long exc_bstride, const float *ceps,
                        long long ceps_bstride, const int *index, const float *f0, long long f0_bstride,
                        int *index_out, const int *n_frames, int max_frames, int batch, float *frames,
                        hipStream_t stream) {
    if (max_frames <= 0 || batch <= 0) return;
    if (c.fft_size == 2 * SW_NC && c.win <= c.fft_size && c.win % 2 == 0) {
        const dim3 grid((max_frames + 3) / 4, batch);
        if (c.win <= 1280 && c.n_ceps <= 256)
            hipLaunchKernelGGL((stft_filter_wave_kernel<10, 2>), grid, dim3(256), 0, stream, c, exc, exc_bstride, ceps, ceps_bstride, index,
                               f0, f0_bstride, index_out, n_frames, max_frames, frames);
        else
            hipLaunchKernelGGL((stft_filter_wave_kernel<16, 16>), grid, dim3(256), 0, stream, c, exc, exc_bstride, ceps, ceps_bstride, index,
                               f0, f0_bstride, index_out, n_frames, max_frames, frames);
        return;
    }
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
            if (c.win <= 4 * c.hop) {
                // at most four frames cover a sample: all four requests first (clamped frame, masked afterwards), added in frame
                // order -- the loop below waits for every load on its own (a trip count the compiler does not know)
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = min(t_lo + j, t_hi);
                    v[j] = fb[(long long)t * c.win + (p - t * c.hop)];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += t_lo + j <= t_hi ? v[j] : 0.f;
            } else {
                for (int t = t_lo; t <= t_hi; ++t) acc += fb[(long long)t * c.win + (p - t * c.hop)];
            }
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
