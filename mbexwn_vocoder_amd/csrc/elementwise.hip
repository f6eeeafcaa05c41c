// Bandwidth-type stages: linear interpolation, activations, WaveNet input fold + start conv.
#include "mbx_kernels.h"

namespace mbx {

// ActivationLayer (reference MBExWN_NVoc/vocoder/model/custom_AE_layers.py:21-109)
__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case 1: return 0.5f + 0.5f * x / (1.0f + fabsf(x));      // soft_sigmoid  :91-99
        case 2: return tanhf(x);
        case 3: return 1.0f / (1.0f + expf(-x));
        case 4: return x / (1.0f + fabsf(x));
        case 5: return x / (1.0f + sqrtf(fabsf(x)));             // soft_sqrt     :80-89
        case 6: return expf(x);
        case 7: return fmaxf(x, 0.f);
        default: return x;
    }
}

// TF2C_LinInterpLayer(num_pad_end=1, drop_last=True)
// (reference MBExWN_NVoc/vocoder/model/tf2_components/layers/support_layers.py:19-27,99-121):
//   out[t*U + u, c] = x[t, c] * (U-u)/U + x[min(t+1, T-1), c] * u/U
// one thread per output element, channel fastest (coalesced on both sides for C >= 64; for C == 1
// consecutive threads walk u, reading two broadcast inputs).
__global__ void lin_interp_kernel(const float *x, long long x_bstride, const int *n_frames, int rows_per_frame,
                                  int max_rows, int channels, int up, const float *w0, const float *w1, int act,
                                  float scale, float offset, float *y, long long y_bstride) {
    const int b = blockIdx.y;
    const int rows = item_rows(n_frames, b, rows_per_frame, max_rows);
    const long long total = (long long)rows * up * channels;
    const float *xb = x + (long long)b * x_bstride;
    float *yb = y + (long long)b * y_bstride;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % channels);
        const long long ro = i / channels;
        const int t = (int)(ro / up), u = (int)(ro - (long long)t * up);
        const int tn = min(t + 1, rows - 1);
        float v = xb[(long long)t * channels + c] * w0[u] + xb[(long long)tn * channels + c] * w1[u];
        v = apply_act(v, act) * scale + offset;
        yb[i] = v;
    }
}

void launch_lin_interp(const float *x, long long x_bstride, const int *n_frames, int rows_per_frame, int max_rows,
                       int batch, int channels, int up, const float *w0, const float *w1, int act, float scale,
                       float offset, float *y, long long y_bstride, hipStream_t stream) {
    if (max_rows <= 0 || batch <= 0) return;
    const long long total = (long long)max_rows * up * channels;
    const int blocks = (int)min((total + 255) / 256, (long long)2048);
    hipLaunchKernelGGL(lin_interp_kernel, dim3(blocks, batch), dim3(256), 0, stream, x, x_bstride, n_frames,
                       rows_per_frame, max_rows, channels, up, w0, w1, act, scale, offset, y, y_bstride);
}

// Head of the F0-net in float64 (mbx_config.f0_accumulate, the default): final 1x1 convolution cin -> 1, linear interpolation
// to the pulse rate, final activation and the affine map onto [f_min, f_max] -- reference custom_pulsed_generator.py:126-146
// (final layer, missing up-sampling factor, final activation), :773-791 (generate_f0), custom_AE_layers.py:91-99 (soft_sigmoid).
// The contour feeds the phase integrator: every step here is float64 on the hidden layer and the float32 constants (bias, the
// interpolator's float32 weight vectors), rounded to float32 ONCE.  Block = 16 rows of the hidden layer (+ the row behind
// them for the interpolation): four threads per row sum every fourth channel each in a fixed order.
__device__ __forceinline__ double apply_act64(double x, int act) {
    switch (act) {
        case 1: return 0.5 + 0.5 * x / (1.0 + fabs(x));
        case 2: return tanh(x);
        case 3: return 1.0 / (1.0 + exp(-x));
        case 4: return x / (1.0 + fabs(x));
        case 5: return x / (1.0 + sqrt(fabs(x)));
        case 6: return exp(x);
        case 7: return fmax(x, 0.0);
        default: return x;
    }
}

constexpr int F0H_ROWS = 16;     // (64: four blocks for a 3 s utterance, 13.5 us of float64 divisions on four CUs)

template <typename XT, typename WT>
__global__ __launch_bounds__(256) void f0_head_kernel(const XT *x, long long x_bstride, int cin, const int *n_frames,
                                                      int rows_per_frame, int max_rows, const WT *w, const float *bias,
                                                      int up, const float *w0, const float *w1, int act, float scale,
                                                      float offset, float *y, long long y_bstride) {
    __shared__ double xs[F0H_ROWS + 1];
    const int b = blockIdx.y;
    const int rows = item_rows(n_frames, b, rows_per_frame, max_rows);
    const int m0 = blockIdx.x * F0H_ROWS;
    if (m0 >= rows) return;
    const XT *xb = x + (long long)b * x_bstride;
    const int tid = threadIdx.x, part = tid & 3;
    auto dot = [&](int row) {
        const XT *xr = xb + (long long)min(row, rows - 1) * cin;
        double s = 0.0;
        for (int c = part; c < cin; c += 4) s = fma((double)xr[c], (double)w[c], s);
        // (s0 + s1) + (s2 + s3) over the row's four threads
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        return s + (bias ? (double)bias[0] : 0.0);
    };
    // rows m0 .. m0 + F0H_ROWS (the last one for the interpolation of the row in front of it): four threads each
    if (tid < 4 * (F0H_ROWS + 1) + 60) {              // (whole waves: the shuffles need their partners)
        const double mine = dot(m0 + (tid >> 2));
        if (part == 0 && (tid >> 2) <= F0H_ROWS) xs[tid >> 2] = mine;
    }
    __syncthreads();
    const int n_here = min(F0H_ROWS, rows - m0);
    float *yb = y + (long long)b * y_bstride;
    for (int i = tid; i < n_here * up; i += 256) {
        const int t = i / up, u = i - t * up;
        const int tn = min(m0 + t + 1, rows - 1) - m0;
        double v = xs[t] * (double)w0[u] + xs[tn] * (double)w1[u];
        v = apply_act64(v, act) * (double)scale + (double)offset;
        yb[(long long)(m0 + t) * up + u] = (float)v;
    }
}

void launch_f0_head(const float *x32, const double *x64, long long x_bstride, int cin, const int *n_frames, int rows_per_frame,
                    int max_rows, int batch, const float *w32, const double *w64, const float *bias, int up, const float *w0,
                    const float *w1, int act, float scale, float offset, float *y, long long y_bstride, hipStream_t stream) {
    if (max_rows <= 0 || batch <= 0) return;
    const dim3 grid((max_rows + F0H_ROWS - 1) / F0H_ROWS, batch);
    if (x64 && w64)
        hipLaunchKernelGGL((f0_head_kernel<double, double>), grid, dim3(256), 0, stream, x64, x_bstride, cin, n_frames,
                           rows_per_frame, max_rows, w64, bias, up, w0, w1, act, scale, offset, y, y_bstride);
    else if (x64)
        hipLaunchKernelGGL((f0_head_kernel<double, float>), grid, dim3(256), 0, stream, x64, x_bstride, cin, n_frames,
                           rows_per_frame, max_rows, w32, bias, up, w0, w1, act, scale, offset, y, y_bstride);
    else if (w64)
        hipLaunchKernelGGL((f0_head_kernel<float, double>), grid, dim3(256), 0, stream, x32, x_bstride, cin, n_frames,
                           rows_per_frame, max_rows, w64, bias, up, w0, w1, act, scale, offset, y, y_bstride);
    else
        hipLaunchKernelGGL((f0_head_kernel<float, float>), grid, dim3(256), 0, stream, x32, x_bstride, cin, n_frames,
                           rows_per_frame, max_rows, w32, bias, up, w0, w1, act, scale, offset, y, y_bstride);
}

__global__ void activation_kernel(const float *x, long long x_bstride, const int *n_frames, int rows_per_frame,
                                  int max_rows, int channels, int act, float scale, float offset, float *y,
                                  long long y_bstride) {
    const int b = blockIdx.y;
    const int rows = item_rows(n_frames, b, rows_per_frame, max_rows);
    const long long total = (long long)rows * channels;
    const float *xb = x + (long long)b * x_bstride;
    float *yb = y + (long long)b * y_bstride;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x)
        yb[i] = apply_act(xb[i], act) * scale + offset;
}

void launch_activation(const float *x, long long x_bstride, const int *n_frames, int rows_per_frame, int max_rows,
                       int batch, int channels, int act, float scale, float offset, float *y, long long y_bstride,
                       hipStream_t stream) {
    if (max_rows <= 0 || batch <= 0) return;
    const long long total = (long long)max_rows * channels;
    const int blocks = (int)min((total + 255) / 256, (long long)2048);
    hipLaunchKernelGGL(activation_kernel, dim3(blocks, batch), dim3(256), 0, stream, x, x_bstride, n_frames,
                       rows_per_frame, max_rows, channels, act, scale, offset, y, y_bstride);
}

// Keras PReLU(shared_axes=[1]) / LeakyReLU (reference custom_pulsed_generator.py:247-253)
__global__ void prelu_kernel(float *x, long long x_bstride, const int *n_frames, int rows_per_frame, int max_rows,
                             int channels, const float *alpha, float leaky) {
    const int b = blockIdx.y;
    const int rows = item_rows(n_frames, b, rows_per_frame, max_rows);
    const long long total = (long long)rows * channels;
    float *xb = x + (long long)b * x_bstride;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const float v = xb[i];
        const float s = alpha ? alpha[i % channels] : leaky;
        xb[i] = v > 0.f ? v : s * v;
    }
}

void launch_prelu(float *x, long long x_bstride, const int *n_frames, int rows_per_frame, int max_rows, int batch,
                  int channels, const float *alpha, float leaky, hipStream_t stream) {
    if (max_rows <= 0 || batch <= 0) return;
    const long long total = (long long)max_rows * channels;
    const int blocks = (int)min((total + 255) / 256, (long long)2048);
    hipLaunchKernelGGL(prelu_kernel, dim3(blocks, batch), dim3(256), 0, stream, x, x_bstride, n_frames,
                       rows_per_frame, max_rows, channels, alpha, leaky);
}

// Sub-band gains (ps_use_stft: false).  The reference interpolates the per-frame gains with factor hop_size (one value per
// audio sample, edge frame repeated) and multiplies the sub-band rows -- steps_per_frame per frame -- with the FIRST rows
// of that tensor (custom_pulsed_generator.py:453,670,916-917): row r takes the gain at frame r / hop, phase r % hop.
__global__ void subband_gain_kernel(float *sub, long long sub_bstride, const float *log_gain, long long gain_bstride,
                                    const int *n_frames, int max_frames, int rows_per_frame, int M, int hop,
                                    const float *w0, const float *w1, int remove_mean) {
    const int b = blockIdx.y;
    const int T = item_rows(n_frames, b, 1, max_frames);
    const long long total = (long long)T * rows_per_frame * M;
    const float *gb = log_gain + (long long)b * gain_bstride;
    float *sb = sub + (long long)b * sub_bstride;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / M), m = (int)(i - (long long)r * M);
        const int f0 = r / hop, j = r - f0 * hop;
        const int f1 = min(f0 + 1, T - 1);
        float m0 = 0.f, m1 = 0.f;
        if (remove_mean) {       // channel_log_gain -= reduce_mean(channel_log_gain, axis=-1) (:868-871)
            for (int k = 0; k < M; ++k) {
                m0 += gb[(long long)f0 * M + k];
                m1 += gb[(long long)f1 * M + k];
            }
            m0 /= (float)M;
            m1 /= (float)M;
        }
        const float g0 = expf(gb[(long long)f0 * M + m] - m0), g1 = expf(gb[(long long)f1 * M + m] - m1);
        sb[i] *= g0 * w0[j] + g1 * w1[j];
    }
}

void launch_subband_gain(float *sub, long long sub_bstride, const float *log_gain, long long gain_bstride, const int *n_frames,
                         int max_frames, int rows_per_frame, int batch, int M, int hop, const float *w0, const float *w1,
                         int remove_mean, hipStream_t stream) {
    if (max_frames <= 0 || batch <= 0) return;
    const long long total = (long long)max_frames * rows_per_frame * M;
    const int blocks = (int)min((total + 255) / 256, (long long)2048);
    hipLaunchKernelGGL(subband_gain_kernel, dim3(blocks, batch), dim3(256), 0, stream, sub, sub_bstride, log_gain, gain_bstride,
                       n_frames, max_frames, rows_per_frame, M, hop, w0, w1, remove_mean);
}

// PQMF analysis bank in front of the WaveNet (pulse_channels_use_pqmf, reference custom_pulsed_generator.py:892-895,
// tf_preprocess.py:188-200): zero-pad taps/2 on both sides, cross-correlate with the K analysis filters, keep every K-th
// sample.  One thread per output value; the bank (taps + 1, K) is a few hundred floats and stays in L1.
__global__ void pulse_analysis_kernel(const float *pulse, long long bstride, const int *n_frames, int samples_per_frame,
                                      int n_max, const float *ana, int taps, int K, float *out) {
    const int b = blockIdx.y;
    const int n = item_rows(n_frames, b, samples_per_frame, n_max);
    const float *pb = pulse + (long long)b * bstride;
    float *ob = out + (long long)b * bstride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int r = i / K, k = i - r * K;
        const int first = r * K - taps / 2;
        float acc = 0.f;
        for (int t = 0; t <= taps; ++t) {
            const int src = first + t;
            if (src >= 0 && src < n) acc = fmaf(ana[t * K + k], pb[src], acc);
        }
        ob[i] = acc;
    }
}

void launch_pulse_analysis(const float *pulse, long long bstride, const int *n_frames, int samples_per_frame, int n_max,
                           int batch, const float *ana, int taps, int K, float *out, hipStream_t stream) {
    if (n_max <= 0 || batch <= 0) return;
    const int blocks = (int)min((long long)(n_max + 255) / 256, (long long)2048);
    hipLaunchKernelGGL(pulse_analysis_kernel, dim3(blocks, batch), dim3(256), 0, stream, pulse, bstride, n_frames,
                       samples_per_frame, n_max, ana, taps, K, out);
}

// Excitation fold + noise channel + WaveNet "start" 1x1 convolution
// (reference custom_pulsed_generator.py:893,905-906 and custom_AE_layers.py:280):
//   x[s, c] = pulse[pc*s + c] (c < pc), x[s, pc] = sigma * noise[s]
//   h[s, co] = bias[co] + sum_c x[s, c] * W[c, co]
// K <= 9, so this is a pure bandwidth kernel: one thread per (step, 4 channels), float4 store.
__global__ void wn_start_kernel(const float *pulse, long long pulse_bstride, const float *noise,
                                long long noise_bstride, float sigma, const int *n_frames, int steps_per_frame,
                                int max_steps, int pc, const float *w, const float *bias, int channels, float *h,
                                long long h_bstride) {
    const int b = blockIdx.y;
    const int steps = item_rows(n_frames, b, steps_per_frame, max_steps);
    const int cq = channels >> 2;   // channels % 4 == 0 (checked on the host)
    const long long total = (long long)steps * cq;
    const float *pb = pulse + (long long)b * pulse_bstride;
    const float *nb = noise ? noise + (long long)b * noise_bstride : nullptr;
    float *hb = h + (long long)b * h_bstride;
    const int cin = pc + (nb ? 1 : 0);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int s = (int)(i / cq), c4 = (int)(i - (long long)s * cq) * 4;
        float4 acc = bias ? *reinterpret_cast<const float4 *>(bias + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c = 0; c < cin; ++c) {
            const float xv = c < pc ? pb[(long long)s * pc + c] : sigma * nb[s];
            const float4 wv = *reinterpret_cast<const float4 *>(w + (long long)c * channels + c4);
            sum.x += xv * wv.x;
            sum.y += xv * wv.y;
            sum.z += xv * wv.z;
            sum.w += xv * wv.w;
        }
        acc.x += sum.x;
        acc.y += sum.y;
        acc.z += sum.z;
        acc.w += sum.w;
        *reinterpret_cast<float4 *>(hb + (long long)s * channels + c4) = acc;
    }
}

void launch_wn_start(const float *pulse, long long pulse_bstride, const float *noise, long long noise_bstride,
                     float sigma, const int *n_frames, int steps_per_frame, int max_steps, int batch,
                     int pulse_channels, const float *w, const float *bias, int channels, float *h,
                     long long h_bstride, hipStream_t stream) {
    if (max_steps <= 0 || batch <= 0) return;
    const long long total = (long long)max_steps * (channels >> 2);
    const int blocks = (int)min((total + 255) / 256, (long long)4096);
    hipLaunchKernelGGL(wn_start_kernel, dim3(blocks, batch), dim3(256), 0, stream, pulse, pulse_bstride, noise,
                       noise_bstride, sigma, n_frames, steps_per_frame, max_steps, pulse_channels, w, bias, channels,
                       h, h_bstride);
}

// Sub-band rows carried between the ticks of a stream (mbx_forward_options.sub_carry): per item b the descriptor
// desc[b] = (slot, inject row, inject rows, extract row, extract rows); `store` (slot, carry_rows, row_floats) is the
// caller's persistent buffer, `sub` (batch, ...) the sub-band tensor of the window.  dir 0: store -> sub rows
// [inject row, + inject rows); dir 1: sub rows [extract row, + extract rows) -> store.
__global__ void sub_carry_kernel(float *sub, long long sub_bstride, float *store, long long slot_stride, const int *desc,
                                 int row_floats, int dir) {
    const int b = blockIdx.y;
    const int *d = desc + 5 * b;
    const int row0 = dir ? d[3] : d[1], n = (dir ? d[4] : d[2]) * row_floats;
    float *sp = sub + (long long)b * sub_bstride + (long long)row0 * row_floats;
    float *st = store + (long long)d[0] * slot_stride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (dir) st[i] = sp[i];
        else sp[i] = st[i];
    }
}

void launch_sub_carry(float *sub, long long sub_bstride, float *store, long long slot_stride, const int *desc, int batch,
                      int max_rows, int row_floats, int dir, hipStream_t stream) {
    if (batch <= 0 || max_rows <= 0) return;
    const int n = max_rows * row_floats;
    hipLaunchKernelGGL(sub_carry_kernel, dim3((n + 255) / 256, batch), dim3(256), 0, stream, sub, sub_bstride, store,
                       slot_stride, desc, row_floats, dir);
}

// Per-layer WaveNet state carried between the ticks of a stream (mbx_forward_options.layer_carry): launched in front
// of the gate launch of layer l >= 1.  Per item b the descriptor desc[b] = (slot, end row of the stored state, end row
// of the state to store), rows at the WaveNet rate inside the window, -1: none.  With `end` the first row behind the
// region an item's WaveNet could read, layer l is exact up to row e_l = end + a.base_off (a staircase: every layer ends
// its own reach in front of the layer before it); the slot keeps, for this layer, the rows [e_l - h_before, + h_rows) of
// its input h_l and the rows [e_l, + acc_rows) of the output accumulator, which hold the contributions of the layers
// in front of l at this point of the sequence.  One thread moves one float both ways (slot -> window for the stored
// state, window -> slot for the new one); the two row ranges of an item must not overlap (checked by the caller:
// the new state ends at least h_rows + acc_rows rows behind the stored one).
__global__ void layer_carry_kernel(LayerCarryArgs a) {
    const int b = blockIdx.y;
    const int *d = a.desc + 3 * b;
    const int in_end = a.inject ? d[1] : -1, out_end = d[2];
    if (in_end < 0 && out_end < 0) return;
    float *st = a.store + (long long)d[0] * a.slot_stride + a.layer_off;
    float *hb = a.h + (long long)b * a.h_bstride;
    float *ab = a.acc + (long long)b * a.acc_bstride;
    const int n_h = a.h_rows * a.C, n = n_h + a.acc_rows * a.n_out;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float *buf;
        long long off_in, off_out;
        if (i < n_h) {
            buf = hb;
            off_in = (long long)(in_end + a.base_off - a.h_before) * a.C + i;
            off_out = (long long)(out_end + a.base_off - a.h_before) * a.C + i;
        } else {
            buf = ab;
            off_in = (long long)(in_end + a.base_off) * a.n_out + (i - n_h);
            off_out = (long long)(out_end + a.base_off) * a.n_out + (i - n_h);
        }
        const bool do_in = in_end >= 0 && off_in >= 0, do_out = out_end >= 0 && off_out >= 0;
        const float v_in = do_in ? st[i] : 0.f;
        const float v_out = do_out ? buf[off_out] : 0.f;
        if (do_in) buf[off_in] = v_in;
        if (do_out) st[i] = v_out;
    }
}

void launch_layer_carry(const LayerCarryArgs &a, int batch, hipStream_t stream) {
    const int n = a.h_rows * a.C + a.acc_rows * a.n_out;
    if (batch <= 0 || n <= 0) return;
    hipLaunchKernelGGL(layer_carry_kernel, dim3(std::min((n + 255) / 256, 64), batch), dim3(256), 0, stream, a);
}

// Front end carried between the ticks of a stream (FrontendCarryArgs): one block row per (frame, item); float4 copies.
__global__ void frontend_carry_kernel(FrontendCarryArgs a) {
    const int f = blockIdx.x, b = blockIdx.y;
    const int n = a.n_frames ? min(a.n_frames[b], a.frames) : a.frames;
    if (f >= n) return;
    const int per_frame = a.cond_floats + a.ceps_floats + a.f0_floats;
    int rf = (a.pos[b] + f) % a.ring_frames;
    if (rf < 0) rf += a.ring_frames;
    float *ring = a.store + ((long long)a.slot_desc[5 * b] * a.ring_frames + rf) * per_frame;
    const bool restore = f < a.first_new;
    float *bufs[3] = {a.cond + ((long long)b * a.frames + f) * a.cond_floats, a.ceps + ((long long)b * a.frames + f) * a.ceps_floats,
                      a.f0 + ((long long)b * a.frames + f) * a.f0_floats};
    const int sizes[3] = {a.cond_floats, a.ceps_floats, a.f0_floats};
    int off = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float4 *win = reinterpret_cast<float4 *>(bufs[k]);
        float4 *rg = reinterpret_cast<float4 *>(ring + off);
        for (int i = threadIdx.x; i < sizes[k] / 4; i += blockDim.x) {
            if (restore) win[i] = rg[i];
            else rg[i] = win[i];
        }
        off += sizes[k];
    }
}

void launch_frontend_carry(const FrontendCarryArgs &a, int batch, hipStream_t stream) {
    if (batch <= 0 || a.frames <= 0) return;
    hipLaunchKernelGGL(frontend_carry_kernel, dim3(a.frames, batch), dim3(128), 0, stream, a);
}

// Streaming windows kept on the device (mbx_window_advance): every row of `win` (batch, frames * row_floats) moves
// `step * row_floats` floats to the left and the freed tail is filled from `fresh` (batch, step * row_floats).  One block
// per (item, buffer); the kept part goes through LDS so that the move is safe in place.
__global__ void window_advance_kernel(float *mel, const float *mel_new, int mel_keep, int mel_step, float *noise,
                                      const float *noise_new, int noise_keep, int noise_step) {
    extern __shared__ float adv_buf[];
    const int b = blockIdx.x, second = blockIdx.y;
    float *win = second ? noise : mel;
    const float *fresh = second ? noise_new : mel_new;
    const int keep = second ? noise_keep : mel_keep, step = second ? noise_step : mel_step;
    if (!win) return;
    float *row = win + (long long)b * (keep + step);
    for (int i = threadIdx.x; i < keep; i += blockDim.x) adv_buf[i] = row[step + i];
    __syncthreads();
    for (int i = threadIdx.x; i < keep; i += blockDim.x) row[i] = adv_buf[i];
    const float *src = fresh + (long long)b * step;
    for (int i = threadIdx.x; i < step; i += blockDim.x) row[keep + i] = src[i];
}

// Delivered shader clock (mbx_clock_probe, bench.py): one wave spins for `real_ticks` ticks of the constant 100 MHz clock
// (s_memrealtime) and reports the shader-clock cycles (s_memtime) that passed: out = {cycles at start, at end, ticks at start,
// at end}.  Launched on a second stream beside the kernels under measurement it needs one wave slot and no LDS.
__global__ void clock_probe_kernel(unsigned long long *out, unsigned long long real_ticks) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < real_ticks) {
        __builtin_amdgcn_s_sleep(64);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[0] = c0;
    out[1] = c1;
    out[2] = r0;
    out[3] = r1;
}

void launch_clock_probe(unsigned long long *out, unsigned long long real_ticks, hipStream_t stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, stream, out, real_ticks);
}

// The general move of a device-resident window (mbx_window_update): rows of `pitch` floats; the floats [shift, shift + keep)
// of every row move to [0, keep), `fresh` (batch, step) lands behind them at [keep, keep + step).  One launch for a tick of a
// schedule whose chunk is not the frames the window moved by (the 80 ms schedule 6 / 6 / 7 / 6 / 7).
__global__ void window_update_kernel(float *mel, const float *mel_new, int mel_pitch, int mel_shift, int mel_keep, int mel_step,
                                     float *noise, const float *noise_new, int noise_pitch, int noise_shift, int noise_keep,
                                     int noise_step) {
    extern __shared__ float adv_buf[];
    const int b = blockIdx.x, second = blockIdx.y;
    float *win = second ? noise : mel;
    const float *fresh = second ? noise_new : mel_new;
    const int pitch = second ? noise_pitch : mel_pitch, shift = second ? noise_shift : mel_shift;
    const int keep = second ? noise_keep : mel_keep, step = second ? noise_step : mel_step;
    if (!win) return;
    float *row = win + (long long)b * pitch;
    if (shift > 0) {
        for (int i = threadIdx.x; i < keep; i += blockDim.x) adv_buf[i] = row[shift + i];
        __syncthreads();
        for (int i = threadIdx.x; i < keep; i += blockDim.x) row[i] = adv_buf[i];
    }
    const float *src = fresh + (long long)b * step;
    for (int i = threadIdx.x; i < step; i += blockDim.x) row[keep + i] = src[i];
}

bool launch_window_update(float *mel, const float *mel_new, float *noise, const float *noise_new, int batch, int win_frames,
                          int shift_frames, int keep_frames, int new_frames, int mel_channels, int steps_per_frame,
                          hipStream_t stream) {
    if (batch <= 0 || win_frames <= 0 || shift_frames < 0 || keep_frames < 0 || new_frames < 0 ||
        shift_frames + keep_frames > win_frames || keep_frames + new_frames > win_frames)
        return false;
    const size_t smem = sizeof(float) * (size_t)keep_frames * std::max(mel_channels, noise ? steps_per_frame : 0);
    if (smem > 64 * 1024) return false;
    hipLaunchKernelGGL(window_update_kernel, dim3(batch, noise ? 2 : 1), dim3(256), smem, stream, mel, mel_new,
                       win_frames * mel_channels, shift_frames * mel_channels, keep_frames * mel_channels, new_frames * mel_channels,
                       noise, noise_new, win_frames * steps_per_frame, shift_frames * steps_per_frame,
                       keep_frames * steps_per_frame, new_frames * steps_per_frame);
    return true;
}

bool launch_window_advance(float *mel, const float *mel_new, float *noise, const float *noise_new, int batch, int frames,
                           int step_frames, int mel_channels, int steps_per_frame, hipStream_t stream) {
    if (batch <= 0 || frames <= 0 || step_frames <= 0 || step_frames > frames) return false;
    const int mel_keep = (frames - step_frames) * mel_channels, noise_keep = (frames - step_frames) * steps_per_frame;
    const size_t smem = sizeof(float) * (size_t)std::max(mel_keep, noise ? noise_keep : 0);
    if (smem > 64 * 1024) return false;
    hipLaunchKernelGGL(window_advance_kernel, dim3(batch, noise ? 2 : 1), dim3(256), smem, stream, mel, mel_new, mel_keep,
                       step_frames * mel_channels, noise, noise_new, noise_keep, step_frames * steps_per_frame);
    return true;
}

}  // namespace mbx
