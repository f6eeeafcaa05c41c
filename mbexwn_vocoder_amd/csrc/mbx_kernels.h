// Internal launch interface between the C ABI (mbx_api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mbx {

// Gate of a WaveNet layer (reference custom_AE_layers.py:312-321): act(zt) * sigmoid(zs) with act = tanh (kind 0, gtu),
// z / (1 + |z|) (1, gfu), z / (1 + sqrt|z|) (2, gsu) or z itself (3, glu: accepted at :156, no branch at :312-318).  One
// reciprocal for the whole product:
//   tanh(zt) sigmoid(zs) = (t - 1) / ((t + 1)(1 + s)),  t = e^(2 zt), s = e^(-zs)   (zt is clamped where tanh is 1 in
//   float32, so t stays finite; s = inf gives 0, as it should);   zt / ((1 + |zt|)(1 + s)) etc. for the others.
// `kind` is a kernel argument: the branch is wave-uniform.
#if defined(__HIPCC__)
// Rows (frames, samples) of batch item b: its own count from the device array -- clamped into [0, bound], so that an item
// that claims more frames than the window holds is cut to the window instead of addressing past its buffers -- or the
// bound itself when the batch has no per-item lengths.
__device__ __forceinline__ int item_rows(const int *n_frames, int b, int per_frame, int bound) {
    if (!n_frames) return bound;
    const long long n = (long long)max(n_frames[b], 0) * per_frame;
    return n < bound ? (int)n : bound;
}

__device__ __forceinline__ float wn_gate_act(int kind, float zt, float zs) {
    const float sg = __builtin_amdgcn_exp2f(zs * -1.4426950408889634f);
    if (kind == 0) {
        const float t = __builtin_amdgcn_exp2f(fminf(zt, 15.f) * 2.885390081777927f);
        const float tp = t + 1.0f;
        return (t - 1.0f) * __builtin_amdgcn_rcpf(fmaf(sg, tp, tp));
    }
    if (kind == 3) return zt * __builtin_amdgcn_rcpf(1.0f + sg);
    const float az = fabsf(zt);
    const float den = 1.0f + (kind == 1 ? az : __builtin_amdgcn_sqrtf(az));
    return zt * __builtin_amdgcn_rcpf(fmaf(sg, den, den));
}
#endif

// ---------------------------------------------------------------------------------------------
// conv1d as an implicit GEMM on the fp32 matrix cores (conv_mfma.hip)
// ---------------------------------------------------------------------------------------------
enum ConvEpilogue {
    EPI_LINEAR = 0,   // y = acc + bias, optional PReLU / leaky slope
    EPI_GATE = 1,     // a = tanh(acc_t + bias_t + cond_t) * sigmoid(acc_s + bias_s + cond_s)
    EPI_RESSKIP = 2   // h += r[:, :C] ; skip (+)= r[:, C:]   (last layer: skip += r)
};

struct ConvArgs {
    // input (batch, rows, cin) channels-last
    const float *x;
    long long x_bstride;      // floats between batch items
    int ldx;                  // floats between rows
    // rows per item: n_frames ? n_frames[b] * rows_per_frame : max_rows
    const int *n_frames;
    int rows_per_frame;
    int max_rows;
    int batch;
    // weights (ks*cin, cout) row-major, bias (cout) or null
    const float *w;
    const float *bias;
    int cin, cout, ks, dil, pad_l, pad_mode;
    // output
    float *out;
    long long out_bstride;
    int ldo;
    // EPI_LINEAR extras
    const float *alpha;       // PReLU slopes (cout) or null
    float leaky;              // slope when use_leaky
    int use_leaky;
    // EPI_GATE extras: conditioning (batch, rows/cond_up, 2*C), interpolated on the fly
    const float *cond;
    long long cond_bstride;
    int cond_up;
    int cond_phase;           // item row r sits at conditioning-rate position r + cond_phase (0 <= cond_phase < cond_up; F(2,3) gate kernel only)
    const float *lerp_w0, *lerp_w1;   // (cond_up) float32 interpolation weights
    int channels;             // C (gate: cout == 2C, out has C columns; res/skip: split point)
    int gate_act;             // EPI_GATE and the gate kernels: 0 gtu, 1 gfu, 2 gsu, 3 glu (wn_gate_act)
    // EPI_RESSKIP extras
    float *h;                 // (batch, rows, C) updated in place
    float *skip;              // (batch, rows, skip_ld)
    int skip_ld;              // floats between rows of skip (0: C)
    long long skip_bstride;   // floats between batch items of skip (0: max_rows * skip_ld, or hs_bstride when skip_ld == 0)
    long long hs_bstride;
    int skip_init;            // 1: skip = s (first layer)  0: skip += s
    int h_init;               // 1: h = r (first layer with the start convolution folded in: x carries [a | x'], cin > C)
    int last_layer;           // 1: cout == C, everything goes to skip
    int acc_preloaded;        // set by the launcher: accumulators start from bias + old value, epilogue only stores
    int remap, n_tiles, m_tiles_per_item, m_tiles_total;   // XCD-aware 1-D grid (set by the launcher)
    int out_row0, out_rows;   // wave-tiled F(2,3) gate kernel: only the rows [out_row0, out_row0 + out_rows) of every item are
                              // computed (out_rows == 0: all rows; out_row0 a multiple of 2 * dil)
    const float *zeros;       // >= 16 bytes of zeros in global memory (LDS-DMA source of padding lanes)
    int fast_dma;             // set by the launcher: 32-bit source offsets are safe (LDS-DMA with a uniform base)
    int vstride;              // set by launch_wn_gate_winograd4w for dilations above 16: every item is treated as vstride interleaved
                              // virtual items (rows s, s + vstride, s + 2 vstride ...) convolved with dilation dil / vstride
    // split half precision (mbx_config.wn_precision): the hidden state as fp16 planes, written by wn_resskip_f16_kernel and read
    // by wn_gate_f16_kernel -- row r: [hi: h_split_ld halves][lo': h_split_ld halves] (h_split_ld = channels rounded up to 8,
    // the padding zero), i.e. h_split_ld float32 words per row
    float *h_split;
    long long h_split_bstride;   // float32 words between batch items
    int h_split_ld;
    int h_planes_only;        // wn_resskip_f16_kernel: the planes ARE the hidden state -- old values are read from them (hi + 2^-11 lo'),
                              // the float32 tensor h is neither read nor written (every consumer of h takes the planes)
    int tune_split;           // wave-tiled res/skip kernel: 1..3 pins its column split (mbx_config.tune_resskip_split; same bits), 0: by launch size
    // EPI_LINEAR, the F0-net (conv1d_f64_tile): contraction on v_mfma_f64_16x16x4_f64
    int precise;              // 1: float64 accumulation; with only this set x, w and out are the float32 ones above (one rounding per output)
    const double *x64;        // input as float64 (same strides, counted in elements) instead of x; needs w64
    const double *w64;        // weights (ks*cin, cout) as float64 instead of w
    double *out64;            // output as float64 (same strides) instead of out
};

void launch_conv1d(const ConvArgs &a, int epilogue, hipStream_t stream);
// n <= 3 independent EPI_LINEAR convolutions; the small (mel-rate, small batch) ones share one launch
void launch_conv1d_group(const ConvArgs *convs, int n, hipStream_t stream);
// Winograd F(4,3) form on v_mfma_f32_16x16x4_f32, wave tile 16 groups x 64 columns (wn_winograd4w.hip); a.w = image of
// engine.pack_winograd4w_weights (ceil(C/32), ceil(C/8), 3072); split: 128-row blocks whose waves split the six products
// (same bits as the 256-row blocks)
// shape: 0 = 256-row blocks, 1 = 128-row product-split blocks, 2 = product-split blocks of half a column tile (same bits, all three)
bool launch_wn_gate_winograd4w(const ConvArgs &a, int shape, hipStream_t stream);
// Winograd F(2,3) form on v_mfma_f32_16x16x4_f32 with wave-granular tiles (wn_winograd2w.hip: streams, per-layer regions,
// MBX_CONV_F23); a.w = image of engine.pack_winograd2w_weights (ceil(C/32), ceil(C/8), 2048)
bool launch_wn_gate_winograd2w(const ConvArgs &a, hipStream_t stream);
// the dilated convolution + gate in split half precision, direct form (wn_gate_f16.hip; opt-in, mbx_config.wn_precision);
// a.w = image of engine.pack_gate_f16_weights (ceil(C/32), ceil(C/32), 6144)
bool launch_wn_gate_f16(const ConvArgs &a, hipStream_t stream);
// First WaveNet layer with the start convolution folded into it (wn_gate0.hip)
struct Gate0Args {
    const float *pulse;       // (batch, rows * pulse_channels): the excitation, folded to pulse_channels per row
    long long pulse_bstride;
    const float *noise;       // (batch, rows) or null
    long long noise_bstride;
    float sigma;
    int pulse_channels;
    const int *n_frames;
    int rows_per_frame, max_rows, batch;
    const float *w;           // image of engine.fold_start_weights (ceil(C/32), 3, 2, 64, 4)
    const float *bias;        // (2C) gate bias or null
    int channels, dil;
    const float *cond;        // (batch, rows/cond_up, 2C)
    long long cond_bstride;
    int cond_up;
    const float *lerp_w0, *lerp_w1;
    int gate_act;             // 0 gtu, 1 gfu, 2 gsu, 3 glu (wn_gate_act)
    float *out;               // (batch, rows, ldo): C gate channels, then x' padded to 16 channels if write_inputs
    long long out_bstride;
    int ldo, write_inputs;
    int n_tiles, m_tiles_per_item;   // set by the launcher
};
bool launch_wn_gate0(const Gate0Args &a, hipStream_t stream);
bool wn_gate0_fits(int channels, int pulse_channels, int dil, int cond_up);
// WaveNet residual/skip layer for large row counts (wn_resskip.hip); a.w = host-packed weights (ceil(cout/128), ceil(C/16), 2048)
bool launch_wn_resskip(const ConvArgs &a, hipStream_t stream);
// the same layer for large launches, one block owning all columns of its rows (wn_resskip_wide.hip); a.w = image of
// engine.pack_resskip_wide_weights (ceil(cin/8), ceil(cout/32), 256)
bool launch_wn_resskip_wide(const ConvArgs &a, hipStream_t stream);
// the same layer for small launches (one utterance, streaming ticks), tiled at wave granularity (wn_resskip_wave.hip);
// a.w = image of engine.pack_resskip_wave_weights (ceil(cin/16), 12, 512)
bool launch_wn_resskip_wave(const ConvArgs &a, hipStream_t stream);
// the same layer in split half precision (wn_resskip_f16.hip; opt-in, mbx_config.wn_precision); a.w = image of
// engine.pack_resskip_f16_weights (ceil(cin/32), 12, 1024); a.gate_act must be set (glu is refused)
bool launch_wn_resskip_f16(const ConvArgs &a, hipStream_t stream);
// WaveNet end convolution + post-net in one pass over the skip tensor (wn_tail.hip); w_end_packed = host-packed
// weights (ceil(C/8), 2, 32, 4); false: shapes do not fit
bool launch_wn_tail(const float *skip, long long skip_bstride, const int *n_frames, int rows_per_frame, int max_rows,
                    int batch, int C, const float *w_end_packed, const float *b_end, int n_out, const float *w_post,
                    const float *b_post, int M, const float *y_acc, float *y, long long y_bstride, float *sub,
                    long long sub_bstride, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// element-wise / bandwidth-type stages (elementwise.hip)
// ---------------------------------------------------------------------------------------------
// linear interpolation (B,rows,C) -> (B,rows*up,C), optional final activation and affine (y*scale+offset)
void launch_lin_interp(const float *x, long long x_bstride, const int *n_frames, int rows_per_frame, int max_rows,
                       int batch, int channels, int up, const float *w0, const float *w1, int act, float scale,
                       float offset, float *y, long long y_bstride, hipStream_t stream);
// Head of the F0-net in float64 (elementwise.hip::f0_head_kernel): 1x1 convolution cin -> 1 (+ bias), linear interpolation by
// `up`, final activation, y * scale + offset -- float32 in, one rounding to float32 at the end
// (reference custom_pulsed_generator.py:126-146, 773-791; custom_AE_layers.py:91-99)
// x32 or x64: the hidden layer as float32 or float64; w32 or w64 likewise
void launch_f0_head(const float *x32, const double *x64, long long x_bstride, int cin, const int *n_frames, int rows_per_frame,
                    int max_rows, int batch, const float *w32, const double *w64, const float *bias, int up, const float *w0,
                    const float *w1, int act, float scale, float offset, float *y, long long y_bstride, hipStream_t stream);
// y = act(x)*scale + offset on (B, rows, C)
// sub-band rows carried between the ticks of a stream (elementwise.hip); desc (batch, 5) int32 on the device
void launch_sub_carry(float *sub, long long sub_bstride, float *store, long long slot_stride, const int *desc, int batch,
                      int max_rows, int row_floats, int dir, hipStream_t stream);
// per-layer WaveNet state carried between the ticks of a stream (elementwise.hip); desc (batch, 3) int32 on the device
struct LayerCarryArgs {
    float *h;                 // (batch, rows, C) hidden state of the window
    long long h_bstride;
    int C;
    float *acc;               // (batch, rows, n_out) output accumulator of the window (folded skip path)
    long long acc_bstride;
    int n_out;
    float *store;             // (slots, slot_stride) caller's persistent buffer
    long long slot_stride, layer_off;   // floats: between slots, of this layer inside a slot
    const int *desc;          // (batch, 3): slot, end row of the stored state, end row of the state to store (-1: none)
    int inject;               // 0: desc[b][1] is ignored (nothing is taken from the store)
    int base_off;             // e_l - end: last exact row (+1) of this layer relative to the end row
    int h_before, h_rows;     // stored rows of h: [e_l - h_before, + h_rows)
    int acc_rows;             // stored rows of acc: [e_l, + acc_rows)
};
void launch_layer_carry(const LayerCarryArgs &a, int batch, hipStream_t stream);
// mel-rate front end (conditioning rows, cepstrum, F0 contour) carried between the ticks of a stream (elementwise.hip):
// per slot a ring of ring_frames frames of [cond | ceps | f0]; window frame f of item b lives at ring frame
// (pos[b] + f) % ring_frames.  Frames [0, first_new) of the window are read from the ring, frames [first_new, frames) --
// [first_new, n_frames[b]) with n_frames -- are written to it.
struct FrontendCarryArgs {
    float *cond, *ceps, *f0;          // (batch, frames * floats) window buffers
    int cond_floats, ceps_floats, f0_floats;
    int frames;
    const int *n_frames;              // (batch) or null
    float *store;                     // (slots, ring_frames, cond_floats + ceps_floats + f0_floats)
    int ring_frames;
    const int *pos;                   // (batch) ring frame of window frame 0
    const int *slot_desc;             // (batch, 5): [0] = slot of the item (the sub-band carry descriptors)
    int first_new;
};
void launch_frontend_carry(const FrontendCarryArgs &a, int batch, hipStream_t stream);
// device-resident streaming windows: shift every item's window left by step_frames and append the new frames
void launch_clock_probe(unsigned long long *out, unsigned long long real_ticks, hipStream_t stream);
bool launch_window_update(float *mel, const float *mel_new, float *noise, const float *noise_new, int batch, int win_frames,
                          int shift_frames, int keep_frames, int new_frames, int mel_channels, int steps_per_frame,
                          hipStream_t stream);
bool launch_window_advance(float *mel, const float *mel_new, float *noise, const float *noise_new, int batch, int frames,
                           int step_frames, int mel_channels, int steps_per_frame, hipStream_t stream);
void launch_activation(const float *x, long long x_bstride, const int *n_frames, int rows_per_frame, int max_rows,
                       int batch, int channels, int act, float scale, float offset, float *y, long long y_bstride,
                       hipStream_t stream);
// PReLU / leaky in place (used when a conv is not directly followed by its activation)
void launch_prelu(float *x, long long x_bstride, const int *n_frames, int rows_per_frame, int max_rows, int batch,
                  int channels, const float *alpha, float leaky, hipStream_t stream);
// sub-band gains of the ps_use_stft: false variant (reference custom_pulsed_generator.py:857-884, 670, 916-917):
// sub[b, r, m] *= lerp_{hop}(exp(log_gain[b, :, m] (- mean over m)))[r], log_gain (B, max_frames, M)
void launch_subband_gain(float *sub, long long sub_bstride, const float *log_gain, long long gain_bstride, const int *n_frames,
                         int max_frames, int rows_per_frame, int batch, int M, int hop, const float *w0, const float *w1,
                         int remove_mean, hipStream_t stream);
// PQMF analysis of the pulse signal (reference tf_preprocess.py:188-200): pulse (B, n_max) -> out (B, n_max) viewed as
// (rows, K): out[r, k] = sum_n ana[n, k] * pulse[r K + n - taps / 2], zero outside the item
void launch_pulse_analysis(const float *pulse, long long bstride, const int *n_frames, int samples_per_frame, int n_max,
                           int batch, const float *ana, int taps, int K, float *out, hipStream_t stream);
// WaveNet input: fold pulse (B, steps*pc) + noise (B, steps) and apply the start 1x1 conv -> h (B, steps, C)
void launch_wn_start(const float *pulse, long long pulse_bstride, const float *noise, long long noise_bstride,
                     float sigma, const int *n_frames, int steps_per_frame, int max_steps, int batch,
                     int pulse_channels, const float *w, const float *bias, int channels, float *h,
                     long long h_bstride, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// wavetable oscillator (wavetable.hip)
// ---------------------------------------------------------------------------------------------
struct WaveTableConsts {
    const float *tables;   // (n_period+1, n_tables)
    int n_period, n_tables;
    float pulse_rate, nominal_f0, min_tf, max_tf, grid_norm;
    int chunk;
    int n_sub;             // add_subharm_chans: extra channels sin(2 pi phase / ii), ii = 2 .. n_sub + 1, behind the pulse
    int sin_fun;           // use_sinusoid_as_fun: pulse = sin(2 pi phase) * 0.5 * (1 - cos(2 pi phase)), no table lookup
};
// carried phase-accumulator state of a stream (== mbx_stream_state of include/mbexwn.h)
struct StreamState {
    float cum;            // running sum of the chunk in progress, just in front of start_sample
    float offset_sum;     // un-wrapped sum of (chunk totals mod 1) of all finished chunks
    int pos_in_chunk;     // position of start_sample inside its 1000-sample chunk
    int start_sample;     // window-relative pulse sample the state applies to
    int save_sample;      // window-relative pulse sample whose state is written to the output (< start: none)
    int reserved;
};
// f0 (B, n_max) -> pulse (B, n_max, 1 + n_sub); cum / chunk_last are scratch: cum (B, n_max), chunk_last (B, n_chunks_max + 1)
void launch_wavetable(const WaveTableConsts &c, const float *f0, long long bstride, const int *n_frames,
                      int samples_per_frame, int n_max, int batch, float *pulse, float *phase_out, float *cum,
                      float *chunk_last, const StreamState *st_in, StreamState *st_out, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// PQMF synthesis (pqmf.hip)
// ---------------------------------------------------------------------------------------------
// x (B, steps, M) -> y (B, steps*M); poly = polyphase table (M, n_dm, M) with dm_min; poly_t = the same table as the MFMA
// B operand (4 * ceil(n_dm * M / 4), 16), zero padded, or null (M > 16)
void launch_pqmf(const float *x, long long x_bstride, const int *n_frames, int steps_per_frame, int max_steps,
                 int batch, int subbands, const float *poly, const float *poly_t, int n_dm, int dm_min, float *y,
                 long long y_bstride, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// STFT-domain envelope filter (stft_filter.hip)
// ---------------------------------------------------------------------------------------------
struct StftConsts {
    int hop, win, fft_size, n_ceps, n_ceps_windows;
    int preserve_energy;              // spect_filters_preserve_energy: cepstral coefficient 0 kept, H / rms_k |H|
    float max_log_range;
    const float *hann, *inv_win;      // (win)
    const float *twiddle;             // (fft_size/2, 2)
    const float *ceps_windows;        // (n_ceps_windows, n_ceps) or null
    const float *ceps_log10f0;        // (n_ceps_windows)
    const float *f0_smooth;           // (2*hop+1)
    int pulse_per_frame;
};
void launch_stft_filter(const StftConsts &c, const float *exc, long long exc_bstride, const float *ceps,
                        long long ceps_bstride, const int *index, const float *f0, long long f0_bstride,
                        int *index_out, const int *n_frames, int max_frames, int batch, float *frames,
                        hipStream_t stream);
// overlap-add of the windowed frames + slice -> audio (B, out_frames*hop), tail zeroed; max_frames = item stride of `frames`
void launch_overlap_add(const StftConsts &c, const float *frames, const int *n_frames, int max_frames, int out_frames, int batch,
                        float *audio, long long audio_bstride, hipStream_t stream);

// audio -> log-mel analysis (mel_analysis.hip); tables from the host module analysis.py
struct MelAnalysisArgs {
    const float *audio;       // (batch, max_samples)
    long long audio_bstride;
    const int *n_samples;     // (batch) or null
    int max_samples, batch;
    int win, hop, fft_size, n_mels;
    const float *window;      // (win)
    const float *twiddle;     // (fft_size/2, 2): exp(-2 pi i m / fft_size)
    const float *basis;       // (n_mels, fft_size/2 + 1) dense rows
    const int *bin_lo, *bin_hi;   // (n_mels) first / last non-zero bin of a row
    float eps;
    float *out;               // (batch, max_frames, n_mels), frames of item b: n_samples[b] / hop + 1
    int max_frames;
};
bool launch_mel_analysis(const MelAnalysisArgs &a, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// optional RMS normalisation of the mel input / de-normalisation of the audio (norm_mel.hip)
// ---------------------------------------------------------------------------------------------
struct NormMelConsts {
    int iters, mel_channels, hop, win, smooth_win, cut;    // cut = smooth_win/2 + 2 hop - win/2
    float rms_norm_fact, rms_floor, compressor_exp, lin_amp_scale, lin_amp_off, mel_amp_scale;
    int use_compressor, use_max_limit;
    const float *inv_enorm;          // (mel_channels)
    const float *pinv;               // normalize_use_pinv: (mel_channels, n_bins) pseudo inverse of the mel filters, else null
    int n_bins;                      // fft_size / 2 + 1
    float win_norm;                  // L2 norm of the analysis window
    const float *gwin;               // (win) unit-sum analysis window
    const float *smooth_win_table;   // (smooth_win)
};
// returns the (B, Tmax) buffer the output gain is built from (pass it to launch_norm_mel_gain)
const float *launch_norm_mel(const NormMelConsts &c, const float *mell, long long mel_bstride, const int *n_frames,
                             int max_frames, int batch, float *rms_a, float *rms_b, float *mell_out,
                             hipStream_t stream);
// audio *= gain (overwrite: audio = gain), gain = max(gain[win/2 + n], eps) of the last smoothing pass
void launch_norm_mel_gain(const NormMelConsts &c, const float *r_last, const int *n_frames, int max_frames, int batch,
                          float *audio, long long audio_bstride, bool overwrite, hipStream_t stream);

}  // namespace mbx
