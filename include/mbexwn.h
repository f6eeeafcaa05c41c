/*
 * mbexwn.h -- C ABI of the MI355X (gfx950) mel-inversion engine.
 *
 * The reference (roebel/MBExWN_Vocoder) has no FFI/plugin boundary: its inference seam is the
 * Python call  model.infer(mell, sigma=None, synth_length=T*hop)  made by
 * MELInverter.synth_from_mel  (reference MBExWN_NVoc/mel_inverter.py:151-154) into
 * PaNWaveNet.infer (reference MBExWN_NVoc/vocoder/model/wavegen_1d.py:483-526), which runs
 * MBExWN.call (reference MBExWN_NVoc/vocoder/model/custom_pulsed_generator.py:556-771) on
 * TensorFlow.  This header is the boundary a maintainer binds instead (ctypes stub in
 * INTEGRATION.md): plain pointers and sizes, no torch / TensorFlow types.
 *
 * Conventions
 *   - every function returns an mbx_status; mbx_last_error() gives the thread-local message
 *   - the caller owns all buffers (mel, noise, audio, workspace: DEVICE memory of the handle's device)
 *   - a handle owns the folded weights and the constant tables; one handle per device; a handle is
 *     not re-entrant; calls only enqueue work on the given stream and never synchronise
 *   - all tensors are float32, channels-last, row-major
 *
 * The library reads NO environment variable: everything that changes numerics or the kernel choice is a field of
 * mbx_config (wn_conv_form, batch_invariant, wn_keep_skip, wn_keep_start, tune_*), so two handles of one process may
 * differ and a stray variable cannot change results.  (The Python host maps the MBX_* variables of its experiment
 * scripts onto these fields and says so on stderr: mbexwn_vocoder_amd/engine.py::experiment_overrides.)
 * The optional operand-order images of the weights ("*.wino2w", "*.wino4w", "*.packed", "*.fold",
 * "*.fold_wide", "*.fold_wave", "*.fold_f16", "*.start_fold", "*.fold_start", "*.fold_start_wide", "*.fold_start_wave",
 * "wn.tail.fold", "wn.end.packed";
 * engine.tensor_table builds them) select the specialised kernels; a handle created from the plain folded weights
 * alone runs the generic ones.
 */
#ifndef MBEXWN_H
#define MBEXWN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MBX_ABI_VERSION 10
#define MBX_MAX_SUBNET_OPS 32
#define MBX_MAX_WN_LAYERS 64
#define MBX_MAX_PRECOND 8
#define MBX_MAX_WN_BLOCKS 4
#define MBX_NAME_LEN 64

typedef enum {
    MBX_OK = 0,
    MBX_ERR_INVALID_ARGUMENT = 1, /* bad config / shape / missing tensor (Python side raises ValueError) */
    MBX_ERR_HIP = 2,              /* a HIP runtime call failed */
    MBX_ERR_WORKSPACE = 3,        /* workspace too small */
    MBX_ERR_UNSUPPORTED = 4       /* reference feature outside the hot path (NotImplementedError) */
} mbx_status;

/* sub-net op kinds: the layer grammar of generate_subnet_from_specs
 * (reference custom_pulsed_generator.py:38-148) flattened by the host */
enum { MBX_OP_CONV = 0, MBX_OP_LIN = 1, MBX_OP_PRELU = 2, MBX_OP_LEAKY = 3, MBX_OP_ACT = 4 };
/* padding of a conv op = TFPad1d modes (reference custom_layers.py:47-71) */
enum { MBX_PAD_ZERO = 0, MBX_PAD_SYMMETRIC = 1, MBX_PAD_EDGE = 2 };
/* final activations = ActivationLayer (reference custom_AE_layers.py:21-109) */
enum { MBX_ACT_LINEAR = 0, MBX_ACT_SOFT_SIGMOID = 1, MBX_ACT_TANH = 2, MBX_ACT_SIGMOID = 3,
       MBX_ACT_SOFT_SIGN = 4, MBX_ACT_SOFT_SQRT = 5, MBX_ACT_EXP = 6, MBX_ACT_RELU = 7 };

typedef struct {
    int32_t kind;            /* MBX_OP_* */
    int32_t ks, cin, cout;   /* conv: kernel size / channels (cout includes the sub-pixel factor) */
    int32_t pad_l, pad_r;    /* conv: samples of padding in front / behind */
    int32_t pad_mode;        /* conv: MBX_PAD_* */
    int32_t up;              /* conv: sub-pixel factor (depth -> time), lin: interpolation factor */
    int32_t act;             /* act: MBX_ACT_* */
    float alpha;             /* leaky: slope */
    char name[MBX_NAME_LEN]; /* conv: tensors "<name>.w" (ks,cin,cout) "<name>.b" ; prelu: "<name>.alpha" */
} mbx_subnet_op;

/* Model topology: the integers MBExWN.__init__ derives from mbexwn_config / preprocess_config
 * (reference custom_pulsed_generator.py:155-504). */
#define MBX_GATE_GTU 0
#define MBX_GATE_GFU 1
#define MBX_GATE_GSU 2
#define MBX_GATE_GLU 3

/* Form of the dilated convolution of the WaveNet layers (mbx_config.wn_conv_form).  All forms are float32 arithmetic of
 * the same function (reference custom_AE_layers.py:305-321); they differ in rounding: the Winograd forms trade
 * multiplications for additions of inputs / outputs, which multiplies the pre-activation rounding error by about 2
 * (F(2,3)) and 5 (F(4,3)) -- an error that grows with the amplitude of the residual stream.
 *   MBX_CONV_AUTO    F(4,3) when the handle's own weights allow it: mbx_create runs a short calibration forward through
 *                    F(4,3), F(2,3) and the direct form and keeps the fastest form whose audio stays within
 *                    calib_fraction of the 1e-4 * max(1, |audio|) parity budget of the direct form's (mbx_conv_form
 *                    reports the decision and the measured differences; mbx_calibrate repeats it on the caller's data)
 *   MBX_CONV_DIRECT  dilated convolution as written (K = 3C contraction)
 *   MBX_CONV_F23     Winograd F(2,3): 4 contractions per 2 outputs
 *   MBX_CONV_F43     Winograd F(4,3): 6 contractions per 4 outputs
 * Streaming calls (state_in / state_out) run F(2,3) unless the handle's form is the direct one. */
#define MBX_PRECISION_F32 0
#define MBX_PRECISION_SPLIT_F16 1
#define MBX_F0_ACC_F64 0
#define MBX_F0_ACC_F32 1
#define MBX_CONV_AUTO 0
#define MBX_CONV_DIRECT 1
#define MBX_CONV_F23 2
#define MBX_CONV_F43 3

typedef struct {
    int32_t struct_size;     /* sizeof(mbx_config), checked by mbx_create */
    int32_t abi_version;     /* MBX_ABI_VERSION */
    int32_t sample_rate, hop_size, mel_channels;
    int32_t subbands, pqmf_taps;
    int32_t pulse_channels, pulse_per_frame, steps_per_frame;
    float pulse_rate;        /* sample_rate / pulse_rate_factor */
    float noise_sigma;       /* pp_mod_subnet_noise_channel_sigma ; 0 => no noise channel */
    float f0_min, f0_max;
    int32_t wn_channels, wn_layers, wn_kernel_size, wn_out_channels, wn_in_channels;
    int32_t wn_dilations[MBX_MAX_WN_LAYERS];
    int32_t cond_kernel_size, cond_conv_upsampling, cond_lin_upsampling;
    int32_t stft_win, fft_size, n_ceps;
    int32_t n_ceps_windows;  /* 0 => ps_env_order_scale unset: no lifter */
    float filter_max_log_range; /* 0 => plain exp(S) */
    int32_t wt_n_period, wt_n_tables;
    float wt_nominal_f0, wt_min_transposition, wt_max_transposition, wt_grid_norm;
    int32_t phase_chunk;     /* 1000: stable_cumsum_and_wrap chunk (reference tf_wavetable.py:429) */
    int32_t n_f0_ops;
    mbx_subnet_op f0_ops[MBX_MAX_SUBNET_OPS];
    int32_t n_vtf_ops;
    mbx_subnet_op vtf_ops[MBX_MAX_SUBNET_OPS];
    /* optional RMS normalisation of the mel input and de-normalisation of the audio: NormMelComponents
     * (reference wavegen_1d.py:578-769) as PaNWaveNet.infer applies it (:493-495, 506-507), smoothing variant.
     * nm_iters == 0 => off.  Tables "table.nm_inv_enorm" (mel_channels), "table.nm_gwin" (stft_win),
     * "table.nm_smooth_win" (nm_smooth_win). */
    int32_t nm_iters;            /* normalize_rms_num_smooth_iters */
    int32_t nm_smooth_win;       /* win_size * normalize_smooth_win_scale */
    int32_t nm_use_compressor;   /* normalize_compressor_exp given */
    int32_t nm_use_max_limit;    /* use_max_limit */
    float nm_rms_norm_fact;      /* fft_size * win_size / 2 */
    float nm_rms_floor;          /* 1 / max_norm_fact, 0 => none */
    float nm_compressor_exp;
    float nm_lin_amp_scale, nm_lin_amp_off, nm_mel_amp_scale;
    /* gate of the WaveNet layers (pp_mod_subnet.activation, reference custom_AE_layers.py:312-321): the first half of a
     * layer's channels goes through MBX_GATE_GTU tanh(z), MBX_GATE_GFU z / (1 + |z|), MBX_GATE_GSU z / (1 + sqrt|z|) or
     * MBX_GATE_GLU z (the reference accepts "glu" at :156 and leaves the half linear), and is multiplied by the sigmoid
     * of the second half */
    int32_t wn_gate_activation;
    /* pp_mod_subnet.disable_conditioning (reference custom_AE_layers.py:203-204,293-294): no conditioning layer, the
     * gates see zeros; the tensors "wn.cond.*" are not needed */
    int32_t wn_disable_conditioning;
    /* pp_mod_subnet.pre_cond_layer_channels (reference custom_AE_layers.py:190-201,283-285): n_precond convolutions
     * "wn.precond_<i>" (kernel size cond_kernel_size, zero SAME padding, no activation) of the mel input in front of the
     * conditioning layer, whose input then has precond_channels[n_precond - 1] channels */
    int32_t n_precond;
    int32_t precond_channels[MBX_MAX_PRECOND];
    /* spect_filters_preserve_energy (reference custom_pulsed_generator.py:817-849): the cepstrum keeps its coefficient
     * 0 and every frame's filter is divided by the root of its mean squared magnitude over the fft_size/2 + 1 bins */
    int32_t spect_preserve_energy;
    /* wavetable_config.add_subharm_chans (reference tf_wavetable.py:520-559): n extra excitation channels
     * sin(2 pi phase / ii), ii = 2 .. n + 1, next to the pulse: the pulse tensor is (samples, 1 + n), a WaveNet row folds
     * pulse_channels samples = pulse_channels * (1 + n) channels (custom_pulsed_generator.py:893), and wn_in_channels
     * counts them.  wt_sinusoid_as_fun (use_sinusoid_as_fun, tf_wavetable.py:522-523): the pulse itself is
     * sin(2 pi phase) * 0.5 * (1 - cos(2 pi phase)) instead of the table lookup */
    int32_t wt_subharm_channels;
    int32_t wt_sinusoid_as_fun;
    /* ps_off (reference custom_pulsed_generator.py:663-672): no VTF-net and no STFT-domain filter, the audio is the
     * excitation itself (n_vtf_ops == 0).  no_pqmf (pp_mod_subnet_use_pqmf: false, :920-923): the sub-band rows are
     * not run through the PQMF synthesis bank but laid out one after the other (a reshape of the post-net output) */
    int32_t ps_off;
    int32_t no_pqmf;
    /* several WaveNet blocks with in-block upsampling (pp_mod_subnet_upsampling_factors / _channel_factors, reference
     * custom_pulsed_generator.py:456-488, custom_AE_layers.py:457-582).  n_wn_blocks == 0: one block without upsampling,
     * the fields above describe it.  n_wn_blocks >= 1: block b has wn_block_channels[b] channels (block 0: == wn_channels) and is followed
     * by a sub-pixel convolution "up<b>" (kernel size 3, wn_out_channels -> wn_out_channels * wn_block_ups[b], depth ->
     * time) when wn_block_ups[b] > 1; block b runs at steps_per_frame / prod(wn_block_ups[b:]) rows per frame
     * (steps_per_frame stays the sub-band rate, pulse_per_frame and cond_conv_upsampling describe block 0); its tensors
     * are "wn<b>.start", "wn<b>.cond", "wn<b>.conv1D_<l>", ... ("wn." for block 0).  Such a handle runs the generic
     * kernels and does not stream. */
    int32_t n_wn_blocks;
    int32_t wn_block_channels[MBX_MAX_WN_BLOCKS];
    int32_t wn_block_ups[MBX_MAX_WN_BLOCKS];
    /* pulse_channels_use_pqmf (reference custom_pulsed_generator.py:499-501, 892-895): the WaveNet rows are not
     * pulse_channels consecutive pulse samples but the pulse_channels sub-bands of a PQMF analysis of the pulse signal
     * (TFPQMF.analysis, tf_preprocess.py:188-200): pulse_pqmf_taps > 0 = taps of that bank, tensor "table.pulse_ana"
     * (pulse_pqmf_taps + 1, pulse_channels).  Whole items only; not together with wt_subharm_channels. */
    int32_t pulse_pqmf_taps;
    /* ps_use_stft: false (reference custom_pulsed_generator.py:427,453,663-672,857-884,916-917): the VTF-net ends in one
     * log gain per sub-band (n_ceps == subbands, mean over the bands removed with spect_preserve_energy); exp of it is
     * linearly interpolated by hop_size -- i.e. to the SAMPLE rate -- and its first rows multiply the sub-band rows
     * (the reference's own indexing, kept as it is); no STFT-domain filter, the audio is the excitation.  Whole items
     * only. */
    int32_t ps_subband_gain;
    /* pp_mod_subnet.padding: CAUSAL (Keras "causal"): the dilated, pre-conditioning, conditioning and up-sampling
     * convolutions of the WaveNet blocks pad dilation * (kernel size - 1) zeros in front and none behind.  Generic
     * kernels, whole items only. */
    int32_t wn_causal;
    /* ---- ABI 7: numerics / kernel policy (all zero = the defaults) ------------------------------------------------ */
    int32_t wn_conv_form;        /* MBX_CONV_* */
    /* 1: the kernels of a forward do not follow its launch size, so an item's result does not depend on the batch it ran
     * in (the reference runs one utterance at a time).  0: large and small launches may take different res/skip kernels,
     * whose results differ by float32 rounding (<= 4e-5 on canonical audio).  The direct form and F(2,3) are batch
     * invariant either way. */
    int32_t batch_invariant;
    int32_t wn_keep_skip;        /* 1: keep the un-folded skip path (C -> 2C res/skip layers, stage "wn_skip") */
    int32_t wn_keep_start;       /* 1: keep the start convolution and the full first layer (default: folded into layer 0) */
    float calib_fraction;        /* MBX_CONV_AUTO: share of the parity budget the form's own rounding may take; 0 = 0.25 */
    /* measurement knobs, none changes what a kernel computes for a given kernel choice (scripts/experiments):
     *   tune_gate_shape        0: by launch size, 1: 256-row F(4,3) blocks, 2: 128-row product-split blocks, 3: product-split
     *                          blocks of half a column tile (same bits, all three)
     *   tune_resskip_wave_tiles 0: default (2048); n > 0: res/skip launches of at most n 16-row tiles run the wave-tiled
     *                          kernel; -1: never
     *   tune_resskip_split     0: by launch size, 1..3: column split of the wave-tiled res/skip kernel (same bits) */
    int32_t tune_gate_shape, tune_resskip_wave_tiles, tune_resskip_split;
    /* normalize_use_pinv (reference wavegen_1d.py:603-608, 683-685): the frame RMS of the normalisation is the energy of
     * mel . pinv(mel filters)^T / nm_win_norm over the fft_size / 2 + 1 bins; table "table.nm_pinv" (mel_channels,
     * fft_size / 2 + 1), nm_win_norm = L2 norm of the analysis window */
    int32_t nm_use_pinv;
    float nm_win_norm;
    /* MBX_PRECISION_F32 (0, the default and the only arithmetic the parity claims and the headline benchmark are made in) or
     * MBX_PRECISION_SPLIT_F16: an opt-in experiment -- the res/skip layers and (whole-item forwards) the gate layers behind
     * the first one run their contractions on the 16-bit matrix pipe with every float32 operand split into two fp16 parts
     * (three products, float32 accumulation: float32-class error; csrc/wn_resskip_f16.hip, csrc/wn_gate_f16.hip; needs the
     * "*.fold_f16" / "*.gate_f16" weight images; not with the glu gate; |h| must stay inside fp16's range).  mbx_create
     * measures the handle in split precision against the float32 direct form on the calibration input and keeps the split
     * kernels only within the calibration threshold (mbx_conv_form_info.err_split / split_rejected) */
    int32_t wn_precision;
    /* ABI 9 (the former reserved word; 0 keeps its meaning "default"): arithmetic of the F0-net, the one stage whose output
     * the graph integrates (phase = running sum of f0 / pulse_rate, reference tf_wavetable.py:429-492), so that its rounding
     * moves every pulse behind it.  MBX_F0_ACC_F64 (0, default): the whole net in float64 -- float32 mel input, float64
     * weights, float64 hidden layers, contractions on v_mfma_f64_16x16x4_f64, the head (final 1x1 convolution,
     * interpolation to the pulse rate, final activation, map onto [f0_min, f0_max]; reference
     * custom_pulsed_generator.py:126-146, 773-791) one float64 kernel, ONE rounding to float32 at the end (contour = the
     * float32 nearest to the exact one).  This full chain needs, for EVERY convolution of the F0-net, the tensor
     * "<layer>.w64" next to "<layer>.w": the weight-norm fold W = g v / sqrt(max(sum v^2, 1e-12)) evaluated in float64,
     * shape (ks, cin, cout) as float64 viewed as 2 x the float32 words (mbexwn_vocoder_amd/engine.py::tensor_table builds
     * them); an op list of the shape (conv [prelu | leaky])* head; channel counts that are multiples of 4.  If any of that
     * is missing the handle falls back -- silently as far as results go, visibly in mbx_conv_form_info.f0_float64_chain --
     * to float32 weights and hidden layers with float64 accumulation per contraction (contour error ~1e-5 Hz instead of
     * half an ulp).  A mel pointer that is not 16-byte aligned is copied into the workspace first (same bits as an aligned
     * one).  MBX_F0_ACC_F32 (1): the float32 kernels of the other mel-rate sub-nets (ABI <= 8 behaviour; contour error
     * ~1e-3 Hz against ~2e-5 Hz) */
    int32_t f0_accumulate;
} mbx_config;

/* A named HOST tensor handed over at creation (weights already weight-norm folded, tables).
 * Required names: see mbexwn_vocoder_amd/engine.py::tensor_table. */
typedef struct {
    const char *name;
    const float *data;
    int32_t ndim;
    int64_t shape[4];
} mbx_tensor;

typedef struct mbx_handle mbx_handle;

/* Thread-local description of the last failure of this library on the calling thread. */
const char *mbx_last_error(void);

/* Replaces: MELInverter.load_model -> create_model -> MBExWN.__init__ + load_weights
 * (reference mel_inverter.py:184-239).  Copies tensors to `device`. */
mbx_status mbx_create(const mbx_config *config, const mbx_tensor *tensors, int32_t n_tensors, int32_t device,
                      mbx_handle **out);
mbx_status mbx_destroy(mbx_handle *handle);

/* What the handle decided about the dilated convolution (mbx_config.wn_conv_form). */
typedef struct {
    int32_t struct_size;      /* sizeof(mbx_conv_form_info), set by the caller */
    int32_t requested;        /* mbx_config.wn_conv_form */
    int32_t form;             /* MBX_CONV_DIRECT | _F23 | _F43: what whole-utterance forwards run */
    int32_t stream_form;      /* what streaming calls run (F(2,3), or the direct form) */
    int32_t calibrated;       /* 0: no calibration ran (form pinned by the config, or no Winograd images); 1: on the
                               * built-in synthetic mel at mbx_create; 2: on the caller's data (mbx_calibrate) */
    int32_t batch_invariant;
    int32_t fold_skip, fold_start;   /* the folds in effect */
    int32_t split_f16_layers; /* res/skip layers that run in split half precision (mbx_config.wn_precision; 0: none) */
    int32_t split_f16_gate_layers; /* gate layers that do (whole-item forwards; streams keep their float32 form) */
    float err_f43, err_f23;   /* max |audio(form) - audio(direct)| of the calibration run; < 0: form not available */
    float ref_max;            /* max |audio(direct)| of the calibration run */
    float threshold;          /* calib_fraction * 1e-4 * max(1, ref_max): a form is accepted at or below it */
    float err_split;          /* ABI 8: max |audio(this handle in split precision) - audio(float32 direct form)| of the calibration
                               * run (mbx_create runs it for every handle that asks for MBX_PRECISION_SPLIT_F16); < 0: not measured */
    int32_t split_rejected;   /* 1: that error was above the threshold (or not finite): the handle runs float32 after all */
    /* ABI 10 */
    int32_t f0_float64_chain; /* 1: the F0-net runs with float64 weights ("<layer>.w64"), float64 hidden layers and the float64
                               * head; 0: mbx_config.f0_accumulate's fallback (float32 weights and hidden layers, float64
                               * accumulation) or MBX_F0_ACC_F32 -- see f0_accumulate */
    int32_t n_gate_layers;    /* entries of gate_kernel that the most recent forward filled (0 before the first one) */
    int32_t gate_kernel[MBX_MAX_WN_LAYERS];   /* MBX_GATE_K_*: what ran the dilated convolution + gate of layer l in the most
                                               * recent forward (first WaveNet block) */
} mbx_conv_form_info;
#define MBX_GATE_K_NONE 0
#define MBX_GATE_K_DIRECT 1          /* conv1d_mfma_dma_kernel<EPI_GATE> */
#define MBX_GATE_K_F23 2             /* wn_gate_winograd2w_kernel */
#define MBX_GATE_K_F43 3             /* wn_gate_winograd4w_kernel, 256-row blocks */
#define MBX_GATE_K_F43_PSPLIT 4      /* wn_gate_winograd4p_kernel, 128-row product-split blocks */
#define MBX_GATE_K_F43_HSPLIT 5      /* wn_gate_winograd4h_kernel, product-split blocks of half a column tile */
#define MBX_GATE_K_F43_STRIDED 6     /* F(4,3) over d / 16 interleaved sub-sequences (d > 16), 256-row blocks */
#define MBX_GATE_K_F43_STRIDED_PSPLIT 7
#define MBX_GATE_K_FOLDED_START 8    /* wn_gate0_kernel: layer 0 with the start convolution folded in */
#define MBX_GATE_K_SPLIT_F16 9       /* wn_gate_f16_kernel (opt-in split half precision) */
mbx_status mbx_conv_form(const mbx_handle *handle, mbx_conv_form_info *info);

/* Re-runs the calibration of MBX_CONV_AUTO on the caller's own input (same argument meaning as mbx_forward) and adopts
 * its decision for the forwards that follow -- for a handle created with any wn_conv_form (a pinned form becomes the
 * calibrated one).  Synchronises the stream, allocates and frees scratch memory: a set-up call, not part of the
 * enqueue-only path.  Do not call it between the ticks of running streams (their form may change). */
mbx_status mbx_calibrate(mbx_handle *handle, const float *mel, const int32_t *n_frames, int32_t batch,
                         int32_t max_frames, const float *noise, void *workspace, size_t workspace_bytes,
                         void *hip_stream);

/* Bytes of device workspace mbx_forward needs for `batch` items of at most `max_frames` mel frames. */
size_t mbx_workspace_size(const mbx_handle *handle, int32_t batch, int32_t max_frames);

/* Replaces: model.infer(mell, synth_length=T*hop) (reference mel_inverter.py:152, wavegen_1d.py:483-526,
 * custom_pulsed_generator.py:556-771).
 *   mel       device (batch, max_frames, mel_channels)
 *   n_frames  device int32 (batch) valid frames per item, or NULL (= max_frames for all); every boundary
 *             op honours the item's own length, so a padded batch equals one-at-a-time runs
 *   noise     device (batch, max_frames*steps_per_frame) N(0,1) draw of the noise channel (with n_wn_blocks >= 1: one value
 *             per row of the FIRST block, i.e. max_frames * steps_per_frame / prod(wn_block_ups))
 *             (reference custom_pulsed_generator.py:905-906); NULL is only legal if noise_sigma == 0
 *   audio     device (batch, max_frames*hop_size); samples behind an item's own length are zeroed */
mbx_status mbx_forward(mbx_handle *handle, const float *mel, const int32_t *n_frames, int32_t batch,
                       int32_t max_frames, const float *noise, float *audio, void *workspace,
                       size_t workspace_bytes, void *hip_stream);

/* Streaming (BASELINE config 5): the graph is non-causal but has a finite receptive field, except for the wavetable
 * phase accumulator (reference tf_wavetable.py:429-492), which is causal with unbounded memory.  A stream is served
 * by windows [t0 - left, t0 + chunk + lookahead) of its mel frames; the caller keeps the middle `chunk` frames of
 * audio.  mbx_forward_stream is mbx_forward plus the carried accumulator state, so that the phase inside the window
 * is bit-identical to the whole-utterance run (state layout: see mbexwn_vocoder_amd/streaming.py).
 *   state_in[b]  : state valid just in front of window pulse-sample start_sample (samples before it give pulse 0)
 *   state_out[b] : state just in front of window pulse-sample save_sample (the next window's start); may be NULL */
typedef struct {
    float cum;             /* running float32 sum of the 1000-sample chunk in progress */
    float offset_sum;      /* un-wrapped float32 sum of (chunk total mod 1) over the finished chunks */
    int32_t pos_in_chunk;  /* position of start_sample inside its chunk (0 .. phase_chunk-1) */
    int32_t start_sample;  /* window-relative pulse-rate sample index the state applies to */
    int32_t save_sample;   /* window-relative sample whose state goes to state_out (< start_sample: none) */
    int32_t reserved;
} mbx_stream_state;

mbx_status mbx_forward_stream(mbx_handle *handle, const float *mel, const int32_t *n_frames, int32_t batch,
                              int32_t max_frames, const float *noise, float *audio, void *workspace,
                              size_t workspace_bytes, const mbx_stream_state *state_in, mbx_stream_state *state_out,
                              void *hip_stream);

/* Component / transposition interface = PaNWaveNet.infer_components (reference wavegen_1d.py:528-557): the F0 contour
 * may be supplied from outside and / or multiplied by a transposition factor before it drives the wavetable and the
 * lifter selection; the components (F0, excitation, cepstrum) are then read back with mbx_stage.
 *   f0             device (batch, max_frames*pulse_per_frame) Hz, or NULL = F0-net output
 *   transposition  factor applied to the F0 contour (1 = none; reference: F0 = transposition_factor * F0)
 *   state_in/out   streaming state as in mbx_forward_stream, or NULL */
typedef struct {
    int32_t struct_size;              /* sizeof(mbx_forward_options) */
    float transposition;
    const float *f0;
    const mbx_stream_state *state_in;
    mbx_stream_state *state_out;
    /* Streaming windows: the mel-rate stages and the phase run on the whole window, every stage from the WaveNet on
     * (WaveNet, PQMF, STFT filter, overlap-add: the expensive ones) only on the frames [active_begin, active_begin +
     * active_frames[b]) of item b, as if that region were the item (its edges get the item-edge padding, so the caller
     * keeps a margin inside it).  `audio` is written for that region only, at the same positions as for a whole window.
     *   active_begin   first frame of the region (0 with active_frames == NULL: whole window)
     *   active_frames  device (batch) int32 frames of each item inside the region, or NULL */
    int32_t active_begin;
    const int32_t *active_frames;
    /* Streaming windows, second level: the WaveNet (first layer .. output stage) runs only on the frames [wn_begin,
     * wn_begin + wn_frames[b]) of the window (inside the active region, which then only bounds PQMF, STFT filter and
     * overlap-add); NULL: the WaveNet runs on the active region.  The sub-band rows the later stages need from in front
     * of that region were computed by the previous tick and are carried:
     *   sub_store   device (slots, sub_store_rows, subbands) persistent buffer of the caller, or NULL
     *   sub_carry   device (batch, 5) int32: slot of the item, first sub-band row (window-relative) and number of rows
     *               to take from its slot before the PQMF, first row and number of rows to save to its slot afterwards */
    int32_t wn_begin;
    const int32_t *wn_frames;
    /* host-side upper bounds of active_frames[] and wn_frames[] (0: up to the end of the window): the launch grids are
     * sized from them */
    int32_t active_max_frames, wn_max_frames;
    float *sub_store;
    int32_t sub_store_rows;
    const int32_t *sub_carry;
    /* Streaming windows, third level: the per-layer state of the WaveNet carried between the ticks of a stream, so that a
     * tick runs every layer only on the rows that are new.  Layer l is exact up to its own reach in front of the rows
     * layer l-1 is exact for (a staircase that ends mbx_layer_state_info().reach_rows in front of the end of the WaveNet
     * region); a slot of the store keeps, per layer, the rows of the layer's input and of the output accumulator the next
     * tick reads from in front of its own rows.
     *   layer_store         device (slots, layer_store_floats) persistent buffer of the caller
     *   layer_store_floats  floats per slot (mbx_layer_state_info)
     *   layer_carry         device (batch, 3) int32: slot of the item; window row (WaveNet rate) the stored state ends at =
     *                       the end of the WaveNet region of the call that stored it, in this window's coordinates (-1: none);
     *                       window row the state stored by this call ends at = the end of this call's region (-1: none)
     *   layer_rows          0: the WaveNet runs on its whole region (state is stored where layer_carry asks for it, none
     *                       is read); > 0: every item reads its stored state and every layer runs on layer_rows rows:
     *                       all items then share one geometry -- region end E = (wn_begin + wn_max_frames) *
     *                       steps_per_frame, stored end = E - layer_rows >= min_rows behind -- and none of them ends
     *                       its utterance inside the window.  The sub-band rows [E - layer_rows - reach_rows,
     *                       E - reach_rows) are produced; wn_begin must be the frame of the first of them. */
    float *layer_store;
    int32_t layer_store_floats;
    const int32_t *layer_carry;
    int32_t layer_rows;
    /* Streaming windows, fourth level: the mel-rate front end (conditioning rows, cepstrum, F0 contour) carried between
     * the ticks of a stream.  Every sub-net has a finite receptive field, so the frames a window shares with the window
     * of the tick before keep their values; only the frames the new mel frames can reach are computed.
     *   fe_store          device (slots, fe_ring_frames, 2 C cond_conv_upsampling + n_ceps + pulse_per_frame) persistent
     *                     ring of the caller, or NULL; the slot of item b is sub_carry[b][0] (sub_carry is then required)
     *   fe_ring_frames    frames of a slot's ring (>= max_frames)
     *   fe_pos            device (batch) int32: ring frame of window frame 0 of each item (its absolute frame modulo
     *                     fe_ring_frames, so that a frame keeps its place from tick to tick)
     *   fe_new_frames     0: the front end runs on the whole window and every frame goes to the ring; > 0: every item spans
     *                     the whole window (a steady tick), the sub-nets run on its last fe_new_frames + fe_margin_frames
     *                     frames, the frames in front of the last fe_new_frames are taken from the ring, the new ones go
     *                     there.  fe_new_frames = frames the window moved by + the frames at the window end that the
     *                     sub-nets' look-ahead leaves inexact; fe_margin_frames = their reach into the past. */
    float *fe_store;
    int32_t fe_ring_frames;
    const int32_t *fe_pos;
    int32_t fe_new_frames, fe_margin_frames;
    /* with fe_new_frames > 0: frames of every item's window (0: max_frames).  A device window that is longer than what a
     * tick uses (one buffer for the phases of a tick schedule, n_frames masks the rest) computes the last fe_new_frames +
     * fe_margin_frames frames IN FRONT OF fe_end_frames */
    int32_t fe_end_frames;
} mbx_forward_options;

/* Geometry of the per-layer state (mbx_forward_options.layer_store): floats per slot (0: the handle cannot carry layer
 * state -- it needs the folded graph and the Winograd F(2,3) images), rows between the end of a WaveNet region and the
 * last sub-band row that is exact, smallest layer_rows a steady tick may use. */
mbx_status mbx_layer_state_info(const mbx_handle *handle, int32_t *floats_per_slot, int32_t *reach_rows,
                                int32_t *min_rows);

mbx_status mbx_forward_ex(mbx_handle *handle, const float *mel, const int32_t *n_frames, int32_t batch,
                          int32_t max_frames, const float *noise, float *audio, void *workspace,
                          size_t workspace_bytes, const mbx_forward_options *options, void *hip_stream);

/* Streaming windows kept on the device: before the steady tick of a set of streams, every item's window
 * mel_window (batch, frames, mel_channels) / noise_window (batch, frames * steps_per_frame) moves step_frames frames to
 * the left and the step_frames new frames mel_new (batch, step_frames, mel_channels) / noise_new (batch, step_frames *
 * steps_per_frame) are appended, in place, in one launch -- so that a tick uploads only the new frames and its launch
 * sequence (this call + mbx_forward_ex with constant arguments) can be replayed as a captured hipGraph (the practical
 * form of the "persistent-kernel path" of BASELINE config 5; streaming.py).  noise_window / noise_new may both be NULL. */
mbx_status mbx_window_advance(mbx_handle *handle, float *mel_window, const float *mel_new, float *noise_window,
                              const float *noise_new, int32_t batch, int32_t frames, int32_t step_frames, void *hip_stream);
/* The general move of the device-resident windows (ABI 10): every item's window is `window_frames` long; its frames
 * [shift_frames, shift_frames + keep_frames) move to the front and `new_frames` frames of mel_new (batch, new_frames,
 * mel_channels) / noise_new (batch, new_frames * steps_per_frame) are written behind them, in one launch -- the tick of a
 * schedule whose chunk differs from the frames the window start moved by (80 ms = 6 / 6 / 7 / 6 / 7 frames). */
mbx_status mbx_window_update(mbx_handle *handle, float *mel_window, const float *mel_new, float *noise_window,
                             const float *noise_new, int32_t batch, int32_t window_frames, int32_t shift_frames,
                             int32_t keep_frames, int32_t new_frames, void *hip_stream);
/* What a tick hands back (ABI 10): the samples [first, first + count) of every item's row of `audio` (batch rows of
 * row_floats floats on the device) as one strided device-to-host copy into host_out (batch, count; pinned memory for an
 * asynchronous copy), enqueued on the stream (capturable into a graph). */
mbx_status mbx_emit_rows(mbx_handle *handle, const float *audio, int64_t row_floats, int32_t batch, int64_t first,
                         int64_t count, float *host_out, void *hip_stream);

/* Intermediate tensors of the most recent mbx_forward (pointers into its workspace), for stage parity
 * tests.  Names: "f0" "pulse" "cond" "wn_hidden" "wn_skip" "wn_out" "subbands" "excitation" "cepstrum"
 * "ceps_index" "frames".  `count` = floats (int32 for ceps_index) per batch item, `stride` = item stride.
 * "wn_skip" (the C-wide skip sum) only exists when the skip path is not folded into the end convolution
 * (mbx_config.wn_keep_skip, or a handle created without the *.fold tensors). */
mbx_status mbx_stage(const mbx_handle *handle, const char *name, const void **device_ptr, int64_t *count,
                     int64_t *stride);

/* Kernel timing for bench.py: while enabled, mbx_forward brackets the launches of every stage of the sequence with
 * HIP events on the caller's stream (not capturable into a graph while enabled).  mbx_profile_read waits for the
 * recorded events, returns the summed device time and the number of bracketed launch groups of one stage since the
 * last read, and recycles the events.  Stages: "gate" (dilated conv + gate, one per layer), "res_skip" (one per layer),
 * "frontend" (F0-net, VTF-net, conditioning conv: the mel-rate launches), "wavetable" (phase + lookup), "start",
 * "tail" (end conv + post-net), "pqmf", "stft_filter", "overlap_add", "norm_mel", "gate0" (first layer with the start
 * convolution folded in), "res_skip_f16" (res/skip launches of the opt-in split half precision). */
mbx_status mbx_profile_enable(mbx_handle *handle, int32_t enabled);
mbx_status mbx_profile_read(mbx_handle *handle, const char *kernel, double *total_ms, int64_t *launches);
/* The same per launch group, in launch order (ABI 10): writes min(capacity, *launches) device times in ms to launch_ms and
 * the number of bracketed launch groups since the last read to *launches, and recycles the events.  With "gate" the
 * entries of one forward are the layers in order (behind the folded first layer, which is "gate0"): the per-layer times
 * of bench.py's geometry sweep. */
mbx_status mbx_profile_read_launches(mbx_handle *handle, const char *kernel, float *launch_ms, int64_t capacity,
                                     int64_t *launches);

/* Delivered shader clock for bench.py (ABI 10): enqueues ONE wave on the stream that spins for real_ticks ticks of the
 * constant 100 MHz clock (10 ns each; at most 1 s) and writes {shader cycles at start, at end, 100 MHz ticks at start, at
 * end} to device_out4 -- launched on a second stream while the kernels under measurement run, (cycles / ticks) x 100 MHz is
 * the shader clock the part delivers under that load (the fp32 MFMA peak of the guide assumes 2.4 GHz). */
mbx_status mbx_clock_probe(mbx_handle *handle, uint64_t *device_out4, int64_t real_ticks, void *hip_stream);

/* ---- stage entry points (unit parity; all pointers device memory) ------------------------------- */

/* TFPQMF.synthesis (reference tf_preprocess.py:204-226): x (batch, n_steps, subbands) -> y (batch, n_steps*subbands) */
mbx_status mbx_pqmf_synthesis(mbx_handle *handle, const float *x, int32_t batch, int32_t n_steps, float *y,
                              void *hip_stream);

/* Weight-normed Conv1D on a padded input (TFPad1d + TF2C_Conv1DWeightNorm, reference custom_layers.py:47-71,
 * conv_layers.py:149-165): x (batch, n_rows, cin), w (ks, cin, cout), b (cout) or NULL,
 * alpha (cout) or NULL (PReLU slopes) -> y (batch, n_rows, cout); dilation d, MBX_PAD_* mode. */
mbx_status mbx_conv1d(mbx_handle *handle, const float *x, int32_t batch, int32_t n_rows, int32_t cin,
                      const float *w, const float *b, const float *alpha, int32_t ks, int32_t cout,
                      int32_t dilation, int32_t pad_l, int32_t pad_mode, float *y, void *hip_stream);

/* The same convolution with float64 accumulation (float32 operands, exact products, float64 sums on
 * v_mfma_f64_16x16x4_f64, one rounding to float32 per output): the arithmetic of the F0-net's layers under
 * mbx_config.f0_accumulate = MBX_F0_ACC_F64 where the net keeps float32 weights and hidden layers (stage parity of
 * csrc/conv_mfma.hip::conv1d_f64_tile).  Needs cin % 4 == 0; launches of >= 12 288 rows take the 32 x 32 tiles of the large
 * mel-rate launches, smaller ones the 16 x 16 tiles -- same bits. */
mbx_status mbx_conv1d_f64acc(mbx_handle *handle, const float *x, int32_t batch, int32_t n_rows, int32_t cin,
                             const float *w, const float *b, const float *alpha, int32_t ks, int32_t cout,
                             int32_t dilation, int32_t pad_l, int32_t pad_mode, float *y, void *hip_stream);

/* TF2C_LinInterpLayer(num_pad_end=1, drop_last=True) (reference support_layers.py:99-121):
 * x (batch, n_rows, channels) -> y (batch, n_rows*up, channels) */
mbx_status mbx_lin_interp(mbx_handle *handle, const float *x, int32_t batch, int32_t n_rows, int32_t channels,
                          int32_t up, float *y, void *hip_stream);

/* PulseWaveTable.call (reference tf_wavetable.py:495-552): f0 (batch, n) Hz -> pulse (batch, n, 1 + wt_subharm_channels);
 * phase (batch, n) optional output of stable_cumsum_and_wrap (may be NULL). scratch >= batch*(n + n/chunk + 3) floats. */
mbx_status mbx_wavetable(mbx_handle *handle, const float *f0, int32_t batch, int32_t n, float *pulse, float *phase,
                         float *scratch, void *hip_stream);

/* STFT -> x envelope -> inverse STFT (reference custom_pulsed_generator.py:681-724, 793-855):
 * excitation (batch, frames*hop), cepstrum (batch, frames, n_ceps) (VTF-net output), ceps_index int32
 * (batch, frames) or NULL, -> audio (batch, frames*hop). scratch >= batch*frames*stft_win floats. */
mbx_status mbx_stft_filter(mbx_handle *handle, const float *excitation, const float *cepstrum,
                           const int32_t *ceps_index, int32_t batch, int32_t frames, float *audio, float *scratch,
                           void *hip_stream);

/* Audio -> log-mel analysis, the step in front of the mel inversion (reference MELInverter.generate_mel_from_snd,
 * mel_inverter.py:156-182 -> compute_mel_spectrogram_internal, vocoder/model/preprocess.py:417-572, with
 * calc_stft(center=True, magnitude), sig_proc/spec/stft.py:14-96).  Needs no handle: every table is an argument.
 *   audio      device (batch, max_samples) float32; n_samples device int32 (batch) or NULL
 *   window     device (win) analysis window; twiddle device (fft_size/2, 2) = exp(-2 pi i m / fft_size)
 *   basis      device (n_mels, fft_size/2+1) mel filters, bin_lo / bin_hi device int32 (n_mels): non-zero range of a row
 *   out        device (batch, max_frames, n_mels): log(max(|STFT| . basis^T, eps)); item b has n_samples[b]/hop + 1 frames */
mbx_status mbx_mel_analysis(const float *audio, const int32_t *n_samples, int32_t batch, int32_t max_samples,
                            int32_t win, int32_t hop, int32_t fft_size, int32_t n_mels, const float *window,
                            const float *twiddle, const float *basis, const int32_t *bin_lo, const int32_t *bin_hi,
                            float eps, float *out, int32_t max_frames, void *hip_stream);

/* NormMelComponents.normalize_inputs_by_rms(None, mell, synth_length) (reference wavegen_1d.py:638-769), only for
 * models with nm_iters > 0: mel (batch, frames, mel_channels) -> mel_out (same shape) and, if gain != NULL,
 * gain (batch, frames*hop) = upsampled_rms (what mbx_forward multiplies onto the audio).  n_frames: device int32
 * (batch) or NULL.  scratch >= 2*batch*frames floats. */
mbx_status mbx_norm_mel(mbx_handle *handle, const float *mel, const int32_t *n_frames, int32_t batch, int32_t frames,
                        float *mel_out, float *gain, float *scratch, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* MBEXWN_H */
