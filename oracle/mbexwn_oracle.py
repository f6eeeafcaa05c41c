"""CPU ORACLE of the MBExWN mel-inversion forward pass -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module; the product package (``mbexwn_vocoder_amd``) never does.

It is a plain numpy restatement (float64 by default, float32 selectable for the CPU timing
baseline) of the reference's inference graph.  Every function cites the reference file:line it
follows (paths relative to /root/reference/MBExWN_NVoc/vocoder/model/).

PINNING STATUS
  * The reference has no tests, fixtures or golden vectors for this path and its
    graph executes inside TensorFlow, which is not installable here.
  * Pinned against the reference itself, run in the build container:
      - constants (PQMF banks, LF wavetables, windows, scale_mel) captured by importing the
        reference's numpy code  -> tests/golden/reference_constants.npz
        (generator: tests/golden/make_reference_constants.py)
      - the full graph mel -> audio and every intermediate stage, produced by executing the
        reference's own Python model code (MBExWN.call and all layers under it) on top of a
        numpy stand-in for the TensorFlow op set -> tests/golden/reference_forward_*.npz
        (generator: tests/golden/make_reference_forward.py, shim: tests/golden/tf_numpy_shim.py)
  * NOT pinned: the TensorFlow kernels themselves (Eigen/oneDNN conv, pocketfft, cumsum ...);
    their published semantics are restated in the shim. "vs TF-CPU" is therefore structural.

The noise channel is an explicit input: TensorFlow's Philox stream cannot be
reproduced, so parity is defined with injected noise.
"""
import re

import numpy as np

LOG_TO_DB = 20 * np.log10(np.exp(1))


# ============================================================================================
# primitive ops
# ============================================================================================
def fold_weight_norm(v, g, dtype):
    """W[k,ci,co] = g[co] * v[k,ci,co] / sqrt(max(sum_{k,ci} v^2, 1e-12)).
    tf2_components/layers/conv_layers.py:149-153 (tf.nn.l2_normalize over axes [0,1])."""
    v = np.asarray(v, dtype=np.float64)
    sq = np.sum(v * v, axis=(0, 1), keepdims=True)
    w = np.asarray(g, dtype=np.float64) * (v / np.sqrt(np.maximum(sq, 1e-12)))
    return w.astype(dtype)


def pad_time(x, left, right, mode):
    """custom_layers.py:47-71 (TFPad1d) ; mode in CONSTANT / SYMMETRIC / EDGE. x: (B,T,C)."""
    if left == 0 and right == 0:
        return x
    np_mode = {"CONSTANT": "constant", "SYMMETRIC": "symmetric", "EDGE": "edge"}[mode]
    return np.pad(x, ((0, 0), (left, right), (0, 0)), mode=np_mode)


def conv1d_valid(x, w, b=None, dilation=1):
    """Keras Conv1D, VALID, stride 1: y[t,co] = b[co] + sum_{j,ci} x[t + j*d, ci] W[j,ci,co]
    (cross-correlation).  Call site conv_layers.py:154."""
    ks = w.shape[0]
    t_out = x.shape[1] - (ks - 1) * dilation
    y = None
    for jj in range(ks):
        part = x[:, jj * dilation: jj * dilation + t_out, :] @ w[jj]
        y = part if y is None else y + part
    if b is not None:
        y = y + b
    return y


def conv1d_same_zero(x, w, b=None, dilation=1):
    """Keras Conv1D padding="same": zero pad d*(k-1) split left-floor / right-ceil.
    custom_AE_layers.py:236-243 (WaveNet layers), :214-227 (cond conv)."""
    total = (w.shape[0] - 1) * dilation
    xp = pad_time(x, total // 2, total - total // 2, "CONSTANT")
    return conv1d_valid(xp, w, b, dilation)


def conv1d_causal(x, w, b=None, dilation=1):
    """Keras Conv1D padding="causal": d*(k-1) zeros in front, none behind (pp_mod_subnet.padding: CAUSAL)."""
    return conv1d_valid(pad_time(x, (w.shape[0] - 1) * dilation, 0, "CONSTANT"), w, b, dilation)


def lin_interp(x, up, weights_dtype=np.float32):
    """TF2C_LinInterpLayer(num_pad_end=1, drop_last=True): support_layers.py:19-27,99-121.
    out[t*U+u] = x[t]*(U-u)/U + x[min(t+1,T-1)]*u/U ; the two weight vectors are float32 constants."""
    uu = np.arange(up)
    w0 = ((up - uu) / up).astype(weights_dtype).astype(x.dtype)
    w1 = (uu / up).astype(weights_dtype).astype(x.dtype)
    nxt = np.concatenate((x[:, 1:], x[:, -1:]), axis=1)
    out = x[:, :, None, :] * w0[None, None, :, None] + nxt[:, :, None, :] * w1[None, None, :, None]
    return out.reshape(x.shape[0], x.shape[1] * up, x.shape[2])


def depth_to_time(x, factor):
    """conv_layers.py:250-256: row-major reshape (B,T,f*F) -> (B,T*f,F)."""
    return x.reshape(x.shape[0], x.shape[1] * factor, -1)


def prelu(x, alpha):
    """Keras PReLU(shared_axes=[1]): x>0 ? x : alpha_c*x. custom_pulsed_generator.py:247-250."""
    return np.where(x > 0, x, alpha * x)


def soft_sigmoid(x):
    """custom_AE_layers.py:91-99."""
    return 0.5 + 0.5 * x / (1 + np.abs(x))


_FINAL_ACTS = {
    "soft_sigmoid": soft_sigmoid,
    "tanh": np.tanh,
    "sigmoid": lambda x: 1 / (1 + np.exp(-x)),
    "soft_sign": lambda x: x / (1 + np.abs(x)),
    "soft_sqrt": lambda x: x / (1 + np.sqrt(np.abs(x))),
    "exp": np.exp,
    "relu": lambda x: np.maximum(x, 0),
    "linear": lambda x: x,
}


# ============================================================================================
# the model
# ============================================================================================
class OracleModel:
    """Numpy restatement of ``MBExWN`` (custom_pulsed_generator.py:151-925) + ``PaNWaveNet.infer``
    (wavegen_1d.py:483-526) for inference.

    config     : dict with the reference's YAML keys (mbexwn_config / preprocess_config)
    raw_weights: dict name -> array with the un-folded variables ``<layer>.v``, ``<layer>.g``,
                 ``<layer>.bias`` and ``<act>.alpha``
    wavetables : object with attributes tables (n_period+1,R) float32, n_period, nominalF0,
                 min_transposition, max_transposition, grid_norm  (init-time constants; golden-pinned)
    """

    def __init__(self, config, raw_weights, wavetables, dtype=np.float64, float32_constants=True):
        """float32_constants=True (default): everything the reference holds or computes as a float32
        quantity that feeds an index, a phase or a constant table is evaluated in float32 exactly like
        the float32 graph does.  False: the same graph entirely in float64 ("structural" mode, used only
        for the comparison with the float64 run of the reference's model code)."""
        self.dtype = dtype
        self.f32 = np.float32 if float32_constants else np.float64
        self.cfg = config
        mb = config["mbexwn_config"]
        self.mb = mb
        pp = config["preprocess_config"]
        self.sample_rate = pp["sample_rate"]
        self.hop = pp["hop_size"]
        self.mel_channels = pp["mel_channels"]
        self.M = mb["multi_band_config"]["subbands"]
        self.pulse_rate = self.sample_rate / mb["pulse_rate_factor"]
        self.pulse_channels = mb["pulse_channels"]
        self.wt_cfg = mb.get("wavetable_config", {}) or {}
        self.steps_per_frame = self.hop // self.M                                   # :265
        self.pulse_per_frame = (self.steps_per_frame * self.pulse_channels) // int(np.prod(mb.get("pp_mod_subnet_upsampling_factors", [1])))   # :266
        self.sigma = mb.get("pp_mod_subnet_noise_channel_sigma", 0.5)
        self.f_min = mb.get("pp_min_frequency", 40.0)
        self.f_max = mb.get("pp_max_frequency", 600.0)
        self.wn = dict(mb["pp_mod_subnet"])
        self.n_ceps = mb.get("ps_max_ceps_coefs", 120)
        self.env_scale = mb.get("ps_env_order_scale", None)
        self.use_ceps_constraint = mb.get("psns_use_cepstral_loss_constraint", False)
        rng_db = mb.get("filter_max_db_range", None)
        self.max_log_range = rng_db / LOG_TO_DB if rng_db is not None else None      # :371
        self.preserve_energy = bool(mb.get("spect_filters_preserve_energy", False))     # :817-849
        self.stft_win = 4 * self.hop                                                # :396
        fft_size = 16
        while fft_size < self.stft_win:
            fft_size *= 2
        self.fft_size = fft_size                                                    # :397-400
        self.wt = wavetables
        self.raw = raw_weights
        self.use_prelu = mb.get("use_prelu", True)
        self.alpha = mb.get("alpha", 0.2)
        self.pp_valid = mb.get("pp_subnet_use_valid_padding", False)
        self.ps_valid = mb.get("ps_subnet_use_valid_padding", False)
        self.pp_specs = mb["pp_subnet"]
        self.ps_specs = mb["ps_subnet"]
        self.pp_activation = mb.get("pp_activation", "soft_sigmoid")
        self._w = {}
        self._build_constants(mb)

    # ------------------------------------------------------------------ constants
    def _build_constants(self, mb):
        dt = self.dtype
        # PQMF synthesis bank -- tf_preprocess.py:30-80,119-161
        mbc = mb["multi_band_config"]
        taps, cutoff, beta = mbc["taps"], mbc["cutoff_ratio"], mbc["beta"]
        nn = np.arange(taps + 1) - 0.5 * taps
        with np.errstate(invalid="ignore", divide="ignore"):
            proto = np.sin(np.pi * cutoff * nn) / (np.pi * nn)
        proto[taps // 2] = cutoff
        proto = proto * np.kaiser(taps + 1, beta)
        kk = np.arange(self.M)[:, None]
        syn = 2 * proto[None, :] * np.cos((2 * kk + 1) * (np.pi / (2 * self.M)) * nn[None, :]
                                          - (-1.0) ** kk * np.pi / 4)
        self.pqmf_taps = taps
        self.pqmf_syn = syn.astype(np.float32).astype(dt)          # (M, taps+1), float32 constants
        # STFT windows -- tf.signal.hann_window(periodic) / inverse_stft_window_fn (TensorFlow)
        n = self.stft_win
        hann = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n))
        self.hann = hann.astype(self.f32).astype(dt)
        den = np.square(self.hann).reshape(n // self.hop, self.hop).sum(axis=0)
        self.inv_win = (self.hann / np.tile(den, n // self.hop)).astype(dt)
        # cepstral lifter table -- custom_pulsed_generator.py:434-450
        if self.env_scale:
            rows, logs = [], []
            for f0 in np.logspace(np.log10(self.f_min), np.log10(self.f_max), 30):
                win_len = int(self.env_scale * 0.5 * self.sample_rate / f0)
                if win_len % 2 == 0:
                    win_len += 1
                logs.append(np.log10(f0))
                half = np.hamming(win_len)[win_len // 2:]
                if win_len // 2 + 1 > self.n_ceps:
                    rows.append(half[:self.n_ceps])
                else:
                    rows.append(np.concatenate((half, np.zeros(self.n_ceps - 1 - win_len // 2))))
            self.ceps_log10f0 = np.asarray(logs, dtype=self.f32)
            self.ceps_windows = np.asarray(rows, dtype=self.f32).astype(dt)
        # F0 smoothing kernel -- :403-406
        sw = np.bartlett(2 * self.hop + 3)[1:-1]
        self.f0_smooth = (sw / np.sum(sw)).astype(self.f32)

    def weight(self, name):
        """folded (W, b) of a conv layer, cached.  conv_layers.py:133-153: weight norm W = g v / ||v||; with
        use_equalized_lr (WaveNetAE layers only, custom_AE_layers.py:177-260) W = g v / sqrt(mean_{k,ci} v^2), or -- without
        weight norm -- the whole layer output is multiplied by g: W = g K, b = g bias."""
        if name not in self._w:
            v = np.asarray(self.raw[name + ".v"], dtype=np.float64)
            b = np.asarray(self.raw[name + ".bias"]).astype(self.dtype)
            eq = re.match(r"wn\d*\.", name) is not None and bool(self.wn.get("use_equalized_lr", False))
            if eq and self.wn.get("use_weight_norm", False):
                g = np.asarray(self.raw[name + ".g"], dtype=np.float64)
                w = (g * v / np.sqrt(np.mean(v * v, axis=(0, 1), keepdims=True))).astype(self.dtype)
            elif eq:
                g = np.asarray(self.raw[name + ".g"], dtype=np.float64)
                w = (g * v).astype(self.dtype)
                b = (g * np.asarray(self.raw[name + ".bias"], dtype=np.float64)).astype(self.dtype)
            elif name + ".g" in self.raw:
                w = fold_weight_norm(v, self.raw[name + ".g"], self.dtype)
            else:                              # a WaveNet layer built without weight norm: the kernel is the weight
                w = v.astype(self.dtype)
            self._w[name] = (w, b)
        return self._w[name]

    # ------------------------------------------------------------------ sub-nets (A2)
    def run_subnet(self, x, specs, base_name, final_n_channels, final_nks, final_activation,
                   target_ups=None, pad_to_valid=False):
        """custom_pulsed_generator.py:38-148 executed directly on the spec list."""
        total_ups = 1
        explicit = "EDGE" if pad_to_valid else "SYMMETRIC"
        for ii, spec in enumerate(specs):
            if spec[0] == "L":                                                   # :57-60
                x = lin_interp(x, int(spec[1]), self.f32)
                continue
            ks, up, linear_up = int(spec[0]), 1, False
            if len(spec) > 2:
                if isinstance(spec[2], str):
                    linear_up = spec[2][0] == "L"
                    up = int(spec[2][1:])
                else:
                    up = int(spec[2])
            w, b = self.weight(f"{base_name}_Layer_{ii}")
            pl, pr = (ks - 1) // 2 + ((ks - 1) % 2), (ks - 1) // 2
            if linear_up:                                                        # :74-89
                x = lin_interp(conv1d_valid(pad_time(x, pl, pr, explicit), w, b), up, self.f32)
            elif up > 1:                                                         # :91-108
                if pad_to_valid:
                    x = conv1d_valid(pad_time(x, pl, pr, "EDGE"), w, b)
                else:
                    x = conv1d_same_zero(x, w, b)
                x = depth_to_time(x, up)
            else:                                                                # :109-122
                x = conv1d_valid(pad_time(x, pl, pr, explicit), w, b)
            if self.use_prelu:                                                   # :123
                x = prelu(x, np.asarray(self.raw[f"{base_name}_ActLayer_{ii}.alpha"]).astype(self.dtype))
            else:
                x = np.where(x > 0, x, self.alpha * x)
            total_ups *= up
        if final_nks is not None:                                                # :126-138
            w, b = self.weight(f"{base_name}_Layer_final")
            if pad_to_valid:
                fk = final_nks
                x = conv1d_valid(pad_time(x, (fk - 1) // 2 + ((fk - 1) % 2), (fk - 1) // 2, "EDGE"), w, b)
            else:
                x = conv1d_same_zero(x, w, b)
            if target_ups is not None and total_ups != target_ups:               # :140-144
                up = target_ups // total_ups
                if total_ups * up != target_ups:
                    raise RuntimeError("Upsampling to target upsampling factor is not possible")
                x = lin_interp(x, up, self.f32)
            if final_activation is not None:                                     # :145-146
                x = _FINAL_ACTS[final_activation](x)
        return x

    # ------------------------------------------------------------------ F0 (A3)
    def generate_f0(self, mel):
        """custom_pulsed_generator.py:773-791 -> (B, T*pulse_per_frame) in Hz."""
        x = self.run_subnet(mel, self.pp_specs, "PulsPar", 1, 1, self.pp_activation,
                            target_ups=self.pulse_per_frame, pad_to_valid=self.pp_valid)
        f0 = x[:, :, 0] * (self.f_max - self.f_min) + self.f_min
        return f0[:, :mel.shape[1] * self.pulse_per_frame]

    # ------------------------------------------------------------------ wavetable (A4)
    def phase_from_f0(self, f0, chunk_size=1000):
        """tf_wavetable.py:429-492 (stable_cumsum_and_wrap), float32 with TensorFlow-CPU's
        sequential accumulation order (tf.cumsum = running sum along the axis)."""
        ft = self.f32
        vel = (np.asarray(f0, dtype=ft) / ft(self.pulse_rate)).astype(ft)          # :516
        n_batch, n_time = vel.shape
        rem = n_time % chunk_size
        if rem:
            vel = np.pad(vel, ((0, 0), (0, chunk_size - rem)))
        n_chunks = vel.shape[1] // chunk_size
        chunks = vel.reshape(n_batch, n_chunks, chunk_size)
        phase = np.cumsum(chunks, axis=2, dtype=ft)                                # sequential running sum
        offsets = np.mod(phase[:, :, -1:], ft(1))
        offsets = np.pad(offsets, ((0, 0), (1, 0), (0, 0)))[:, :-1]
        offsets = np.mod(np.cumsum(offsets, axis=1, dtype=ft), ft(1))
        phase = np.mod(phase + offsets, ft(1)).astype(ft)
        return phase.reshape(n_batch, -1)[:, :n_time]

    def wavetable(self, f0):
        """tf_wavetable.py:495-552 + _linear_lookup :605-638, float32 like the reference
        (positions / indices are float32-exact, the two lerps are evaluated in self.dtype)."""
        dt = self.dtype
        ft = self.f32
        f0_32 = np.asarray(f0, dtype=ft)
        phase = self.phase_from_f0(f0_32)
        n_sub = int(self.wt_cfg.get("add_subharm_chans", 0) or 0)
        if self.wt_cfg.get("use_sinusoid_as_fun", False) or n_sub:
            w2pi = ((phase * ft(2)) * ft(np.pi)).astype(ft)                         # :521, float32 left to right
        if n_sub:                                                                  # :554-559: (B, N, 1 + n_sub)
            subs = [np.sin((w2pi / ft(ii)).astype(ft).astype(dt)) for ii in range(2, n_sub + 2)]
            return np.stack([self._pulse_channel(f0_32, phase, w2pi if self.wt_cfg.get("use_sinusoid_as_fun", False) else None)] + subs, axis=-1)
        return self._pulse_channel(f0_32, phase, w2pi if self.wt_cfg.get("use_sinusoid_as_fun", False) else None)

    def _pulse_channel(self, f0_32, phase, w2pi):
        dt, ft = self.dtype, self.f32
        if w2pi is not None:                                                       # :522-523 use_sinusoid_as_fun
            w = w2pi.astype(dt)
            return np.sin(w) * 0.5 * (1.0 - np.cos(w))
        pos = (phase * ft(self.wt.n_period)).astype(ft)                            # :619
        base = np.floor(pos)
        rem = (pos - base).astype(ft).astype(dt)                                   # :630
        idx = base.astype(np.int64)
        tab = np.asarray(self.wt.tables).astype(dt)
        samples = tab[idx] * (1.0 - rem)[..., None] + tab[idx + 1] * rem[..., None]  # (B,N,R)  :633-638
        ratio = np.maximum(ft(self.wt.min_transposition),
                           np.minimum(ft(self.wt.max_transposition), f0_32 / ft(self.wt.nominalF0)))
        q = np.log(ratio.astype(ft)).astype(ft) * ft(self.wt.grid_norm)            # :539-543
        diff = q.astype(dt)[..., None] - np.arange(tab.shape[1], dtype=dt)
        mix = np.maximum(1 - np.abs(diff), 0)
        return np.sum(samples * mix, axis=2)                                       # :548

    # ------------------------------------------------------------------ WaveNet (A5-A9)
    def dilation(self, index):
        step = self.wn.get("dilation_rate_step", 1)
        mx = self.wn.get("max_log2_dilation_rate", None)
        if mx is not None:
            return 2 ** (int(index // step) % mx)
        return 2 ** int(index // step)                                            # custom_AE_layers.py:229-233

    def wn_conv(self, x, w, b=None, dilation=1):
        """The convolutions that take pp_mod_subnet.padding (custom_AE_layers.py:192-260, 519-524): SAME or CAUSAL."""
        if str(self.wn.get("padding", "SAME")).upper() == "CAUSAL":
            return conv1d_causal(x, w, b, dilation)
        return conv1d_same_zero(x, w, b, dilation)

    def conditioning(self, mel, prefix="wn.", rate_factor=1):
        """custom_AE_layers.py:214-227,287-289: sub-pixel conv (factor cond_conv_upsampling) then LinInterp.  ``prefix`` /
        ``rate_factor``: the block's tensors and its rate relative to the first block (custom_pulsed_generator.py:484,488)."""
        lin_up = self.wn.get("cond_lin_upsampling", 16)
        conv_up = int((self.pulse_rate / self.pulse_channels * rate_factor) // ((self.sample_rate / self.hop) * lin_up))
        x = mel
        for ii in range(len(self.wn.get("pre_cond_layer_channels", None) or [])):    # :192-201, 283-285: plain convolutions
            w, b = self.weight(f"{prefix}precond_{ii}")
            x = self.wn_conv(x, w, b)
        w, b = self.weight(prefix + "cond")
        c = depth_to_time(self.wn_conv(x, w, b), conv_up)
        return lin_interp(c, lin_up, self.f32)

    def wavenet(self, x, mel, return_layers=False, prefix="wn.", channels=None, rate_factor=1):
        """custom_AE_layers.py:273-346 (WaveNetAE.call), activation gtu / gfu / gsu / glu; n_ch_groups independent channel groups between
        the shared start and end convolutions (:303-340; layers of group g > 0 are named "<layer>g<g>", :249,260)."""
        C = self.wn["n_channels"] if channels is None else channels
        L = self.wn.get("n_layers", 12)
        G = int(self.wn.get("n_ch_groups", 1))
        Cg = C // G
        w, b = self.weight(prefix + "start")
        started = np.split(conv1d_valid(x, w, b), G, axis=-1)                     # :280, :303-304
        started = [np.array(ss) for ss in started]
        if self.wn.get("disable_conditioning", False):                            # :293-294: zeros
            cond = [np.zeros((), dtype=self.dtype)] * G
        else:
            cond = np.split(self.conditioning(mel, prefix, rate_factor), G, axis=-1)   # :287-289
        output = [None] * G
        acts = []
        for ll in range(L):
            for gg in range(G):
                sfx = f"g{gg}" if gg else ""
                w, b = self.weight(f"{prefix}conv1D_{ll}{sfx}")
                z = self.wn_conv(started[gg], w, b, dilation=self.dilation(ll)) + cond[gg]   # :307-309
                zt = z[..., :Cg]
                act = self.wn.get("activation", "gtu")
                if act == "gtu":                                                  # :312-318
                    half = np.tanh(zt)
                elif act == "gfu":
                    half = zt / (1 + np.abs(zt))
                elif act == "gsu":
                    half = zt / (1 + np.sqrt(np.abs(zt)))
                elif act == "glu":                                                # accepted at :156, no branch: linear half
                    half = zt
                else:
                    raise NotImplementedError(f"WaveNetAE activation {act}")
                a = half * (1 / (1 + np.exp(-z[..., Cg:])))                       # :320-321
                w, b = self.weight(f"{prefix}res_skip_{ll}{sfx}")
                r = conv1d_valid(a, w, b)                                         # :324
                if ll < L - 1:
                    started[gg] = started[gg] + r[..., :Cg]                       # :326-328
                    s = r[..., Cg:]
                else:
                    s = r                                                         # :330
                output[gg] = s if output[gg] is None else output[gg] + s          # :332-335
                if return_layers:
                    acts.append(a)
        skip = np.concatenate(output, axis=-1) if G > 1 else output[0]            # :337-340
        w, b = self.weight(prefix + "end")
        out = conv1d_valid(skip, w, b)
        if return_layers:
            h = np.concatenate(started, axis=-1) if G > 1 else started[0]
            if G > 1:       # per layer: the groups' gate outputs side by side, as the dense layout of the HIP path holds them
                acts = [np.concatenate(acts[ll * G:(ll + 1) * G], axis=-1) for ll in range(L)]
                cond = [np.concatenate([cc[..., :Cg] for cc in cond] + [cc[..., Cg:] for cc in cond], axis=-1)]
            return out, h, skip, acts, cond[0]
        return out

    # ------------------------------------------------------------------ PQMF (A10)
    def pqmf_synthesis(self, x):
        """tf_preprocess.py:208-226: zero-stuff by M with gain M, zero-pad taps/2, cross-correlate."""
        B, S, K = x.shape
        M, taps = self.M, self.pqmf_taps
        up = np.zeros((B, S * M + taps, K), dtype=self.dtype)
        up[:, taps // 2: taps // 2 + S * M: M, :] = M * x
        y = np.zeros((B, S * M), dtype=self.dtype)
        for b in range(B):
            for k in range(K):
                y[b] += np.correlate(up[b, :, k], self.pqmf_syn[k], mode="valid")
        return y

    # ------------------------------------------------------------------ excitation (A5)
    def generate_excitation(self, mel, f0, noise):
        """custom_pulsed_generator.py:886-925 ; noise (B, T*steps_per_frame) ~ N(0,1) or None (sigma=0)."""
        pulse = self.wavetable(f0)                                                # :889
        n_sub = int(self.wt_cfg.get("add_subharm_chans", 0) or 0)
        if self.mb.get("pulse_channels_use_pqmf", False):                         # :894-895, tf_preprocess.py:188-200
            pq = self.mb["pulse_channels_multi_band_config"]
            ana = pqmf_analysis_bank(pq["subbands"], pq["taps"], pq["cutoff_ratio"], pq["beta"]).astype(self.dtype)
            K, taps = ana.shape[1], ana.shape[0] - 1
            padded = np.pad(pulse.astype(self.dtype), ((0, 0), (taps // 2, taps // 2)))
            x = np.stack([np.stack([np.correlate(padded[bb], ana[:, kk], mode="valid")[::K] for kk in range(K)], axis=-1)
                          for bb in range(pulse.shape[0])], axis=0)
        else:
            x = pulse.reshape(pulse.shape[0], -1, self.pulse_channels * (1 + n_sub)).astype(self.dtype)   # :893
        if self.sigma:
            if noise is None:
                raise ValueError("noise must be given when pp_mod_subnet_noise_channel_sigma != 0")
            nz = np.asarray(noise).astype(self.dtype)[:, :x.shape[1], None]
            x = np.concatenate((x, self.sigma * nz), axis=-1)                     # :905-906
        # :908-910, 456-488: one WaveNet block per up-sampling factor; block b has n_channels * channel_factors[b] channels
        # and an up-sampling convolution (k = 3, SAME, depth -> time; custom_AE_layers.py:519-524, 574-582) behind it
        ups = [int(uu) for uu in self.mb.get("pp_mod_subnet_upsampling_factors", [1])]
        chf = list(self.mb.get("pp_mod_subnet_channel_factors", [1]))
        y, rate = x, 1
        for bb, (uu, ff) in enumerate(zip(ups, chf)):
            prefix = "wn." if bb == 0 else f"wn{bb}."
            y = self.wavenet(y, mel, prefix=prefix, channels=int(self.wn["n_channels"] * ff), rate_factor=rate)
            if uu > 1:
                w, b = self.weight(f"up{bb}")
                y = depth_to_time(self.wn_conv(y, w, b), uu)
            rate *= uu
        w, b = self.weight("post")
        y = conv1d_valid(y, w, b)                                                 # :913-914
        if not self.mb.get("ps_use_stft", True) and not self.mb.get("ps_off", False):
            # :857-884: one log gain per sub-band from the VTF-net (mean over the bands removed with preserve_energy), exp;
            # :453,670: interpolated by hop_size (edge frame repeated); :916-917: its FIRST rows multiply the sub-band rows
            lg = self.run_subnet(mel, self.ps_specs, "PS", self.M, 1, None, pad_to_valid=self.ps_valid)
            if self.preserve_energy:
                lg = lg - np.mean(lg, axis=-1, keepdims=True)
            gain = lin_interp(np.exp(lg), self.hop, self.f32)
            y = y * gain[:, :y.shape[1]]
        if not self.mb.get("pp_mod_subnet_use_pqmf", True):                       # :922-923: no PQMF, a reshape
            return y.reshape(y.shape[0], y.shape[1] * y.shape[2])
        return self.pqmf_synthesis(y)                                             # :920-921

    # ------------------------------------------------------------------ envelope (A12)
    def cepstral_window_index(self, f0, return_position=False):
        """custom_pulsed_generator.py:507-525 (float32 like the reference; returns int indices (B,T)).  return_position: also the
        fractional row position the index is rounded from (the selection is the nearest row: discontinuous at the midpoints)."""
        ft = self.f32
        f0 = np.asarray(f0, dtype=ft)
        half = self.f0_smooth.shape[0] // 2
        padded = np.concatenate((np.repeat(f0[:, :1], half, axis=1), f0, np.repeat(f0[:, -1:], half, axis=1)), axis=1)
        stride = self.pulse_per_frame
        n_out = (padded.shape[1] - self.f0_smooth.shape[0]) // stride + 1
        sm = np.empty((f0.shape[0], n_out), dtype=ft)
        for t in range(n_out):
            seg = padded[:, t * stride: t * stride + self.f0_smooth.shape[0]]
            sm[:, t] = np.sum(seg.astype(np.float64) * self.f0_smooth.astype(np.float64), axis=1)
        lg = (ft(1 / np.log(10)) * np.log(sm)).astype(ft)
        lg = np.minimum(np.maximum(lg, self.ceps_log10f0[0]), self.ceps_log10f0[-1])
        ratio = (lg - self.ceps_log10f0[0]) / (self.ceps_log10f0[-1] - self.ceps_log10f0[0])
        pos = ratio * ft(self.ceps_log10f0.shape[0] - 1)
        idx = np.rint(pos).astype(np.int64)                                            # tf.round: half to even
        return (idx, pos) if return_position else idx

    def generate_specenv(self, mel, f0, window_index=None):
        """custom_pulsed_generator.py:793-855 -> complex (B,T,fft/2+1)."""
        x = self.run_subnet(mel, self.ps_specs, "PS", self.n_ceps, 1, None, pad_to_valid=self.ps_valid)
        if self.env_scale and not self.use_ceps_constraint:
            if window_index is None:
                window_index = self.cepstral_window_index(f0)
            x = x * self.ceps_windows[window_index]                               # :813
        ceps = np.zeros(x.shape[:2] + (self.fft_size,), dtype=self.dtype)
        first = 0 if self.preserve_energy else 1                                  # :817-826
        ceps[:, :, first:self.n_ceps] = x[:, :, first:]
        spec = np.fft.rfft(ceps, axis=-1)                                         # :829
        if self.max_log_range:
            filt = np.exp(self.max_log_range * np.tanh(spec.real) + 1j * spec.imag)   # :831-834
        else:
            filt = np.exp(spec)
        if self.preserve_energy:                                                  # :838-843
            filt = filt / np.sqrt(np.mean(np.square(np.abs(filt)), axis=-1, keepdims=True))
        return filt

    # ------------------------------------------------------------------ STFT filter (A11,A13)
    def stft(self, exc, n_frames):
        """custom_pulsed_generator.py:681-694 (tf.signal.stft, pad_end=False, first n_frames kept)."""
        n, hop = self.stft_win, self.hop
        padded = np.pad(exc, ((0, 0), (n // 2, n // 2 + hop + 1)))
        total = 1 + (padded.shape[1] - n) // hop
        idx = np.arange(n)[None, :] + hop * np.arange(total)[:, None]
        frames = padded[:, idx] * self.hann
        return np.fft.rfft(frames, n=self.fft_size, axis=-1)[:, :n_frames]

    def istft(self, spec, out_len):
        """custom_pulsed_generator.py:716-724 (tf.signal.inverse_stft + inverse_stft_window_fn + slice)."""
        n, hop = self.stft_win, self.hop
        frames = np.fft.irfft(spec, n=self.fft_size, axis=-1)[..., :n] * self.inv_win
        B, T = frames.shape[:2]
        sig = np.zeros((B, (T - 1) * hop + n), dtype=self.dtype)
        for t in range(T):
            sig[:, t * hop: t * hop + n] += frames[:, t]
        return sig[:, n // 2: n // 2 + out_len]

    # ------------------------------------------------------------------ full graph (A1)
    def forward(self, mel, noise=None, return_stages=False):
        """MBExWN.call (custom_pulsed_generator.py:556-771, inference branch) wrapped like
        PaNWaveNet.infer (wavegen_1d.py:483-526) with synth_length = T*hop.
        mel (B,T,80) ; noise (B, T*steps_per_frame). Returns audio (B, T*hop)."""
        mel = np.asarray(mel).astype(self.dtype)
        T = mel.shape[1]
        f0 = self.generate_f0(mel)                                                # :567
        exc = self.generate_excitation(mel, f0, noise)                            # :676
        if self.mb.get("ps_off", False) or not self.mb.get("ps_use_stft", True):  # :663-672: the signal is the excitation
            audio = exc[:, :T * self.hop]
            return (audio, {"f0": f0, "excitation": exc}) if return_stages else audio
        src = self.stft(exc, T)                                                   # :681-694
        env = self.generate_specenv(mel, f0)                                      # :704
        out_len = f0.shape[1] * int(self.sample_rate // self.pulse_rate)
        audio = self.istft(src * env, out_len)[:, :T * self.hop]                  # :715-724, wavegen_1d.py:504-510
        if return_stages:
            return audio, {"f0": f0, "excitation": exc, "envelope": env}
        return audio


def pqmf_analysis_bank(subbands, taps, cutoff, beta):
    """Cosine-modulated analysis bank (taps + 1, subbands), float32 constants -- tf_preprocess.py:30-80,119-150."""
    nn = np.arange(taps + 1) - 0.5 * taps
    with np.errstate(invalid="ignore", divide="ignore"):
        proto = np.sin(np.pi * cutoff * nn) / (np.pi * nn)
    proto[taps // 2] = cutoff
    proto = proto * np.kaiser(taps + 1, beta)
    kk = np.arange(subbands)[:, None]
    ana = 2 * proto[None, :] * np.cos((2 * kk + 1) * (np.pi / (2 * subbands)) * nn[None, :] + (-1.0) ** kk * np.pi / 4)
    return ana.T.astype(np.float32)


def synthetic_mel(rng, batch, frames, channels=80):
    """SURVEY.md section 8(d) synthetic input: mell = log(exp(N(-5,2^2)) + 1e-5), float32, clipped."""
    mell = np.log(np.exp(rng.normal(-5.0, 2.0, size=(batch, frames, channels))) + 1e-5)
    return np.clip(mell, -11.5, 2.0).astype(np.float32)


# ============================================================================================
# optional RMS normalisation (row A14)
# ============================================================================================
def slaney_mel_frequencies(n_mels, fmin, fmax):
    """librosa.mel_frequencies(htk=False) restated from the published Slaney formulas (third party, absent here):
    200/3 Hz per mel below 1 kHz, logarithmic above with step ln(6.4)/27."""
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp

    def to_mel(f):
        return min_log_mel + np.log(f / min_log_hz) / logstep if f >= min_log_hz else f / f_sp

    mels = np.linspace(to_mel(float(fmin)), to_mel(float(fmax)), n_mels)
    return np.where(mels >= min_log_mel, min_log_hz * np.exp(logstep * (mels - min_log_mel)), f_sp * mels)


def slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(htk=False, norm="slaney") restated from the published construction (third party, absent
    here; reference preprocess.py:52-74 calls it): triangles between neighbouring Slaney mel frequencies over the
    n_fft // 2 + 1 bin frequencies, each divided by half its band width.  Returns (n_mels, n_fft // 2 + 1) float32."""
    freqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_f = slaney_mel_frequencies(n_mels + 2, fmin, fmax)
    basis = np.zeros((n_mels, freqs.shape[0]))
    for ii in range(n_mels):
        up = (freqs - mel_f[ii]) / (mel_f[ii + 1] - mel_f[ii])
        down = (mel_f[ii + 2] - freqs) / (mel_f[ii + 2] - mel_f[ii + 1])
        basis[ii] = np.maximum(0.0, np.minimum(up, down)) * (2.0 / (mel_f[ii + 2] - mel_f[ii]))
    return basis.astype(np.float32)


def normalize_inputs_by_rms(mell, config, synth_length, dtype=np.float64):
    """wavegen_1d.py:638-769 (NormMelComponents.normalize_inputs_by_rms, audio=None, smoothing variant; both RMS
    estimates: the band-width weighted one and normalize_use_pinv, :603-611, 683-689) with the constructor constants of
    :580-636.  Returns (mell', gain (B, synth_length))."""
    pp, mb = config["preprocess_config"], config["mbexwn_config"]
    hop, win, n_mels = pp["hop_size"], pp.get("win_size", pp["fft_size"]), pp["mel_channels"]
    iters = mb.get("normalize_rms_num_smooth_iters", 0)
    assert 4 * hop == win and iters > 0
    eps = 1e-7

    def hann(n):     # sig_proc/Mwindows.py:176-185 (symmetric, zero end points)
        w = np.zeros(n)
        mid = (n - 1) // 2
        half = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(mid + 1) / (n - 1))
        w[:mid + 1] = half
        w[n - 1:n - 2 - mid:-1] = half
        return w

    mel_f = slaney_mel_frequencies(n_mels + 2, pp["fmin"], pp["fmax"])
    inv_enorm = ((mel_f[2:] - mel_f[:n_mels]) / 2.0).astype(np.float32).astype(dtype)          # :611
    rms_norm_fact = pp["fft_size"] * win * 0.5                                                   # :598
    gwin = hann(win).astype(np.float32).astype(dtype)
    gwin = gwin / np.sum(gwin)                                                                   # :622
    sws = int(win * mb.get("normalize_smooth_win_scale", 1))
    ssw = hann(sws).astype(np.float32).astype(dtype)
    if mb.get("normalize_smooth_with_squared_win", True):
        ssw = ssw ** 2                                                                           # :626-627

    def ola(frames):
        n_fr, flen = frames.shape[-2:]
        out = np.zeros(frames.shape[:-2] + ((n_fr - 1) * hop + flen,), dtype=dtype)
        for tt in range(n_fr):
            out[..., tt * hop: tt * hop + flen] += frames[..., tt, :]
        return out

    mell = np.asarray(mell).astype(dtype)
    T = mell.shape[1]
    mel = np.exp(mell)
    if mb.get("normalize_use_pinv", False):
        # :603-608: the mel filters inverted (pseudo inverse, float32 like the basis), the window's L2 norm; :684-685: the
        # minimum-energy spectrum that explains the mel frame, its energy over all bins
        win_norm = np.sqrt(np.sum(hann(win).astype(np.float32) ** 2))
        basis = slaney_mel_basis(pp["sample_rate"], pp["fft_size"], n_mels, pp["fmin"], pp["fmax"])
        inverted = np.linalg.pinv(basis).T.astype(dtype)                                         # (n_mels, bins)
        spec = np.tensordot(mel, inverted, axes=1) / win_norm
        rms = np.sqrt(np.sum(np.square(spec), axis=-1) / rms_norm_fact)
    else:
        rms = np.sqrt(np.sum(np.square(mel * inv_enorm), axis=-1) / rms_norm_fact)               # :689
    if mb.get("max_norm_fact", None):
        rms = np.maximum(rms, 1.0 / mb["max_norm_fact"])                                         # :690-691
    if mb.get("normalize_compressor_exp", None) is not None:
        rms = np.power(rms, mb["normalize_compressor_exp"])                                      # :692-693
    cut = sws // 2 + 2 * hop - win // 2
    norm_gain = ola(np.ones((1, T + 4, 1), dtype) * ssw)[:, cut:]                                # :700-705
    gain = None
    for _ in range(iters):                                                                       # :714-726
        ext = np.concatenate((rms[:, :1], rms[:, :1], rms, rms[:, -1:], rms[:, -1:]), axis=1)
        gain = ola(ext[:, :, None] * ssw)[:, cut:] / np.maximum(eps, norm_gain)
        n_out = (gain.shape[1] - win) // hop + 1
        idx = np.arange(win)[None, :] + hop * np.arange(n_out)[:, None]
        rms = np.sum(gain[:, idx] * gwin, axis=-1)[:, :T]
    mel = mel / np.maximum(eps, rms[:, :, None]) * mb.get("lin_amp_scale", 1.0)                  # :731
    off = mb.get("lin_amp_off", 1.0e-5)
    if mb.get("use_max_limit", False):
        out = mb.get("mel_amp_scale", 1.0) * np.log(np.maximum(mel, off))
    else:
        out = mb.get("mel_amp_scale", 1.0) * np.log(mel + off)                                   # :733-736
    up = np.maximum(gain[:, win // 2: win // 2 + synth_length], eps)                             # :740-742
    if up.shape[1] < synth_length:
        up = np.concatenate((up, np.repeat(up[:, -1:], synth_length - up.shape[1], axis=1)), axis=1)
    return out, up
