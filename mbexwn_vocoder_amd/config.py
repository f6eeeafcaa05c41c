"""Model configuration for the MBExWN mel-inversion path.

The reference reads a per-model ``config.yaml`` with ``config_utils.read_config``
(reference: MBExWN_NVoc/vocoder/model/config_utils.py:314-339) and hands
``mbexwn_config`` / ``preprocess_config`` to ``MBExWN.__init__``
(reference: MBExWN_NVoc/vocoder/model/custom_pulsed_generator.py:155-230).
The pretrained model directories are not part of the reference tree, so this module also
provides the canonical builder configuration of SURVEY.md section 8 (values marked
[INFERRED] there are builder defaults, never facts about the shipped models).

Everything below is config-driven: the same keys the reference consumes are consumed here.
"""
import copy
import os
import re

import numpy as np
import yaml

# model registry tokens -> WaveNet channel count (reference: MBExWN_NVoc/__init__.py:19-31)
_MODEL_CHANNELS = {"SING": 320, "SPEECH": 320, "VOICE": 340}


def canonical_config(voice_type="SPEECH", **overrides):
    """Builder-default configuration in the reference's YAML key layout.

    voice_type: SING / SPEECH (C=320) or VOICE (C=340), the three model families
    of reference MBExWN_NVoc/__init__.py:19-31.
    """
    voice_type = voice_type.upper()
    if voice_type not in _MODEL_CHANNELS:
        raise ValueError(f"unknown voice type {voice_type}")
    n_channels = _MODEL_CHANNELS[voice_type]
    cfg = {
        "use_tf25_compatible_implementation": True,
        "preprocess_config": {
            "sample_rate": 24000,
            "hop_size": 300,
            "win_size": 1200,
            "fft_size": 2048,
            "mel_channels": 80,
            "fmin": 0.0,
            "fmax": 12000.0,
            "lin_amp_scale": 1,
            "lin_amp_off": 1.0e-5,
            "mel_amp_scale": 1,
            "use_max_limit": False,
            "segment_length": 24000,
        },
        "training_config": {"ftype": "float32"},
        "mbexwn_config": {
            "pulse_rate_factor": 3,
            "pulse_channels": 5,
            "pp_subnet": [[3, 128], [3, 128], [3, 64]],
            "ps_subnet": [[3, 256], [3, 256]],
            "pp_mod_subnet": {
                "n_channels": n_channels,
                "n_layers": 5,
                "kernel_size": 3,
                "n_out_channels": 30,
                "cond_lin_upsampling": 10,
                "cond_kernel_size": 3,
                "dilation_rate_step": 1,
                "max_log2_dilation_rate": None,
                "n_ch_groups": 1,
                "activation": "gtu",
                "use_weight_norm": True,
            },
            "pp_mod_subnet_upsampling_factors": [1],
            "pp_mod_subnet_channel_factors": [1],
            "multi_band_config": {"subbands": 15, "taps": 120, "cutoff_ratio": 0.0421, "beta": 9.0},
            "pp_min_frequency": 40.0,
            "pp_max_frequency": 600.0,
            "pp_activation": "soft_sigmoid",
            "pp_mod_subnet_noise_channel_sigma": 0.5,
            "ps_max_ceps_coefs": 240,
            "ps_env_order_scale": 1.0,
            "filter_max_db_range": 50.0,
            "spect_filters_preserve_energy": False,
            "psns_use_cepstral_loss_constraint": False,
            "use_prelu": True,
            "alpha": 0.2,
            "wavetable_config": {
                "nominalF0": 40.0,
                "maxF0": 600.0,
                "F0GridFactor": 1.25,
                "wt_oversampling": 2,
                "Oq": 0.5,
                "am": 0.8,
                "rta": 0.05,
                "use_radiation": True,
            },
        },
    }
    for kk, vv in overrides.items():
        _set_path(cfg, kk, vv)
    return cfg


def _set_path(cfg, path, value):
    """``a:b:c`` style override, the subset of the reference override mini-language
    (reference config_utils.py:193-229) that addresses nested dict keys."""
    keys = path.split(":")
    node = cfg
    for kk in keys[:-1]:
        node = node[kk]
    node[keys[-1]] = value


# --------------------------------------------------------------------------------------
# YAML reading with the reference's ``__defaults__`` / include / env-expansion semantics
# --------------------------------------------------------------------------------------
_TYPE_NAMES = {"tf.float32": "float32", "tf.float16": "float16", "np.float32": "float32",
               "np.float16": "float16", "None": None}


def _expand(value, base_dir):
    # reference config_utils.py:33-60 (_fill_format)
    if isinstance(value, str):
        if value in _TYPE_NAMES:
            return _TYPE_NAMES[value]
        if "$" in value:
            value = os.path.expandvars(value)
        if "~" in value:
            value = os.path.expanduser(value)
        stripped = value.strip()
        mapped = re.sub("<@CONFIG_DIR@/(.*)>$", f"{base_dir}/\\1", stripped)
        if mapped != stripped:
            file_name, *keys = mapped.split(":")
            value = read_config(file_name, config_base_dir=base_dir)
            for kk in keys:
                value = value[kk]
        return value
    if isinstance(value, dict):
        return {kk: _expand(vv, base_dir) for kk, vv in value.items()}
    if isinstance(value, list):
        return [_expand(vv, base_dir) for vv in value]
    return value


def _fill_defaults(config):
    # reference config_utils.py:271-312
    for kk, vv in list(config.items()):
        if kk == "__defaults__":
            for dk, dv in vv.items():
                config.setdefault(dk, dv)
            config.pop("__defaults__")
        elif isinstance(vv, dict):
            _fill_defaults(vv)
        elif isinstance(vv, list):
            defaults = [ee for ee in vv if isinstance(ee, dict) and list(ee.keys()) == ["__defaults__"]]
            if len(defaults) > 1:
                raise RuntimeError(f"read_config::error::multiple __defaults__ entries in list {vv}")
            if defaults:
                vv.remove(defaults[0])
                for entry in vv:
                    if not isinstance(entry, dict):
                        raise RuntimeError("read_config::error::cannot use default values for list "
                                           f"entries that are not dicts {entry}")
                    for dk, dv in defaults[0]["__defaults__"].items():
                        entry.setdefault(dk, copy.deepcopy(dv))
            for entry in vv:
                if isinstance(entry, dict):
                    _fill_defaults(entry)


def read_config(config_file, config_base_dir=None):
    """Read one (or a concatenation of several) YAML config file(s).

    Mirrors reference config_utils.py:314-339: files are concatenated before parsing,
    string values get env/``~`` expansion and ``<@CONFIG_DIR@/file:key>`` includes,
    ``__defaults__`` entries are merged.
    """
    if config_base_dir is None:
        config_base_dir = os.path.dirname(os.path.abspath(
            config_file[0] if isinstance(config_file, (list, tuple)) else config_file))
    files = config_file if isinstance(config_file, (list, tuple)) else [config_file]
    text = ""
    for ff in files:
        with open(ff, "r") as fi:
            text += fi.read()
    config = yaml.safe_load(text)
    config = {kk: _expand(vv, config_base_dir) for kk, vv in config.items()}
    _fill_defaults(config)
    return config


def dump_config(config_file, config):
    with open(config_file, "w") as fo:
        yaml.safe_dump(_plain(config), fo, default_flow_style=None, sort_keys=False)


def _plain(obj):
    if isinstance(obj, dict):
        return {kk: _plain(vv) for kk, vv in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(vv) for vv in obj]
    if isinstance(obj, np.generic):
        return obj.item()
    return obj


# --------------------------------------------------------------------------------------
# Derived rates and consistency checks
# --------------------------------------------------------------------------------------
class ModelDims:
    """Derived integer rates of the generator; raises on the reference's config errors.

    reference custom_pulsed_generator.py:256-267 (rates), :344-353 (sample-rate identity),
    :469-472 (conditioning-rate divisibility), :393-400 (STFT sizes).
    """

    def __init__(self, config):
        mb = config["mbexwn_config"]
        pp = config["preprocess_config"]
        if not config.get("use_tf25_compatible_implementation", False):
            # reference custom_pulsed_generator.py:50-51,272-273
            raise NotImplementedError("MBExWN::error::implmentations not selecting "
                                      "use_tf25_compatible_implementation are not supported")
        self.sample_rate = int(pp["sample_rate"])
        self.hop_size = int(pp["hop_size"])
        self.mel_channels = int(pp["mel_channels"])
        self.segment_length = int(pp.get("segment_length", 0))     # infer(synth_length=0) falls back to it (wavegen_1d.py:489)
        self.subbands = int(mb["multi_band_config"]["subbands"])
        self.pulse_rate_factor = int(mb.get("pulse_rate_factor", 2))
        self.pulse_rate = self.sample_rate / self.pulse_rate_factor
        self.pulse_channels = int(mb.get("pulse_channels", 8))
        # one WaveNet block per entry, block b with n_channels * channel_factors[b] channels and an up-sampling convolution
        # of factor upsampling_factors[b] behind it (reference custom_pulsed_generator.py:456-488)
        ups = [int(uu) for uu in mb["pp_mod_subnet_upsampling_factors"]]
        chf = list(mb["pp_mod_subnet_channel_factors"])
        if len(ups) != len(chf) or not ups or any(uu < 1 for uu in ups):
            raise RuntimeError("MBExWN::config_error::pp_mod_subnet_upsampling_factors / _channel_factors must be non-empty "
                               "lists of the same length")
        self.steps_per_frame = self.hop_size // self.subbands
        self.pulse_per_frame = (self.steps_per_frame * self.pulse_channels) // int(np.prod(ups))
        self.f0_down_sampling_factor = int(self.sample_rate // self.pulse_rate)
        gen_rate = self.pulse_rate / self.pulse_channels * np.prod(ups) * self.subbands
        if gen_rate != self.sample_rate:
            raise RuntimeError(f"MBExWN::config_error::the generated sample rate {gen_rate} != {self.sample_rate}")
        wn = mb["pp_mod_subnet"]
        self.wn_channels = int(int(wn["n_channels"]) * chf[0])             # the first block (reference :479)
        self.wn_block_channels = [int(int(wn["n_channels"]) * ff) for ff in chf]
        self.wn_block_ups = ups
        self.n_wn_blocks = len(ups)
        self.wn_multi = len(ups) > 1 or ups[0] > 1                # several blocks / in-block upsampling: the generic path
        # rows per frame of the first block; the noise channel has one value per such row
        self.wn_in_rows_per_frame = self.steps_per_frame // int(np.prod(ups))
        if self.wn_in_rows_per_frame * int(np.prod(ups)) != self.steps_per_frame:
            raise RuntimeError("MBExWN::config_error::the product of pp_mod_subnet_upsampling_factors must divide the "
                               "sub-band rows per frame")
        self.wn_layers = int(wn.get("n_layers", 12))
        self.wn_kernel_size = int(wn.get("kernel_size", 3))
        self.wn_out_channels = int(wn["n_out_channels"])
        self.wn_groups = int(wn.get("n_ch_groups", 1))
        self.wn_dilation_rate_step = int(wn.get("dilation_rate_step", 1))
        self.wn_max_log2_dilation = wn.get("max_log2_dilation_rate", None)
        self.wn_activation = wn.get("activation", "gtu")
        if self.wn_kernel_size % 2 != 1 or self.wn_channels % 2 != 0:
            # reference custom_AE_layers.py:134-135
            raise AssertionError("WaveNet kernel_size must be odd and n_channels even")
        if self.wn_groups < 1 or self.wn_channels % self.wn_groups or (self.wn_channels // self.wn_groups) % 2:
            # reference custom_AE_layers.py:165-166; the groups run as block-diagonal dense layers (weights.merge_channel_groups)
            raise RuntimeError(f"WaveNetAE::error::n_channels parameter {self.wn_channels} has to be a multiple of chanel "
                               f"groups parameter {self.wn_groups}")
        if self.wn_multi and self.wn_groups > 1:
            raise NotImplementedError("n_ch_groups > 1 together with several WaveNet blocks is not supported")
        if self.wn_activation not in ("gtu", "glu", "gfu", "gsu"):
            # reference custom_AE_layers.py:156-158 (glu passes the check and has no branch at :312-318: the half stays linear)
            raise RuntimeError(f"WaveNetAE::error::unsupported wavenet activation {self.wn_activation} selected. "
                               f"For gated units please select one of gtu, gfu, gsu, or glu.")
        # keys of WaveNetAE.__init__ (reference custom_AE_layers.py:120-131) that change the arithmetic and are not built:
        # silently ignoring them would load a valid model and produce wrong audio
        self.wn_equalized_lr = bool(wn.get("use_equalized_lr", False))    # folded on the host (weights.fold_weights)
        # convolutions of the mel input in front of the conditioning layer (reference custom_AE_layers.py:190-201,283-285)
        self.wn_pre_cond_channels = [int(cc) for cc in (wn.get("pre_cond_layer_channels", None) or [])]
        # no conditioning layer at all: the gates see zeros (reference custom_AE_layers.py:203-204,293-294)
        self.wn_disable_conditioning = bool(wn.get("disable_conditioning", False))
        # Keras padding of the WaveNet's convolutions: SAME, or CAUSAL (all padding in front).  VALID shortens the dilated
        # convolutions' output against the conditioning rows, which the reference's own graph cannot add up (:309)
        self.wn_padding = str(wn.get("padding", "SAME")).upper()
        if self.wn_padding not in ("SAME", "CAUSAL"):
            raise NotImplementedError(f"pp_mod_subnet.padding {self.wn_padding}: SAME and CAUSAL are supported")
        self.wn_use_weight_norm = bool(wn.get("use_weight_norm", False))
        self.cond_lin_upsampling = int(wn.get("cond_lin_upsampling", 16))
        self.cond_kernel_size = int(wn.get("cond_kernel_size", 3))
        curr_rate = self.pulse_rate / self.pulse_channels
        spect_rate = self.sample_rate / self.hop_size
        conv_up = curr_rate // (spect_rate * self.cond_lin_upsampling)
        if curr_rate != conv_up * spect_rate * self.cond_lin_upsampling:
            raise RuntimeError(f"MBExWN::config_error:: cannot achieve conditioning rate {curr_rate} by means of "
                               f"integer usampling of spectrum rate {spect_rate} with linear up "
                               f"{self.cond_lin_upsampling}")
        self.cond_conv_upsampling = int(conv_up)
        rate = curr_rate
        for uu in ups:                                             # every block's rate must be reachable (reference :469-472)
            if rate != (rate // (spect_rate * self.cond_lin_upsampling)) * spect_rate * self.cond_lin_upsampling:
                raise RuntimeError(f"MBExWN::config_error:: cannot achieve conditioning rate {rate} by means of "
                                   f"integer usampling of spectrum rate {spect_rate} with linear up "
                                   f"{self.cond_lin_upsampling}")
            rate *= uu
        self.noise_sigma = float(mb.get("pp_mod_subnet_noise_channel_sigma", 0.5) or 0.0)
        # wavetable options that change the excitation tensor (reference tf_wavetable.py:520-559,
        # custom_pulsed_generator.py:893): n sub-harmonic sinusoid channels next to the pulse; the pulse as a function
        wtc = mb.get("wavetable_config", {}) or {}
        self.wt_subharm = int(wtc.get("add_subharm_chans", 0) or 0)
        self.wt_sinusoid_as_fun = bool(wtc.get("use_sinusoid_as_fun", False))
        self.pulse_channels_eff = self.pulse_channels * (1 + self.wt_subharm)
        self.wn_in_channels = self.pulse_channels_eff + (1 if self.noise_sigma else 0)
        self.f0_min = float(mb.get("pp_min_frequency", 40.0))
        self.f0_max = float(mb.get("pp_max_frequency", 600.0))
        win_s = mb.get("internal_win_size_s", None)
        self.stft_win = int(win_s * self.sample_rate) if win_s else 4 * self.hop_size
        fft_size = 16
        while fft_size < self.stft_win:
            fft_size *= 2
        self.fft_size = fft_size * (2 ** int(mb.get("internal_fft_over", 0)))
        self.n_ceps = int(mb.get("ps_max_ceps_coefs", 120))
        self.ps_env_order_scale = mb.get("ps_env_order_scale", None)
        fr = mb.get("filter_max_db_range", None)
        self.filter_max_log_range = (fr / (20 * np.log10(np.exp(1)))) if fr else 0.0
        # cepstral coefficient 0 kept, every frame's filter divided by its rms magnitude (custom_pulsed_generator.py:817-849)
        self.preserve_energy = bool(mb.get("spect_filters_preserve_energy", False))
        # ps_off: no VTF-net, no STFT-domain filter, the audio is the excitation (reference custom_pulsed_generator.py:663-672)
        self.ps_off = bool(mb.get("ps_off", False))
        # ps_use_stft: false -- the VTF-net ends in one log gain per sub-band; the gains multiply the sub-band rows and there
        # is no STFT-domain filter (reference :427,453,663-672,857-884,916-917).  n_ceps then is the number of sub-bands.
        self.ps_subband_gain = (not mb.get("ps_use_stft", True)) and not self.ps_off
        if self.ps_subband_gain:
            self.n_ceps = self.subbands
        self.no_envelope = self.ps_off or self.ps_subband_gain     # neither PSig nor PS among the returned parameters
        # pp_mod_subnet_use_pqmf: false -- the sub-band rows are laid out one after the other instead (reference :920-923)
        self.no_pqmf = not mb.get("pp_mod_subnet_use_pqmf", True)
        # pulse_channels_use_pqmf: the WaveNet rows are the sub-bands of a PQMF analysis of the pulse signal instead of
        # consecutive samples (reference custom_pulsed_generator.py:499-501, 892-895)
        self.pulse_pqmf = None
        if mb.get("pulse_channels_use_pqmf", False):
            self.pulse_pqmf = dict(mb["pulse_channels_multi_band_config"])
            if int(self.pulse_pqmf["subbands"]) != self.pulse_channels:
                raise RuntimeError("MBExWN::config_error::pulse_channels_multi_band_config.subbands must equal pulse_channels")
            if self.wt_subharm:
                raise NotImplementedError("pulse_channels_use_pqmf together with add_subharm_chans is not supported")
        self.alpha = float(mb.get("alpha", 0.2))
        # NormMelComponents (reference wavegen_1d.py:578-769, row A14): device kernels csrc/norm_mel.hip, tables norm_mel.py
        self.normalize_rms_from_mell = bool(mb.get("normalize_rms_from_mell", False))

    def wn_dilation(self, index):
        # reference custom_AE_layers.py:229-233
        if self.wn_max_log2_dilation is not None:
            return 2 ** (int(index // self.wn_dilation_rate_step) % int(self.wn_max_log2_dilation))
        return 2 ** int(index // self.wn_dilation_rate_step)
