"""Shared builders for the tests (CPU side)."""
import functools

import numpy as np

from mbexwn_vocoder_amd.config import ModelDims, canonical_config
from mbexwn_vocoder_amd.tables import WaveTables
from mbexwn_vocoder_amd.weights import synthetic_weights

# the golden cases of tests/golden/make_reference_forward.py
GOLDEN_CASES = {
    "small": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3}, 2, 23),
    "canon": ("SPEECH", {}, 1, 12),
    "canon60": ("SPEECH", {}, 1, 60),
    "voice": ("VOICE", {}, 2, 41),
    "grammar": ("SPEECH", {"mbexwn_config:pp_subnet": [[5, 32, 2], [3, 64, "L2"], ["L", 5]],
                           "mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3}, 2, 9),
    "groups": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                          "mbexwn_config:pp_mod_subnet:n_ch_groups": 2}, 2, 9),
    "gfu": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                       "mbexwn_config:pp_mod_subnet:activation": "gfu"}, 2, 9),
    "gsu_eqlr": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                            "mbexwn_config:pp_mod_subnet:activation": "gsu",
                            "mbexwn_config:pp_mod_subnet:use_equalized_lr": True}, 2, 9),
    "eqlr_plain": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                              "mbexwn_config:pp_mod_subnet:use_weight_norm": False,
                              "mbexwn_config:pp_mod_subnet:use_equalized_lr": True}, 2, 9),
    "glu": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                       "mbexwn_config:pp_mod_subnet:activation": "glu"}, 2, 9),
    "precond": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                           "mbexwn_config:pp_mod_subnet:pre_cond_layer_channels": [48, 40]}, 2, 9),
    "nocond": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                          "mbexwn_config:pp_mod_subnet:disable_conditioning": True}, 2, 9),
    "energy": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                          "mbexwn_config:spect_filters_preserve_energy": True}, 2, 9),
    "mixed_a": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 5,
                           "mbexwn_config:pp_mod_subnet:dilation_rate_step": 2,
                           "mbexwn_config:pp_mod_subnet:max_log2_dilation_rate": 2,
                           "mbexwn_config:pp_mod_subnet:cond_kernel_size": 1,
                           "mbexwn_config:pp_mod_subnet_noise_channel_sigma": 0,
                           "mbexwn_config:filter_max_db_range": None,
                           "mbexwn_config:ps_env_order_scale": None}, 2, 9),
    "mixed_b": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                           "mbexwn_config:use_prelu": False, "mbexwn_config:alpha": 0.1,
                           "mbexwn_config:pp_subnet_use_valid_padding": True,
                           "mbexwn_config:ps_subnet_use_valid_padding": True,
                           "mbexwn_config:psns_use_cepstral_loss_constraint": True,
                           "mbexwn_config:filter_max_db_range": 12.0,
                           "mbexwn_config:pp_mod_subnet:cond_lin_upsampling": 20}, 2, 9),
    "subharm": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                           "mbexwn_config:wavetable_config:add_subharm_chans": 1}, 2, 9),
    "sinfun": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                          "mbexwn_config:wavetable_config:use_sinusoid_as_fun": True,
                          "mbexwn_config:wavetable_config:add_subharm_chans": 2}, 2, 9),
    "psoff": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                         "mbexwn_config:ps_off": True}, 2, 9),
    "nopqmf": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                          "mbexwn_config:pp_mod_subnet_use_pqmf": False}, 2, 9),
    "blocks": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                          "mbexwn_config:pp_mod_subnet_upsampling_factors": [2, 1],
                          "mbexwn_config:pp_mod_subnet_channel_factors": [1, 0.5],
                          "mbexwn_config:pulse_channels": 10,
                          "mbexwn_config:pp_mod_subnet:cond_lin_upsampling": 5}, 2, 9),
    "pulsepqmf": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                             "mbexwn_config:pulse_channels_use_pqmf": True,
                             "mbexwn_config:pulse_channels_multi_band_config": {"subbands": 5, "taps": 40,
                                                                                "cutoff_ratio": 0.12, "beta": 9.0}}, 2, 9),
    "subgain": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                           "mbexwn_config:ps_use_stft": False}, 2, 9),
    "subgain_e": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                             "mbexwn_config:ps_use_stft": False, "mbexwn_config:spect_filters_preserve_energy": True}, 1, 40),
    "causal": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                          "mbexwn_config:pp_mod_subnet:padding": "CAUSAL",
                          "mbexwn_config:pp_mod_subnet_upsampling_factors": [2, 1],
                          "mbexwn_config:pp_mod_subnet_channel_factors": [1, 1],
                          "mbexwn_config:pulse_channels": 10,
                          "mbexwn_config:pp_mod_subnet:cond_lin_upsampling": 5}, 2, 9),
    "causal1": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                           "mbexwn_config:pp_mod_subnet:padding": "CAUSAL"}, 2, 9),
}
LEAN_GOLDEN_CASES = {"canon60", "voice", "grammar", "groups", "gfu", "gsu_eqlr", "eqlr_plain", "glu", "precond", "nocond", "energy", "mixed_a", "mixed_b", "subharm", "sinfun", "psoff", "nopqmf", "blocks", "pulsepqmf", "subgain", "subgain_e", "causal", "causal1"}     # cond subsampled [:, ::37], no envelope / wavetables (see the generator)


@functools.lru_cache(maxsize=None)
def wavetables_for(voice="SPEECH"):
    cfg = canonical_config(voice)
    dims = ModelDims(cfg)
    return WaveTables(sample_rate=dims.pulse_rate, **cfg["mbexwn_config"]["wavetable_config"])


def build_case(voice, overrides, seed=1234, bias_std=0.05, alpha_jitter=0.05):
    cfg = canonical_config(voice, **overrides)
    raw = synthetic_weights(cfg, seed=seed, bias_std=bias_std, alpha_jitter=alpha_jitter)
    return cfg, raw, wavetables_for(voice)


def synthetic_inputs(seed, batch, frames, steps_per_frame=20):
    rng = np.random.default_rng(seed)
    mell = np.log(np.exp(rng.normal(-5.0, 2.0, size=(batch, frames, 80))) + 1e-5)
    mell = np.clip(mell, -11.5, 2.0).astype(np.float32)
    noise = rng.normal(size=(batch, frames * steps_per_frame)).astype(np.float32)
    return mell, noise


def form_kwargs(form):
    """Engine arguments of the convolution forms the parity tests pin: "default" (auto), "0" direct, "2" Winograd F(2,3),
    "4" F(4,3), "44" F(4,3) with batch-invariant kernels (the names are those of the experiment variable MBX_WINOGRAD, which
    the library no longer reads: mbx_config.wn_conv_form / batch_invariant)."""
    return {"default": {}, "0": {"conv_form": "direct"}, "2": {"conv_form": "f23"}, "4": {"conv_form": "f43"},
            "44": {"conv_form": "f43", "batch_invariant": True}}[str(form)]
