#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own model code (build container only).

Imports ``MBExWN_NVoc`` from /root/reference with ``tf_numpy_shim`` registered as ``tensorflow``,
instantiates the reference ``MBExWN`` layer from this repo's canonical configuration, loads the
seeded synthetic variables into the reference's layer objects (v / g / bias / alpha), injects the
noise draw and executes ``MBExWN.call`` and its stage methods unmodified.

Outputs (committed, small):
  reference_forward_f32.npz  tf.float32 := numpy float32  (emulation of the float32 TF-CPU run)
  reference_forward_f64.npz  tf.float32 := numpy float64  (same graph, rounding noise removed)

Usage: python tests/golden/make_reference_forward.py     (needs /root/reference)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import tf_numpy_shim as shim  # noqa: E402

CASES = {
    # name: (voice type, config overrides, batch, frames)
    "small": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3}, 2, 23),
    "canon": ("SPEECH", {}, 1, 12),
    # long enough for several 1000-sample phase chunks, the chunk-offset chain and F0-dependent lifter rows
    "canon60": ("SPEECH", {}, 1, 60),
    # MW-VO-FD geometry: C = 340 (340 % 32 != 0, 340 % 16 = 4: partial column tile, masked staging paths)
    "voice": ("VOICE", {}, 2, 41),
    # sub-net grammar variants (reference custom_pulsed_generator.py:57-60, 74-108): sub-pixel convolution with Keras SAME
    # zero padding, "L<up>" interpolation behind a convolution, bare ["L", up] (total_ups not updated -> the F0-net runs
    # at 5x the pulse rate and generate_f0 cuts it, :787)
    "grammar": ("SPEECH", {"mbexwn_config:pp_subnet": [[5, 32, 2], [3, 64, "L2"], ["L", 5]],
                           "mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3}, 2, 9),
    # two independent channel groups between the shared start and end convolutions (reference custom_AE_layers.py:303-340)
    "groups": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                          "mbexwn_config:pp_mod_subnet:n_ch_groups": 2}, 2, 9),
    # the other two gates of WaveNetAE (reference custom_AE_layers.py:312-318) and use_equalized_lr (conv_layers.py:133-153),
    # with weight norm (W = g v / sqrt(mean v^2)) and without it (the layer output is multiplied by g)
    "gfu": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                       "mbexwn_config:pp_mod_subnet:activation": "gfu"}, 2, 9),
    "gsu_eqlr": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                            "mbexwn_config:pp_mod_subnet:activation": "gsu",
                            "mbexwn_config:pp_mod_subnet:use_equalized_lr": True}, 2, 9),
    "eqlr_plain": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                              "mbexwn_config:pp_mod_subnet:use_weight_norm": False,
                              "mbexwn_config:pp_mod_subnet:use_equalized_lr": True}, 2, 9),
    # "glu" (accepted at custom_AE_layers.py:156, no branch at :312-318: linear half x sigmoid), pre-conditioning
    # convolutions (:190-201,283-285), a WaveNet without conditioning (:203-204,293-294) and energy preserving spectral
    # filters (custom_pulsed_generator.py:817-849)
    "glu": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                       "mbexwn_config:pp_mod_subnet:activation": "glu"}, 2, 9),
    "precond": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                           "mbexwn_config:pp_mod_subnet:pre_cond_layer_channels": [48, 40]}, 2, 9),
    "nocond": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                          "mbexwn_config:pp_mod_subnet:disable_conditioning": True}, 2, 9),
    "energy": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                          "mbexwn_config:spect_filters_preserve_energy": True}, 2, 9),
    # branches that had only the oracle: no noise channel, plain exp(S) filters without the F0-dependent lifter, a dilation
    # cycle with two layers per dilation, a 1-tap conditioning layer; leaky ReLU instead of PReLU, VALID-padded sub-nets
    # (edge padding in front of the net), the cepstral loss-constraint switch (no lifter at inference), 3 x 20 conditioning
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
    # wavetable options that change the excitation tensor (reference tf_wavetable.py:520-559, custom_pulsed_generator.py:893):
    # one sub-harmonic sinusoid channel next to the LF pulse; the pulse as sin * (1 - cos) / 2 with two of them
    "subharm": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                           "mbexwn_config:wavetable_config:add_subharm_chans": 1}, 2, 9),
    "sinfun": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                          "mbexwn_config:wavetable_config:use_sinusoid_as_fun": True,
                          "mbexwn_config:wavetable_config:add_subharm_chans": 2}, 2, 9),
    # no STFT-domain filter (and no VTF-net): the audio is the excitation; no PQMF bank: the sub-band rows laid out one
    # after the other (reference custom_pulsed_generator.py:663-672, 920-923)
    "psoff": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                         "mbexwn_config:ps_off": True}, 2, 9),
    "nopqmf": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                          "mbexwn_config:pp_mod_subnet_use_pqmf": False}, 2, 9),
    # two WaveNet blocks with in-block upsampling (reference custom_pulsed_generator.py:456-488, custom_AE_layers.py:457-582):
    # 32 channels at 800 Hz (10 folded pulse samples per row), x2 sub-pixel convolution, 16 channels at 1 600 Hz
    "blocks": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                          "mbexwn_config:pp_mod_subnet_upsampling_factors": [2, 1],
                          "mbexwn_config:pp_mod_subnet_channel_factors": [1, 0.5],
                          "mbexwn_config:pulse_channels": 10,
                          "mbexwn_config:pp_mod_subnet:cond_lin_upsampling": 5}, 2, 9),
    # PQMF analysis of the pulse signal in front of the WaveNet instead of folding consecutive samples (reference
    # custom_pulsed_generator.py:499-501, 892-895)
    "pulsepqmf": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                             "mbexwn_config:pulse_channels_use_pqmf": True,
                             "mbexwn_config:pulse_channels_multi_band_config": {"subbands": 5, "taps": 40,
                                                                                "cutoff_ratio": 0.12, "beta": 9.0}}, 2, 9),
    # ps_use_stft: false: sub-band gains instead of the STFT-domain filter, with and without the mean removal of
    # spect_filters_preserve_energy (reference custom_pulsed_generator.py:453,663-672,857-884,916-917)
    "subgain": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                           "mbexwn_config:ps_use_stft": False}, 2, 9),
    "subgain_e": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                             "mbexwn_config:ps_use_stft": False, "mbexwn_config:spect_filters_preserve_energy": True}, 1, 40),
    # pp_mod_subnet.padding: CAUSAL -- every padded convolution of the WaveNet blocks (dilated, conditioning, up-sampling)
    # pads in front only (Keras "causal"); two blocks so that the up-sampling convolution is covered
    "causal": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                          "mbexwn_config:pp_mod_subnet:padding": "CAUSAL",
                          "mbexwn_config:pp_mod_subnet_upsampling_factors": [2, 1],
                          "mbexwn_config:pp_mod_subnet_channel_factors": [1, 1],
                          "mbexwn_config:pulse_channels": 10,
                          "mbexwn_config:pp_mod_subnet:cond_lin_upsampling": 5}, 2, 9),
    "causal1": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                           "mbexwn_config:pp_mod_subnet:padding": "CAUSAL"}, 2, 9),
}
# the long cases keep the small stage outputs only (the full conditioning / envelope tensors would be megabytes)
LEAN = {"canon60", "voice", "grammar", "groups", "gfu", "gsu_eqlr", "eqlr_plain", "glu", "precond", "nocond", "energy", "mixed_a", "mixed_b", "subharm", "sinfun", "psoff", "nopqmf", "blocks", "pulsepqmf", "subgain", "subgain_e", "causal", "causal1"}


def assign_conv(layer, raw, name):
    # a layer built with use_equalized_lr but without weight norm keeps its kernel in the Keras layer and only adds g
    (layer.v if hasattr(layer, "v") else layer.conv1d_layer.kernel).assign(raw[name + ".v"])
    layer.g.assign(raw[name + ".g"])
    layer.conv1d_layer.bias.assign(raw[name + ".bias"])


def load_into_reference(model, raw):
    used = set()
    for sub in (model.pp_subnet_layers, model.ps_subnet_layers):
        for ll in sub:
            if hasattr(ll, "v"):
                name = ll.conv1d_layer.name
                assign_conv(ll, raw, name)
                used.update({name + ".v", name + ".g", name + ".bias"})
            elif type(ll).__name__ == "PReLU":
                ll.alpha.assign(np.reshape(raw[ll.name + ".alpha"], ll.alpha.shape))
                used.add(ll.name + ".alpha")
    pairs = [(model.wn_post_net[0], "post")]
    # one WaveNetAEBlock per up-sampling factor (reference custom_pulsed_generator.py:456-488): "wn." / "wn1." ..., the
    # up-sampling convolution behind block b is "up<b>"
    for bb, block in enumerate(model.pp_waveNetBlocks):
        wn, pre = block.wavenet, ("wn." if bb == 0 else f"wn{bb}.")
        pairs += [(wn.start, pre + "start"), (wn.end, pre + "end")]
        if wn.cond_layer is not None:
            pairs += [(wn.cond_layer, pre + "cond")] + [(ll, f"{pre}precond_{ii}") for ii, ll in enumerate(wn.pre_cond_layers)]
        # layer list index = layer * n_ch_groups + group; group g > 0 is named "<layer>g<g>" (reference custom_AE_layers.py:249,260)
        ng = wn.n_ch_groups

        def wn_name(base, ii, pre=pre, ng=ng):
            return f"{pre}{base}_{ii // ng}" + (f"g{ii % ng}" if ii % ng else "")
        pairs += [(ll, wn_name("conv1D", ii)) for ii, ll in enumerate(wn.conv_layers)]
        pairs += [(ll, wn_name("res_skip", ii)) for ii, ll in enumerate(wn.res_skip_layers)]
        if block.up_down_sample is not None:
            pairs.append((block.up_down_sample, f"up{bb}"))
    for layer, name in pairs:
        assign_conv(layer, raw, name)
        used.update({name + ".v", name + ".g", name + ".bias"})
    missing = set(raw.keys()) - used
    if missing:
        raise RuntimeError(f"variables not consumed by the reference model: {sorted(missing)}")


def run_case(voice, overrides, batch, frames, float_type):
    from mbexwn_vocoder_amd.config import canonical_config
    from mbexwn_vocoder_amd.weights import synthetic_weights
    from MBExWN_NVoc.vocoder.model.custom_pulsed_generator import MBExWN

    cfg = canonical_config(voice, **overrides)
    raw = synthetic_weights(cfg, seed=1234, bias_std=0.05, alpha_jitter=0.05)
    model = MBExWN(**cfg["mbexwn_config"], preprocess_config=cfg["preprocess_config"], quiet=True,
                   use_tf25_compatible_implementation=True)
    model.build(shim.Shape((batch, frames, cfg["preprocess_config"]["mel_channels"])))
    load_into_reference(model, raw)

    rng = np.random.default_rng(42)
    mell = np.log(np.exp(rng.normal(-5.0, 2.0, size=(batch, frames, 80))) + 1e-5)
    mell = np.clip(mell, -11.5, 2.0).astype(np.float32)
    # one noise value per row of the first WaveNet block (reference :905-906)
    steps = frames * model.spect_to_pulse_upsampling_factor // model.pulse_channels
    noise = rng.normal(size=(batch, steps)).astype(np.float32)

    mel_t = shim.Tensor(mell.astype(float_type))
    out = {"mell": mell, "noise": noise}
    # stage outputs through the reference's own methods
    f0 = model.generate_f0(mel_t)
    out["f0"] = np.asarray(f0)
    out["phase"] = np.asarray(model.pulse_generator.stable_cumsum_and_wrap(f0 / model.pulse_generator.sample_rate))
    pulse = np.asarray(model.pulse_generator(f0))            # (B, N, 1 + add_subharm_chans)
    out["pulse"] = pulse[:, :, 0] if pulse.shape[2] == 1 else pulse
    wn = model.pp_waveNetBlocks[0].wavenet
    if wn.cond_layer is not None:
        cond_in = mel_t
        for ll in wn.pre_cond_layers:
            cond_in = ll(cond_in)
        out["cond"] = np.asarray(wn.cond_lin_upsampling_layer(wn.cond_layer(cond_in)))
    shim.INJECTED_NOISE["normal"] = noise
    mb_gain = None
    if not model.ps_off and not model.ps_use_stft:      # the call path of the sub-band gain variant (reference :667-672)
        mb_gain = model.ps_gain_interpolator(model.generate_multiband_gain(mel=mel_t, training=False))
    out["excitation"] = np.asarray(model.generate_excitation(mel_t, pulse_frequency=f0, mb_gain=mb_gain))
    if not model.ps_off and model.ps_use_stft:
        env = model.generate_specenv(mel=mel_t, pulse_frequency=f0, training=False)
        out["envelope_re"] = np.asarray(env).real
        out["envelope_im"] = np.asarray(env).imag
    if model.ps_env_order_scale and not model.ps_off and model.ps_use_stft:
        win = model._get_cepstral_windows(f0, model.ps_cepstral_windows_log10f0, model.ps_cepstral_windows,
                                          smooth_stride=model.spect_to_pulse_upsampling_factor)
        out["ceps_window_sum"] = np.asarray(win).sum(axis=-1)
    # the full graph exactly as PaNWaveNet.infer drives it (reference wavegen_1d.py:504-526)
    shim.INJECTED_NOISE["normal"] = noise
    signals, _ = model(mel_t, None, training=False, return_PP=False)
    out["audio"] = np.asarray(signals[0])[:, :frames * model.spect_hop_size]
    # wavetable constants the oracle takes as input
    out["wavetables"] = np.asarray(model.pulse_generator.wavetables)
    out["wt_nominalF0"] = np.float64(model.pulse_generator.nominalF0)
    return out


def main():
    shim.install("/root/reference")
    for tag, float_type in (("f32", np.float32), ("f64", np.float64)):
        shim.set_float(float_type)
        bundle = {}
        for name, (voice, overrides, batch, frames) in CASES.items():
            res = run_case(voice, overrides, batch, frames, float_type)
            if name in LEAN:
                if "cond" in res:
                    res["cond"] = np.asarray(res["cond"])[:, ::37]
                for kk in ("envelope_re", "envelope_im", "wavetables"):
                    res.pop(kk, None)
                if tag == "f64":
                    res = {kk: res[kk] for kk in ("mell", "noise", "f0", "excitation", "audio")}
            for kk, vv in res.items():
                arr = np.asarray(vv)
                if arr.dtype == np.float64 and tag == "f32":
                    arr = arr.astype(np.float32) if kk not in ("wt_nominalF0",) else arr
                bundle[f"{name}/{kk}"] = arr
            print(tag, name, "audio", res["audio"].shape, float(np.abs(res["audio"]).max()))
        path = os.path.join(HERE, f"reference_forward_{tag}.npz")
        np.savez_compressed(path, **bundle)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
