#!/usr/bin/env python3
"""Golden vectors of the model-level entry points: the reference's own ``PaNWaveNet.infer`` / ``infer_components``
(reference MBExWN_NVoc/vocoder/model/wavegen_1d.py:483-557) executed, unmodified, on a stand-in ``self`` that carries the
four attributes those two methods read (``segment_length``, ``spect_hop_size``, ``norm_mel_components``, ``block`` = the
reference's MBExWN layer with the seeded synthetic weights loaded).  PaNWaveNet's constructor itself builds the training
losses and is not needed for inference.  Pins: synth_length handling (repeat of the last mel frame, slicing), the
parameter list of return_F0 (F0 sub-sampling, PSig, |PS|), infer_components with a transposition factor and with an
external F0 contour, and the RMS-normalised variant.

Writes tests/golden/reference_infer.npz.   (build container only; needs /root/reference)
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import tf_numpy_shim as shim  # noqa: E402
from make_reference_forward import load_into_reference  # noqa: E402

SMALL = {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3}
CASES = {
    # name: (config overrides, frames, synth_length offsets relative to frames * hop)
    "plain": (SMALL, 14),
    "rmsnorm": (dict(SMALL, **{"mbexwn_config:normalize_rms_from_mell": True, "mbexwn_config:normalize_rms_num_smooth_iters": 1}), 14),
}


def main():
    shim.install("/root/reference")
    shim.set_float(np.float32)
    from mbexwn_vocoder_amd import analysis
    from mbexwn_vocoder_amd.config import canonical_config
    from mbexwn_vocoder_amd.weights import synthetic_weights
    sys.modules["librosa.core.convert"].mel_frequencies = \
        lambda n_mels=128, fmin=0.0, fmax=11025.0, htk=False: analysis.mel_frequencies(n_mels, fmin, fmax)
    from MBExWN_NVoc.vocoder.model.custom_pulsed_generator import MBExWN
    from MBExWN_NVoc.vocoder.model.wavegen_1d import NormMelComponents, PaNWaveNet
    out = {}
    for name, (over, frames) in CASES.items():
        cfg = canonical_config("SPEECH", **over)
        raw = synthetic_weights(cfg, seed=1234, bias_std=0.05, alpha_jitter=0.05)
        mb = cfg["mbexwn_config"]
        gen_keys = {kk: vv for kk, vv in mb.items() if not kk.startswith("normalize_") and kk not in ("max_norm_fact", "lin_amp_scale", "mel_amp_scale")}
        model = MBExWN(**gen_keys, preprocess_config=cfg["preprocess_config"], quiet=True, use_tf25_compatible_implementation=True)
        hop = cfg["preprocess_config"]["hop_size"]
        model.build(shim.Shape((2, frames + 1, 80)))
        load_into_reference(model, raw)
        nm = None
        if mb.get("normalize_rms_from_mell", False):
            nm = NormMelComponents(preprocess_config=cfg["preprocess_config"], dtype=shim.tf.float32, **mb)
        me = types.SimpleNamespace(segment_length=frames * hop, spect_hop_size=hop, norm_mel_components=nm, block=model)
        rng = np.random.default_rng(77)
        mell = np.clip(np.log(np.exp(rng.normal(-5.0, 2.0, size=(2, frames, 80))) + 1e-5), -11.5, 2.0).astype(np.float32)
        out[f"{name}/mell"] = mell
        steps = model.spect_to_subband_upsampling_factor
        for tag, synth_length in (("short", frames * hop - 123), ("exact", 0), ("long", frames * hop + 150)):
            n_fr = frames + (1 if synth_length > frames * hop else 0)          # infer repeats the last frame once
            noise = rng.normal(size=(2, n_fr * steps)).astype(np.float32)
            shim.INJECTED_NOISE["normal"] = noise
            audio, params = PaNWaveNet.infer(me, shim.Tensor(mell), synth_length=synth_length, return_F0=True)
            out[f"{name}/{tag}/noise"] = noise
            out[f"{name}/{tag}/audio"] = np.asarray(audio)
            for pname, val in params:
                arr = np.asarray(val)
                out[f"{name}/{tag}/{pname}"] = arr[:, :, ::8] if pname == "PS" else arr      # every 8th bin: small fixtures
                out[f"{name}/{tag}/{pname}_shape"] = np.asarray(arr.shape)
            print(name, tag, np.asarray(audio).shape, [(pp[0], np.asarray(pp[1]).shape) for pp in params])
        # infer_components: transposed F0, then an external contour (shorter than the mel: the tail repeats its last value
        # in this build; the reference takes the contour as it is, so the golden uses a full-length one)
        noise = rng.normal(size=(2, frames * steps)).astype(np.float32)
        shim.INJECTED_NOISE["normal"] = noise
        f0, exc, env, gain = PaNWaveNet.infer_components(me, shim.Tensor(mell), synth_length=frames * hop, transposition_factor=1.25)
        out[f"{name}/comp/noise"] = noise
        out[f"{name}/comp/f0"], out[f"{name}/comp/excitation"] = np.asarray(f0), np.asarray(exc)
        out[f"{name}/comp/env_abs"] = np.abs(np.asarray(env))[:, :, ::8]
        if gain is not None:
            out[f"{name}/comp/gain"] = np.asarray(gain)
        ext = (110.0 + 40.0 * np.sin(np.arange(frames * model.spect_to_pulse_upsampling_factor) / 97.0))[None].repeat(2, 0).astype(np.float32)
        shim.INJECTED_NOISE["normal"] = noise
        f0b, excb, envb, _ = PaNWaveNet.infer_components(me, shim.Tensor(mell), F0=shim.Tensor(ext))
        out[f"{name}/ext/f0_in"] = ext
        out[f"{name}/ext/excitation"] = np.asarray(excb)
        out[f"{name}/ext/env_abs"] = np.abs(np.asarray(envb))[:, :, ::8]
        print(name, "components", np.asarray(f0).shape, np.asarray(exc).shape, np.asarray(env).shape, None if gain is None else np.asarray(gain).shape)
    path = os.path.join(HERE, "reference_infer.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
