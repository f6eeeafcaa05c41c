#!/usr/bin/env python3
"""Golden vectors at the BASELINE lengths and at the reference's default WaveNet depth (build container only).

Same mechanism as make_reference_forward.py (whose ``run_case`` this script calls): the REFERENCE's own ``MBExWN`` layer is
imported from /root/reference with ``tf_numpy_shim`` registered as ``tensorflow``, loaded with the seeded synthetic
variables and executed unmodified -- once with tf.float32 := numpy float32 (the emulation of the float32 TF-CPU run) and
once with tf.float32 := numpy float64 (the same graph without rounding noise).

Why these cases exist:
  * the phase of the oscillator is a float32 running sum (reference tf_wavetable.py:429-492), so the distance between two
    float32 evaluations of the graph GROWS with the utterance: parity against the reference's float32 run has to be pinned
    at the lengths the BASELINE configs name (240 frames = 3 s, 800 frames = 10 s), not only at 0.75 s;
  * ``WaveNetAE.__init__`` defaults to n_layers = 12 without a dilation cycle (reference custom_AE_layers.py:120-123,
    229-233): dilations 1 .. 2048.  The canonical L = 5 is an inference of SURVEY.md; the default depth needs its own pin.

Outputs (committed; lean: inputs, F0 contour, phase, excitation, audio):
  reference_long_f32.npz   per case: mell, noise, f0, phase, excitation, audio            (float32)
  reference_long_f64.npz   per case: f0, audio                                           (float64)

Usage: python tests/golden/make_reference_long.py     (needs /root/reference; about two minutes)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import tf_numpy_shim as shim  # noqa: E402
from make_reference_forward import run_case  # noqa: E402

SMALL_WN = {"mbexwn_config:pp_mod_subnet:n_channels": 32}
CASES = {
    # name: (voice type, config overrides, batch, frames)
    # BASELINE configs[0] / configs[1]: MW-SP-FD (C = 320), one utterance of 3 s / 10 s (reference bin/resynth_mel.py:65-88)
    "speech240": ("SPEECH", {}, 1, 240),
    "speech800": ("SPEECH", {}, 1, 800),
    # MW-VO-FD (C = 340), 5 s
    "voice400": ("VOICE", {}, 1, 400),
    # the reference's default depth: 12 layers, no dilation cycle -> d = 1, 2, 4 .. 2048 (custom_AE_layers.py:229-233).
    # 240 frames = 4800 rows: the widest layer (+-2048 rows) has real rows on both sides of the middle of the item;
    # 60 frames = 1200 rows: shorter than the dilation of the last two layers, whose outer taps then read zero padding only
    "deep12": ("SPEECH", dict(SMALL_WN, **{"mbexwn_config:pp_mod_subnet:n_layers": 12}), 1, 240),
    "deep12_short": ("SPEECH", dict(SMALL_WN, **{"mbexwn_config:pp_mod_subnet:n_layers": 12}), 1, 60),
    # 12 layers in three cycles of d = 1, 2, 4, 8 (max_log2_dilation_rate = 4: custom_AE_layers.py:229-231)
    "cycle12": ("SPEECH", dict(SMALL_WN, **{"mbexwn_config:pp_mod_subnet:n_layers": 12,
                                            "mbexwn_config:pp_mod_subnet:max_log2_dilation_rate": 4}), 2, 31),
}
KEEP32 = ("mell", "noise", "f0", "phase", "excitation", "audio")
KEEP64 = ("f0", "audio")


def main():
    shim.install("/root/reference")
    for tag, float_type, keep in (("f32", np.float32, KEEP32), ("f64", np.float64, KEEP64)):
        shim.set_float(float_type)
        bundle = {}
        for name, (voice, overrides, batch, frames) in CASES.items():
            res = run_case(voice, overrides, batch, frames, float_type)
            for kk in keep:
                arr = np.asarray(res[kk])
                if tag == "f32" and arr.dtype == np.float64:
                    arr = arr.astype(np.float32)
                bundle[f"{name}/{kk}"] = arr
            print(tag, name, "audio", res["audio"].shape, float(np.abs(res["audio"]).max()), flush=True)
        path = os.path.join(HERE, f"reference_long_{tag}.npz")
        np.savez_compressed(path, **bundle)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
