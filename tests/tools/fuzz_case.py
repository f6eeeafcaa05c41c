#!/usr/bin/env python3
"""Re-run one case of config_fuzz.py (same seed arithmetic) and compare stage by stage, also against the float32 oracle:
a difference that the float32 oracle shows too is conditioning of the random model, not the kernels."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import importlib.util
spec = importlib.util.spec_from_file_location("cf", os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_fuzz.py"))
src = open(spec.origin).read()
seed0, case = int(sys.argv[1]), int(sys.argv[2])
# reproduce the draw of the case by executing the loop body's prefix
from mbexwn_vocoder_amd.config import ModelDims, canonical_config
from mbexwn_vocoder_amd.engine import MBExWNEngine
from mbexwn_vocoder_amd.tables import WaveTables
from mbexwn_vocoder_amd.weights import synthetic_weights
from oracle.mbexwn_oracle import OracleModel
M, W = "mbexwn_config:", "mbexwn_config:pp_mod_subnet:"
body = src[src.index("    rng = np.random.default_rng(seed0 + case)"):src.index("    try:\n        cfg = canonical_config")]
ns = {"np": np, "seed0": seed0, "case": case, "M": M, "W": W, "sys": sys}      # (a third argument "structure" is read by the body)
exec("if True:\n" + body, ns)
rng, over = ns["rng"], ns["over"]
cfg = canonical_config(str(rng.choice(["SPEECH", "VOICE"])), **over)
dims = ModelDims(cfg)
raw = synthetic_weights(cfg, seed=int(rng.integers(1, 10 ** 6)), bias_std=0.05, alpha_jitter=0.05)
wt = WaveTables(sample_rate=dims.pulse_rate, **cfg["mbexwn_config"]["wavetable_config"])
form = str(rng.choice(["auto", "auto", "auto", "auto", "f43", "f23", "direct", "f43i"]))      # as config_fuzz.py draws it
eng = MBExWNEngine(cfg, raw, wt, conv_form=form.rstrip("i"), batch_invariant=form.endswith("i"))
print("conv form:", eng.conv_form_info())
om64, om32 = OracleModel(cfg, raw, wt), OracleModel(cfg, raw, wt, dtype=np.float32)
B, T = int(rng.integers(1, 5)), int(rng.integers(1, 45))
lengths = [T] + [int(rng.integers(1, T + 1)) for _ in range(B - 1)]
mel = np.clip(np.log(np.exp(rng.normal(-5.0, 2.0, size=(B, T, 80))) + 1e-5), -11.5, 2.0).astype(np.float32)
rpf = dims.wn_in_rows_per_frame
noise = rng.normal(size=(B, T * rpf)).astype(np.float32)
print("form", form, "B", B, "T", T, "lengths", lengths, over)
got = eng.forward(torch.as_tensor(mel).cuda(), n_frames=torch.tensor(lengths, dtype=torch.int32, device="cuda"),
                  noise=torch.as_tensor(noise).cuda() if dims.noise_sigma else None).cpu().numpy()
exc = eng.stage("excitation").cpu().numpy(); f0 = eng.stage("f0").cpu().numpy()
for ii, ll in enumerate(lengths):
    nz = noise[ii:ii + 1, :ll * rpf] if dims.noise_sigma else None
    a64, s64 = om64.forward(mel[ii:ii + 1, :ll], nz, return_stages=True)
    a32, s32 = om32.forward(mel[ii:ii + 1, :ll], nz, return_stages=True)
    amp = max(1.0, float(np.abs(a64).max()))
    print("item", ii, "frames", ll, "amp", round(amp, 2), "exc amp", round(float(np.abs(s64["excitation"]).max()), 2),
          "| hip-f64 audio", f"{np.abs(got[ii, :ll*300]-a64[0]).max()/amp:.1e}", "f32-f64 audio", f"{np.abs(a32[0]-a64[0]).max()/amp:.1e}",
          "| hip-f64 exc", f"{np.abs(exc[ii, :ll*300]-s64['excitation'][0]).max():.1e}", "f32-f64 exc", f"{np.abs(s32['excitation'][0]-s64['excitation'][0]).max():.1e}",
          "| f0 hip-f64", f"{np.abs(f0[ii, :ll*dims.pulse_per_frame]-s64['f0'][0]).max():.1e}", "f0 f32-f64", f"{np.abs(s32['f0'][0]-s64['f0'][0]).max():.1e}")
    dex = np.abs(exc[ii, :ll*300]-s64['excitation'][0])
    worst = int(dex.argmax())
    ph_hip = eng.wavetable(eng.stage("f0")[ii:ii+1])[1].cpu().numpy()[0]
    ph_ref = om64.phase_from_f0(np.asarray(s64['f0'], dtype=np.float32))[0]
    flips = np.nonzero(np.abs(ph_hip[:ll*dims.pulse_per_frame] - ph_ref[:ll*dims.pulse_per_frame]) > 0.5)[0]
    print("   largest excitation difference at sample", worst, "= pulse sample", worst * dims.pulse_per_frame // 300,
          "| pulse samples where the wrapped phases differ by a whole turn:", flips[:10])
    # the STFT filter: the error of the audio frame by frame, and the size of the envelope the oracle applies
    if "envelope" in s64 and getattr(om64, "ceps_windows", None) is not None:
        # the lifter row of a frame is the NEAREST of n_ceps_windows rows to the frame's smoothed log F0: a discontinuous function,
        # a contour that lands within rounding of a midpoint selects the neighbouring row (in the engine or in the oracle)
        ix_hip = eng.stage("ceps_index").cpu().numpy().view(np.int32)[ii, :ll]
        ix_64, ix_32 = om64.cepstral_window_index(s64["f0"])[0], om32.cepstral_window_index(s32["f0"])[0]
        print("   lifter rows that differ from the float64 oracle's: hip", np.nonzero(ix_hip != ix_64)[0], "float32 port", np.nonzero(ix_32 != ix_64)[0])
    if "envelope" in s64:
        env = np.abs(s64["envelope"][0])
        print("   max |H|", f"{env.max():.2e}", "min |H|", f"{env.min():.2e}", "| audio error per frame:",
              " ".join(f"{np.abs(got[ii, t*300:(t+1)*300]-a64[0, t*300:(t+1)*300]).max():.0e}" for t in range(ll)))
