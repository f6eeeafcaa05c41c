#!/usr/bin/env python3
"""One-off probe (GPU box): model geometries far from the canonical one (other band counts, hop sizes, sample rates, channel
counts) through the engine against the float64 oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mbexwn_vocoder_amd.config import ModelDims, canonical_config
from mbexwn_vocoder_amd.engine import MBExWNEngine
from mbexwn_vocoder_amd.tables import WaveTables
from mbexwn_vocoder_amd.weights import synthetic_weights
from oracle.mbexwn_oracle import OracleModel
P, M = "preprocess_config:", "mbexwn_config:"
W = M + "pp_mod_subnet:"
CASES = {
    "12 bands, 4 folded samples": {M + "multi_band_config": {"subbands": 12, "taps": 96, "cutoff_ratio": 0.05, "beta": 9.0},
                                   M + "pulse_channels": 4, W + "cond_lin_upsampling": 5, W + "n_channels": 48, W + "n_layers": 3},
    "6 bands, 2 folded samples": {M + "multi_band_config": {"subbands": 6, "taps": 48, "cutoff_ratio": 0.1, "beta": 9.0},
                                  M + "pulse_channels": 2, W + "cond_lin_upsampling": 10, W + "n_channels": 32, W + "n_layers": 3},
    "30 bands (generic PQMF / tail)": {M + "multi_band_config": {"subbands": 30, "taps": 240, "cutoff_ratio": 0.02, "beta": 9.0},
                                        M + "pulse_channels": 10, W + "cond_lin_upsampling": 5, W + "n_channels": 32,
                                        W + "n_layers": 2, W + "n_out_channels": 60},
    "16 kHz, hop 200, 10 bands": {P + "sample_rate": 16000, P + "hop_size": 200, P + "win_size": 800, P + "fft_size": 1024,
                                  M + "pulse_rate_factor": 2, M + "pulse_channels": 5,
                                  M + "multi_band_config": {"subbands": 10, "taps": 80, "cutoff_ratio": 0.06, "beta": 9.0},
                                  W + "cond_lin_upsampling": 10, W + "n_channels": 64, W + "n_layers": 3, M + "ps_max_ceps_coefs": 120},
    "40 mel channels, 44 out channels": {P + "mel_channels": 40, W + "n_out_channels": 44, W + "n_channels": 40, W + "n_layers": 3},
    "kernel size 5": {W + "kernel_size": 5, W + "n_channels": 32, W + "n_layers": 3},
}
only = sys.argv[1:] 
for name, over in CASES.items():
    if only and not any(oo in name for oo in only):
        continue
    cfg = canonical_config("SPEECH", **over)
    dims = ModelDims(cfg)
    raw = synthetic_weights(cfg, seed=77, bias_std=0.05, alpha_jitter=0.05)
    wt = WaveTables(sample_rate=dims.pulse_rate, **cfg["mbexwn_config"]["wavetable_config"])
    eng, om = MBExWNEngine(cfg, raw, wt), OracleModel(cfg, raw, wt)
    rng = np.random.default_rng(5)
    B, T = 2, 21
    mel = np.clip(np.log(np.exp(rng.normal(-5.0, 2.0, size=(B, T, dims.mel_channels))) + 1e-5), -11.5, 2.0).astype(np.float32)
    noise = rng.normal(size=(B, T * dims.wn_in_rows_per_frame)).astype(np.float32)
    lengths = (T, 8)
    got = eng.forward(torch.as_tensor(mel).cuda(), n_frames=torch.tensor(lengths, dtype=torch.int32, device="cuda"),
                      noise=torch.as_tensor(noise).cuda()).cpu().numpy()
    worst = 0.0
    for ii, ll in enumerate(lengths):
        ref = om.forward(mel[ii:ii + 1, :ll], noise[ii:ii + 1, :ll * dims.wn_in_rows_per_frame])[0]
        dd = float(np.abs(got[ii, :ll * dims.hop_size] - ref).max()) / max(1.0, float(np.abs(ref).max()))
        worst = max(worst, dd)
        assert np.all(got[ii, ll * dims.hop_size:] == 0.0)
    print(f"{name:36s} rel max diff {worst:.2e}", "OK" if worst <= 1e-4 else "FAIL", flush=True)
    assert worst <= 1e-4, name
print("OK")
