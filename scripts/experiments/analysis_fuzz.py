#!/usr/bin/env python3
"""One-off fuzz (GPU box): audio -> log-mel on the device against the host analysis for random ragged batches and a few
analysis geometries (window / hop / FFT size / mel channels)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mbexwn_vocoder_amd import analysis
from mbexwn_vocoder_amd.config import canonical_config
base = canonical_config("SPEECH")["preprocess_config"]
geoms = [{}, {"win_size": 800, "hop_size": 200, "fft_size": 1024, "sample_rate": 16000, "fmax": 8000},
         {"win_size": 512, "hop_size": 128, "fft_size": 512, "mel_channels": 40}, {"win_size": 1024, "hop_size": 256, "fft_size": 2048}]
fails = 0
for case in range(int(sys.argv[1])):
    rng = np.random.default_rng(int(sys.argv[2]) + case)
    cfg = dict(base, **geoms[int(rng.integers(0, len(geoms)))])
    win, hop = int(cfg.get("win_size", cfg["fft_size"])), int(cfg["hop_size"])
    B = int(rng.integers(1, 6))
    N = int(rng.integers(win // 2 + 1, 40 * hop))
    lens = [N] + [int(rng.integers(0, N + 1)) for _ in range(B - 1)]
    snd = np.zeros((B, N), dtype=np.float32)
    for ii, ll in enumerate(lens):
        snd[ii, :ll] = (10 ** rng.uniform(-4, 0)) * rng.normal(size=ll)
    dev, _ = analysis.compute_log_mel_device(torch.as_tensor(snd).cuda(), cfg, n_samples=torch.tensor(lens, dtype=torch.int32, device="cuda"))
    dev = dev.cpu().numpy()
    worst = 0.0
    for ii, ll in enumerate(lens):
        if ll < win // 2 + 1:
            continue                                       # shorter than the reflect padding: numpy cannot pad it either
        ref, _ = analysis.compute_log_mel(snd[ii:ii + 1, :ll], cfg)
        nfr = ll // hop + 1
        worst = max(worst, float(np.abs(dev[ii, :nfr] - ref[0, :nfr]).max()))
    ok = worst < 2e-3 and np.all(np.isfinite(dev))
    fails += not ok
    print(case, "OK  " if ok else "FAIL", f"{worst:.1e}", "B", B, "N", N, "win/hop/fft", win, hop, cfg["fft_size"], flush=True)
print("failures:", fails)
sys.exit(1 if fails else 0)
