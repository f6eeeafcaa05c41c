#!/usr/bin/env python3
"""Round 6: one utterance (3 s, 10 s) through the 12-layer model (C = 320, d <= 2048): ms per forward, kernel per gate layer."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch
cfg, raw, wt, dims, eng = bench.build_engine("SPEECH", {"mbexwn_config:pp_mod_subnet:n_layers": 12})
for batch, frames, reps in ((1, 240, 200), (1, 800, 200), (4, 800, 50)):
    mel_h, noise_h = bench.synthetic_batch(np.random.default_rng(1), batch, frames, 20)
    mel, noise = torch.as_tensor(mel_h).cuda(), torch.as_tensor(noise_h).cuda()
    out = torch.empty((batch, frames * 300), device="cuda")
    for _ in range(30):
        eng.forward(mel, noise=noise, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.forward(mel, noise=noise, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    eng.profile_enable(True)
    for _ in range(3):
        eng.forward(mel, noise=noise, out=out)
    torch.cuda.synchronize()
    per = eng.profile_read_launches("gate")
    eng.profile_enable(False)
    kern = eng.conv_form_info()["gate_kernels"]
    n = len(kern) - 1
    layer_us = [round(float(np.mean([per[f * n + i] for f in range(3)])) * 1e3, 1) for i in range(n)]
    print(f"{batch} x {frames} frames, 12 layers: {ms:.3f} ms per forward = {batch * frames / 80 / ms * 1e3:.0f} x real time; gate kernels {kern[1:]}; us per gate layer {layer_us}", flush=True)
