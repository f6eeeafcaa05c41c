#!/usr/bin/env python3
"""Round 6: fixed cost per block of the split-precision gate kernel.  One block per CU: its launch time is rounds x (F + nk S)
with nk = C / 32 K steps per block and rounds = blocks / CUs; C = 160, 224, 320 (16 x 10 s) give F and S."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch
pts = []
for C in (160, 224, 320):
    cfg, raw, wt, dims, eng = bench.build_engine("SING", {"mbexwn_config:pp_mod_subnet:n_channels": C}, precision="split_f16")
    mel_h, noise_h = bench.synthetic_batch(np.random.default_rng(1), 16, 800, 20)
    mel, noise = torch.as_tensor(mel_h).cuda(), torch.as_tensor(noise_h).cuda()
    for _ in range(3):
        eng.forward(mel, noise=noise)
    eng.profile_enable(True)
    for _ in range(10):
        eng.forward(mel, noise=noise)
    torch.cuda.synchronize()
    g_ms, g_n = eng.profile_read("gate")
    r_ms, r_n = eng.profile_read("res_skip_f16")
    eng.profile_enable(False)
    info = eng.conv_form_info()
    blocks = 1000 * ((C + 31) // 32)
    rounds = blocks / 256.0
    us = g_ms / max(g_n, 1) * 1e3
    pts.append((C // 32, us / rounds))
    print(f"C {C}: split gate layers {info['split_f16_gate_layers']}, kernels {info['gate_kernels']}, gate {us:.1f} us per launch, "
          f"{blocks} blocks = {rounds:.1f} rounds -> {us / rounds:.2f} us per block (nk = {C // 32}); res/skip {r_ms / max(r_n, 1) * 1e3:.1f} us", flush=True)
    del eng
    bench._ENGINES.clear()
(n0, t0), (n2, t2) = pts[0], pts[-1]
S = (t2 - t0) / (n2 - n0)
print(f"per K step S = {S:.3f} us, fixed per block F = {t0 - n0 * S:.2f} us")
