#!/usr/bin/env python3
"""One-off probe (GPU box): per-item frame counts outside [0, frames] must behave like their clamped values."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from helpers import build_case, synthetic_inputs
from mbexwn_vocoder_amd.engine import MBExWNEngine
for over in ({}, {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3}):
    cfg, raw, wt = build_case("SPEECH", over)
    eng = MBExWNEngine(cfg, raw, wt)
    mel, noise = synthetic_inputs(3, 4, 40)
    m, n = torch.as_tensor(mel).cuda(), torch.as_tensor(noise).cuda()
    good = eng.forward(m, n_frames=torch.tensor([40, 0, 17, 40], dtype=torch.int32, device="cuda"), noise=n).cpu().numpy()
    bad = eng.forward(m, n_frames=torch.tensor([41, -5, 17, 2 ** 30], dtype=torch.int32, device="cuda"), noise=n).cpu().numpy()
    assert np.array_equal(good, bad)
    print("C", eng.dims.wn_channels, "ok")
print("OK")
