#!/usr/bin/env python3
"""One-off probe (GPU box): items of zero frames inside a batch, every convolution form."""
import os, sys
os.environ["MBX_EXPERIMENT"] = "1"      # opt in to the MBX_* experiment variables (engine.experiment_overrides)
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from helpers import build_case, synthetic_inputs
from mbexwn_vocoder_amd.engine import MBExWNEngine
for over in ({}, {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3}):
    cfg, raw, wt = build_case("SPEECH", over)
    for form in ("4", "2", "0", "44"):
        os.environ["MBX_WINOGRAD"] = form
        eng = MBExWNEngine(cfg, raw, wt)
        mel, noise = synthetic_inputs(3, 4, 30)
        nf = torch.tensor([30, 0, 11, 1], dtype=torch.int32, device="cuda")
        got = eng.forward(torch.as_tensor(mel).cuda(), n_frames=nf, noise=torch.as_tensor(noise).cuda()).cpu().numpy()
        assert np.all(np.isfinite(got)) and np.all(got[1] == 0.0), form
        for ii, ll in ((0, 30), (2, 11), (3, 1)):
            one = eng.forward(torch.as_tensor(mel[ii:ii + 1, :ll]).cuda(), noise=torch.as_tensor(noise[ii:ii + 1, :ll * 20]).cuda()).cpu().numpy()[0]
            d = np.abs(got[ii, :ll * 300] - one).max()
            assert d <= 2e-5 * max(1.0, np.abs(one).max()), (form, ii, d)
            assert np.all(got[ii, ll * 300:] == 0.0)
        print("C", eng.dims.wn_channels, "form", form, "ok", flush=True)
        del eng
print("OK")
