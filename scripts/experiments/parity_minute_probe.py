#!/usr/bin/env python3
"""Round 6, one-off: the distance HIP <-> reference float32 run at 30 s and 60 s (C = 320).  The fixtures are generated in the
build container by the same mechanism as tests/golden/make_reference_long.py (run_case("SPEECH", {}, 1, 2400 | 4800, float32)) into
tests/golden/_scratch_long60_f32.npz -- 13 MB, NOT committed; this probe prints what DESIGN.md section 5 quotes."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import build_case
from mbexwn_vocoder_amd.engine import MBExWNEngine
from oracle.mbexwn_oracle import OracleModel
g = np.load(os.path.join(ROOT, "tests", "golden", "_scratch_long60_f32.npz"))
cfg, raw, wt = build_case("SPEECH", {})
eng = MBExWNEngine(cfg, raw, wt)
om = OracleModel(cfg, raw, wt)
for case in ("speech2400", "speech4800"):
    mel, noise, ref = g[case + "/mell"], g[case + "/noise"], g[case + "/audio"]
    got = eng.forward(torch.as_tensor(mel).cuda(), noise=torch.as_tensor(noise).cuda()).cpu().numpy()
    eng.infer_components(mel, F0=g[case + "/f0"], noise=noise)
    inj = eng.last_audio.cpu().numpy()
    amp = float(np.abs(ref).max())
    d = np.abs(got.astype(np.float64) - ref)[0]
    per10 = [float(d[i:i + 240000].max()) for i in range(0, d.size, 240000)]
    orc = om.forward(mel, noise)
    print(f"{case}: A {amp:.2f}  HIP-ref32 {d.max():.3e} = {d.max() / amp / 1e-4:.2f} x 1e-4 A   per 10 s: {['%.1e' % v for v in per10]}   "
          f"HIP given ref32 contour - ref32 {np.abs(inj.astype(np.float64) - ref).max():.3e}   HIP-oracle {np.abs(got - orc).max():.3e}   "
          f"oracle-ref32 {np.abs(orc - ref).max():.3e}", flush=True)
