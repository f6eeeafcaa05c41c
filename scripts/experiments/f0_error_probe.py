#!/usr/bin/env python3
"""GPU box: error of the engine's F0 contour and audio against the float64 oracle and against the numpy float32 port of the
graph (canonical SPEECH model, the bench's weights and inputs; 1 x FRAMES frames).  VERDICT round 4, item 3."""
import os, sys, json
import numpy as np, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
import bench
from oracle.mbexwn_oracle import OracleModel
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 240
kw = {}
if len(sys.argv) > 2:
    kw = json.loads(sys.argv[2])
cfg, raw, wt, dims, eng = bench.build_engine("SPEECH", **kw)
mel, noise = bench.synthetic_batch(np.random.default_rng(42), 1, frames, dims.steps_per_frame)
got = eng.forward(torch.as_tensor(mel).cuda(), noise=torch.as_tensor(noise).cuda()).cpu().numpy()
f0 = eng.stage("f0").cpu().numpy()
om = OracleModel(cfg, raw, wt)
ref, st = om.forward(mel, noise, return_stages=True)
om32 = OracleModel(cfg, raw, wt, dtype=np.float32)
ref32, st32 = om32.forward(mel, noise, return_stages=True)
out = {"frames": frames, "engine_kw": kw, "f0_range": [float(st["f0"].min()), float(st["f0"].max())],
       "f0_err_hip": float(np.abs(f0 - st["f0"]).max()), "f0_err_numpy_f32_port": float(np.abs(st32["f0"] - st["f0"]).max()),
       "audio_err_hip": float(np.abs(got - ref).max()), "audio_err_numpy_f32_port": float(np.abs(ref32 - ref).max()),
       "audio_max": float(np.abs(ref).max()), "conv_form": eng.conv_form_info()["form"]}
print(json.dumps(out))
