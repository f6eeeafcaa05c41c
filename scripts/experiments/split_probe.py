#!/usr/bin/env python3
"""One-off measurement (GPU box): res/skip launch time of the float32 kernels and of the opt-in split half precision
(mbx_config.wn_precision) at the config-3 size, HIP events per launch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch
for batch, frames in ((16, 800), (1, 800)):
    for prec in ("f32", "split_f16"):
        cfg, raw, wt, dims, eng = bench.build_engine("SING", None, precision=prec)
        mel_h, noise_h = bench.synthetic_batch(np.random.default_rng(1), batch, frames, 20)
        mel, noise = torch.as_tensor(mel_h).cuda(), torch.as_tensor(noise_h).cuda()
        for _ in range(3):
            eng.forward(mel, noise=noise)
        eng.profile_enable(True)
        for _ in range(10):
            eng.forward(mel, noise=noise)
        torch.cuda.synchronize()
        out = {kk: eng.profile_read(kk) for kk in ("res_skip", "res_skip_f16", "gate")}
        eng.profile_enable(False)
        print(eng.conv_form_info()["split_f16_gate_layers"], end=" ")
        print(batch, frames, prec, {kk: (round(vv[0] / vv[1] * 1e3, 1) if vv[1] else None, vv[1]) for kk, vv in out.items()}, flush=True)
