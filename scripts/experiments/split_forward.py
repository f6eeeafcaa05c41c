#!/usr/bin/env python3
"""Five config-3 forwards with the opt-in split half precision of the res/skip layers (for rocprofv3 passes)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch
cfg, raw, wt, dims, eng = bench.build_engine("SING", None, precision=sys.argv[1] if len(sys.argv) > 1 else "split_f16")
mel_h, noise_h = bench.synthetic_batch(np.random.default_rng(1), 16, 800, 20)
mel, noise = torch.as_tensor(mel_h).cuda(), torch.as_tensor(noise_h).cuda()
for _ in range(5):
    eng.forward(mel, noise=noise)
torch.cuda.synchronize()
