#!/usr/bin/env python3
"""N forwards of one benchmark geometry (for rocprofv3 passes): python one_forward.py <voice> <batch> <frames> [n]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch
voice, batch, frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg, raw, wt, dims, eng = bench.build_engine(voice, None)
mel_h, noise_h = bench.synthetic_batch(np.random.default_rng(1), batch, frames, dims.steps_per_frame)
mel, noise = torch.as_tensor(mel_h).cuda(), torch.as_tensor(noise_h).cuda()
for _ in range(int(sys.argv[4]) if len(sys.argv) > 4 else 8):
    eng.forward(mel, noise=noise)
torch.cuda.synchronize()
