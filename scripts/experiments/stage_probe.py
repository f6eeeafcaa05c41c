#!/usr/bin/env python3
"""One-off measurement (GPU box): step time and per-stage launch times (HIP events around every launch, mbx_profile) of a
forward of the canonical model, for A/B runs of ablation libraries through MBX_LIB_PATH:
    python scripts/experiments/stage_probe.py [batch frames [repeats]]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch
batch, frames = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 800)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg, raw, wt, dims, eng = bench.build_engine("SING", None)
mel_h, noise_h = bench.synthetic_batch(np.random.default_rng(1), batch, frames, 20)
mel, noise = torch.as_tensor(mel_h).cuda(), torch.as_tensor(noise_h).cuda()
for _ in range(3):
    eng.forward(mel, noise=noise)
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(reps):
    eng.forward(mel, noise=noise)
t1.record()
torch.cuda.synchronize()
step = t0.elapsed_time(t1) / reps
eng.profile_enable(True)
for _ in range(reps):
    eng.forward(mel, noise=noise)
torch.cuda.synchronize()
out = {}
for kk in ("gate", "res_skip", "frontend", "wavetable", "start", "tail", "pqmf", "stft_filter", "overlap_add"):
    ms, n = eng.profile_read(kk)
    out[kk] = round(ms / n * 1e3, 1) if n else None
eng.profile_enable(False)
print(os.environ.get("MBX_LIB_PATH", "product").split("/")[-1], f"step {step:.3f} ms", out, flush=True)
