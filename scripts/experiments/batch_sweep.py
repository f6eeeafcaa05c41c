#!/usr/bin/env python3
"""One-off measurement (GPU box): throughput of the canonical model against batch size and utterance length."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from helpers import build_case
from mbexwn_vocoder_amd.engine import MBExWNEngine
cfg, raw, wt = build_case("SPEECH", {})
eng = MBExWNEngine(cfg, raw, wt)
for T in (240, 800, 2400):
    for B in (1, 2, 4, 8, 16, 32, 64, 128):
        if B * T > 128 * 800:
            continue
        mel = torch.randn((B, T, 80), device="cuda") * 2 - 5
        noise = torch.randn((B, T * 20), device="cuda")
        for _ in range(3):
            eng.forward(mel, noise=noise)
        torch.cuda.synchronize()
        n = max(5, int(0.2 / (0.0007 * B * T / 800)))
        t0 = time.perf_counter()
        for _ in range(n):
            eng.forward(mel, noise=noise)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"{T / 80:5.1f} s x {B:3d}: {ms:8.3f} ms  {B * T * 300 / 24000 / (ms / 1e3):9.0f} x real time", flush=True)
