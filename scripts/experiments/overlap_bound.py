#!/usr/bin/env python3
"""Round 6: what overlap across the layer boundary could buy a single utterance, at most.

Alternates, on ONE box, the product library and `mkexp.py ovl:overlap` (EXP_FILE=mbx_api.hip: res/skip of layer l on a second
stream, the gate of layer l+1 issued without waiting for it -- wrong audio, timing only) in fresh processes and prints the
forward time of one utterance of 3 s and 10 s (and 16 x 10 s for scale):

    EXP_FILE=mbx_api.hip python scripts/experiments/mkexp.py ovl:overlap ovl2:overlap+overlap2      (ovl2: no events per layer)
    gpurun -- 'python scripts/experiments/overlap_bound.py > gpurun_out/overlap_bound.txt'
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %r)
import bench, torch
cfg, raw, wt, dims, eng = bench.build_engine("SPEECH", None)
for batch, frames, reps in ((1, 240, 400), (1, 800, 300), (16, 800, 20)):
    mel_h, noise_h = bench.synthetic_batch(np.random.default_rng(1), batch, frames, 20)
    mel, noise = torch.as_tensor(mel_h).cuda(), torch.as_tensor(noise_h).cuda()
    out = torch.empty((batch, frames * 300), device="cuda")
    for _ in range(30):
        eng.forward(mel, noise=noise, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.forward(mel, noise=noise, out=out)
    torch.cuda.synchronize()
    print("%%s  %%2d x %%4d frames: %%.4f ms per forward" %% (os.environ.get("MBX_LIB_PATH", "product").split("/")[-1], batch, frames, (time.perf_counter() - t0) / reps * 1e3), flush=True)
''' % ROOT


def main():
    libs = [os.path.join(ROOT, "scripts", "experiments", "libs", nn) for nn in ("lib_ovl.so", "lib_ovl2.so")]
    for rnd in range(2):
        for path in [None] + [pp for pp in libs if os.path.exists(pp)]:
            env = dict(os.environ)
            env.pop("MBX_LIB_PATH", None)
            if path:
                env["MBX_LIB_PATH"] = path
            subprocess.run([sys.executable, "-c", CHILD], env=env, check=True)


if __name__ == "__main__":
    main()
