#!/usr/bin/env python3
"""One-off measurement (GPU box): gate / res-skip time per launch against the number of blocks of the launch, around whole
multiples of the resident block count (768 gate blocks = 3 per CU, 512 res/skip blocks), to see how much of a large launch
is round quantisation (fill / drain) and how much is per-row work.  Canonical model, batch 16, frames swept.
    python scripts/experiments/gate_staircase.py [frames ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    frames_list = [int(vv) for vv in sys.argv[1:]] or [230, 246, 262, 290, 300, 307, 308, 320, 340, 360, 384, 385, 400, 614, 615, 700, 768, 769, 800]
    cfg, raw, wt, dims, eng = bench.build_engine("SING")
    print("form:", eng.conv_form_info())
    for frames in frames_list:
        batch = 16
        rng = np.random.default_rng(1)
        mel_h, noise_h = bench.synthetic_batch(rng, batch, frames, dims.steps_per_frame)
        mel, noise = torch.as_tensor(mel_h).cuda(), torch.as_tensor(noise_h).cuda()
        for _ in range(3):
            eng.forward(mel, noise=noise)
        eng.profile_enable(True)
        for _ in range(8):
            eng.forward(mel, noise=noise)
        torch.cuda.synchronize()
        gms, gcnt = eng.profile_read("gate")
        rms, rcnt = eng.profile_read("res_skip")
        eng.profile_enable(False)
        rows = frames * dims.steps_per_frame
        gblocks = -(-rows // 256) * batch * 10
        rblocks = -(-rows // 128) * batch
        print(f"16 x {frames:4d} frames: gate {gms / gcnt * 1e3:7.1f} us  {gblocks:6d} blocks = {gblocks / 768:6.2f} rounds  "
              f"{gms / gcnt * 1e6 / (rows * batch):6.3f} ns/row | res/skip {rms / rcnt * 1e3:7.1f} us {rblocks:6d} blocks = "
              f"{rblocks / 512:6.2f} rounds {rms / rcnt * 1e6 / (rows * batch):6.3f} ns/row", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
