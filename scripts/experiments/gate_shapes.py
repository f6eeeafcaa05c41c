"""Gate-kernel time per launch for the F(4,3) block shapes at several launch sizes (tune_gate_shape pins the shape of
launches below 4 x 768 full blocks; larger launches always take the 256-row shape).  Run on the GPU box:
    python scripts/experiments/gate_shapes.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    sizes = [(1, 80), (1, 160), (1, 240), (1, 320), (1, 400), (1, 480), (1, 560), (1, 640), (1, 800), (1, 1200), (2, 800), (4, 800), (8, 800)]
    for shape in (1, 2, 3, 0):            # 256-row, product-split, product-split half column tiles, the launch-size rule
        bench._ENGINES.clear()
        cfg, raw, wt, dims, eng = bench.build_engine("SPEECH", None, tune={"gate_shape": shape})
        row = []
        for batch, frames in sizes:
            rng = np.random.default_rng(1)
            mel_h, noise_h = bench.synthetic_batch(rng, batch, frames, dims.steps_per_frame)
            mel, noise = torch.as_tensor(mel_h).cuda(), torch.as_tensor(noise_h).cuda()
            for _ in range(3):
                eng.forward(mel, noise=noise)
            eng.profile_enable(True)
            for _ in range(10):
                eng.forward(mel, noise=noise)
            torch.cuda.synchronize()
            ms, cnt = eng.profile_read("gate")
            eng.profile_enable(False)
            row.append(f"{batch}x{frames}: {ms / cnt * 1e3:7.1f} us ({eng.gate_form(batch, frames)[13:] or 'f43'})")
        print(f"tune_gate_shape={shape}  " + "  ".join(row), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
