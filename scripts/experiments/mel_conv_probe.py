"""Time of each mel-rate convolution of the front end as a launch of its own (16 x 800 frames), beside its matrix-pipe time:
which member of the three shared launches of the large-launch front end (profiles: conv1d_mel_group_kernel<2>) costs what.
Run on the GPU box:  python scripts/experiments/mel_conv_probe.py [batch frames]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONVS = [  # name, ks, cin, cout, float64 accumulation
    ("PulsPar_0 (f64 acc)", 3, 80, 128, True), ("PulsPar_1 (f64 acc)", 3, 128, 128, True), ("PulsPar_2 (f64 acc)", 3, 128, 64, True),
    ("PS_0", 3, 80, 256, False), ("PS_1", 3, 256, 256, False), ("PS_final", 1, 256, 240, False), ("wn.cond", 3, 80, 1280, False)]


def main():
    import torch
    batch, frames = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 800)
    cfg, raw, wt, dims, eng = bench.build_engine("SING")
    g = torch.Generator(device="cuda").manual_seed(1)
    for name, ks, cin, cout, f64 in CONVS:
        x = torch.randn((batch, frames, cin), device="cuda", generator=g)
        w = torch.randn((ks, cin, cout), device="cuda", generator=g) * 0.05
        b = torch.zeros(cout, device="cuda")
        for _ in range(5):
            eng.conv1d(x, w, b, pad_l=ks // 2, f64_accumulate=f64)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        e0.record()
        for _ in range(n):
            eng.conv1d(x, w, b, pad_l=ks // 2, f64_accumulate=f64)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        flop = 2.0 * batch * frames * ks * cin * cout
        peak = 78.6e12 if f64 else 157.3e12
        print(f"{name:22s} K {ks * cin:4d} N {cout:5d}  {us:7.1f} us   matrix pipe {flop / peak * 1e6:6.1f} us   frac {flop / peak * 1e6 / us:5.2f}", flush=True)


if __name__ == "__main__":
    main()
