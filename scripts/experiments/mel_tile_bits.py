"""The large-launch mel-rate tile against the same rows in small launches: same bits (ad-hoc check; the committed test is
tests/test_gpu_parity.py::test_mel_tile_large_launch_same_bits)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    cfg, raw, wt, dims, eng = bench.build_engine("SING")
    g = torch.Generator(device="cuda").manual_seed(3)
    bad = 0
    for ks, cin, cout, pad_mode, frames in [(3, 80, 256, 0, 800), (3, 256, 256, 0, 800), (1, 256, 240, 0, 800), (3, 80, 1280, 0, 800),
                                            (3, 80, 128, 1, 777), (3, 88, 132, 2, 801), (5, 24, 36, 0, 790), (3, 16, 4, 0, 800)]:
        batch = 16
        x = torch.randn((batch, frames, cin), device="cuda", generator=g)
        w = torch.randn((ks, cin, cout), device="cuda", generator=g) * 0.05
        b = torch.randn(cout, device="cuda", generator=g)
        al = torch.rand(cout, device="cuda", generator=g)
        big = eng.conv1d(x, w, b, alpha=al, pad_l=ks // 2, pad_mode=pad_mode)
        small = torch.cat([eng.conv1d(x[i:i + 1], w, b, alpha=al, pad_l=ks // 2, pad_mode=pad_mode) for i in range(batch)])
        ref = torch.nn.functional.conv1d(torch.nn.functional.pad(x.double().transpose(1, 2), (ks // 2, ks - 1 - ks // 2),
                                         mode={0: "constant", 1: "reflect", 2: "replicate"}[pad_mode]), w.double().permute(2, 1, 0)) .transpose(1, 2) + b.double()
        ref = torch.where(ref > 0, ref, ref * al.double())
        same = torch.equal(big, small)
        err = float((big.double() - ref).abs().max())
        print(f"ks {ks} cin {cin} cout {cout} pad_mode {pad_mode} frames {frames}: same bits {same}  max|big - small| {float((big - small).abs().max()):.3g}  err vs f64 {err:.3g}", flush=True)
        bad += not same
    print("FAIL" if bad else "OK")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
