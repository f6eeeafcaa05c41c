#!/usr/bin/env python3
"""Three-way parity table at the BASELINE lengths: reference float32 run <-> reference float64 run,
HIP <-> each, and HIP given the float32 run's contour -- tests/golden/reference_long_*.npz, default handle.  GPU box:

    python scripts/parity_table.py gpurun_out/r06_parity_lengths.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch
    import test_gpu_long as tl
    g32 = np.load(os.path.join(ROOT, "tests", "golden", "reference_long_f32.npz"))
    g64 = np.load(os.path.join(ROOT, "tests", "golden", "reference_long_f64.npz"))
    table = {}
    for case in tl.LONG_CASES:
        eng = tl.get_engine(case)
        res, _ = tl.three_way(torch, eng, g32, g64, case)
        res["form"] = eng.conv_form_info()["form"]
        table[case] = res
        print(case, json.dumps(res), flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            json.dump(table, fh, indent=1)


if __name__ == "__main__":
    main()
