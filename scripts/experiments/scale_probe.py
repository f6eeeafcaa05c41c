#!/usr/bin/env python3
"""One-off scale probe (GPU box): a batch of long utterances through the default kernels and through the direct form
(MBX_WINOGRAD=0), which address differently; the two must agree over the whole length.  Not part of the test suite (tens of
GB of workspace); tests/test_gpu_parity.py::test_item_longer_than_4_gib_of_activation_rows is the permanent, smaller case."""
import os
os.environ["MBX_EXPERIMENT"] = "1"      # opt in to the MBX_* experiment variables (engine.experiment_overrides)
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from helpers import build_case, synthetic_inputs  # noqa: E402
from mbexwn_vocoder_amd.engine import MBExWNEngine  # noqa: E402

B, T = int(sys.argv[1]), int(sys.argv[2])
cfg, raw, wt = build_case(os.environ.get("PROBE_VOICE_TYPE", "SPEECH"), {})      # VOICE: C = 340 (partial column tiles)
period = min(500, T)
base_mel, _ = synthetic_inputs(8, B, period)
mel = torch.as_tensor(np.tile(base_mel, (1, T // period, 1))).cuda()
noise = torch.randn((B, T * 20), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
nf = torch.tensor([max(T - (977 * ii) % T, 1) for ii in range(B)], dtype=torch.int32, device="cuda")
outs = {}
generic = os.environ.get("PROBE_GENERIC", "0") == "1"      # second path: a handle without weight images (generic kernels)
for form in ("4", "0"):
    os.environ["MBX_WINOGRAD"] = "4" if generic else form
    eng = MBExWNEngine(cfg, raw, wt, weight_images=not (generic and form == "0"))
    outs[form] = eng.forward(mel, n_frames=nf, noise=noise)
    torch.cuda.synchronize()
    print(form, "workspace GiB", round(eng._lib.mbx_workspace_size(eng._handle, B, T) / 2 ** 30, 1), flush=True)
    del eng
bad = ((outs["4"] - outs["0"]).abs() > 1e-3).nonzero()
if bad.numel():
    first = int(bad[:, 1].min())
    print("first divergence at sample", first, "= frame", first // 300, "= row", first // 15, flush=True)
    # which path is off?  the prefix of a shorter run (verified size) is the reference for the samples it covers
    Tp = min(T, 200000)
    os.environ["MBX_WINOGRAD"] = "4"
    eng = MBExWNEngine(cfg, raw, wt)
    pre = eng.forward(mel[:, :Tp].contiguous(), noise=noise[:, :Tp * 20].contiguous())
    for kk in ("4", "0"):
        dd = (outs[kk][:, :(Tp - 20) * 300] - pre[:, :(Tp - 20) * 300]).abs()
        print("path", kk, "vs the", Tp, "frame run: max diff", float(dd.max()), flush=True)
    del eng
amp = float(outs["0"].abs().max())
diff = (outs["4"] - outs["0"]).abs()
print("batch", B, "frames", T, "amp", amp, "max diff", float(diff.max()), "per item (first 8)", [float(dd.max()) for dd in diff[:8]])
for ii in range(B):
    ll = int(nf[ii]) * 300
    assert float(outs["4"][ii, ll:].abs().max()) == 0.0 if ll < T * 300 else True
assert float(diff.max()) <= 5e-5 * max(1.0, amp)
print("OK")
