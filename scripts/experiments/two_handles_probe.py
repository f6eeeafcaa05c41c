#!/usr/bin/env python3
"""One-off probe (GPU box): two handles driven from two host threads on two HIP streams at the same time."""
import os, sys, threading
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from helpers import build_case, synthetic_inputs
from mbexwn_vocoder_amd.engine import MBExWNEngine
cfgs = [build_case("SPEECH", {}), build_case("VOICE", {})]
engs = [MBExWNEngine(*cc) for cc in cfgs]
inputs = [synthetic_inputs(40 + ii, 3, 200 + 37 * ii) for ii in range(2)]
refs = [engs[ii].forward(torch.as_tensor(inputs[ii][0]).cuda(), noise=torch.as_tensor(inputs[ii][1]).cuda()).cpu().numpy() for ii in range(2)]
errs = []
def work(ii):
    try:
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            mel, noise = torch.as_tensor(inputs[ii][0]).cuda(), torch.as_tensor(inputs[ii][1]).cuda()
            for rep in range(150):
                out = engs[ii].forward(mel, noise=noise)
                if rep % 10 == 0:
                    st.synchronize()
                    if not np.array_equal(out.cpu().numpy(), refs[ii]):
                        errs.append((ii, rep))
            st.synchronize()
    except Exception as ee:          # noqa: BLE001
        errs.append((ii, repr(ee)))
ths = [threading.Thread(target=work, args=(ii,)) for ii in range(2)]
[tt.start() for tt in ths]; [tt.join() for tt in ths]
print("errors:", errs)
assert not errs
print("OK")
