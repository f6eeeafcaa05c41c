#!/usr/bin/env python3
"""Round 6: where a wave of wn_gate_f16w_kernel (split precision, 256 x 128 tiles) spends its cycles, from in-kernel stamps
(mkexp.py gw_stamp: s_memtime around the wait, the barrier, the request code and the operand reads + MFMAs of every tap):

    EXP_FILE=wn_gate_f16.hip python scripts/experiments/mkexp.py gw_stamp:gw_stamp
    gpurun -- 'MBX_LIB_PATH=$PWD/scripts/experiments/libs/lib_gw_stamp.so python scripts/experiments/gate_f16w_account.py'
"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch
from mbexwn_vocoder_amd import engine
cfg, raw, wt, dims, eng = bench.build_engine("SING", None, precision="split_f16")
lib = engine.load_library()
lib.mbx_exp_stamps.restype = ctypes.c_int
lib.mbx_exp_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
mel_h, noise_h = bench.synthetic_batch(np.random.default_rng(1), 16, 800, 20)
mel, noise = torch.as_tensor(mel_h).cuda(), torch.as_tensor(noise_h).cuda()
for _ in range(5):
    eng.forward(mel, noise=noise)
torch.cuda.synchronize()
eng.profile_enable(True)
for _ in range(3):
    eng.forward(mel, noise=noise)
torch.cuda.synchronize()
gms, gn = eng.profile_read("gate")
eng.profile_enable(False)
buf = np.zeros((8192, 8, 8), dtype=np.uint64)
assert lib.mbx_exp_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
ok = buf[:, :, 3] != 0
st = buf[ok].astype(np.int64)
print("kernels", eng.conv_form_info()["gate_kernels"], "gate launch (stamped build) %.1f us, blocks with stamps %d" % (gms / gn * 1e3, int(ok[:, 0].sum())))
t0, t1, t2, t3, aw, ab, ai, ac = (st[:, ii] for ii in range(8))
def med(x):
    return float(np.median(x))
tot = med(t3 - t0)
print("per wave, median cycles: block %.0f = prologue %.0f + K loop %.0f + epilogue %.0f" % (tot, med(t1 - t0), med(t2 - t1), med(t3 - t2)))
print("K loop (30 taps): s_waitcnt vmcnt %.0f, barrier %.0f, request code %.0f, operand reads + MFMAs %.0f (matrix-pipe time of a wave: %d)" %
      (med(aw), med(ab), med(ai), med(ac), 30 * 48 * 16))
print("per tap: wait %.0f barrier %.0f requests %.0f reads+MFMAs %.0f (48 MFMAs = 768 cycles; two waves share a SIMD)" % (med(aw) / 30, med(ab) / 30, med(ai) / 30, med(ac) / 30))
