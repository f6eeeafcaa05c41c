import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")); sys.path.insert(0, '' + os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests") + '')
import numpy as np, torch
from helpers import build_case, synthetic_inputs
from test_gpu_forms import stressed_weights
from mbexwn_vocoder_amd.engine import MBExWNEngine
for gain in (3.0, 4.0, 5.0, 6.0, 8.0):
    for bias in (0.0, 0.3):
        cfg, raw, wt = build_case("SING", {})
        raw = stressed_weights(raw, gain, bias)
        eng = MBExWNEngine(cfg, raw, wt)
        i0 = eng.conv_form_info()
        mel, noise = synthetic_inputs(4242, 2, 60)
        loud = np.clip(mel + np.float32(4.0), -11.5, 2.0).astype(np.float32)
        i1 = eng.calibrate(torch.as_tensor(loud).cuda(), noise=torch.as_tensor(noise).cuda())
        i2 = eng.calibrate(torch.as_tensor(mel).cuda(), noise=torch.as_tensor(noise).cuda())
        f = lambda i: (i['form'], None if i['err_f43'] is None else round(i['err_f43'],6), None if i['err_f23'] is None else round(i['err_f23'],6), round(i['threshold'],6), round(i['ref_max'],2))
        print(gain, bias, 'create', f(i0), 'loud', f(i1), 'plain', f(i2), flush=True)
        eng.close()
