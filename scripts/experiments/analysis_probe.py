import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from mbexwn_vocoder_amd.config import canonical_config
from mbexwn_vocoder_amd import analysis
cfg = canonical_config("SPEECH")["preprocess_config"]
n = int(sys.argv[1])
rng = np.random.default_rng(0)
snd = (0.1 * rng.normal(size=(1, n))).astype(np.float32)
t=time.time(); ref, _ = analysis.compute_log_mel(snd, cfg); print('host', ref.shape, round(time.time()-t,1), flush=True)
dev, _ = analysis.compute_log_mel_device(torch.as_tensor(snd).cuda(), cfg)
d = np.abs(dev.cpu().numpy() - ref)
print('max diff', float(d.max()), 'first/last frames', float(d[:, :10].max()), float(d[:, -10:].max()))
assert d.max() < 1e-3
print('OK')
