import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
sys.argv=['bench.py']
import bench
from mbexwn_vocoder_amd.streaming import StreamingSynthesizer
cfg, raw, wt, dims, eng = bench.build_engine("SPEECH")
syn = StreamingSynthesizer(eng, chunk_frames=8)
n_streams=64; n_ticks=14
total=(n_ticks+1)*8+syn.right+8
for sid in range(n_streams):
    syn.open(sid)
    mm, nn = bench.synthetic_batch(np.random.default_rng(sid), 1, total, dims.steps_per_frame)
    syn.push(sid, mm[0], nn[0])
for t in range(6): syn.tick()
eng.profile_enable(True)
for t in range(6): syn.tick()
torch.cuda.synchronize()
tot=0
for k in ("frontend","wavetable","gate0","gate","res_skip","tail","pqmf","stft_filter","overlap_add"):
    ms,n=eng.profile_read(k); tot+=ms
    print(f"{k:12s} {ms/6*1e3:8.1f} us per tick  ({n//6} launches)")
print('sum', tot/6*1e3, 'layer_rows', syn.last_tick_layer_rows)
print('wavenet frames per tick per stream', syn.last_tick_wavenet_frames/64, 'active', syn.last_tick_active_frames/64, 'window', syn.last_tick_frames/64)
