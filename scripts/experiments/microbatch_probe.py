#!/usr/bin/env python3
"""Round 6: config 4 (256 utterances of 2-15 s, C = 340, one GPU) against the micro-batch size of the sharded synthesizer."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch
from mbexwn_vocoder_amd.sharding import ShardedSynthesizer
cfg, raw, wt, dims, eng = bench.build_engine("VOICE")
rng = np.random.default_rng(4242)
lengths = [int(vv) for vv in rng.integers(160, 1201, size=256)]
mels, noises = [], []
for ii, ll in enumerate(lengths):
    mm, nn = bench.synthetic_batch(np.random.default_rng(1000 + ii), 1, ll, dims.steps_per_frame)
    mels.append(mm[0]); noises.append(nn[0])
for rnd in range(2):
    for mb in (16, 32, 64, 128):
        syn = ShardedSynthesizer(lambda mel, nfr, noise: eng.forward(mel, n_frames=nfr, noise=noise), dims.hop_size, dims.steps_per_frame,
                                 rank=0, world_size=1, max_batch=mb, max_padded_frames=mb * 1200, device=torch.device("cuda", 0))
        plan = syn.stage(mels, noises)
        for _ in range(2):
            syn.run_staged(plan, gather="rank0")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            syn.run_staged(plan, gather="rank0")
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        padded = sum(int(bb[1].shape[0]) * int(bb[1].shape[1]) for bb in plan["batches"])
        print(f"max_batch {mb:3d}: {len(plan['batches']):2d} micro-batches, padding {padded / sum(lengths):.3f}, {ms:.1f} ms per step = {sum(lengths) / 80 / ms * 1e3:.0f} x real time", flush=True)
        del plan, syn
        torch.cuda.empty_cache()
