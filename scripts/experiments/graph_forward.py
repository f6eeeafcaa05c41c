#!/usr/bin/env python3
"""One-off measurement (GPU box): a whole forward of one utterance launched kernel by kernel against the same forward
replayed as one hipGraph (torch.cuda.CUDAGraph): does the graph shorten the dependent-launch latency between the ~22 short
kernels of a 3 s / 10 s utterance?   python scripts/experiments/graph_forward.py [frames ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch
cfg, raw, wt, dims, eng = bench.build_engine("SPEECH", None)
for frames in [int(a) for a in sys.argv[1:]] or [240, 800]:
    mel_h, noise_h = bench.synthetic_batch(np.random.default_rng(1), 1, frames, 20)
    mel, noise = torch.as_tensor(mel_h).cuda(), torch.as_tensor(noise_h).cuda()
    for _ in range(5):
        out = eng.forward(mel, noise=noise)
    torch.cuda.synchronize()
    def timed(fn, reps=300):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps):
            fn()
        t1.record()
        torch.cuda.synchronize()
        return t0.elapsed_time(t1) / reps
    plain = timed(lambda: eng.forward(mel, noise=noise))
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        for _ in range(3):
            eng.forward(mel, noise=noise)
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            gout = eng.forward(mel, noise=noise)
    torch.cuda.synchronize()
    replay = timed(graph.replay)
    same = bool(torch.equal(gout, out))
    print(f"{frames} frames: launched {plain:.4f} ms, graph replay {replay:.4f} ms, same bits {same}", flush=True)
