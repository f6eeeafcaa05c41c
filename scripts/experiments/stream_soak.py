#!/usr/bin/env python3
"""One-off soak (GPU box): long-running streams (ring wrap-arounds, input-row drops, thousands of graph replays) against the
offline synthesis of the same utterances in the streams' convolution form: bit equality is the criterion."""
import os
os.environ["MBX_EXPERIMENT"] = "1"      # opt in to the MBX_* experiment variables (engine.experiment_overrides)
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from helpers import build_case, synthetic_inputs  # noqa: E402
from mbexwn_vocoder_amd.engine import MBExWNEngine  # noqa: E402
from mbexwn_vocoder_amd.streaming import StreamingSynthesizer  # noqa: E402

n_streams, frames, chunk = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
over = {} if os.environ.get("SOAK_CANON", "0") == "1" else {"mbexwn_config:pp_mod_subnet:n_channels": 32,
                                                             "mbexwn_config:pp_mod_subnet:n_layers": 3}
cfg, raw, wt = build_case("SPEECH", over)
os.environ["MBX_WINOGRAD"] = "2"
off = MBExWNEngine(cfg, raw, wt)
del os.environ["MBX_WINOGRAD"]
eng = MBExWNEngine(cfg, raw, wt)
syn = StreamingSynthesizer(eng, chunk_frames=chunk)
rng = np.random.default_rng(0)
data, offline, got, pos = {}, {}, {}, {}
for sid in range(n_streams):
    ll = frames - int(rng.integers(0, frames // 10))
    mel, noise = synthetic_inputs(900 + sid, 1, ll)
    offline[sid] = off.forward(torch.as_tensor(mel).cuda(), noise=torch.as_tensor(noise).cuda()).cpu().numpy()[0]
    data[sid], got[sid], pos[sid] = (mel[0], noise[0]), [], 0
    syn.open(sid)
t0 = time.time()
ticks = 0
while not all(syn.finished(sid) for sid in data):
    for sid, (mel, noise) in data.items():                  # frames arrive in packets of irregular size
        if pos[sid] < mel.shape[0]:
            end = min(pos[sid] + int(rng.integers(chunk - 3, chunk + 4)), mel.shape[0])
            syn.push(sid, mel[pos[sid]:end], noise[pos[sid] * 20:end * 20], last=end == mel.shape[0])
            pos[sid] = end
    for sid, audio in syn.tick().items():
        got[sid].append(np.array(audio))
    ticks += 1
    if ticks > 20 * frames:
        raise SystemExit("streams do not finish")
print("streams", n_streams, "frames", frames, "ticks", ticks, "graph ticks", syn.graph_ticks, "wall s", round(time.time() - t0, 1), flush=True)
for sid in data:
    out = np.concatenate(got[sid])
    assert out.shape == offline[sid].shape, (sid, out.shape, offline[sid].shape)
    if not np.array_equal(out, offline[sid]):
        bad = np.nonzero(out != offline[sid])[0]
        raise SystemExit(f"stream {sid}: {bad.size} samples differ, first at {bad[0]} (frame {bad[0] // 300}), max {np.abs(out - offline[sid]).max()}")
print("OK: bit equal")
