#!/usr/bin/env python3
"""One-off fuzz (GPU box): random single-block models x random stream sets / chunk sizes / packet sizes; every stream must
be bit equal to the offline synthesis of its utterance in the streams' convolution form."""
import os, sys, traceback
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mbexwn_vocoder_amd.config import ModelDims, canonical_config
from mbexwn_vocoder_amd.engine import MBExWNEngine
from mbexwn_vocoder_amd.streaming import StreamingSynthesizer
from mbexwn_vocoder_amd.tables import WaveTables
from mbexwn_vocoder_amd.weights import synthetic_weights
M, W = "mbexwn_config:", "mbexwn_config:pp_mod_subnet:"
n_cases, seed0 = int(sys.argv[1]), int(sys.argv[2])
fails = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    over = {W + "n_channels": int(rng.choice([32, 48, 64])), W + "n_layers": int(rng.integers(1, 6)),
            W + "activation": str(rng.choice(["gtu", "gfu", "gsu", "glu"])),
            W + "cond_lin_upsampling": int(rng.choice([10, 20, 5])), W + "cond_kernel_size": int(rng.choice([1, 3, 5, 2, 4]))}   # (even sizes: one frame more reach to the right)
    if rng.random() < 0.3:
        over[W + "max_log2_dilation_rate"] = int(rng.integers(1, 4))
    if rng.random() < 0.2:
        over[W + "pre_cond_layer_channels"] = [int(rng.choice([16, 24]))]
    if rng.random() < 0.15:
        over[W + "disable_conditioning"] = True
    if rng.random() < 0.2:
        over[M + "spect_filters_preserve_energy"] = True
    if rng.random() < 0.15:
        over[M + "wavetable_config:add_subharm_chans"] = 1
    if rng.random() < 0.15:
        over[M + "ps_off"] = True
    if rng.random() < 0.15:
        over[M + "pp_mod_subnet_use_pqmf"] = False
    if rng.random() < 0.2:
        over[M + "pp_subnet"] = [[int(rng.choice([3, 5, 7])), 32]]
    if rng.random() < 0.25:
        over[M + "ps_subnet"] = [[int(rng.choice([3, 5, 7])), 48]] * int(rng.integers(1, 5))
    if rng.random() < 0.15:
        over[M + "normalize_rms_from_mell"] = True
        over[M + "normalize_rms_num_smooth_iters"] = 1
    try:
        cfg = canonical_config("SPEECH", **over)
        dims = ModelDims(cfg)
        raw = synthetic_weights(cfg, seed=int(rng.integers(1, 10 ** 6)), bias_std=0.05, alpha_jitter=0.05)
        wt = WaveTables(sample_rate=dims.pulse_rate, **cfg["mbexwn_config"]["wavetable_config"])
        off = MBExWNEngine(cfg, raw, wt, conv_form="f23")        # the form the streams run
        eng = MBExWNEngine(cfg, raw, wt)
        chunk = int(rng.integers(2, 13))
        sched = chunk
        if rng.random() < 0.35:
            # a cyclic tick schedule (BASELINE config 5's 80 ms are 6 / 6 / 7 / 6 / 7 frames); some of them periodic in the
            # window alignment, so that their phases are captured as graphs
            sched = [(6, 6, 7, 6, 7), (3, 5), (4, 4, 8), (2, 3, 3), (7, 9), (5, 6, 5)][int(rng.integers(0, 6))]
            chunk = max(sched)
        syn = StreamingSynthesizer(eng, chunk_frames=sched)
        n_streams = int(rng.integers(1, 6))
        data, offline, got, pos = {}, {}, {}, {}
        for sid in range(n_streams):
            ll = int(rng.integers(1, 150))
            mel = np.clip(np.log(np.exp(rng.normal(-5.0, 2.0, size=(1, ll, 80))) + 1e-5), -11.5, 2.0).astype(np.float32)
            noise = rng.normal(size=(1, ll * 20)).astype(np.float32)
            offline[sid] = off.forward(torch.as_tensor(mel).cuda(), noise=torch.as_tensor(noise).cuda()).cpu().numpy()[0]
            data[sid], got[sid], pos[sid] = (mel[0], noise[0]), [], 0
        opened, ticks, steady = set(), 0, 0
        while not all(sid in opened and syn.finished(sid) for sid in data):
            for sid, (mel, noise) in data.items():
                if sid not in opened:
                    if rng.random() < 0.5:           # streams join at different ticks
                        syn.open(sid)
                        opened.add(sid)
                    else:
                        continue
                if pos[sid] < mel.shape[0]:
                    end = min(pos[sid] + int(rng.integers(1, 2 * chunk + 1)), mel.shape[0])
                    syn.push(sid, mel[pos[sid]:end], noise[pos[sid] * 20:end * 20], last=end == mel.shape[0])
                    pos[sid] = end
            for sid, audio in syn.tick().items():
                got[sid].append(np.array(audio))
            steady += syn.last_tick_layer_rows > 0
            ticks += 1
            if ticks > 5000:
                raise RuntimeError("streams do not finish")
        bad = [sid for sid in data if not np.array_equal(np.concatenate(got[sid]) if got[sid] else np.zeros(0, np.float32), offline[sid])]
        fails += bool(bad)
        print(case, "FAIL" if bad else "OK  ", "streams", n_streams, "chunk", sched, "ticks", ticks, "steady", steady, "graph", syn.graph_ticks,
              {kk.split(':')[-1]: vv for kk, vv in over.items()}, "bad", bad, flush=True)
        del eng, off, syn
    except Exception:                                        # noqa: BLE001
        fails += 1
        print(case, "EXC", over, flush=True)
        traceback.print_exc()
print("failures:", fails)
sys.exit(1 if fails else 0)
