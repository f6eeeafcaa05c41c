import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
from helpers import build_case, synthetic_inputs
from mbexwn_vocoder_amd.engine import MBExWNEngine
from mbexwn_vocoder_amd.streaming import StreamingSynthesizer
for L in (8, 12):
    cfg, raw, wt = build_case("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": L})
    eng = MBExWNEngine(cfg, raw, wt)
    print("L", L, "layer_state_info", eng.layer_state_info(), eng.conv_form_info()["form"], eng.conv_form_info()["stream_form"])
    T = 300
    mel, noise = synthetic_inputs(5, 2, T)
    try:
        syn = StreamingSynthesizer(eng, chunk_frames=8)
        print("  margins: left", syn.left, "right", getattr(syn, "right", None), "lookahead ms", getattr(syn, "lookahead_ms", None))
        outs = {0: [], 1: []}
        for sid in (0, 1):
            syn.open(sid)
            syn.push(sid, mel[sid], noise[sid], last=True)
        for _ in range(200):
            res = syn.tick()
            if not res:
                break
            for sid, au in res.items():
                outs[sid].append(np.asarray(au))
        got = [np.concatenate(outs[s]) for s in (0, 1)]
        eng23 = MBExWNEngine(cfg, raw, wt, conv_form=eng.conv_form_info()["stream_form"])
        ref = eng23.forward(torch.as_tensor(mel).cuda(), noise=torch.as_tensor(noise).cuda()).cpu().numpy()
        for s in (0, 1):
            n = min(len(got[s]), ref.shape[1])
            print("  stream", s, "samples", len(got[s]), "bit-equal to offline:", np.array_equal(got[s][:n], ref[s][:n]), "max diff", float(np.abs(got[s][:n]-ref[s][:n]).max()))
    except Exception as e:
        print("  streaming raised:", type(e).__name__, str(e)[:300])
