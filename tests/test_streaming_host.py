"""Host-side geometry of the streaming driver (no GPU): margins derived from the layer geometry."""
from mbexwn_vocoder_amd.config import ModelDims, canonical_config
from mbexwn_vocoder_amd.streaming import pack_state, stream_margins

P = "mbexwn_config:pp_mod_subnet:"


def margins(**over):
    cfg = canonical_config("SPEECH", **over)
    return stream_margins(ModelDims(cfg), cfg)


def test_canonical_margins():
    # F0-net 3 convs k=3 (+1 interpolation) -> pulses valid from frame 4; WaveNet 31 steps + 9 rows of conditioning
    # interpolation = 40 rows = 2 frames; PQMF 1 frame; STFT / overlap-add 3 / 4 frames
    assert margins() == (10, 11, 4, 6, 7, 2)


def test_wavenet_reach_includes_the_conditioning_interpolation():
    # dilation cycle 1 2 4 1 2 4 1: 15 steps of receptive field, but the last 9 rows of a region interpolate the
    # conditioning towards a clamped row and that spreads back through the layers: 24 rows = 2 frames, not 1
    assert margins(**{P + "n_layers": 7, P + "max_log2_dilation_rate": 3})[5] == 2
    # two layers (1, 2): 3 + 9 = 12 rows -> 1 frame
    assert margins(**{P + "n_layers": 2})[5] == 1
    # a finer split of the conditioning up-sampling shortens the clamp: 7 + 4 rows
    assert margins(**{P + "n_layers": 3, P + "cond_lin_upsampling": 5})[5] == 1
    # a coarser one lengthens it: 15 + 19 rows -> 2 frames
    assert margins(**{P + "n_layers": 4, P + "cond_lin_upsampling": 20})[5] == 2


def test_margins_follow_the_kernel_sizes():
    left, right, lead, act_l, act_r, wn = margins(**{"mbexwn_config:pp_subnet": [[7, 48]]})
    assert lead == 4 and left == max(lead + wn + 1 + 3, lead + 3 + 1)       # one conv k=7: +-3 frames, like three k=3
    left5, right5, *_ = margins(**{P + "cond_kernel_size": 5})
    assert right5 >= 3 + 2 + 1 + 4                                             # conditioning conv reaches 2 (+1) frames ahead


def test_even_conditioning_kernels_reach_further_right():
    """The library pads a conditioning convolution (k - 1) // 2 frames in front and k // 2 behind: an even kernel size
    reaches one frame more to the right, per convolution of the chain -- in the window margins and in the reach of the
    carried front end alike (one shared helper)."""
    from mbexwn_vocoder_amd.config import ModelDims, canonical_config
    from mbexwn_vocoder_amd.streaming import cond_chain_reach, frontend_reach
    for ks, pre, want in ((3, [], (1, 1)), (4, [], (1, 2)), (4, [48, 40], (3, 6)), (2, [16], (0, 2)), (5, [16], (4, 4))):
        cfg = canonical_config("SPEECH", **{P + "cond_kernel_size": ks, P + "pre_cond_layer_channels": pre,
                                            P + "n_channels": 32})
        dims = ModelDims(cfg)
        assert cond_chain_reach(dims) == want, (ks, pre)
        fe_l, fe_r = frontend_reach(dims, cfg)
        assert fe_l >= want[0] and fe_r >= want[1] + 1


def test_pack_state_layout():
    st = pack_state(0.25, 3.5, 17, 400, 1200)
    assert st.dtype.name == "int32" and st.shape == (6,)
    assert list(st[2:]) == [17, 400, 1200, 0]
    assert st[:2].view("float32").tolist() == [0.25, 3.5]


class _FakeEngine:
    """CPU test double of the engine surface the streaming driver uses (NOT the oracle, not a product path): the audio of
    frame t is sum(mel[t]) + noise[t * spf] on every sample of the frame, so the streamed output is known in closed form
    and any mistake in the window assembly / the shared input rows shows up."""

    def __init__(self):
        import torch
        cfg = canonical_config("SPEECH")
        self.config, self.dims, self.device = cfg, ModelDims(cfg), torch.device("cpu")

    def layer_state_info(self):
        return 0, 0, 0

    def conv_form_info(self):
        return {"split_f16_layers": 0, "split_f16_gate_layers": 0}

    def forward(self, mel, n_frames=None, noise=None, stream_state=None, **_):
        import torch
        hop, spf = self.dims.hop_size, self.dims.steps_per_frame
        val = mel.sum(dim=2) + noise[:, ::spf]
        return val.repeat_interleave(hop, dim=1), torch.zeros_like(stream_state)


def test_shared_input_rows_drop_and_grow():
    """push() keeps the frames of all streams in shared rows: frames in front of the next window are dropped when a row
    runs full, the rows grow when that is not enough; the windows a tick assembles are unaffected."""
    import numpy as np
    from mbexwn_vocoder_amd.streaming import StreamingSynthesizer
    syn = StreamingSynthesizer(_FakeEngine(), chunk_frames=8)
    syn.use_graph = False
    assert syn._in_cap == 256
    rng = np.random.default_rng(0)
    lengths = {"a": 700, "b": 45, "c": 300}
    data = {sid: (rng.normal(size=(ll, 80)).astype(np.float32), rng.normal(size=(ll * 20,)).astype(np.float32))
            for sid, ll in lengths.items()}
    got = {sid: [] for sid in lengths}
    pos = {sid: 0 for sid in lengths}
    syn.open("a")
    syn.open("b")
    syn.push("a", data["a"][0][:300], data["a"][1][:6000])        # more than a row holds: the rows grow
    pos["a"] = 300
    assert syn._in_cap >= 300
    for tick in range(400):
        if tick == 3:
            syn.open("c")
        for sid in list(syn.streams):
            ll = lengths[sid]
            if pos[sid] < ll:
                nn = min(int(rng.integers(1, 13)), ll - pos[sid])      # slower than the ticks consume
                mel, noise = data[sid]
                syn.push(sid, mel[pos[sid]:pos[sid] + nn], noise[pos[sid] * 20:(pos[sid] + nn) * 20], last=pos[sid] + nn >= ll)
                pos[sid] += nn
        for sid, audio in syn.tick().items():
            got[sid].append(audio)
        if len(syn.streams) == 3 and all(syn.finished(sid) for sid in lengths):
            break
    assert syn._in_cap == 512                            # stream "a" never held its 700 frames at once: old frames were dropped
    assert syn.streams["a"].base > 0
    for sid, ll in lengths.items():
        mel, noise = data[sid]
        want = np.repeat(mel.sum(axis=1) + noise[::20], 300)
        out = np.concatenate(got[sid])
        assert out.shape == (ll * 300,)
        np.testing.assert_allclose(out, want, rtol=0, atol=2e-5)      # float32 sums of the test double
