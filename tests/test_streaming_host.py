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


def test_pack_state_layout():
    st = pack_state(0.25, 3.5, 17, 400, 1200)
    assert st.dtype.name == "int32" and st.shape == (6,)
    assert list(st[2:]) == [17, 400, 1200, 0]
    assert st[:2].view("float32").tolist() == [0.25, 3.5]
