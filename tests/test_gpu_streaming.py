"""Streaming (BASELINE config 5 in miniature): chunked synthesis with carried phase state equals the offline run."""
import numpy as np
import pytest

from helpers import build_case, synthetic_inputs

pytestmark = pytest.mark.gpu

SMALL = {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 5}


@pytest.fixture(scope="module")
def engine():
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    cfg, raw, wt = build_case("SPEECH", SMALL)
    return MBExWNEngine(cfg, raw, wt)


def test_margins(engine):
    from mbexwn_vocoder_amd.streaming import stream_margins
    left, right, lead = stream_margins(engine.dims, engine.config)[:3]
    assert (left, right, lead) == (10, 11, 4)


def test_phase_state_carry_is_bit_exact(engine):
    """mbx_forward_stream on a window starting mid-utterance reproduces the offline phase bit for bit."""
    import torch
    from mbexwn_vocoder_amd.streaming import pack_state
    mel, noise = synthetic_inputs(21, 1, 60)
    full = engine.forward(torch.as_tensor(mel).cuda(), noise=torch.as_tensor(noise).cuda())
    pulse_full = engine.stage("pulse").cpu().numpy()[0]
    # state at frame 13 (sample 1300: inside the second 1000-sample chunk), taken from an offline-style run of a prefix
    st0 = torch.as_tensor(pack_state(0.0, 0.0, 0, 0, 13 * 100)[None]).cuda()
    _, st13 = engine.forward(torch.as_tensor(mel[:, :30]).cuda(), noise=torch.as_tensor(noise[:, :600]).cuda(),
                             stream_state=st0)
    st13 = st13.cpu().numpy()[0]
    assert st13[2] == 300          # position inside the chunk
    # window [9, 45): F0 is reproducible from frame 13 on; start the accumulator there
    ws, we = 9, 45
    state = st13.copy()
    state[3] = (13 - ws) * 100
    state[4] = -1
    win_audio, _ = engine.forward(torch.as_tensor(mel[:, ws:we]).cuda(), noise=torch.as_tensor(noise[:, ws * 20:we * 20]).cuda(),
                                  stream_state=torch.as_tensor(state[None]).cuda())
    pulse_win = engine.stage("pulse").cpu().numpy()[0]
    lo, hi = (13 - ws) * 100, (we - ws - 5) * 100          # F0 of the last frames of the window feels the right edge
    assert np.array_equal(pulse_win[lo:hi], pulse_full[13 * 100: 13 * 100 + hi - lo])
    assert np.all(pulse_win[:lo] == 0.0)


def test_layer_state_geometry_and_argument_checks(engine):
    """mbx_layer_state_info: per layer l >= 1 a slot keeps 2 d_l rows of the hidden state and d_l rows of the 30-wide
    output accumulator; the staircase ends 40 rows (2 frames) in front of the region end.  Malformed layer options are
    refused with the argument error of the C ABI."""
    import torch
    from mbexwn_vocoder_amd.streaming import pack_state
    floats, reach, min_rows = engine.layer_state_info()
    assert (reach, min_rows) == (40, 32)
    assert floats == sum(2 * d * 32 + d * 30 for d in (2, 4, 8, 16))
    mel, noise = synthetic_inputs(3, 1, 40)
    mel_d, noise_d = torch.as_tensor(mel).cuda(), torch.as_tensor(noise).cuda()
    st = torch.as_tensor(pack_state(0.0, 0.0, 0, 0, -1)[None]).cuda()
    act = torch.full((1,), 24, dtype=torch.int32, device="cuda")
    store = torch.zeros((2, floats), dtype=torch.float32, device="cuda")
    desc = torch.tensor([[0, -1, 30 * 20]], dtype=torch.int32, device="cuda")
    # a whole-region run that stores the state: fine
    engine.forward(mel_d, noise=noise_d, stream_state=st, active=(8, act, 24), layers=(store, desc, 0))
    assert float(store[0].abs().sum()) > 0 and float(store[1].abs().sum()) == 0
    with pytest.raises(ValueError):          # wrong slot size
        engine.forward(mel_d, noise=noise_d, stream_state=st, active=(8, act, 24), layers=(store[:, :-1].contiguous(), desc, 0))
    with pytest.raises(ValueError):          # a steady tick needs the WaveNet region and >= min_rows new rows
        engine.forward(mel_d, noise=noise_d, stream_state=st, active=(8, act, 24), layers=(store, desc, 160))
    wn = torch.full((1,), 10, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        engine.forward(mel_d, noise=noise_d, stream_state=st, active=(8, act, 24), wavenet=(20, wn, 10), layers=(store, desc, 20))
    with pytest.raises(ValueError):          # regions without the active one
        engine.forward(mel_d, noise=noise_d, stream_state=st, layers=(store, desc, 0))


@pytest.fixture(scope="module")
def offline_f23_engine():
    """Offline engine pinned to the convolution form the streams use (Winograd F(2,3): mbx_config.wn_conv_form): bit-equality
    needs the same arithmetic on both sides, the default offline form is calibrated and picked by size."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    cfg, raw, wt = build_case("SPEECH", SMALL)
    return MBExWNEngine(cfg, raw, wt, conv_form="f23")


@pytest.mark.parametrize("chunk", [8, 5, 2, (6, 6, 7, 6, 7)], ids=["8", "5", "2", "80ms_schedule"])
def test_streaming_equals_offline(engine, offline_f23_engine, chunk):
    import torch
    from mbexwn_vocoder_amd.streaming import StreamingSynthesizer
    lengths = [97, 40, 8, 23]
    syn = StreamingSynthesizer(engine, chunk_frames=chunk)
    assert abs(syn.lookahead_ms - 137.5) < 1e-9
    offline, pending = {}, {}
    for sid, ll in enumerate(lengths):
        mel, noise = synthetic_inputs(100 + sid, 1, ll)
        mel_d, noise_d = torch.as_tensor(mel).cuda(), torch.as_tensor(noise).cuda()
        offline[sid] = offline_f23_engine.forward(mel_d, noise=noise_d).cpu().numpy()[0]
        # the default offline form (picked by launch size) is the same function up to float32 rounding
        default = engine.forward(mel_d, noise=noise_d).cpu().numpy()[0]
        assert np.max(np.abs(default - offline[sid])) <= 2e-5 * max(1.0, np.abs(default).max())
        pending[sid] = (mel[0], noise[0], 0)
        syn.open(sid)
    got = {sid: [] for sid in offline}
    rng = np.random.default_rng(0)
    kinds = set()
    for _ in range(400):
        for sid, (mel, noise, pos) in pending.items():          # frames arrive in irregular packets
            if pos < mel.shape[0]:
                nn = int(rng.integers(1, 13))
                end = min(pos + nn, mel.shape[0])
                syn.push(sid, mel[pos:end], noise[pos * 20:end * 20], last=end == mel.shape[0])
                pending[sid] = (mel, noise, end)
        out = syn.tick()
        if out:
            kinds.add(syn.last_tick_layer_rows > 0)
        for sid, audio in out.items():
            got[sid].append(audio)
        if all(syn.finished(sid) for sid in offline):
            break
    assert kinds == {False, True}          # ticks with the per-layer WaveNet state carried, and whole-region ticks
    for sid in offline:
        stream_audio = np.concatenate(got[sid])
        assert stream_audio.shape == offline[sid].shape
        assert np.array_equal(stream_audio, offline[sid]), f"stream {sid} differs from the offline synthesis"


def test_streaming_without_layer_state(monkeypatch):
    """A handle that cannot carry the per-layer state (direct form of the dilated convolution: conv_form="direct") still
    streams -- every tick runs the WaveNet on its whole region -- and is bit-equal to its own offline synthesis."""
    import torch
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    from mbexwn_vocoder_amd.streaming import StreamingSynthesizer
    cfg, raw, wt = build_case("SPEECH", SMALL)
    eng = MBExWNEngine(cfg, raw, wt, conv_form="direct")
    assert eng.layer_state_info()[0] == 0
    syn = StreamingSynthesizer(eng, chunk_frames=8)
    assert not syn.layer_carry
    mel, noise = synthetic_inputs(77, 1, 61)
    offline = eng.forward(torch.as_tensor(mel).cuda(), noise=torch.as_tensor(noise).cuda()).cpu().numpy()[0]
    syn.open(0)
    syn.push(0, mel[0], noise[0], last=True)
    got = []
    for _ in range(50):
        out = syn.tick()
        assert syn.last_tick_layer_rows == 0
        if 0 in out:
            got.append(out[0])
        if syn.finished(0):
            break
    assert np.array_equal(np.concatenate(got), offline)


@pytest.mark.parametrize("extra", [{"mbexwn_config:pp_mod_subnet:pre_cond_layer_channels": [48, 40]},
                                   {"mbexwn_config:pp_mod_subnet:disable_conditioning": True},
                                   {"mbexwn_config:spect_filters_preserve_energy": True,
                                    "mbexwn_config:pp_mod_subnet:activation": "glu"},
                                   {"mbexwn_config:wavetable_config:add_subharm_chans": 1},
                                   {"mbexwn_config:wavetable_config:use_sinusoid_as_fun": True},
                                   {"mbexwn_config:ps_off": True},
                                   {"mbexwn_config:pp_mod_subnet_use_pqmf": False}])
def test_streaming_with_the_second_batch_of_options(monkeypatch, extra):
    """Streams of models with pre-conditioning layers (the mel-rate front end reaches two more frames per layer: margins
    and the carried front end follow streaming.frontend_reach), without conditioning, and with energy preserving
    filters + the glu gate: bit-equal to the offline synthesis in the streams' convolution form, steady ticks included."""
    import torch
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    from mbexwn_vocoder_amd.streaming import StreamingSynthesizer, frontend_reach
    cfg, raw, wt = build_case("SPEECH", dict(SMALL, **extra))
    eng = MBExWNEngine(cfg, raw, wt, conv_form="f23")
    if "mbexwn_config:pp_mod_subnet:pre_cond_layer_channels" in extra:
        base_cfg, _, _ = build_case("SPEECH", SMALL)
        from mbexwn_vocoder_amd.config import ModelDims
        assert frontend_reach(ModelDims(cfg), cfg)[0] >= 3 > (ModelDims(base_cfg).cond_kernel_size - 1) // 2
    syn = StreamingSynthesizer(eng, chunk_frames=8)
    offline, got = {}, {}
    for sid, ll in enumerate([97, 33]):
        mel, noise = synthetic_inputs(300 + sid, 1, ll)
        offline[sid] = eng.forward(torch.as_tensor(mel).cuda(), noise=torch.as_tensor(noise).cuda()).cpu().numpy()[0]
        got[sid] = []
        syn.open(sid)
        syn.push(sid, mel[0], noise[0], last=True)
    steady = 0
    for _ in range(60):
        out = syn.tick()
        steady += syn.last_tick_layer_rows > 0
        for sid, audio in out.items():
            got[sid].append(np.array(audio))
        if all(syn.finished(sid) for sid in offline):
            break
    # (10 excitation channels do not fit the folded first layer: such a handle carries no per-layer state and every tick
    # runs the WaveNet on its whole region)
    assert steady >= 3 or not syn.layer_carry
    assert syn.layer_carry or "mbexwn_config:wavetable_config:add_subharm_chans" in extra
    for sid in offline:
        assert np.array_equal(np.concatenate(got[sid]), offline[sid]), f"stream {sid}"


def test_streaming_dilation_cycle_model(monkeypatch):
    """Per-layer state with a dilation cycle (7 layers: 1 2 4 1 2 4 1): the staircase of exact rows has repeated steps,
    two inner layers of dilation 1 whose reach is rounded up to 2 rows; 15 rows of reach + 9 rows of conditioning clamp = 2 frames.  Steady ticks must occur and the stream must be
    bit-equal to the offline F(2,3) synthesis."""
    import torch
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    from mbexwn_vocoder_amd.streaming import StreamingSynthesizer
    over = {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 7,
            "mbexwn_config:pp_mod_subnet:max_log2_dilation_rate": 3}
    cfg, raw, wt = build_case("SPEECH", over)
    eng = MBExWNEngine(cfg, raw, wt, conv_form="f23")
    floats, reach, min_rows = eng.layer_state_info()
    dil = [2, 4, 1, 2, 4, 1]                                   # layers 1..6
    assert floats == sum((d + d + d % 2) * 32 + (d + d % 2) * 30 for d in dil) and reach == 40 and min_rows == 8
    syn = StreamingSynthesizer(eng, chunk_frames=8)
    assert syn.layer_carry
    lengths = [70, 52]
    offline, got = {}, {0: [], 1: []}
    for sid, ll in enumerate(lengths):
        mel, noise = synthetic_inputs(300 + sid, 1, ll)
        offline[sid] = eng.forward(torch.as_tensor(mel).cuda(), noise=torch.as_tensor(noise).cuda()).cpu().numpy()[0]
        syn.open(sid)
        syn.push(sid, mel[0], noise[0], last=True)
    steady = 0
    for _ in range(60):
        out = syn.tick()
        steady += syn.last_tick_layer_rows > 0
        for sid, audio in out.items():
            got[sid].append(audio)
        if all(syn.finished(sid) for sid in offline):
            break
    assert steady >= 3
    for sid in offline:
        assert np.array_equal(np.concatenate(got[sid]), offline[sid])


def test_steady_ticks_replay_a_graph_and_stay_bit_equal(engine, offline_f23_engine):
    """Steady ticks are served by ONE replayed hipGraph (device-resident windows advanced by mbx_window_advance, constant
    integer arguments): the streams stay bit-equal to the offline synthesis and to the launch-by-launch driver, graph ticks
    take over from the second steady tick on, and a change of the stream set (one stream ends earlier, one joins later)
    drops back to the launch-by-launch path and re-captures."""
    import torch
    from mbexwn_vocoder_amd.streaming import StreamingSynthesizer
    lengths = {0: 140, 1: 140, 2: 93, 3: 140}
    data, offline = {}, {}
    for sid, ll in lengths.items():
        mel, noise = synthetic_inputs(500 + sid, 1, ll)
        data[sid] = (mel[0], noise[0])
        offline[sid] = offline_f23_engine.forward(torch.as_tensor(mel).cuda(), noise=torch.as_tensor(noise).cuda()).cpu().numpy()[0]

    def run(use_graph):
        syn = StreamingSynthesizer(engine, chunk_frames=8)
        syn.use_graph = use_graph
        got = {sid: [] for sid in lengths}
        pos = {sid: 0 for sid in lengths}
        opened = set()
        replayed = []
        for tick in range(60):
            for sid, ll in lengths.items():
                if sid == 3 and tick < 4:
                    continue                                   # stream 3 joins four ticks late
                if sid not in opened:
                    syn.open(sid)
                    opened.add(sid)
                if pos[sid] < ll:
                    nn = min(8, ll - pos[sid])
                    mel, noise = data[sid]
                    syn.push(sid, mel[pos[sid]:pos[sid] + nn], noise[pos[sid] * 20:(pos[sid] + nn) * 20], last=pos[sid] + nn >= ll)
                    pos[sid] += nn
            out = syn.tick()
            replayed.append(syn.last_tick_replayed if out else None)
            for sid, audio in out.items():
                got[sid].append(np.array(audio))
            if len(opened) == len(lengths) and all(syn.finished(sid) for sid in lengths):
                break
        return {sid: np.concatenate(vv) for sid, vv in got.items()}, syn.graph_ticks, replayed

    plain, n_plain, _ = run(False)
    graphed, n_graph, replayed = run(True)
    assert n_plain == 0 and n_graph >= 6
    # replay stops when the stream set changes and resumes afterwards
    flips = sum(1 for aa, bb in zip(replayed, replayed[1:]) if aa is True and bb is False)
    assert flips >= 1 and replayed.count(True) == n_graph
    for sid in lengths:
        assert np.array_equal(graphed[sid], plain[sid]), f"stream {sid}: graph replay differs from the launch-by-launch ticks"
        assert np.array_equal(graphed[sid], offline[sid]), f"stream {sid} differs from the offline synthesis"


@pytest.mark.parametrize("layers", [8, 12])
def test_streaming_a_model_with_dilations_above_16(layers):
    """The reference's default depth (12 layers, d <= 2048; and 8 layers, d <= 128): no per-layer state (the F(2,3) stream kernel
    stops at d = 16), so every tick runs its whole window -- look-ahead 2.7 s at 12 layers, what the receptive field is -- and
    the layers above d = 16 run the direct form; the streams stay bit-equal to the offline synthesis in the stream form."""
    import torch
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    from mbexwn_vocoder_amd.streaming import StreamingSynthesizer
    cfg, raw, wt = build_case("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": layers})
    eng = MBExWNEngine(cfg, raw, wt)
    assert eng.layer_state_info()[0] == 0
    syn = StreamingSynthesizer(eng, chunk_frames=8)
    assert not syn.layer_carry
    frames = 300 if layers == 12 else 120
    mel, noise = synthetic_inputs(5, 2, frames)
    got = {0: [], 1: []}
    for sid in (0, 1):
        syn.open(sid)
        syn.push(sid, mel[sid], noise[sid], last=True)
    for _ in range(200):
        out = syn.tick()
        for sid, audio in out.items():
            got[sid].append(audio)
        if all(syn.finished(sid) for sid in (0, 1)):
            break
    offline = MBExWNEngine(cfg, raw, wt, conv_form=eng.conv_form_info()["stream_form"]).forward(
        torch.as_tensor(mel).cuda(), noise=torch.as_tensor(noise).cuda()).cpu().numpy()
    for sid in (0, 1):
        assert np.array_equal(np.concatenate(got[sid]), offline[sid])


def test_window_advance(engine):
    """mbx_window_advance: in-place shift of the device-resident windows + append of the new frames."""
    import torch
    rng = np.random.default_rng(0)
    B, T, step = 3, 35, 8
    mel = rng.normal(size=(B, T, 80)).astype(np.float32)
    noise = rng.normal(size=(B, T * 20)).astype(np.float32)
    mel_new = rng.normal(size=(B, step, 80)).astype(np.float32)
    noise_new = rng.normal(size=(B, step * 20)).astype(np.float32)
    mel_d, noise_d = torch.as_tensor(mel).cuda(), torch.as_tensor(noise).cuda()
    engine.window_advance(mel_d, torch.as_tensor(mel_new).cuda(), noise_d, torch.as_tensor(noise_new).cuda())
    assert np.array_equal(mel_d.cpu().numpy(), np.concatenate((mel[:, step:], mel_new), axis=1))
    assert np.array_equal(noise_d.cpu().numpy(), np.concatenate((noise[:, step * 20:], noise_new), axis=1))
    engine.window_advance(mel_d, torch.as_tensor(mel_new).cuda())          # mel only
    assert np.array_equal(mel_d.cpu().numpy()[:, -step:], mel_new)
    with pytest.raises(ValueError):
        engine.window_advance(mel_d, torch.as_tensor(mel_new[:, :, :40]).cuda())
    # mbx_window_update: frames [shift, shift + keep) to the front, the new frames behind them, the rest of the window untouched
    mel2, noise2 = rng.normal(size=(B, T, 80)).astype(np.float32), rng.normal(size=(B, T * 20)).astype(np.float32)
    mel_d, noise_d = torch.as_tensor(mel2).cuda(), torch.as_tensor(noise2).cuda()
    shift, keep, new = 5, 20, 7
    engine.window_update(mel_d, torch.as_tensor(mel_new[:, :new]).cuda(), noise_d, torch.as_tensor(noise_new[:, :new * 20]).cuda(), shift, keep)
    want = mel2.copy()
    want[:, :keep] = mel2[:, shift:shift + keep]
    want[:, keep:keep + new] = mel_new[:, :new]
    assert np.array_equal(mel_d.cpu().numpy(), want)
    wantn = noise2.copy()
    wantn[:, :keep * 20] = noise2[:, shift * 20:(shift + keep) * 20]
    wantn[:, keep * 20:(keep + new) * 20] = noise_new[:, :new * 20]
    assert np.array_equal(noise_d.cpu().numpy(), wantn)
    # mbx_emit_rows: a strided device-to-host copy of a slice of every row
    host = torch.empty((B, 100), dtype=torch.float32).pin_memory()
    engine.emit_rows(noise_d, 37, 100, host)
    torch.cuda.synchronize()
    assert np.array_equal(host.numpy(), wantn[:, 37:137])


def test_frontend_ring_equals_whole_window_and_checks_its_arguments(engine):
    """mbx_forward_options.fe_store: a call that computes only the last fe_new + fe_margin frames of the front end and
    takes the frames in front of them from the ring must give the conditioning rows / cepstrum / F0 contour (and the
    audio) of the call that computed the whole window; malformed options are refused."""
    import torch
    from mbexwn_vocoder_amd.streaming import pack_state
    assert engine.frontend_carry_supported
    B, T, ring_frames = 2, 40, 64
    mel, noise = synthetic_inputs(61, B, T + 8)
    # the phase starts at frame 4 of the window: the F0 contour in front of it feels the window's left edge in a
    # whole-window call and does not in a carried one
    st = torch.as_tensor(np.stack([pack_state(0.0, 0.0, 0, 4 * 100, -1)] * B)).cuda()
    act = torch.full((B,), T - 12, dtype=torch.int32, device="cuda")
    store = torch.zeros((4, 8, 15), dtype=torch.float32, device="cuda")
    desc = torch.tensor([[2, 0, 0, 0, 0], [1, 0, 0, 0, 0]], dtype=torch.int32, device="cuda")        # slots 2 and 1
    ring = torch.zeros((4, ring_frames, engine.frontend_frame_floats), dtype=torch.float32, device="cuda")

    def window(first):
        return (torch.as_tensor(mel[:, first:first + T]).cuda(), torch.as_tensor(noise[:, first * 20:(first + T) * 20]).cuda())

    def run(first, frontend):
        mel_d, noise_d = window(first)
        audio, _ = engine.forward(mel_d, noise=noise_d, stream_state=st, active=(8, act, T - 12), carry=(store, desc),
                                  frontend=frontend)
        return audio.cpu().numpy(), {kk: engine.stage(kk).cpu().numpy() for kk in ("cond", "cepstrum", "f0")}

    pos0 = torch.tensor([10, 50], dtype=torch.int32, device="cuda")          # ring frames of window frame 0 (one wraps)
    run(0, (ring, pos0, 0, 0))                                               # window [0, 40): every frame goes to the ring
    want_audio, want = run(8, None)                                          # window [8, 48) computed as a whole
    pos8 = torch.tensor([18, 58], dtype=torch.int32, device="cuda")
    got_audio, got = run(8, (ring, pos8, 8 + 5, 3))                          # the same window: 16 frames computed, 27 carried
    for kk in want:
        per = want[kk].shape[1] // T
        # frames 4 .. T - 5: exact in both calls (the first frames of a window feel its left edge, the last ones its right edge)
        lo, hi = 4 * per, (T - 5) * per
        assert np.array_equal(got[kk][:, lo:hi], want[kk][:, lo:hi]), kk
    assert np.array_equal(got_audio[:, 12 * 300:(T - 12) * 300], want_audio[:, 12 * 300:(T - 12) * 300])
    mel_d, noise_d = window(8)
    with pytest.raises(ValueError):                                          # ring shorter than the window
        engine.forward(mel_d, noise=noise_d, stream_state=st, active=(8, act, T - 12), carry=(store, desc),
                       frontend=(ring[:, :32].contiguous(), pos8, 13, 3))
    with pytest.raises(ValueError):                                          # more new + margin frames than the window has
        engine.forward(mel_d, noise=noise_d, stream_state=st, active=(8, act, T - 12), carry=(store, desc),
                       frontend=(ring, pos8, 38, 3))
    with pytest.raises(ValueError):                                          # the slots come from the carry descriptors
        engine.forward(mel_d, noise=noise_d, stream_state=st, active=(8, act, T - 12), frontend=(ring, pos8, 13, 3))

def test_random_models_and_stream_sets_stay_bit_equal():
    """Forty draws of tests/tools/stream_fuzz.py (which ran 800: random single-block models incl. deep conditioning
    and VTF chains, RMS normalisation, ps_off, no PQMF bank; 1-5 streams joining at different ticks; chunks of 2-12 frames;
    irregular packets): every stream bit equal to the offline synthesis of its utterance.  Seed 20292 + is the region where
    the first run found the margins of a two-convolution conditioning chain one frame short."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "stream_fuzz.py"), "40", "20280"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "failures: 0" in res.stdout, res.stdout[-3000:] + res.stderr[-2000:]
    assert res.stdout.count(" OK ") == 40
