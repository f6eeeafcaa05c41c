"""The BASELINE configurations at their real geometry, through the C ABI (task 4):

  config 4  MW-VO-FD (C = 340: 340 % 32 = 20 -> partial column tile, 340 % 16 = 4 -> the masked LDS-DMA staging paths):
            ragged batch of 16, lengths U[160, 1200] frames, every convolution form
  config 5  MW-SP-FD (C = 320) streaming: 64 concurrent streams, 8-frame ticks, bit-equal to offline synthesis
  A2        sub-net grammar variants on the device: sub-pixel conv (Keras SAME zero padding), "L<up>", bare ["L", up]
  A14       RMS normalisation on the device against the reference-generated goldens

Tolerances as in tests/test_gpu_parity.py: end to end 1e-4 * max(1, max|ref|); bit exact where the arithmetic is the same.
"""
import os

import numpy as np
import pytest

from oracle import mbexwn_oracle as orc
from helpers import form_kwargs, GOLDEN_CASES, build_case, synthetic_inputs

pytestmark = pytest.mark.gpu

E2E_TOL = 1e-4


def _tol(ref, rel=E2E_TOL):
    return rel * max(1.0, float(np.max(np.abs(ref))))


def _maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


@pytest.fixture(scope="module")
def torch():
    import torch as _torch
    assert _torch.cuda.is_available(), "GPU tests need an MI355X"
    return _torch


def dev(torch, arr, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(arr), dtype=dtype or torch.float32).cuda()


# ------------------------------------------------------------------------------------------------
# config 4: VOICE geometry, ragged batch of 16
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def voice_case():
    cfg, raw, wt = build_case("VOICE", {})
    rng = np.random.default_rng(44)
    lengths = [int(vv) for vv in rng.integers(160, 1201, size=16)]
    lengths[3], lengths[11] = 160, 163                       # two short items for the oracle
    lengths[7] = 1200                                        # and the longest possible one
    mel, noise = synthetic_inputs(404, 16, max(lengths))
    return cfg, raw, wt, lengths, mel, noise


@pytest.mark.parametrize("form", ["default", "0", "2", "44"])
def test_voice_ragged_batch_of_16(torch, monkeypatch, voice_case, form):
    """Every item of the ragged batch equals its one-at-a-time run: bit for bit when the convolution form is pinned
    (conv_form direct / f23, or f43 with batch_invariant: "0", "2", "44"); with the default policy the batch
    runs the large-launch F(4,3) kernel and a single item the channel-split one (two K halves summed), which agree to
    float32 rounding: 4e-5 relative to the peak (measured 2.1e-5; each is within 1e-4 of the float64 oracle).
    The two shortest items are held to the float64 oracle, the longest to the prefix property."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    cfg, raw, wt, lengths, mel, noise = voice_case
    eng = MBExWNEngine(cfg, raw, wt, **form_kwargs(form))
    assert eng.dims.wn_channels == 340
    nf = torch.as_tensor(lengths, dtype=torch.int32).cuda()
    batch = eng.forward(dev(torch, mel), n_frames=nf, noise=dev(torch, noise)).cpu().numpy()
    assert np.all(np.isfinite(batch))
    scale = max(1.0, float(np.abs(batch).max()))
    for ii, ll in enumerate(lengths):
        single = eng.forward(dev(torch, mel[ii:ii + 1, :ll]), noise=dev(torch, noise[ii:ii + 1, :ll * 20])).cpu().numpy()[0]
        if form == "default":
            assert _maxdiff(batch[ii, :ll * 300], single) <= 4e-5 * scale, f"item {ii}"
        else:
            assert np.array_equal(batch[ii, :ll * 300], single), f"item {ii} (form {form}) differs from its single run"
        assert np.all(batch[ii, ll * 300:] == 0.0)
    om = orc.OracleModel(cfg, raw, wt)
    for ii in (3, 11):
        ll = lengths[ii]
        ref = om.forward(mel[ii:ii + 1, :ll], noise[ii:ii + 1, :ll * 20])[0]
        assert _maxdiff(batch[ii, :ll * 300], ref) <= _tol(ref), f"item {ii} vs oracle"
    # prefix property on the longest item: the first 300 frames do not depend on what follows (finite receptive
    # field + causal phase), up to the rounding of the launch-size dependent kernel choice
    cut, margin = 300, 12
    part = eng.forward(dev(torch, mel[7:8, :cut]), noise=dev(torch, noise[7:8, :cut * 20])).cpu().numpy()[0]
    keep = (cut - margin) * 300
    assert _maxdiff(batch[7, :keep], part[:keep]) <= 4e-5 * scale
    if form == "default":
        # the longest item (1200 frames = 15 s, inside the 16-batch, large-launch kernels) against
        # the float64 oracle over its WHOLE length -- with the contour exact (F0-net in float64) a 2e-5 check
        ref = om.forward(mel[7:8, :1200], noise[7:8, :1200 * 20])[0]
        assert _maxdiff(batch[7, :1200 * 300], ref) <= _tol(ref, 2e-5), "1200-frame item of the batch vs oracle, whole length"


@pytest.fixture(scope="module")
def canon_case():
    cfg, raw, wt = build_case("SING", {})
    rng = np.random.default_rng(33)
    lengths = [int(vv) for vv in rng.integers(400, 801, size=16)]
    lengths[2], lengths[13] = 160, 163                       # two short items for the oracle
    lengths[5] = 800                                         # the padded length: 16 x 800 frames = the config-3 launch sizes
    mel, noise = synthetic_inputs(303, 16, max(lengths))
    return cfg, raw, wt, lengths, mel, noise


def test_canon_ragged_batch_of_16(torch, monkeypatch, canon_case):
    """the DEFAULT policy at config-3 size (C = 320, 16 x 800 frames padded) -- the kernels of the
    driver's bench line: wn_gate_winograd4w_kernel (256-row blocks), wn_resskip_wide_kernel<11>, conv1d_mel_group_kernel<2> -- held
    to the float64 oracle on the two short items, to the prefix property on the longest, and every item to its
    one-at-a-time run (which takes the small-launch kernels: equal to float32 rounding)."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    cfg, raw, wt, lengths, mel, noise = canon_case
    eng = MBExWNEngine(cfg, raw, wt)
    info = eng.conv_form_info()
    assert info["requested"] == "auto" and info["calibrated"] == 1 and info["form"] == "f43"      # the default earned F(4,3)
    assert eng.dims.wn_channels == 320 and eng.gate_form(16, 800) == "winograd_f43" and eng.folds_start
    nf = torch.as_tensor(lengths, dtype=torch.int32).cuda()
    out = torch.full((16, 800 * 300), float("nan"), dtype=torch.float32).cuda()
    batch = eng.forward(dev(torch, mel), n_frames=nf, noise=dev(torch, noise), out=out).cpu().numpy()
    assert np.all(np.isfinite(batch))
    scale = max(1.0, float(np.abs(batch).max()))
    om = orc.OracleModel(cfg, raw, wt)
    for ii in (2, 13):
        ll = lengths[ii]
        ref = om.forward(mel[ii:ii + 1, :ll], noise[ii:ii + 1, :ll * 20])[0]
        assert _maxdiff(batch[ii, :ll * 300], ref) <= _tol(ref), f"item {ii} vs oracle"
    for ii, ll in enumerate(lengths):
        assert np.all(batch[ii, ll * 300:] == 0.0)
        if ii % 5 == 0 or ii in (2, 13):
            single = eng.forward(dev(torch, mel[ii:ii + 1, :ll]), noise=dev(torch, noise[ii:ii + 1, :ll * 20])).cpu().numpy()[0]
            assert _maxdiff(batch[ii, :ll * 300], single) <= 4e-5 * scale, f"item {ii}"
    # the longest item: its first 80 frames against the oracle on the 92-frame prefix (prefix property, margin 12)
    ref = om.forward(mel[5:6, :92], noise[5:6, :92 * 20])[0][:80 * 300]
    assert _maxdiff(batch[5, :80 * 300], ref) <= _tol(ref), "longest item vs oracle on its prefix"
    cut, margin = 300, 12
    part = eng.forward(dev(torch, mel[5:6, :cut]), noise=dev(torch, noise[5:6, :cut * 20])).cpu().numpy()[0]
    keep = (cut - margin) * 300
    assert _maxdiff(batch[5, :keep], part[:keep]) <= 4e-5 * scale
    # the 800-frame item of the batch against the float64 oracle over its WHOLE length (the
    # large-launch kernels of the bench line on a full-length item; 2e-5 with the exact contour)
    ref = om.forward(mel[5:6, :800], noise[5:6, :800 * 20])[0]
    assert _maxdiff(batch[5], ref) <= _tol(ref, 2e-5), "800-frame item of the batch vs oracle, whole length"


def test_voice_reference_golden(torch, golden_dir):
    """C = 340, 2 x 41 frames, against the float32 run of the reference's own MBExWN.call."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    gold = np.load(os.path.join(golden_dir, "reference_forward_f32.npz"))
    voice, overrides, batch, frames = GOLDEN_CASES["voice"]
    eng = MBExWNEngine(*build_case(voice, overrides))
    got = eng.forward(dev(torch, gold["voice/mell"]), noise=dev(torch, gold["voice/noise"])).cpu().numpy()
    assert _maxdiff(got, gold["voice/audio"]) <= _tol(gold["voice/audio"])
    assert _maxdiff(eng.stage("f0").cpu().numpy(), gold["voice/f0"]) <= 1e-3


# ------------------------------------------------------------------------------------------------
# config 5: canonical model, 64 streams
# ------------------------------------------------------------------------------------------------
def test_canonical_streaming_64_streams_80ms_schedule_bit_equal(torch):
    """BASELINE config 5 as written: 80 ms ticks.  80 ms are 6.4 mel frames, so the streams walk the cyclic schedule
    6 / 6 / 7 / 6 / 7 frames (32 frames = 400 ms per period: exactly 80 ms per tick on average).  64 concurrent streams of
    the canonical C = 320 model, packets arriving in the same rhythm; the concatenated output of every stream must be
    bit-equal to the offline synthesis with the same convolution form (F(2,3)), and the per-layer state must have been
    carried (ticks that run every WaveNet layer on the tick's new rows only)."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    from mbexwn_vocoder_amd.streaming import StreamingSynthesizer
    cfg, raw, wt = build_case("SPEECH", {})
    eng = MBExWNEngine(cfg, raw, wt, conv_form="f23")
    n_streams, schedule = 64, (6, 6, 7, 6, 7)
    rng = np.random.default_rng(6)
    lengths = [int(vv) for vv in rng.integers(110, 140, size=n_streams)]
    syn = StreamingSynthesizer(eng, chunk_frames=schedule)
    assert not syn.uniform and syn.schedule == list(schedule)
    utts = []
    for sid, ll in enumerate(lengths):
        mm, nn = synthetic_inputs(1900 + sid, 1, ll)
        utts.append((mm[0], nn[0]))
        syn.open(sid)
    got = {sid: [] for sid in range(n_streams)}
    pos = [0] * n_streams
    ticks, layered, sizes = 0, 0, set()
    while not all(syn.finished(sid) for sid in range(n_streams)):
        for sid, ll in enumerate(lengths):
            if pos[sid] < ll:
                nn = min(schedule[ticks % len(schedule)], ll - pos[sid])
                syn.push(sid, utts[sid][0][pos[sid]:pos[sid] + nn], utts[sid][1][pos[sid] * 20:(pos[sid] + nn) * 20],
                         last=pos[sid] + nn >= ll)
                pos[sid] += nn
        out = syn.tick()
        for sid, audio in out.items():
            got[sid].append(audio)
            sizes.add(audio.shape[0] // 300)
        ticks += 1
        layered += syn.last_tick_layer_rows > 0
        assert ticks < 300
    assert {6, 7} <= sizes and layered >= 3
    assert syn.graph_ticks >= 3                            # phases recorded in the first period are replayed as graphs afterwards
    for sid in range(n_streams):
        ll = lengths[sid]
        offline = eng.forward(dev(torch, utts[sid][0][None]), noise=dev(torch, utts[sid][1][None])).cpu().numpy()[0]
        stream = np.concatenate(got[sid])
        assert stream.shape == (ll * 300,)
        assert np.array_equal(stream, offline), f"stream {sid} is not bit-equal to the offline synthesis"


def test_canonical_streaming_64_streams_bit_equal(torch):
    """What `bench.py --workload config5_sp_stream64` times: 64 concurrent streams of the canonical C = 320 model,
    8-frame ticks.  The concatenated stream output must be bit-equal to the offline synthesis of the same utterance
    with the same convolution form (streams run Winograd F(2,3); conv_form="f23" pins the offline engine to it)."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    from mbexwn_vocoder_amd.streaming import StreamingSynthesizer
    cfg, raw, wt = build_case("SPEECH", {})
    eng = MBExWNEngine(cfg, raw, wt, conv_form="f23")
    n_streams, chunk = 64, 8
    rng = np.random.default_rng(5)
    lengths = [int(vv) for vv in rng.integers(50, 75, size=n_streams)]      # >= 5 ticks each, ragged ends
    syn = StreamingSynthesizer(eng, chunk_frames=chunk)
    utts = []
    for sid, ll in enumerate(lengths):
        mm, nn = synthetic_inputs(900 + sid, 1, ll)
        utts.append((mm[0], nn[0]))
        syn.open(sid)
    got = {sid: [] for sid in range(n_streams)}
    pos = [0] * n_streams
    ticks = steady = 0
    while not all(syn.finished(sid) for sid in range(n_streams)):
        for sid, ll in enumerate(lengths):                 # packets of 8 frames arrive, the last one closes the stream
            if pos[sid] < ll:
                nn = min(chunk, ll - pos[sid])
                syn.push(sid, utts[sid][0][pos[sid]:pos[sid] + nn], utts[sid][1][pos[sid] * 20:(pos[sid] + nn) * 20],
                         last=pos[sid] + nn >= ll)
                pos[sid] += nn
        for sid, audio in syn.tick().items():
            got[sid].append(audio)
        ticks += 1
        steady += syn.last_tick_layer_rows == chunk * 20
        assert ticks < 200
    assert ticks >= 5
    # both kinds of tick ran: steady ones (per-layer WaveNet state carried, every layer on the 160 new rows only) and
    # whole-region ones (first tick, ticks in which a stream ends)
    assert 2 <= steady < ticks
    assert syn.graph_ticks >= 1                            # steady ticks behind the first one replay the captured graph
    for sid in range(0, n_streams, 1):
        ll = lengths[sid]
        offline = eng.forward(dev(torch, utts[sid][0][None]), noise=dev(torch, utts[sid][1][None])).cpu().numpy()[0]
        stream = np.concatenate(got[sid])
        assert stream.shape == (ll * 300,)
        assert np.array_equal(stream, offline), f"stream {sid} is not bit-equal to the offline synthesis"


# ------------------------------------------------------------------------------------------------
# A2: sub-net grammar variants
# ------------------------------------------------------------------------------------------------
def test_subnet_grammar_variants_end_to_end(torch, golden_dir):
    """pp_subnet = [[5,32,2], [3,64,"L2"], ["L",5]] (reference custom_pulsed_generator.py:57-60, 74-108): sub-pixel
    convolution with SAME zero padding, interpolation behind a convolution, bare interpolation whose factor the
    reference forgets in total_ups -- the F0-net then runs at 5x the pulse rate and generate_f0 cuts it (:787).
    Held to the oracle, to a ragged batch, and to the float32 run of the reference's own code."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    voice, overrides, batch, frames = GOLDEN_CASES["grammar"]
    cfg, raw, wt = build_case(voice, overrides)
    eng, om = MBExWNEngine(cfg, raw, wt), orc.OracleModel(cfg, raw, wt)
    mel, noise = synthetic_inputs(12, 3, 21)
    got = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()
    ref, st = om.forward(mel, noise, return_stages=True)
    assert _maxdiff(eng.stage("f0").cpu().numpy(), st["f0"]) <= 1e-3
    assert _maxdiff(got, ref) <= _tol(ref)
    lengths = [21, 4, 13]
    nf = torch.as_tensor(lengths, dtype=torch.int32).cuda()
    rag = eng.forward(dev(torch, mel), n_frames=nf, noise=dev(torch, noise)).cpu().numpy()
    for ii, ll in enumerate(lengths):
        single = eng.forward(dev(torch, mel[ii:ii + 1, :ll]), noise=dev(torch, noise[ii:ii + 1, :ll * 20])).cpu().numpy()[0]
        assert np.array_equal(rag[ii, :ll * 300], single)
    gold = np.load(os.path.join(golden_dir, "reference_forward_f32.npz"))
    out = eng.forward(dev(torch, gold["grammar/mell"]), noise=dev(torch, gold["grammar/noise"])).cpu().numpy()
    assert _maxdiff(eng.stage("f0").cpu().numpy(), gold["grammar/f0"]) <= 1e-3
    assert _maxdiff(out, gold["grammar/audio"]) <= _tol(gold["grammar/audio"])


def test_long_canonical_reference_golden(torch, golden_dir):
    """60 frames of the canonical model against the float32 run of the reference: six 1000-sample phase chunks (bit
    exact phase from the reference's own F0), the chunk-offset chain, F0-dependent lifter rows."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    gold = np.load(os.path.join(golden_dir, "reference_forward_f32.npz"))
    voice, overrides, batch, frames = GOLDEN_CASES["canon60"]
    eng = MBExWNEngine(*build_case(voice, overrides))
    got = eng.forward(dev(torch, gold["canon60/mell"]), noise=dev(torch, gold["canon60/noise"])).cpu().numpy()
    assert _maxdiff(got, gold["canon60/audio"]) <= _tol(gold["canon60/audio"])
    assert _maxdiff(eng.stage("excitation").cpu().numpy(), gold["canon60/excitation"]) <= _tol(gold["canon60/excitation"])
    pulse, phase = eng.wavetable(dev(torch, gold["canon60/f0"]))
    assert np.array_equal(phase.cpu().numpy(), gold["canon60/phase"])
    assert _maxdiff(pulse.cpu().numpy(), gold["canon60/pulse"]) <= 2e-6


# ------------------------------------------------------------------------------------------------
# A14: RMS normalisation on the device
# ------------------------------------------------------------------------------------------------
NORM_CASES = {
    "iters1": {"normalize_rms_num_smooth_iters": 1},
    "iters2_comp": {"normalize_rms_num_smooth_iters": 2, "normalize_compressor_exp": 0.8, "max_norm_fact": 200.0},
    "scaled_win": {"normalize_rms_num_smooth_iters": 1, "normalize_smooth_win_scale": 2,
                   "normalize_smooth_with_squared_win": False, "lin_amp_scale": 1.5, "mel_amp_scale": 0.5},
    "pinv": {"normalize_rms_num_smooth_iters": 1, "normalize_use_pinv": True},      # reference wavegen_1d.py:603-608, 683-685
}


def _norm_engine(extra):
    from mbexwn_vocoder_amd.config import canonical_config
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    from mbexwn_vocoder_amd.tables import WaveTables
    from mbexwn_vocoder_amd.weights import synthetic_weights
    from mbexwn_vocoder_amd.config import ModelDims
    cfg = canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3})
    cfg["mbexwn_config"].update(normalize_rms_from_mell=True, **extra)
    raw = synthetic_weights(cfg, seed=1234, bias_std=0.05, alpha_jitter=0.05)
    wt = WaveTables(sample_rate=ModelDims(cfg).pulse_rate, **cfg["mbexwn_config"]["wavetable_config"])
    return cfg, raw, wt, MBExWNEngine(cfg, raw, wt)


@pytest.mark.parametrize("case", sorted(NORM_CASES))
def test_norm_mel_device_vs_reference_goldens(torch, golden_dir, case):
    """mbx_norm_mel (NormMelComponents.normalize_inputs_by_rms on the device) against the float32 run of the
    reference's own class (tests/golden/make_reference_normmel.py): normalised log-mel and up-sampled gain."""
    gold = np.load(os.path.join(golden_dir, "reference_normmel.npz"))
    cfg, raw, wt, eng = _norm_engine(NORM_CASES[case])
    mell = gold[f"f32/{case}/mell"]
    out, gain = eng.norm_mel_stage(dev(torch, mell))
    tol = 2e-4 if case == "pinv" else 2e-5          # pinv: a float32 contraction over 1025 bins of +-1e3 entries
    np.testing.assert_allclose(out.cpu().numpy(), gold[f"f32/{case}/mell_norm"], rtol=0, atol=tol)
    np.testing.assert_allclose(gain.cpu().numpy(), gold[f"f32/{case}/gain"], rtol=tol, atol=0)
    # ragged: the smoothing uses the item's own edges
    nf = torch.as_tensor([17, 9], dtype=torch.int32).cuda()
    out_r, gain_r = eng.norm_mel_stage(dev(torch, mell), n_frames=nf)
    out_s, gain_s = eng.norm_mel_stage(dev(torch, mell[1:2, :9]))
    assert np.array_equal(out_r.cpu().numpy()[1, :9], out_s.cpu().numpy()[0])
    assert np.array_equal(gain_r.cpu().numpy()[1, :9 * 300], gain_s.cpu().numpy()[0])
    assert np.array_equal(out_r.cpu().numpy()[0], out.cpu().numpy()[0])


def test_norm_mel_inside_forward_matches_oracle(torch):
    """A model with normalize_rms_from_mell: mbx_forward normalises the mel on the device, synthesises and multiplies
    the gain onto the audio (reference wavegen_1d.py:493-507) -- no host round trip; vs the float64 oracle."""
    cfg, raw, wt, eng = _norm_engine(NORM_CASES["iters2_comp"])
    assert eng.normalizes_rms
    om = orc.OracleModel(cfg, raw, wt)
    mel, noise = synthetic_inputs(8, 2, 19)
    got = eng.infer(mel, synth_length=19 * 300, noise=noise).numpy()
    mel_n, gain = orc.normalize_inputs_by_rms(mel, cfg, 19 * 300)
    ref = om.forward(mel_n.astype(np.float32), noise) * gain
    assert _maxdiff(eng.stage("mel_norm").cpu().numpy().reshape(2, 19, 80), mel_n) <= 2e-5
    assert _maxdiff(got, ref) <= _tol(ref)
    f0, exc, env, rms = eng.infer_components(mel, synth_length=19 * 300, noise=noise)
    np.testing.assert_allclose(rms, gain, rtol=2e-5)


def test_streaming_with_rms_normalisation(torch, monkeypatch):
    """A model with normalize_rms_from_mell streams: the smoothing of the RMS contour reaches norm_reach frames, which the
    window margins cover (streaming.py::stream_margins), so the stream is bit-equal to the offline synthesis (F(2,3) form)
    although every window normalises its own frames; steady ticks replay the captured graph."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    from mbexwn_vocoder_amd.streaming import StreamingSynthesizer, norm_reach, stream_margins
    cfg, raw, wt, _ = _norm_engine(NORM_CASES["iters2_comp"])
    eng = MBExWNEngine(cfg, raw, wt, conv_form="f23")
    assert eng.normalizes_rms and norm_reach(eng.dims, cfg) == 8
    import copy
    plain = copy.deepcopy(cfg)
    plain["mbexwn_config"]["normalize_rms_from_mell"] = False
    from mbexwn_vocoder_amd.config import ModelDims
    base = stream_margins(ModelDims(plain), plain)
    assert stream_margins(eng.dims, cfg)[:3] == (base[0] + 8, base[1] + 8, base[2] + 8)
    syn = StreamingSynthesizer(eng, chunk_frames=8)
    assert not syn.fe_carry                                  # the front-end ring is not used with the normalisation
    lengths = [150, 97]
    offline, got = {}, {0: [], 1: []}
    for sid, ll in enumerate(lengths):
        mel, noise = synthetic_inputs(700 + sid, 1, ll)
        offline[sid] = eng.infer(mel, synth_length=ll * 300, noise=noise).numpy()[0]
        syn.open(sid)
        syn.push(sid, mel[0], noise[0], last=True)
    for _ in range(60):
        for sid, audio in syn.tick().items():
            got[sid].append(np.array(audio))
        if all(syn.finished(sid) for sid in offline):
            break
    assert syn.graph_ticks >= 3
    for sid in offline:
        assert np.array_equal(np.concatenate(got[sid]), offline[sid]), f"stream {sid}"


# ------------------------------------------------------------------------------------------------
# WaveNet options: the gfu / gsu gates and use_equalized_lr
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["gfu", "gsu_eqlr", "eqlr_plain", "glu", "precond", "nocond", "energy"])
def test_gate_variants_and_equalized_lr_vs_reference_goldens(torch, golden_dir, case):
    """pp_mod_subnet.activation = gfu / gsu / glu (reference custom_AE_layers.py:156,312-318), use_equalized_lr with and
    without weight norm (conv_layers.py:133-153), pre_cond_layer_channels (:190-201,283-285), disable_conditioning
    (:203-204,293-294) and spect_filters_preserve_energy (custom_pulsed_generator.py:817-849), against the float32 run of
    the reference's own MBExWN.call and the oracle."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    gold = np.load(os.path.join(golden_dir, "reference_forward_f32.npz"))
    voice, overrides, batch, frames = GOLDEN_CASES[case]
    cfg, raw, wt = build_case(voice, overrides)
    eng = MBExWNEngine(cfg, raw, wt)
    mel, noise = gold[f"{case}/mell"], gold[f"{case}/noise"]
    got = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()
    assert _maxdiff(got, gold[f"{case}/audio"]) <= _tol(gold[f"{case}/audio"])
    assert _maxdiff(eng.stage("excitation").cpu().numpy(), gold[f"{case}/excitation"]) <= _tol(gold[f"{case}/excitation"])
    ref = orc.OracleModel(cfg, raw, wt).forward(mel, noise)
    assert _maxdiff(got, ref) <= _tol(ref)


def test_pulse_pqmf_model(torch):
    """pulse_channels_use_pqmf: the excitation rows are the PQMF analysis of the pulse signal (reference
    custom_pulsed_generator.py:892-895); ragged batch vs the oracle (pinned by the golden case "pulsepqmf"), the stage
    "pulse" stays the oscillator's output, streams are refused."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    from mbexwn_vocoder_amd.streaming import StreamingSynthesizer
    cfg, raw, wt = build_case(*GOLDEN_CASES["pulsepqmf"][:2])
    eng, om = MBExWNEngine(cfg, raw, wt), orc.OracleModel(cfg, raw, wt)
    mel, noise = synthetic_inputs(77, 2, 19)
    lengths = (19, 6)
    got = eng.forward(dev(torch, mel), noise=dev(torch, noise),
                      n_frames=torch.tensor(lengths, dtype=torch.int32, device="cuda")).cpu().numpy()
    for ii, ll in enumerate(lengths):
        ref, st = om.forward(mel[ii:ii + 1, :ll], noise[ii:ii + 1, :ll * 20], return_stages=True)
        assert _maxdiff(got[ii, :ll * 300], ref[0]) <= _tol(ref)
    pulse = eng.stage("pulse").cpu().numpy()
    assert _maxdiff(pulse[0], om.wavetable(eng.stage("f0").cpu().numpy()[:1])[0]) <= 5e-4
    syn = StreamingSynthesizer(eng, chunk_frames=8)
    syn.open(0)
    with pytest.raises((NotImplementedError, ValueError)):
        syn.push(0, mel[0], noise[0], last=True)
        for _ in range(4):
            syn.tick()


@pytest.mark.parametrize("case", ["subgain", "subgain_e"])
def test_subband_gain_model(torch, golden_dir, case):
    """ps_use_stft: false -- sub-band gains, interpolated and indexed exactly as the reference does it (by hop_size, first
    rows), against the reference's own run and the oracle; a ragged batch; only ["F0", .] among the parameters."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    gold = np.load(os.path.join(golden_dir, "reference_forward_f32.npz"))
    voice, overrides, batch, frames = GOLDEN_CASES[case]
    cfg, raw, wt = build_case(voice, overrides)
    eng, om = MBExWNEngine(cfg, raw, wt), orc.OracleModel(cfg, raw, wt)
    mel, noise = gold[f"{case}/mell"], gold[f"{case}/noise"]
    got = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()
    assert _maxdiff(got, gold[f"{case}/audio"]) <= _tol(gold[f"{case}/audio"])
    mel2, noise2 = synthetic_inputs(5, 2, 47)
    lengths = (47, 16)
    got = eng.forward(dev(torch, mel2), noise=dev(torch, noise2),
                      n_frames=torch.tensor(lengths, dtype=torch.int32, device="cuda")).cpu().numpy()
    for ii, ll in enumerate(lengths):
        ref = om.forward(mel2[ii:ii + 1, :ll], noise2[ii:ii + 1, :ll * 20])[0]
        assert _maxdiff(got[ii, :ll * 300], ref) <= _tol(ref)
    _, params = eng.infer(mel2, synth_length=47 * 300, return_F0=True, noise=noise2)
    assert [pp[0] for pp in params] == ["F0"]


def test_ps_off_model_returns_only_the_f0_parameter(torch):
    """ps_off: MBExWN.call returns the excitation as the signal and only ["F0", .] as parameter (neither PSig nor PS exist,
    reference custom_pulsed_generator.py:663-672, 756-767); infer_components has no envelope to return."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    cfg, raw, wt = build_case(*GOLDEN_CASES["psoff"][:2])
    assert not any(kk.startswith("PS_") for kk in raw)
    eng = MBExWNEngine(cfg, raw, wt)
    mel, noise = synthetic_inputs(31, 2, 12)
    audio, params = eng.infer(mel, synth_length=12 * 300 - 7, return_F0=True, noise=noise)
    assert [pp[0] for pp in params] == ["F0"] and audio.numpy().shape == (2, 12 * 300 - 7)
    assert np.array_equal(audio.numpy(), eng.stage("excitation").cpu().numpy()[:, :12 * 300 - 7])
    ref = orc.OracleModel(cfg, raw, wt).forward(mel, noise)
    assert _maxdiff(audio.numpy(), ref[:, :12 * 300 - 7]) <= _tol(ref)
    with pytest.raises(NotImplementedError):
        eng.infer_components(mel, synth_length=12 * 300, noise=noise)
    # a ragged batch: samples behind an item's own length are zero
    nf = torch.tensor([12, 5], dtype=torch.int32, device="cuda")
    got = eng.forward(dev(torch, mel), n_frames=nf, noise=dev(torch, noise)).cpu().numpy()
    assert np.all(got[1, 5 * 300:] == 0.0) and np.any(got[1, :5 * 300] != 0.0)


@pytest.mark.parametrize("act", ["gfu", "gsu", "glu"])
def test_gate_variants_at_full_width(torch, monkeypatch, act):
    """The other two gates through every gate kernel of the canonical geometry (C = 320): the folded first layer, F(4,3) in
    both block shapes, F(2,3) and the direct form (mbx_config.wn_conv_form / tune_gate_shape), each against the float64 oracle."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    cfg, raw, wt = build_case("SPEECH", {"mbexwn_config:pp_mod_subnet:activation": act})
    mel, noise = synthetic_inputs(55, 2, 60)
    ref = orc.OracleModel(cfg, raw, wt).forward(mel, noise)
    for env in ({"conv_form": "f43"}, {"conv_form": "f43", "tune": {"gate_shape": 1}}, {"conv_form": "f23"}, {"conv_form": "direct"}):
        eng = MBExWNEngine(cfg, raw, wt, **env)
        got = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()
        assert _maxdiff(got, ref) <= _tol(ref), f"{act} {env}"
        del eng

# ------------------------------------------------------------------------------------------------
# geometries far from the canonical one: every size is configuration driven
# ------------------------------------------------------------------------------------------------
_P, _M = "preprocess_config:", "mbexwn_config:"
_W = _M + "pp_mod_subnet:"
_GEOMETRIES = {
    "bands12_fold4": {_M + "multi_band_config": {"subbands": 12, "taps": 96, "cutoff_ratio": 0.05, "beta": 9.0},
                      _M + "pulse_channels": 4, _W + "cond_lin_upsampling": 5, _W + "n_channels": 48, _W + "n_layers": 3},
    "bands6_fold2": {_M + "multi_band_config": {"subbands": 6, "taps": 48, "cutoff_ratio": 0.1, "beta": 9.0},
                     _M + "pulse_channels": 2, _W + "cond_lin_upsampling": 10, _W + "n_channels": 32, _W + "n_layers": 3},
    # more than 16 bands / 32 output channels: the generic PQMF kernel and the un-fused end + post-net convolutions
    "bands30_out60": {_M + "multi_band_config": {"subbands": 30, "taps": 240, "cutoff_ratio": 0.02, "beta": 9.0},
                      _M + "pulse_channels": 10, _W + "cond_lin_upsampling": 5, _W + "n_channels": 32, _W + "n_layers": 2,
                      _W + "n_out_channels": 60},
    "sr16k_hop200": {_P + "sample_rate": 16000, _P + "hop_size": 200, _P + "win_size": 800, _P + "fft_size": 1024,
                     _M + "pulse_rate_factor": 2, _M + "pulse_channels": 5,
                     _M + "multi_band_config": {"subbands": 10, "taps": 80, "cutoff_ratio": 0.06, "beta": 9.0},
                     _W + "cond_lin_upsampling": 10, _W + "n_channels": 64, _W + "n_layers": 3, _M + "ps_max_ceps_coefs": 120},
    # the wave-per-frame STFT filter without its pruned first passes (more than 256 cepstral coefficients; a window of more
    # than 1 280 samples), and with the energy-preserving gain (coefficient 0 kept, wave-wide sum of |H|^2)
    "ceps400": {_M + "ps_max_ceps_coefs": 400, _W + "n_channels": 32, _W + "n_layers": 2},
    "win1600": {_P + "win_size": 1600, _P + "hop_size": 300, _W + "n_channels": 32, _W + "n_layers": 2,
                _M + "spect_filters_preserve_energy": True},
    "mel40_out44": {_P + "mel_channels": 40, _W + "n_out_channels": 44, _W + "n_channels": 40, _W + "n_layers": 3},
    "kernel5": {_W + "kernel_size": 5, _W + "n_channels": 32, _W + "n_layers": 3},
}


@pytest.mark.parametrize("name", sorted(_GEOMETRIES))
def test_other_model_geometries(torch, name):
    """Band counts, folded samples per row, sample rate / hop / FFT size, mel and output channel counts, kernel size: a
    ragged batch through the engine against the float64 oracle."""
    from mbexwn_vocoder_amd.config import ModelDims, canonical_config
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    from mbexwn_vocoder_amd.tables import WaveTables
    from mbexwn_vocoder_amd.weights import synthetic_weights
    cfg = canonical_config("SPEECH", **_GEOMETRIES[name])
    dims = ModelDims(cfg)
    raw = synthetic_weights(cfg, seed=77, bias_std=0.05, alpha_jitter=0.05)
    wt = WaveTables(sample_rate=dims.pulse_rate, **cfg["mbexwn_config"]["wavetable_config"])
    eng, om = MBExWNEngine(cfg, raw, wt), orc.OracleModel(cfg, raw, wt)
    rng = np.random.default_rng(5)
    frames, rpf, hop = 21, dims.wn_in_rows_per_frame, dims.hop_size
    mel = np.clip(np.log(np.exp(rng.normal(-5.0, 2.0, size=(2, frames, dims.mel_channels))) + 1e-5), -11.5, 2.0).astype(np.float32)
    noise = rng.normal(size=(2, frames * rpf)).astype(np.float32)
    lengths = (frames, 8)
    got = eng.forward(dev(torch, mel), n_frames=torch.tensor(lengths, dtype=torch.int32, device="cuda"),
                      noise=dev(torch, noise)).cpu().numpy()
    for ii, ll in enumerate(lengths):
        ref = om.forward(mel[ii:ii + 1, :ll], noise[ii:ii + 1, :ll * rpf])[0]
        assert _maxdiff(got[ii, :ll * hop], ref) <= _tol(ref)
        assert np.all(got[ii, ll * hop:] == 0.0)

def test_random_configurations_against_the_oracle(torch):
    """Thirty random model configurations x random ragged batches x a random convolution form (the first cases of
    tests/tools/config_fuzz.py, which ran 800 of them) against the float64 oracle; the numpy float32 port of the
    graph is the yardstick for draws that float32 itself conditions badly."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "config_fuzz.py"), "30", "1000"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "failures: 0" in res.stdout, res.stdout[-3000:] + res.stderr[-2000:]
    assert res.stdout.count(" OK ") >= 25
    # and twenty draws with the structural options on top: WaveNet blocks, causal padding, ps_off, sub-band gains, no PQMF
    # bank, PQMF analysis of the pulse signal, sinusoid-as-function (400 of them ran in the script)
    res = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "config_fuzz.py"), "20", "9000", "structure"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "failures: 0" in res.stdout, res.stdout[-3000:] + res.stderr[-2000:]
    assert res.stdout.count(" OK ") >= 18
