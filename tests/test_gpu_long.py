"""HIP engine against the reference's own float32 and float64 runs at the BASELINE lengths (3 s, 10 s; 5 s for C = 340)
and at the reference's default WaveNet depth (12 layers, dilations to 2048) -- tests/golden/make_reference_long.py.

What is stated here, max|.| over the whole utterance, A = max|audio| (measured: DESIGN.md section 5):

  HIP <-> reference float32 run (the emulation of TF-CPU): <= 1e-4 max(1, A) at 3 s AND at 10 s.  The HIP path evaluates the
      F0-net in float64 and rounds the contour once; the reference's float32 contour is off by <= 8e-5 Hz in most samples and
      its float32 phase integrator (reference tf_wavetable.py:429-492) turns that into a pulse-position difference that grows
      with the length: the measured distance grows from ~3e-5 (3 s) to ~2.6e-4 (10 s) at A = 4.3.
  HIP given the reference's float32 contour <-> reference float32 run: phase bit-equal, audio <= 2e-5 max(1, A) at every
      length -- the rest of the graph (WaveNet, PQMF, STFT filter) does not drift.
  float64 run of the reference graph: its phase integrator runs in float64 as well, so it is a DIFFERENT object from any
      float32 evaluation -- ref32 <-> ref64 is 3.3e-4 (3 s) / 8.6e-4 (10 s), HIP <-> ref64 the same class.  Reported, bounded
      by 3e-4 max(1, A), not a parity target: the reference computes in float32.
"""
import os

import numpy as np
import pytest

from helpers import build_case

pytestmark = pytest.mark.gpu

E2E_TOL = 1e-4
E2E_TIGHT = 2e-5

SMALL12 = {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 12}
LONG_CASES = {
    # name: (voice, overrides, batch, frames)
    "speech240": ("SPEECH", {}, 1, 240),
    "speech800": ("SPEECH", {}, 1, 800),
    "voice400": ("VOICE", {}, 1, 400),
    "deep12": ("SPEECH", SMALL12, 1, 240),
    "deep12_short": ("SPEECH", SMALL12, 1, 60),
    "cycle12": ("SPEECH", dict(SMALL12, **{"mbexwn_config:pp_mod_subnet:max_log2_dilation_rate": 4}), 2, 31),
}


@pytest.fixture(scope="module")
def torch():
    import torch as _torch
    assert _torch.cuda.is_available(), "GPU tests need an MI355X"
    return _torch


@pytest.fixture(scope="module")
def gold(golden_dir):
    return (np.load(os.path.join(golden_dir, "reference_long_f32.npz")), np.load(os.path.join(golden_dir, "reference_long_f64.npz")))


_ENGINES = {}


def get_engine(case, **kwargs):
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    voice, overrides = LONG_CASES[case][:2]
    key = (voice, tuple(sorted((kk, str(vv)) for kk, vv in overrides.items())), tuple(sorted(kwargs.items())))
    if key not in _ENGINES:
        _ENGINES[key] = MBExWNEngine(*build_case(voice, overrides), **kwargs)
    return _ENGINES[key]


def dev(torch, arr, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(arr), dtype=dtype or torch.float32).cuda()


def _maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def three_way(torch, eng, g32, g64, case):
    """The numbers of the three-way table for one case (also used by scripts/parity_table.py)."""
    mel, noise = g32[f"{case}/mell"], g32[f"{case}/noise"]
    ref32, ref64 = g32[f"{case}/audio"], g64[f"{case}/audio"]
    got = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()
    f0 = eng.stage("f0").cpu().numpy()
    exc = eng.stage("excitation").cpu().numpy()
    phase = eng.wavetable(dev(torch, f0))[1].cpu().numpy()
    part = np.flatnonzero((phase != g32[f"{case}/phase"]).ravel())
    res = {"frames": int(mel.shape[1]), "amplitude": float(np.abs(ref64).max()),
           "ref32_ref64": _maxdiff(ref32, ref64), "hip_ref64": _maxdiff(got, ref64), "hip_ref32": _maxdiff(got, ref32),
           "f0_hip_ref64_hz": _maxdiff(f0, g64[f"{case}/f0"]), "f0_ref32_ref64_hz": _maxdiff(g32[f"{case}/f0"], g64[f"{case}/f0"]),
           "f0_hip_within_an_ulp_of_ref64": bool(np.all(np.abs(f0.astype(np.float64) - g64[f"{case}/f0"]) <= np.spacing(f0))),
           "excitation_hip_ref32": _maxdiff(exc, g32[f"{case}/excitation"]),
           "first_pulse_sample_where_the_phase_chains_part": int(part[0]) if part.size else -1,
           "phase_samples_that_differ": int(part.size), "phase_samples": int(phase.size)}
    # the same handle given the reference's float32 contour (infer_components' F0 argument, reference wavegen_1d.py:528-557)
    if not eng.dims.no_envelope:
        eng.infer_components(mel, F0=g32[f"{case}/f0"], noise=noise)
        inj = eng.last_audio.cpu().numpy()
        f0i = eng.stage("f0").cpu().numpy()
        res["hip_with_ref32_contour_ref32"] = _maxdiff(inj, ref32)
        res["phase_with_ref32_contour_bit_equal"] = bool(np.array_equal(eng.wavetable(dev(torch, f0i))[1].cpu().numpy(), g32[f"{case}/phase"]))
    return res, got


@pytest.mark.parametrize("case", ["speech240", "speech800", "voice400"])
def test_baseline_lengths_against_the_reference_runs(torch, gold, case):
    """The default handle (float64 F0-net, auto form) at 3 s / 10 s (C = 320) and 5 s (C = 340) against both runs of the
    reference graph (module docstring): inside the stated tolerance of the float32 run at every length; bit-equal phase and
    2e-5 when given that run's contour."""
    g32, g64 = gold
    eng = get_engine(case)
    res, _ = three_way(torch, eng, g32, g64, case)
    print(case, res)
    amp = max(1.0, res["amplitude"])
    # the contour: within one float32 ulp of the float64 run's (whose constants are float64 too; against the oracle, which
    # keeps the reference's float32 constants, it is the nearest float32: tests/test_gpu_parity.py) -- the float32 run is ~8 x off
    assert res["f0_hip_within_an_ulp_of_ref64"] and res["f0_hip_ref64_hz"] <= 2e-5 < res["f0_ref32_ref64_hz"]
    assert res["hip_ref32"] <= E2E_TOL * amp, "the stated tolerance against the float32 run, at this length"
    assert res["hip_ref32"] < res["ref32_ref64"], "the HIP path is the float32 run without its rounding noise, not the float64 graph"
    assert res["hip_ref64"] <= 3e-4 * amp and res["ref32_ref64"] <= 3e-4 * amp
    # with the reference's own float32 contour the phase chains coincide and what is left is float32 rounding of the
    # WaveNet and the filters on both sides: no growth with the length
    assert res["phase_with_ref32_contour_bit_equal"]
    assert res["hip_with_ref32_contour_ref32"] <= E2E_TIGHT * amp


FORM_KW = {"default": {}, "direct": {"conv_form": "direct"}, "f23": {"conv_form": "f23"}, "f43": {"conv_form": "f43"},
           "f43_invariant": {"conv_form": "f43", "batch_invariant": True}}


@pytest.mark.parametrize("form", ["default", "direct", "f43_invariant"])
def test_default_depth_12_layers_dilations_to_2048(torch, gold, form):
    """WaveNetAE's own defaults (reference custom_AE_layers.py:120-123, 229-233): 12 layers, d = 1 .. 2048.  One item of
    240 frames (4800 rows: the widest layers have real rows on both sides), one of 60 frames (1200 rows: shorter than the
    dilation of the last two layers, only their centre tap sees data) -- each against the reference's runs, as single items
    and as a ragged batch of two (bit-equal to the single runs)."""
    g32, g64 = gold
    eng = get_engine("deep12", **FORM_KW[form])
    outs = {}
    for case in ("deep12", "deep12_short"):
        mel, noise = g32[f"{case}/mell"], g32[f"{case}/noise"]
        got = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()
        outs[case] = got
        ref32, ref64 = g32[f"{case}/audio"], g64[f"{case}/audio"]
        amp = max(1.0, float(np.abs(ref64).max()))
        print(form, case, "hip-ref64 %.3e hip-ref32 %.3e ref32-ref64 %.3e amp %.2f" % (_maxdiff(got, ref64), _maxdiff(got, ref32), _maxdiff(ref32, ref64), amp))
        f0 = eng.stage("f0").cpu().numpy()
        assert np.all(np.abs(f0.astype(np.float64) - g64[f"{case}/f0"]) <= np.spacing(f0))
        assert _maxdiff(got, ref32) <= E2E_TOL * amp
        assert _maxdiff(got, ref64) <= 3e-4 * amp
    mel = np.zeros((2, 240, 80), dtype=np.float32)
    noise = np.zeros((2, 4800), dtype=np.float32)
    mel[0], noise[0] = g32["deep12/mell"][0], g32["deep12/noise"][0]
    mel[1, :60], noise[1, :1200] = g32["deep12_short/mell"][0], g32["deep12_short/noise"][0]
    nf = torch.tensor([240, 60], dtype=torch.int32, device="cuda")
    rag = eng.forward(dev(torch, mel), n_frames=nf, noise=dev(torch, noise)).cpu().numpy()
    if form != "default":       # a pinned form: an item's bits do not depend on its batch (the default picks kernels by launch size)
        assert np.array_equal(rag[0], outs["deep12"][0])
        assert np.array_equal(rag[1, :18000], outs["deep12_short"][0])
    else:
        assert _maxdiff(rag[0], outs["deep12"][0]) <= E2E_TIGHT * 12 and _maxdiff(rag[1, :18000], outs["deep12_short"][0]) <= E2E_TIGHT * 12
    assert np.all(rag[1, 18000:] == 0.0)


@pytest.mark.parametrize("form", ["default", "direct", "f43", "f23"])
def test_12_layers_in_dilation_cycles(torch, gold, form):
    """max_log2_dilation_rate = 4: d = 1, 2, 4, 8 three times (reference custom_AE_layers.py:229-231)."""
    g32, g64 = gold
    eng = get_engine("cycle12", **FORM_KW[form])
    mel, noise = g32["cycle12/mell"], g32["cycle12/noise"]
    got = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()
    ref32, ref64 = g32["cycle12/audio"], g64["cycle12/audio"]
    amp = max(1.0, float(np.abs(ref64).max()))
    print(form, "hip-ref64 %.3e hip-ref32 %.3e ref32-ref64 %.3e amp %.2f" % (_maxdiff(got, ref64), _maxdiff(got, ref32), _maxdiff(ref32, ref64), amp))
    assert _maxdiff(got, ref32) <= E2E_TOL * amp
    assert _maxdiff(got, ref64) <= 3e-4 * amp


def test_f43_at_dilations_above_16_runs_the_interleaved_sub_sequences(torch):
    """Dilations 32 .. 2048 in Winograd F(4,3) form (csrc/wn_winograd4w.hip, VS kernels: an item is d / 16 interleaved
    virtual items at dilation 16).  MW-VO-FD width (C = 340: partial column tile, partial last slice), 12 layers, ragged
    batch of two whose items end inside blocks of every sub-sequence: both block shapes give the same bits; they, the default
    policy (F(4,3) where padding the sub-sequences to blocks costs less than the direct form's multiplies) and the direct
    form agree with the oracle; under batch_invariant a ragged batch is bit-equal to single runs."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    from oracle.mbexwn_oracle import OracleModel
    from helpers import synthetic_inputs
    cfg, raw, wt = build_case("VOICE", {"mbexwn_config:pp_mod_subnet:n_layers": 12})
    lengths = [400, 173]
    mel, noise = synthetic_inputs(123, 2, 400)
    nf = torch.as_tensor(lengths, dtype=torch.int32).cuda()
    outs = {}
    for name, kw in (("shape0", {"conv_form": "f43", "tune": {"gate_shape": 1}}), ("shape1", {"conv_form": "f43", "tune": {"gate_shape": 2}}),
                     ("default", {}), ("direct", {"conv_form": "direct"}), ("invariant", {"conv_form": "f43", "batch_invariant": True})):
        eng = MBExWNEngine(cfg, raw, wt, **kw)
        outs[name] = eng.forward(dev(torch, mel), n_frames=nf, noise=dev(torch, noise)).cpu().numpy()
        if name == "invariant":
            for ii, ll in enumerate(lengths):
                single = eng.forward(dev(torch, mel[ii:ii + 1, :ll]), noise=dev(torch, noise[ii:ii + 1, :ll * 20])).cpu().numpy()
                assert np.array_equal(outs[name][ii, :ll * 300], single[0])
        del eng
    assert np.array_equal(outs["shape0"], outs["shape1"]), "256-row and product-split blocks: the same sums in the same order"
    om = OracleModel(cfg, raw, wt)
    for ii, ll in enumerate(lengths):
        ref = om.forward(mel[ii:ii + 1, :ll], noise[ii:ii + 1, :ll * 20])[0]
        amp = max(1.0, float(np.abs(ref).max()))
        for name, got in outs.items():
            err = _maxdiff(got[ii, :ll * 300], ref)
            print(name, "item", ii, "max|d| %.3e (amp %.2f)" % (err, amp))
            assert err <= E2E_TOL * amp, name
            assert np.all(got[ii, ll * 300:] == 0.0)
    assert not np.array_equal(outs["shape0"], outs["direct"])
