"""The oracle against golden vectors produced by the reference's own model code
(tests/golden/make_reference_forward.py: MBExWN.call executed on a numpy stand-in for TensorFlow)."""
import os

import numpy as np
import pytest

from oracle.mbexwn_oracle import OracleModel
from helpers import GOLDEN_CASES, LEAN_GOLDEN_CASES, build_case


def _load(golden_dir, tag):
    return np.load(os.path.join(golden_dir, f"reference_forward_{tag}.npz"))


def _maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))))


@pytest.mark.parametrize("case", sorted(GOLDEN_CASES))
def test_structural_float64(golden_dir, case):
    """Same graph, float64 end to end: every stage must agree to rounding noise of float64."""
    gold = _load(golden_dir, "f64")
    voice, overrides, batch, frames = GOLDEN_CASES[case]
    cfg, raw, wt = build_case(voice, overrides)
    om = OracleModel(cfg, raw, wt, dtype=np.float64, float32_constants=False)
    mel, noise = gold[f"{case}/mell"], gold[f"{case}/noise"]
    audio, st = om.forward(mel, noise, return_stages=True)
    assert audio.shape == (batch, frames * 300)
    assert _maxdiff(st["f0"], gold[f"{case}/f0"]) < 1e-10
    if case in LEAN_GOLDEN_CASES:          # long cases: the float64 file keeps f0 / excitation / audio only
        assert _maxdiff(st["excitation"], gold[f"{case}/excitation"]) < 1e-8
        assert _maxdiff(audio, gold[f"{case}/audio"]) < 1e-8
        return
    assert _maxdiff(wt.tables, gold[f"{case}/wavetables"]) == 0.0
    assert _maxdiff(om.phase_from_f0(gold[f"{case}/f0"]), gold[f"{case}/phase"]) == 0.0
    assert _maxdiff(om.wavetable(gold[f"{case}/f0"]), gold[f"{case}/pulse"]) < 1e-7     # grid_norm is a float32 constant
    assert _maxdiff(om.conditioning(mel.astype(np.float64)), gold[f"{case}/cond"]) < 1e-12
    assert _maxdiff(st["excitation"], gold[f"{case}/excitation"]) < 1e-8
    assert _maxdiff(st["envelope"], gold[f"{case}/envelope_re"] + 1j * gold[f"{case}/envelope_im"]) < 1e-11
    assert _maxdiff(audio, gold[f"{case}/audio"]) < 1e-8


@pytest.mark.parametrize("case", sorted(GOLDEN_CASES))
def test_float32_emulation(golden_dir, case):
    """float64 oracle (float32-faithful phase/index arithmetic) vs the float32 run of the reference code.
    Tolerance = float32 rounding of a 4-amplitude signal through ~15 layers: 1e-4 absolute."""
    gold = _load(golden_dir, "f32")
    voice, overrides, batch, frames = GOLDEN_CASES[case]
    cfg, raw, wt = build_case(voice, overrides)
    om = OracleModel(cfg, raw, wt)
    mel, noise = gold[f"{case}/mell"], gold[f"{case}/noise"]
    audio, st = om.forward(mel, noise, return_stages=True)
    # the phase accumulator is bit exact given the same float32 F0
    assert _maxdiff(om.phase_from_f0(gold[f"{case}/f0"]), gold[f"{case}/phase"]) == 0.0
    assert _maxdiff(om.wavetable(gold[f"{case}/f0"]), gold[f"{case}/pulse"]) < 1e-6
    assert _maxdiff(st["f0"], gold[f"{case}/f0"]) < 1e-3          # Hz, float32 sub-net on values up to 600
    assert _maxdiff(st["excitation"], gold[f"{case}/excitation"]) < 1e-4
    assert _maxdiff(audio, gold[f"{case}/audio"]) < 1e-4
    if case in LEAN_GOLDEN_CASES:
        if f"{case}/cond" in gold.files:                            # absent with disable_conditioning
            cond = om.conditioning(mel.astype(np.float64))[:, ::37]
            assert _maxdiff(cond, gold[f"{case}/cond"]) < 1e-5
        if case in ("canon60", "voice"):                           # several phase chunks and the offset chain are pinned
            assert (frames * 100 + 999) // 1000 >= 5
    if f"{case}/ceps_window_sum" in gold.files:
        idx = om.cepstral_window_index(gold[f"{case}/f0"])
        assert _maxdiff(om.ceps_windows[idx].sum(axis=-1), gold[f"{case}/ceps_window_sum"]) < 1e-4


def test_float32_oracle_mode(golden_dir):
    """The float32 mode of the oracle (the CPU timing baseline) stays within float32 tolerance."""
    gold = _load(golden_dir, "f32")
    voice, overrides, batch, frames = GOLDEN_CASES["small"]
    cfg, raw, wt = build_case(voice, overrides)
    om = OracleModel(cfg, raw, wt, dtype=np.float32)
    audio = om.forward(gold["small/mell"], gold["small/noise"])
    assert audio.dtype == np.float32
    assert _maxdiff(audio, gold["small/audio"]) < 2e-4


def test_torch_cpu_port_matches_the_float64_oracle():
    """oracle/mbexwn_oracle_torch.py (the CPU baseline of bench.py: the WaveNet on torch-CPU float32 ops, the rest the numpy
    float32 port) against the float64 oracle, at the path's tolerance; canonical model and a gfu variant."""
    import numpy as np
    from helpers import build_case, synthetic_inputs
    from oracle.mbexwn_oracle import OracleModel
    from oracle.mbexwn_oracle_torch import TorchOracleModel
    for over in ({"mbexwn_config:pp_mod_subnet:n_channels": 64}, {"mbexwn_config:pp_mod_subnet:n_channels": 32,
                                                                  "mbexwn_config:pp_mod_subnet:activation": "gfu"}):
        cfg, raw, wt = build_case("SPEECH", over)
        mel, noise = synthetic_inputs(3, 2, 21)
        ref = OracleModel(cfg, raw, wt).forward(mel, noise)
        got = TorchOracleModel(cfg, raw, wt).forward(mel, noise)
        assert got.dtype == np.float32
        assert np.max(np.abs(got - ref)) <= 1e-4 * max(1.0, float(np.abs(ref).max()))


# the BASELINE lengths and the reference's default depth (tests/golden/make_reference_long.py)
LONG_CASES = {
    "speech240": ("SPEECH", {}, 1, 240),
    "speech800": ("SPEECH", {}, 1, 800),
    "voice400": ("VOICE", {}, 1, 400),
    "deep12": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 12}, 1, 240),
    "deep12_short": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 12}, 1, 60),
    "cycle12": ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 12,
                           "mbexwn_config:pp_mod_subnet:max_log2_dilation_rate": 4}, 2, 31),
}


@pytest.mark.parametrize("case", sorted(LONG_CASES))
def test_long_cases_structural_float64_and_float32_phase(golden_dir, case):
    """The oracle at 3 s / 10 s / 5 s (C = 340) and with 12 layers (dilations to 2048, and in cycles of four): float64 mode
    against the float64 run of the reference graph to rounding noise; the float32-faithful phase integrator against the
    float32 run's phase, bit for bit, over every 1000-sample chunk of the utterance."""
    g32 = np.load(os.path.join(golden_dir, "reference_long_f32.npz"))
    g64 = np.load(os.path.join(golden_dir, "reference_long_f64.npz"))
    voice, overrides, batch, frames = LONG_CASES[case]
    cfg, raw, wt = build_case(voice, overrides)
    mel, noise = g32[f"{case}/mell"], g32[f"{case}/noise"]
    assert mel.shape == (batch, frames, 80)
    om64 = OracleModel(cfg, raw, wt, dtype=np.float64, float32_constants=False)
    audio, st = om64.forward(mel, noise, return_stages=True)
    assert _maxdiff(st["f0"], g64[f"{case}/f0"]) < 1e-10
    assert _maxdiff(audio, g64[f"{case}/audio"]) < 1e-8
    om = OracleModel(cfg, raw, wt)
    assert _maxdiff(om.phase_from_f0(g32[f"{case}/f0"]), g32[f"{case}/phase"]) == 0.0
    # the default oracle (float64 arithmetic on the float32-cast contour, float32-faithful phase / index / table work) is the
    # float32 run without its rounding noise: inside the stated tolerance of the float32 run at every length.  The float64 run
    # of the graph is a different object -- its phase integrator runs in float64 too -- and BOTH float32 phase chains sit a few
    # 1e-4 from it (the float32 running sum of f0 / 8000, reference tf_wavetable.py:429-492, rounds at 1e-6 cycles per add)
    got = om.forward(mel, noise)
    ref32, ref64 = g32[f"{case}/audio"], g64[f"{case}/audio"]
    amp = max(1.0, float(np.abs(ref64).max()))
    assert _maxdiff(got, ref32) < 1e-4 * amp
    assert _maxdiff(got, ref64) < 3e-4 * amp and _maxdiff(ref32, ref64) < 3e-4 * amp
    assert _maxdiff(got, ref32) < _maxdiff(ref32, ref64)
