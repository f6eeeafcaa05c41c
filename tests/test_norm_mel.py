"""Row A14 (optional RMS normalisation): oracle and product host code against the reference's own
NormMelComponents executed on the numpy TensorFlow stand-in (tests/golden/make_reference_normmel.py)."""
import os

import numpy as np
import pytest

from mbexwn_vocoder_amd.config import canonical_config
from mbexwn_vocoder_amd.norm_mel import NormMel
from oracle.mbexwn_oracle import normalize_inputs_by_rms

CASES = {
    "iters1": {"normalize_rms_num_smooth_iters": 1},
    "iters2_comp": {"normalize_rms_num_smooth_iters": 2, "normalize_compressor_exp": 0.8, "max_norm_fact": 200.0},
    "scaled_win": {"normalize_rms_num_smooth_iters": 1, "normalize_smooth_win_scale": 2,
                   "normalize_smooth_with_squared_win": False, "lin_amp_scale": 1.5, "mel_amp_scale": 0.5},
    # normalize_use_pinv (reference wavegen_1d.py:603-608, 683-685): the RMS of the minimum-energy spectrum behind a mel frame
    "pinv": {"normalize_rms_num_smooth_iters": 1, "normalize_use_pinv": True},
}


def make_config(extra):
    cfg = canonical_config("SPEECH")
    cfg["mbexwn_config"].update(normalize_rms_from_mell=True, **extra)
    return cfg


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "reference_normmel.npz"))


@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_structural(gold, case):
    cfg = make_config(CASES[case])
    out, gain = normalize_inputs_by_rms(gold[f"f64/{case}/mell"], cfg, 17 * 300)
    # float32 constant tables; the pseudo inverse (float32 in the reference, its entries reach 1e3 with alternating signs)
    # loses a little more in the 1025-term contraction
    tol = 2e-5 if case == "pinv" else 1e-6
    np.testing.assert_allclose(out, gold[f"f64/{case}/mell_norm"], rtol=0, atol=tol)
    np.testing.assert_allclose(gain, gold[f"f64/{case}/gain"], rtol=tol, atol=0)


@pytest.mark.parametrize("case", sorted(CASES))
def test_product_host_code(gold, case):
    cfg = make_config(CASES[case])
    out, gain = NormMel(cfg).normalize(gold[f"f32/{case}/mell"], 17 * 300)
    assert out.dtype == np.float32 and gain.dtype == np.float32 and gain.shape == (2, 5100)
    tol = 2e-4 if case == "pinv" else 2e-5          # float32 contraction over 1025 bins of +-1e3 pseudo-inverse entries
    np.testing.assert_allclose(out, gold[f"f32/{case}/mell_norm"], rtol=0, atol=tol)
    np.testing.assert_allclose(gain, gold[f"f32/{case}/gain"], rtol=tol, atol=0)
    ref, ref_gain = normalize_inputs_by_rms(gold[f"f32/{case}/mell"], cfg, 17 * 300)
    np.testing.assert_allclose(out, ref, rtol=0, atol=tol)
    np.testing.assert_allclose(gain, ref_gain, rtol=tol, atol=0)


def test_unsupported_variants():
    with pytest.raises(NotImplementedError):
        NormMel(make_config({"normalize_rms_num_smooth_iters": 0}))
    nm = NormMel(make_config({"normalize_rms_num_smooth_iters": 1, "normalize_use_pinv": True}))   # built since round 4
    assert nm.use_pinv and nm.pinv.shape == (80, 1025) and nm.pinv.dtype == np.float32 and nm.win_norm > 1.0
    cfg = make_config({"normalize_rms_num_smooth_iters": 1})
    cfg["preprocess_config"]["win_size"] = 1024
    with pytest.raises(RuntimeError, match="4 \\* hop_size"):
        NormMel(cfg)
