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
    np.testing.assert_allclose(out, gold[f"f64/{case}/mell_norm"], rtol=0, atol=1e-6)    # float32 constant tables
    np.testing.assert_allclose(gain, gold[f"f64/{case}/gain"], rtol=1e-6, atol=0)


@pytest.mark.parametrize("case", sorted(CASES))
def test_product_host_code(gold, case):
    cfg = make_config(CASES[case])
    out, gain = NormMel(cfg).normalize(gold[f"f32/{case}/mell"], 17 * 300)
    assert out.dtype == np.float32 and gain.dtype == np.float32 and gain.shape == (2, 5100)
    np.testing.assert_allclose(out, gold[f"f32/{case}/mell_norm"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(gain, gold[f"f32/{case}/gain"], rtol=2e-5, atol=0)
    ref, ref_gain = normalize_inputs_by_rms(gold[f"f32/{case}/mell"], cfg, 17 * 300)
    np.testing.assert_allclose(out, ref, rtol=0, atol=2e-5)
    np.testing.assert_allclose(gain, ref_gain, rtol=2e-5, atol=0)


def test_unsupported_variants():
    with pytest.raises(NotImplementedError):
        NormMel(make_config({"normalize_rms_num_smooth_iters": 0}))
    with pytest.raises(NotImplementedError):
        NormMel(make_config({"normalize_rms_num_smooth_iters": 1, "normalize_use_pinv": True}))
    cfg = make_config({"normalize_rms_num_smooth_iters": 1})
    cfg["preprocess_config"]["win_size"] = 1024
    with pytest.raises(RuntimeError, match="4 \\* hop_size"):
        NormMel(cfg)
