"""The float64 weights of the F0-net (engine.tensor_table "<layer>.w64", mbx_config.f0_accumulate = MBX_F0_ACC_F64) are the
oracle's own weight-norm fold: the contour can only be the float32 nearest to the oracle's if both contract the same numbers."""
import numpy as np

from helpers import build_case
from mbexwn_vocoder_amd import engine
from oracle import mbexwn_oracle as orc


def test_f0_net_w64_tensors_are_the_oracle_fold():
    cfg, raw, wt = build_case("SPEECH", {})
    table = engine.tensor_table(cfg, raw, wt)
    om = orc.OracleModel(cfg, raw, wt)
    names = [kk[:-4] for kk in table if kk.endswith(".w64")]
    assert sorted(names) == ["PulsPar_Layer_0", "PulsPar_Layer_1", "PulsPar_Layer_2", "PulsPar_Layer_final"]
    for name in names:
        w64 = table[name + ".w64"]
        assert w64.dtype == np.float32 and w64.shape[-1] == 2 and w64.flags["C_CONTIGUOUS"]
        as_double = w64.view(np.float64).reshape(w64.shape[:-1])
        ref, _ = om.weight(name)
        assert as_double.shape == ref.shape and np.array_equal(as_double, ref)          # bit for bit
        # the float32 weights of the same layer are the reference's float32 fold (TensorFlow's arithmetic), a few ulps away
        assert np.max(np.abs(table[name + ".w"].astype(np.float64) - ref)) <= 4e-7 * np.max(np.abs(ref))


def test_f0_accumulate_is_a_config_field():
    cfg, raw, wt = build_case("SPEECH", {})
    cc, _ = engine.make_config(cfg, wt)
    assert cc.f0_accumulate == 0                                        # MBX_F0_ACC_F64, the default
    cc, _ = engine.make_config(cfg, wt, f0_accumulate="f32")
    assert cc.f0_accumulate == 1
    import pytest
    with pytest.raises(ValueError):
        engine.make_config(cfg, wt, f0_accumulate="f16")
