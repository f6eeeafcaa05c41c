"""TF-checkpoint reader (mbexwn_vocoder_amd/tf_checkpoint.py): building blocks against known answers, tables with
several prefix-compressed blocks, and a full round trip of a model's variables under the reference's variable names.
No TensorFlow-written file exists in this environment, so compatibility rests on the published formats the module
restates (its docstring says so)."""
import os
import struct

import numpy as np
import pytest

from mbexwn_vocoder_amd import tf_checkpoint as tfc
from mbexwn_vocoder_amd.config import canonical_config
from mbexwn_vocoder_amd.weights import layer_table, synthetic_weights


def test_crc32c_known_answers():
    assert tfc.crc32c(b"123456789") == 0xE3069283                  # Castagnoli check value
    assert tfc.crc32c(b"\x00" * 32) == 0x8A9136AA                   # RFC 3720 B.4
    assert tfc.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert tfc.mask_crc(0) == 0xA282EAD8


def test_varint_and_message_round_trip():
    for value in (0, 1, 127, 128, 300, 2 ** 32 + 5, 2 ** 63 - 1):
        buf = tfc.write_varint(value)
        assert tfc.read_varint(buf, 0) == (value, len(buf))
    entry = tfc.encode_bundle_entry(1, (3, 80, 128), 0, 4096, 3 * 80 * 128 * 4, 0xDEADBEEF)
    got = tfc.parse_bundle_entry(entry)
    assert got["dtype"] == 1 and got["shape"] == (3, 80, 128) and got["offset"] == 4096
    assert got["size"] == 3 * 80 * 128 * 4 and got["crc32c"] == 0xDEADBEEF and not got["sliced"]


def test_snappy_decoder():
    # literal "abcd", copy (1-byte offset form) of 8 bytes from offset 4 -> "abcdabcdabcd"; then a 2-byte-offset copy
    stream = bytes([16]) + bytes([3 << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4]) + bytes([((4 - 1) << 2) | 2, 12, 0])
    assert tfc.snappy_decompress(stream) == b"abcdabcdabcdabcd"
    with pytest.raises(ValueError):
        tfc.snappy_decompress(bytes([5]) + bytes([3 << 2]) + b"abcd")


def test_table_with_many_blocks_and_shared_prefixes(tmp_path):
    items = [(f"model/layer_{ii:03d}/kernel/.ATTRIBUTES/VARIABLE_VALUE".encode(), os.urandom(ii % 7 + 1))
             for ii in range(300)] + [(b"", b"header")]
    path = str(tmp_path / "t.index")
    tfc.write_table(path, items, block_entries=17)
    assert tfc.read_table(path) == sorted(items)
    raw = bytearray(open(path, "rb").read())
    raw[10] ^= 0xFF                                                  # corrupt a data block
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        tfc.read_table(path)
    with pytest.raises(ValueError, match="magic"):
        open(path, "wb").write(b"\x00" * 100)
        tfc.read_table(path)


_reference_names = tfc.to_reference_variables


def test_checkpoint_round_trip_restores_the_engine_weights(tmp_path):
    cfg = canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:n_channels": 32})
    raw = synthetic_weights(cfg, seed=5)
    prefix = str(tmp_path / "weights.tf")
    named = _reference_names(raw)
    named["save_counter"] = np.asarray(3, dtype=np.int64)             # a variable the engine does not know
    tfc.write_checkpoint(prefix, named)
    reader = tfc.CheckpointReader(prefix)
    assert tfc.OBJECT_GRAPH_KEY in reader.entries and len(reader.variables()) == len(named)
    key = reader.variables()["mb_ex_wn/PulsPar_Layer_0/kernel"]
    assert np.array_equal(reader.get_tensor(key, verify=True), raw["PulsPar_Layer_0.v"])
    got = tfc.load_reference_checkpoint(prefix, cfg)
    assert sorted(got) == sorted(raw)
    for name in raw:
        assert got[name].dtype == np.float32 and np.array_equal(got[name], raw[name]), name
    convs, prelus = layer_table(cfg)
    assert len(got) == 3 * len(convs) + len(prelus)


def test_checkpoint_mismatch_is_reported(tmp_path):
    cfg = canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:n_channels": 32})
    raw = synthetic_weights(cfg, seed=5)
    named = _reference_names(raw)
    del named["mb_ex_wn/PS_Layer_final_base/g"]
    named["mb_ex_wn/PulsPar_Layer_0/kernel"] = named["mb_ex_wn/PulsPar_Layer_0/kernel"][:, :, :5]
    prefix = str(tmp_path / "weights.tf")
    tfc.write_checkpoint(prefix, named)
    with pytest.raises(ValueError, match="missing PS_Layer_final.g|PulsPar_Layer_0.v has shape"):
        tfc.load_reference_checkpoint(prefix, cfg)
    with pytest.raises(FileNotFoundError):
        tfc.CheckpointReader(str(tmp_path / "nothing.tf"))


def test_both_keras_name_scopes_of_the_wrapped_convolution_load(tmp_path):
    """TF2C_Conv1DWeightNorm builds its inner Conv1D from inside its own build() (reference conv_layers.py:72-75), so
    the inner kernel / bias can appear under `<layer>/...` or under `<layer>_base/...`: both spellings must map."""
    cfg = canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:n_channels": 32})
    raw = synthetic_weights(cfg, seed=7)
    named = {}
    for name, arr in _reference_names(raw).items():
        scope, leaf = name.rsplit("/", 1)
        if leaf in ("kernel", "bias") and not scope.endswith("_base"):
            name = f"{scope}_base/{leaf}"                            # the alternative spelling
        named[name] = arr
    assert any(kk.endswith("start_base/kernel") for kk in named)
    prefix = str(tmp_path / "weights.tf")
    tfc.write_checkpoint(prefix, named)
    got = tfc.load_reference_checkpoint(prefix, cfg)
    assert sorted(got) == sorted(raw)
    for name in raw:
        assert np.array_equal(got[name], raw[name]), name


def test_wavenet_without_weight_norm_loads_and_folds(tmp_path):
    """pp_mod_subnet.use_weight_norm = False (the WaveNetAE default, reference custom_AE_layers.py:123): those layers
    have no `g`; the kernel is the weight."""
    from mbexwn_vocoder_amd.weights import fold_weights
    cfg = canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:n_channels": 32,
                                        "mbexwn_config:pp_mod_subnet:use_weight_norm": False})
    raw = {kk: vv for kk, vv in synthetic_weights(cfg, seed=9).items() if not (kk.startswith("wn.") and kk.endswith(".g"))}
    prefix = str(tmp_path / "weights.tf")
    tfc.write_checkpoint(prefix, _reference_names(raw))
    got = tfc.load_reference_checkpoint(prefix, cfg)
    assert sorted(got) == sorted(raw) and "wn.start.g" not in got and "PS_Layer_0.g" in got
    folded = fold_weights(got)
    assert np.array_equal(folded["wn.conv1D_0.w"], raw["wn.conv1D_0.v"])
    # with weight norm configured the missing g is an error
    cfg_wn = canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:n_channels": 32})
    with pytest.raises(ValueError, match="missing wn"):
        tfc.load_reference_checkpoint(prefix, cfg_wn)


def test_pre_conditioning_layers_and_missing_conditioning_round_trip(tmp_path):
    """`precond_<i>` layers (reference custom_AE_layers.py:192-201) map to "wn.precond_<i>"; a model built with
    disable_conditioning has no `cond_` layer at all (:203-204)."""
    for extra, present, absent in (({"mbexwn_config:pp_mod_subnet:pre_cond_layer_channels": [48, 40]}, "wn.precond_1.v", None),
                                   ({"mbexwn_config:pp_mod_subnet:disable_conditioning": True}, "wn.start.v", "wn.cond.v")):
        cfg = canonical_config("SPEECH", **dict({"mbexwn_config:pp_mod_subnet:n_channels": 32}, **extra))
        raw = synthetic_weights(cfg, seed=11)
        assert present in raw and (absent is None or absent not in raw)
        if "wn.precond_1.v" in raw:
            assert raw["wn.precond_0.v"].shape == (3, 80, 48) and raw["wn.precond_1.v"].shape == (3, 48, 40)
            assert raw["wn.cond.v"].shape[1] == 40
        prefix = str(tmp_path / ("w" + str(len(raw))))
        named = _reference_names(raw)
        if "wn.precond_1.v" in raw:
            assert any(kk.endswith("PP_waveNetBlock_ups1_0_WNBlock_WN/precond_1/kernel") for kk in named)
        tfc.write_checkpoint(prefix, named)
        got = tfc.load_reference_checkpoint(prefix, cfg)
        assert sorted(got) == sorted(raw)
        for name in raw:
            assert np.array_equal(got[name], raw[name]), name


def test_several_wavenet_blocks_round_trip(tmp_path):
    """Layers of the WaveNet blocks behind the first one carry the same names inside another "PP_waveNetBlock_ups<u>_<i>"
    scope; the up-sampling convolution of block i is "<block>_WNBlock_UP_<factor>" (reference custom_pulsed_generator.py:487,
    custom_AE_layers.py:519-524)."""
    cfg = canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 2,
                                        "mbexwn_config:pp_mod_subnet_upsampling_factors": [2, 1],
                                        "mbexwn_config:pp_mod_subnet_channel_factors": [1, 0.5],
                                        "mbexwn_config:pulse_channels": 10, "mbexwn_config:pp_mod_subnet:cond_lin_upsampling": 5})
    raw = synthetic_weights(cfg, seed=13)
    assert raw["wn1.start.v"].shape == (1, 30, 16) and raw["up0.v"].shape == (3, 30, 60) and raw["wn1.cond.v"].shape == (3, 80, 2 * 16 * 4)
    named = _reference_names(raw, config=cfg)
    # names built the way the reference builds them: block "PP_waveNetBlock_ups{ups}_{iwn}" (custom_pulsed_generator.py:487),
    # its WaveNet self.name + "_WNBlock_WN" (custom_AE_layers.py:516), its up layer self.name + f"_WNBlock_UP_{factor}" --
    # which exists only for ups > 1 (custom_AE_layers.py:518-526)
    ups = cfg["mbexwn_config"]["pp_mod_subnet_upsampling_factors"]
    blocks = ["PP_waveNetBlock_ups{}_{}".format(uu, ii) for ii, uu in enumerate(ups)]
    for ii, (blk, uu) in enumerate(zip(blocks, ups)):
        assert any(f"/{blk}/{blk}_WNBlock_WN/conv1D_0/kernel" in kk for kk in named), blk
        has_up = any(f"/{blk}/{blk}_WNBlock_UP_{uu}/kernel" in kk for kk in named)
        assert has_up == (uu > 1), blk
    assert not any("_WNBlock_UP_1" in kk or "ups1_0" in kk for kk in named)
    with pytest.raises(ValueError, match="pass config"):
        _reference_names(raw)                                          # two blocks need the configuration's factors
    prefix = str(tmp_path / "blocks")
    tfc.write_checkpoint(prefix, named)
    got = tfc.load_reference_checkpoint(prefix, cfg)
    assert sorted(got) == sorted(raw)
    for name in raw:
        assert np.array_equal(got[name], raw[name]), name


def test_unbuilt_wavenet_options_raise():
    """Keys of WaveNetAE.__init__ that change the arithmetic and are not built must not be ignored silently."""
    from mbexwn_vocoder_amd.config import ModelDims
    with pytest.raises(NotImplementedError):    # VALID cannot add the conditioning rows up in the reference's own graph
        ModelDims(canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:padding": "VALID"}))
    assert ModelDims(canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:padding": "causal"})).wn_padding == "CAUSAL"
    with pytest.raises(RuntimeError, match="unsupported wavenet activation"):   # the reference's own check and message
        ModelDims(canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:activation": "relu"}))
    # built since round 3 (second batch): pre-conditioning layers, a WaveNet without conditioning, the glu gate
    dims = ModelDims(canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:pre_cond_layer_channels": [64, 48]}))
    assert dims.wn_pre_cond_channels == [64, 48]
    assert ModelDims(canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:disable_conditioning": True})).wn_disable_conditioning
    assert ModelDims(canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:activation": "glu"})).wn_activation == "glu"
    assert ModelDims(canonical_config("SPEECH", **{"mbexwn_config:spect_filters_preserve_energy": True})).preserve_energy
    # built since round 3: the gfu / gsu gates and use_equalized_lr (folded on the host)
    assert ModelDims(canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:use_equalized_lr": True})).wn_equalized_lr
    assert ModelDims(canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:activation": "gsu"})).wn_activation == "gsu"
    # channel groups are built (block-diagonal dense layers); the reference's divisibility check stays
    assert ModelDims(canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:n_ch_groups": 2})).wn_groups == 2
    with pytest.raises(RuntimeError, match="multiple of chanel groups"):
        ModelDims(canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:n_ch_groups": 3}))
