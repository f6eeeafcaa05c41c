"""Host-side drop-in surface (no GPU needed): scale_mel against reference I/O pairs, model registry,
config reading, file I/O, sub-net grammar."""
import os
import pickle

import numpy as np
import pytest

import mbexwn_vocoder_amd as pkg
from mbexwn_vocoder_amd import fileio, subnet
from mbexwn_vocoder_amd.config import ModelDims, canonical_config, dump_config, read_config
from mbexwn_vocoder_amd.mel_inverter import MELInverter

BASE = {"nfft": 2048, "hoplen": 300, "winlen": 1200, "nmels": 80, "sr": 24000, "fmin": 0.0, "fmax": 12000.0,
        "lin_spec_offset": 1e-5, "lin_spec_scale": 1, "log_spec_offset": 0.0, "log_spec_scale": 1, "time_axis": 1}


def make_inverter(**attrs):
    inv = MELInverter(None)
    inv.hop_size, inv._srate, inv.fft_size, inv.fmin, inv.fmax = 300, 24000, 2048, 0.0, 12000.0
    for kk, vv in attrs.items():
        setattr(inv, kk, vv)
    return inv


@pytest.mark.parametrize("tag,cfg_updates,inv_updates", [
    ("plain", {}, {}),
    ("max_limit", {}, {"use_max_limit": True, "lin_amp_off": 1e-4}),
    ("nfft1024", {"nfft": 1024}, {}),
    ("hop256", {"hoplen": 256}, {}),
    ("scaled", {"lin_spec_scale": 2.0, "log_spec_scale": 0.5, "log_spec_offset": 0.3},
     {"lin_amp_scale": 1.5, "mel_amp_scale": 0.25}),
])
def test_scale_mel_matches_reference(golden_dir, tag, cfg_updates, inv_updates):
    gold = np.load(os.path.join(golden_dir, "reference_scale_mel.npz"))
    dd = dict(BASE)
    dd.update(cfg_updates)
    dd["mell"] = gold[f"{tag}/in_mell"].copy()
    keep = dd["mell"].copy()
    out = make_inverter(**inv_updates).scale_mel(dd)
    ref = gold[f"{tag}/out"]
    assert out.dtype == ref.dtype == np.float32 and out.shape == ref.shape
    assert np.array_equal(out, ref)
    assert np.array_equal(dd["mell"], keep)          # the caller's array is left alone


def test_scale_mel_errors(golden_dir):
    gold = np.load(os.path.join(golden_dir, "reference_scale_mel.npz"))
    assert int(gold["fmin_mismatch_raises"]) == 1
    inv = make_inverter()
    dd = dict(BASE, fmin=50.0, mell=np.zeros((80, 3), np.float32))
    with pytest.raises(RuntimeError, match="fmin"):
        inv.scale_mel(dd)
    dd = dict(BASE, fmax=8000.0, mell=np.zeros((80, 3), np.float32))
    with pytest.raises(RuntimeError, match="fmax"):
        inv.scale_mel(dd)
    dd = dict(BASE)
    with pytest.raises(RuntimeError, match="no supported mel"):
        inv.scale_mel(dd)
    # fmax None is accepted when the model fmax is Nyquist
    dd = dict(BASE, fmax=None, mell=np.zeros((80, 3), np.float32))
    assert inv.scale_mel(dd).shape == (1, 3, 80)


def test_linear_mel_key():
    inv = make_inverter()
    rng = np.random.default_rng(0)
    mel = np.exp(rng.normal(-5, 1, size=(80, 5))).astype(np.float32)
    out = inv.scale_mel(dict(BASE, mel=mel, lin_spec_offset=0))
    np.testing.assert_allclose(out[0], np.log(mel.T + 1e-5), rtol=1e-6)


def test_registry_and_config_file(tmp_path, monkeypatch):
    models = pkg.list_models()
    assert set(models) == {"SING", "SPEECH", "VOICE"} and pkg.mbexwn_version == (1, 2, 3)
    models["SING"].append("x")
    assert "x" not in pkg.list_models()["SING"]            # deep copy like the reference
    with pytest.raises(FileNotFoundError):
        pkg.get_config_file("no-such-model")
    monkeypatch.setenv("MBEXWN_MODELS_DIR", str(tmp_path))
    name = pkg.list_models()["SPEECH"][0]
    os.makedirs(tmp_path / name)
    with pytest.raises(FileNotFoundError):
        pkg.get_config_file("SPEECH")
    dump_config(str(tmp_path / name / "config.yaml"), canonical_config("SPEECH"))
    assert pkg.get_config_file("SPEECH") == str(tmp_path / name / "config.yaml")
    assert pkg.get_config_file("SPEECH_IMP0").endswith("config.yaml")
    assert pkg.get_config_file(str(tmp_path / name)) == str(tmp_path / name / "config.yaml")


def test_read_config_defaults_and_includes(tmp_path):
    (tmp_path / "inc.yaml").write_text("pp:\n  hop_size: 300\n")
    (tmp_path / "main.yaml").write_text(
        "a:\n  __defaults__: {x: 1, y: 2}\n  x: 5\n"
        "lst:\n  - __defaults__: {k: 7}\n  - {name: one}\n  - {name: two, k: 9}\n"
        "inc: <@CONFIG_DIR@/inc.yaml:pp>\n"
        "dtype: tf.float32\n")
    cfg = read_config(str(tmp_path / "main.yaml"))
    assert cfg["a"] == {"x": 5, "y": 2}
    assert cfg["lst"] == [{"name": "one", "k": 7}, {"name": "two", "k": 9}]
    assert cfg["inc"] == {"hop_size": 300} and cfg["dtype"] == "float32"
    round_trip = tmp_path / "rt.yaml"
    dump_config(str(round_trip), canonical_config("VOICE"))
    assert read_config(str(round_trip)) == canonical_config("VOICE")


def test_model_dims_checks():
    dims = ModelDims(canonical_config("VOICE"))
    assert (dims.wn_channels, dims.steps_per_frame, dims.pulse_per_frame, dims.cond_conv_upsampling) == (340, 20, 100, 2)
    assert [dims.wn_dilation(ll) for ll in range(5)] == [1, 2, 4, 8, 16] and dims.fft_size == 2048
    with pytest.raises(RuntimeError, match="sample rate"):
        ModelDims(canonical_config(**{"mbexwn_config:pulse_channels": 4}))
    with pytest.raises(RuntimeError, match="conditioning rate"):
        ModelDims(canonical_config(**{"mbexwn_config:pp_mod_subnet:cond_lin_upsampling": 16}))
    with pytest.raises(NotImplementedError):
        ModelDims(canonical_config(**{"use_tf25_compatible_implementation": False}))
    with pytest.raises(AssertionError):
        ModelDims(canonical_config(**{"mbexwn_config:pp_mod_subnet:kernel_size": 4}))
    cfg = canonical_config(**{"mbexwn_config:pp_mod_subnet:max_log2_dilation_rate": 3,
                              "mbexwn_config:pp_mod_subnet:n_layers": 7})
    assert [ModelDims(cfg).wn_dilation(ll) for ll in range(7)] == [1, 2, 4, 1, 2, 4, 1]


def test_subnet_grammar():
    ops, ups, cout = subnet.build_subnet([[3, 128], [3, 64, "L2"], [5, 32, 2], ["L", 5]], "PulsPar", 80, 1, 1,
                                         "soft_sigmoid", target_ups=100)
    kinds = [op["kind"] for op in ops]
    assert kinds == ["conv", "prelu", "conv", "lin", "prelu", "conv", "prelu", "lin", "conv", "lin", "act"]
    assert ops[0]["pad_mode"] == subnet.PAD_SYMMETRIC and (ops[0]["pad_l"], ops[0]["pad_r"]) == (1, 1)
    assert ops[5]["cout"] == 64 and ops[5]["up"] == 2 and ops[5]["pad_mode"] == subnet.PAD_ZERO
    assert ops[9]["up"] == 25 and ups == 100 and cout == 1       # bare ["L",5] is not counted (reference quirk)
    assert subnet.subnet_time_factor(ops) == 500
    with pytest.raises(RuntimeError):
        subnet.build_subnet([[3, 8, 3]], "X", 80, 1, 1, None, target_ups=100)
    ops, _, _ = subnet.build_subnet([[4, 8]], "PS", 80, 240, 1, None, pad_to_valid=True)
    assert ops[0]["pad_mode"] == subnet.PAD_EDGE and (ops[0]["pad_l"], ops[0]["pad_r"]) == (2, 1)


def test_fileio_roundtrip(tmp_path):
    dd = dict(BASE, mell=np.arange(160, dtype=np.float32).reshape(80, 2))
    for name in ("a.mell", "a.mell.gz"):
        fileio.save_var(str(tmp_path / name), dd)
        back = fileio.load_var(str(tmp_path / name))
        assert back.keys() == dd.keys() and np.array_equal(back["mell"], dd["mell"])
    with open(tmp_path / "plain.mell", "wb") as fo:      # a file written by the reference's save_var
        pickle.dump(dd, fo, -1)
    assert fileio.load_var(str(tmp_path / "plain.mell"))["hoplen"] == 300


def test_generate_mel_from_snd():
    inv = make_inverter(mel_channels=80, win_len=1200, preprocess_config=canonical_config()["preprocess_config"])
    rng = np.random.default_rng(3)
    dd = inv.generate_mel_from_snd(rng.normal(size=(6000,)).astype(np.float32), 24000)
    assert dd["mell"].shape == (80, 21) and dd["hoplen"] == 300 and dd["nfft"] == 2048 and dd["sr"] == 24000
    # the dictionary feeds straight back into scale_mel (round trip of the CLI's -v check)
    assert inv.scale_mel(dd).shape == (1, 21, 80)
    # another sample rate: resampled to the model rate first (the reference calls a function it never imports there);
    # a 1 kHz sine analysed at 16 kHz and at 24 kHz gives the same mel frames away from the edges
    tt = np.arange(24000) / 24000.0
    ref = inv.generate_mel_from_snd(np.sin(2 * np.pi * 1000.0 * tt).astype(np.float32), 24000)["mell"]
    tt16 = np.arange(16000) / 16000.0
    got = inv.generate_mel_from_snd(np.sin(2 * np.pi * 1000.0 * tt16).astype(np.float32), 16000)["mell"]
    assert got.shape == ref.shape
    assert np.max(np.abs(np.exp(got[:, 5:-5]) - np.exp(ref[:, 5:-5]))) < 2e-3 * np.max(np.exp(ref))
