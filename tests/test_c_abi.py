"""The C-ABI library loads on a CPU-only host and exports every entry point include/mbexwn.h declares;
the ctypes mirror of the structs has the layout the C compiler gives them (no compute calls here)."""
import ctypes
import os
import re
import subprocess

import pytest

from mbexwn_vocoder_amd import engine
from mbexwn_vocoder_amd.build import LIB_PATH, build_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mbexwn.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mbx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    build_library()
    assert os.path.exists(LIB_PATH)
    lib = engine.load_library()
    names = declared_functions()
    assert "mbx_forward" in names and "mbx_create" in names and len(names) >= 12
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in mbexwn.h but not exported"
    assert sorted(engine.EXPORTED_SYMBOLS) == names


def test_struct_layout_matches_c(tmp_path):
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mbexwn.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(mbx_config), sizeof(mbx_subnet_op),'
                   ' sizeof(mbx_tensor), offsetof(mbx_config, n_f0_ops), offsetof(mbx_config, vtf_ops),'
                   ' offsetof(mbx_config, wt_nominal_f0), sizeof(mbx_forward_options), offsetof(mbx_forward_options, wn_frames),'
                   ' offsetof(mbx_forward_options, sub_carry), offsetof(mbx_forward_options, layer_rows),'
                   ' offsetof(mbx_config, wn_conv_form), offsetof(mbx_config, tune_resskip_split), sizeof(mbx_conv_form_info),'
                   ' offsetof(mbx_conv_form_info, err_f23)); return 0;}\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    got = [int(vv) for vv in out]
    cc = engine.mbx_config
    assert got == [ctypes.sizeof(cc), ctypes.sizeof(engine.mbx_subnet_op), ctypes.sizeof(engine.mbx_tensor),
                   cc.n_f0_ops.offset, cc.vtf_ops.offset, cc.wt_nominal_f0.offset,
                   ctypes.sizeof(engine.mbx_forward_options), engine.mbx_forward_options.wn_frames.offset,
                   engine.mbx_forward_options.sub_carry.offset, engine.mbx_forward_options.layer_rows.offset,
                   cc.wn_conv_form.offset, cc.tune_resskip_split.offset, ctypes.sizeof(engine.mbx_conv_form_info),
                   engine.mbx_conv_form_info.err_f23.offset]


def test_policy_fields_and_experiment_variables(monkeypatch, capsys):
    """The library reads no environment variable: the policy is mbx_config fields, which the Python host fills from its
    arguments -- or, for the experiment scripts, from the MBX_* variables, naming them on stderr."""
    from helpers import build_case
    cfg, raw, wt = build_case("SPEECH", {})
    for kk in ("MBX_WINOGRAD", "MBX_FOLD_SKIP", "MBX_FOLD_START", "MBX_WG_SMALL", "MBX_RV_TILES", "MBX_RV_SPLIT", "MBX_EXPERIMENT"):
        monkeypatch.delenv(kk, raising=False)
    cc, _ = engine.make_config(cfg, wt)
    assert (cc.wn_conv_form, cc.batch_invariant, cc.wn_keep_skip, cc.wn_keep_start, cc.tune_gate_shape) == (0, 0, 0, 0, 0)
    assert capsys.readouterr().err == ""
    cc, _ = engine.make_config(cfg, wt, conv_form="f23", batch_invariant=True, keep_start=True, tune={"resskip_split": 2})
    assert (cc.wn_conv_form, cc.batch_invariant, cc.wn_keep_start, cc.tune_resskip_split) == (2, 1, 1, 2)
    monkeypatch.setenv("MBX_WINOGRAD", "44")
    monkeypatch.setenv("MBX_FOLD_START", "0")
    monkeypatch.setenv("MBX_WG_SMALL", "1")
    monkeypatch.delenv("MBX_EXPERIMENT", raising=False)
    cc, _ = engine.make_config(cfg, wt)                                # without the opt-in the variables are ignored, loudly
    assert (cc.wn_conv_form, cc.batch_invariant, cc.wn_keep_start, cc.tune_gate_shape) == (0, 0, 0, 0)
    assert "ignoring MBX_WINOGRAD=44" in capsys.readouterr().err
    monkeypatch.setenv("MBX_EXPERIMENT", "1")
    cc, _ = engine.make_config(cfg, wt)
    assert (cc.wn_conv_form, cc.batch_invariant, cc.wn_keep_start, cc.tune_gate_shape) == (3, 1, 1, 2)
    assert "MBX_WINOGRAD=44" in capsys.readouterr().err
    cc, _ = engine.make_config(cfg, wt, conv_form="direct")           # an explicit argument wins over the variable
    assert cc.wn_conv_form == 1
    with pytest.raises(ValueError):
        engine.make_config(cfg, wt, conv_form="f63")
    src = open(os.path.join(ROOT, "mbexwn_vocoder_amd", "csrc", "mbx_api.hip")).read()
    assert "getenv" not in src


def test_make_config_and_tensor_table():
    from helpers import build_case
    cfg, raw, wt = build_case("VOICE", {})
    cconf, dims = engine.make_config(cfg, wt)
    assert cconf.struct_size == ctypes.sizeof(engine.mbx_config) and cconf.wn_channels == 340
    assert [cconf.wn_dilations[ii] for ii in range(5)] == [1, 2, 4, 8, 16]
    assert cconf.n_f0_ops == 9 and cconf.n_vtf_ops == 5   # 3x(conv,prelu) + final conv + lin + act ; 2x(conv,prelu) + final conv
    assert cconf.f0_ops[0].name == b"PulsPar_Layer_0" and cconf.f0_ops[8].act == 1
    tensors = engine.tensor_table(cfg, raw, wt)
    assert tensors["wn.conv1D_0.w"].shape == (3, 340, 680) and tensors["table.pqmf_syn"].shape == (121, 15)
    assert tensors["wn.res_skip_4.w"].shape == (1, 340, 340) and tensors["table.wavetables"].shape == (513, 15)
    assert all(vv.dtype.name == "float32" and vv.flags["C_CONTIGUOUS"] for vv in tensors.values())


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from helpers import build_case
    cfg, raw, wt = build_case("SPEECH", {})
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.MBExWNEngine(cfg, raw, wt)


def test_gate_kernel_codes_match_the_header():
    """mbx_conv_form_info.gate_kernel[] (ABI 10): the MBX_GATE_K_* codes of include/mbexwn.h and the names engine.py reports."""
    import re
    from mbexwn_vocoder_amd import engine
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mbexwn.h")).read()
    codes = {name: int(val) for name, val in re.findall(r"#define MBX_GATE_K_(\w+) (\d+)", header)}
    assert sorted(codes.values()) == sorted(engine.GATE_KERNEL_NAMES)
    want = {"NONE": "none", "DIRECT": "direct", "F23": "f23", "F43": "f43", "F43_PSPLIT": "f43_psplit", "F43_HSPLIT": "f43_hsplit",
            "F43_STRIDED": "f43_strided", "F43_STRIDED_PSPLIT": "f43_strided_psplit", "FOLDED_START": "folded_start",
            "SPLIT_F16": "split_f16"}
    assert {engine.GATE_KERNEL_NAMES[vv]: kk for kk, vv in codes.items()} == {vv: kk for kk, vv in want.items()}
    assert int(re.search(r"#define MBX_ABI_VERSION (\d+)", header).group(1)) == engine.MBX_ABI_VERSION == 10
    # bench.py's executed-FLOP factors know every kernel that multiplies
    import bench
    assert set(bench.GATE_EXECUTED) == set(engine.GATE_KERNEL_NAMES.values()) - {"none", "folded_start", "split_f16"}
