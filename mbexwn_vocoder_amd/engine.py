"""ctypes binding of the HIP engine (include/mbexwn.h) and its host-side driver.

``MBExWNEngine`` plays the role of the reference's Keras model object: it is what
``MELInverter.model`` holds and what ``model.infer(mell, synth_length=...)`` is called on
(reference MBExWN_NVoc/mel_inverter.py:151-154, MBExWN_NVoc/vocoder/model/wavegen_1d.py:483-526).
PyTorch is used for device memory and streams only.

There is NO CPU fallback: the engine refuses to load without the compiled HIP library, and refuses
to run without a GPU.
"""
import ctypes
import os

import numpy as np

from . import tables as tb
from .build import LIB_PATH
from .config import ModelDims
from .subnet import build_subnet
from .weights import fold_weights, merge_channel_groups

MBX_ABI_VERSION = 10
MBX_MAX_SUBNET_OPS = 32
MBX_MAX_WN_LAYERS = 64
MBX_MAX_PRECOND = 8
MBX_MAX_WN_BLOCKS = 4
MBX_NAME_LEN = 64

_OP_KIND = {"conv": 0, "lin": 1, "prelu": 2, "leaky": 3, "act": 4}
_ACT = {"linear": 0, "soft_sigmoid": 1, "tanh": 2, "sigmoid": 3, "soft_sign": 4, "soft_sqrt": 5, "exp": 6, "relu": 7}


class mbx_subnet_op(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("ks", ctypes.c_int32), ("cin", ctypes.c_int32), ("cout", ctypes.c_int32),
                ("pad_l", ctypes.c_int32), ("pad_r", ctypes.c_int32), ("pad_mode", ctypes.c_int32),
                ("up", ctypes.c_int32), ("act", ctypes.c_int32), ("alpha", ctypes.c_float),
                ("name", ctypes.c_char * MBX_NAME_LEN)]


class mbx_config(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_int32), ("abi_version", ctypes.c_int32),
                ("sample_rate", ctypes.c_int32), ("hop_size", ctypes.c_int32), ("mel_channels", ctypes.c_int32),
                ("subbands", ctypes.c_int32), ("pqmf_taps", ctypes.c_int32),
                ("pulse_channels", ctypes.c_int32), ("pulse_per_frame", ctypes.c_int32),
                ("steps_per_frame", ctypes.c_int32),
                ("pulse_rate", ctypes.c_float), ("noise_sigma", ctypes.c_float),
                ("f0_min", ctypes.c_float), ("f0_max", ctypes.c_float),
                ("wn_channels", ctypes.c_int32), ("wn_layers", ctypes.c_int32), ("wn_kernel_size", ctypes.c_int32),
                ("wn_out_channels", ctypes.c_int32), ("wn_in_channels", ctypes.c_int32),
                ("wn_dilations", ctypes.c_int32 * MBX_MAX_WN_LAYERS),
                ("cond_kernel_size", ctypes.c_int32), ("cond_conv_upsampling", ctypes.c_int32),
                ("cond_lin_upsampling", ctypes.c_int32),
                ("stft_win", ctypes.c_int32), ("fft_size", ctypes.c_int32), ("n_ceps", ctypes.c_int32),
                ("n_ceps_windows", ctypes.c_int32), ("filter_max_log_range", ctypes.c_float),
                ("wt_n_period", ctypes.c_int32), ("wt_n_tables", ctypes.c_int32),
                ("wt_nominal_f0", ctypes.c_float), ("wt_min_transposition", ctypes.c_float),
                ("wt_max_transposition", ctypes.c_float), ("wt_grid_norm", ctypes.c_float),
                ("phase_chunk", ctypes.c_int32),
                ("n_f0_ops", ctypes.c_int32), ("f0_ops", mbx_subnet_op * MBX_MAX_SUBNET_OPS),
                ("n_vtf_ops", ctypes.c_int32), ("vtf_ops", mbx_subnet_op * MBX_MAX_SUBNET_OPS),
                ("nm_iters", ctypes.c_int32), ("nm_smooth_win", ctypes.c_int32), ("nm_use_compressor", ctypes.c_int32),
                ("nm_use_max_limit", ctypes.c_int32), ("nm_rms_norm_fact", ctypes.c_float),
                ("nm_rms_floor", ctypes.c_float), ("nm_compressor_exp", ctypes.c_float),
                ("nm_lin_amp_scale", ctypes.c_float), ("nm_lin_amp_off", ctypes.c_float),
                ("nm_mel_amp_scale", ctypes.c_float), ("wn_gate_activation", ctypes.c_int32),
                ("wn_disable_conditioning", ctypes.c_int32), ("n_precond", ctypes.c_int32),
                ("precond_channels", ctypes.c_int32 * MBX_MAX_PRECOND), ("spect_preserve_energy", ctypes.c_int32),
                ("wt_subharm_channels", ctypes.c_int32), ("wt_sinusoid_as_fun", ctypes.c_int32),
                ("ps_off", ctypes.c_int32), ("no_pqmf", ctypes.c_int32), ("n_wn_blocks", ctypes.c_int32),
                ("wn_block_channels", ctypes.c_int32 * MBX_MAX_WN_BLOCKS), ("wn_block_ups", ctypes.c_int32 * MBX_MAX_WN_BLOCKS),
                ("pulse_pqmf_taps", ctypes.c_int32), ("ps_subband_gain", ctypes.c_int32),
                ("wn_causal", ctypes.c_int32),
                ("wn_conv_form", ctypes.c_int32), ("batch_invariant", ctypes.c_int32), ("wn_keep_skip", ctypes.c_int32),
                ("wn_keep_start", ctypes.c_int32), ("calib_fraction", ctypes.c_float), ("tune_gate_shape", ctypes.c_int32),
                ("tune_resskip_wave_tiles", ctypes.c_int32), ("tune_resskip_split", ctypes.c_int32),
                ("nm_use_pinv", ctypes.c_int32), ("nm_win_norm", ctypes.c_float), ("wn_precision", ctypes.c_int32),
                ("f0_accumulate", ctypes.c_int32)]


class mbx_conv_form_info(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_int32), ("requested", ctypes.c_int32), ("form", ctypes.c_int32),
                ("stream_form", ctypes.c_int32), ("calibrated", ctypes.c_int32), ("batch_invariant", ctypes.c_int32),
                ("fold_skip", ctypes.c_int32), ("fold_start", ctypes.c_int32), ("split_f16_layers", ctypes.c_int32),
                ("split_f16_gate_layers", ctypes.c_int32), ("err_f43", ctypes.c_float),
                ("err_f23", ctypes.c_float), ("ref_max", ctypes.c_float), ("threshold", ctypes.c_float),
                ("err_split", ctypes.c_float), ("split_rejected", ctypes.c_int32), ("f0_float64_chain", ctypes.c_int32),
                ("n_gate_layers", ctypes.c_int32), ("gate_kernel", ctypes.c_int32 * MBX_MAX_WN_LAYERS)]


GATE_KERNEL_NAMES = {0: "none", 1: "direct", 2: "f23", 3: "f43", 4: "f43_psplit", 5: "f43_hsplit", 6: "f43_strided",
                     7: "f43_strided_psplit", 8: "folded_start", 9: "split_f16"}


CONV_FORMS = {"auto": 0, "direct": 1, "f23": 2, "f43": 3}
_CONV_FORM_NAMES = {vv: kk for kk, vv in CONV_FORMS.items()}


class mbx_forward_options(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_int32), ("transposition", ctypes.c_float), ("f0", ctypes.c_void_p),
                ("state_in", ctypes.c_void_p), ("state_out", ctypes.c_void_p), ("active_begin", ctypes.c_int32),
                ("active_frames", ctypes.c_void_p), ("wn_begin", ctypes.c_int32), ("wn_frames", ctypes.c_void_p),
                ("active_max_frames", ctypes.c_int32), ("wn_max_frames", ctypes.c_int32), ("sub_store", ctypes.c_void_p), ("sub_store_rows", ctypes.c_int32), ("sub_carry", ctypes.c_void_p),
                ("layer_store", ctypes.c_void_p), ("layer_store_floats", ctypes.c_int32), ("layer_carry", ctypes.c_void_p),
                ("layer_rows", ctypes.c_int32), ("fe_store", ctypes.c_void_p), ("fe_ring_frames", ctypes.c_int32),
                ("fe_pos", ctypes.c_void_p), ("fe_new_frames", ctypes.c_int32), ("fe_margin_frames", ctypes.c_int32),
                ("fe_end_frames", ctypes.c_int32)]


class mbx_tensor(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.POINTER(ctypes.c_float)), ("ndim", ctypes.c_int32),
                ("shape", ctypes.c_int64 * 4)]


_STATUS_EXC = {1: ValueError, 2: RuntimeError, 3: RuntimeError, 4: NotImplementedError}

_lib = None


def load_library():
    """Load libmbexwn_hip.so (built in-tree by mbexwn_vocoder_amd.build). Fails loudly when absent."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch must be loaded first: the engine works on torch's device pointers and streams, so both have to
    # live in ONE HIP runtime instance (torch bundles libamdhip64.so.7; loading ours first would give the
    # process a second, separate runtime).
    import torch  # noqa: F401
    # MBX_LIB_PATH: another build of the same library (kernel experiments, scripts/experiments/, whose ablated kernels
    # produce wrong audio on purpose) -- never a fallback; said on stderr so that a stale variable cannot go unnoticed
    path = os.environ.get("MBX_LIB_PATH") or LIB_PATH
    if path != LIB_PATH:
        import sys
        print(f"mbexwn_vocoder_amd: MBX_LIB_PATH overrides the product library: loading {path}", file=sys.stderr)
    if not os.path.exists(path):
        raise RuntimeError(f"HIP extension {path} is missing: run `python -m mbexwn_vocoder_amd.build` "
                           "(there is no CPU fallback for the mel-inversion path)")
    lib = ctypes.CDLL(path)
    vp, i32, i64p, fp = ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p
    lib.mbx_last_error.restype = ctypes.c_char_p
    lib.mbx_last_error.argtypes = []
    lib.mbx_create.restype = i32
    lib.mbx_create.argtypes = [ctypes.POINTER(mbx_config), ctypes.POINTER(mbx_tensor), i32, i32,
                               ctypes.POINTER(ctypes.c_void_p)]
    lib.mbx_destroy.restype = i32
    lib.mbx_destroy.argtypes = [vp]
    lib.mbx_conv_form.restype = i32
    lib.mbx_conv_form.argtypes = [vp, ctypes.POINTER(mbx_conv_form_info)]
    lib.mbx_calibrate.restype = i32
    lib.mbx_calibrate.argtypes = [vp, fp, vp, i32, i32, fp, vp, ctypes.c_size_t, vp]
    lib.mbx_workspace_size.restype = ctypes.c_size_t
    lib.mbx_workspace_size.argtypes = [vp, i32, i32]
    lib.mbx_forward.restype = i32
    lib.mbx_forward.argtypes = [vp, fp, vp, i32, i32, fp, fp, vp, ctypes.c_size_t, vp]
    lib.mbx_forward_stream.restype = i32
    lib.mbx_forward_stream.argtypes = [vp, fp, vp, i32, i32, fp, fp, vp, ctypes.c_size_t, vp, vp, vp]
    lib.mbx_forward_ex.restype = i32
    lib.mbx_forward_ex.argtypes = [vp, fp, vp, i32, i32, fp, fp, vp, ctypes.c_size_t,
                                   ctypes.POINTER(mbx_forward_options), vp]
    lib.mbx_layer_state_info.restype = i32
    lib.mbx_layer_state_info.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    lib.mbx_window_advance.restype = i32
    lib.mbx_window_advance.argtypes = [vp, fp, fp, fp, fp, i32, i32, i32, vp]
    lib.mbx_clock_probe.restype = i32
    lib.mbx_clock_probe.argtypes = [vp, vp, ctypes.c_int64, vp]
    lib.mbx_window_update.restype = i32
    lib.mbx_window_update.argtypes = [vp, fp, fp, fp, fp, i32, i32, i32, i32, i32, vp]
    lib.mbx_emit_rows.restype = i32
    lib.mbx_emit_rows.argtypes = [vp, fp, ctypes.c_int64, i32, ctypes.c_int64, ctypes.c_int64, fp, vp]
    lib.mbx_stage.restype = i32
    lib.mbx_stage.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), i64p, i64p]
    lib.mbx_profile_enable.restype = i32
    lib.mbx_profile_enable.argtypes = [vp, i32]
    lib.mbx_profile_read.restype = i32
    lib.mbx_profile_read.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), i64p]
    lib.mbx_profile_read_launches.restype = i32
    lib.mbx_profile_read_launches.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_float), ctypes.c_int64, i64p]
    lib.mbx_pqmf_synthesis.restype = i32
    lib.mbx_pqmf_synthesis.argtypes = [vp, fp, i32, i32, fp, vp]
    lib.mbx_conv1d.restype = i32
    lib.mbx_conv1d.argtypes = [vp, fp, i32, i32, i32, fp, fp, fp, i32, i32, i32, i32, i32, fp, vp]
    lib.mbx_conv1d_f64acc.restype = i32
    lib.mbx_conv1d_f64acc.argtypes = [vp, fp, i32, i32, i32, fp, fp, fp, i32, i32, i32, i32, i32, fp, vp]
    lib.mbx_lin_interp.restype = i32
    lib.mbx_lin_interp.argtypes = [vp, fp, i32, i32, i32, i32, fp, vp]
    lib.mbx_wavetable.restype = i32
    lib.mbx_wavetable.argtypes = [vp, fp, i32, i32, fp, fp, fp, vp]
    lib.mbx_stft_filter.restype = i32
    lib.mbx_stft_filter.argtypes = [vp, fp, fp, vp, i32, i32, fp, fp, vp]
    lib.mbx_norm_mel.restype = i32
    lib.mbx_norm_mel.argtypes = [vp, fp, vp, i32, i32, fp, fp, fp, vp]
    lib.mbx_mel_analysis.restype = i32
    lib.mbx_mel_analysis.argtypes = [fp, vp, i32, i32, i32, i32, i32, i32, fp, fp, fp, vp, vp, ctypes.c_float, fp, i32, vp]
    _lib = lib
    return lib


EXPORTED_SYMBOLS = ["mbx_last_error", "mbx_create", "mbx_destroy", "mbx_conv_form", "mbx_calibrate", "mbx_workspace_size", "mbx_forward",
                    "mbx_forward_stream", "mbx_forward_ex", "mbx_layer_state_info", "mbx_window_advance", "mbx_window_update", "mbx_emit_rows", "mbx_stage",
                    "mbx_profile_enable", "mbx_profile_read", "mbx_profile_read_launches", "mbx_clock_probe", "mbx_pqmf_synthesis", "mbx_conv1d", "mbx_conv1d_f64acc", "mbx_lin_interp", "mbx_wavetable", "mbx_stft_filter", "mbx_norm_mel", "mbx_mel_analysis"]


def _check(status):
    if status != 0:
        msg = load_library().mbx_last_error().decode("utf-8", "replace")
        raise _STATUS_EXC.get(status, RuntimeError)(f"mbexwn_hip: {msg}")


# ------------------------------------------------------------------------------------------------
# host-side model description
# ------------------------------------------------------------------------------------------------
def subnet_ops(config):
    """(f0 ops, vtf ops): the reference's sub-net grammar flattened (see subnet.build_subnet)."""
    dims = ModelDims(config)
    mb = config["mbexwn_config"]
    use_prelu = mb.get("use_prelu", True)
    f0_ops, _, _ = build_subnet(mb["pp_subnet"], "PulsPar", dims.mel_channels, 1, 1,
                                mb.get("pp_activation", "soft_sigmoid"), target_ups=dims.pulse_per_frame,
                                pad_to_valid=mb.get("pp_subnet_use_valid_padding", False),
                                remove_inactive_pad_layers=mb.get("remove_inactive_pad_layers", False),
                                use_prelu=use_prelu, alpha=dims.alpha, force_causal=mb.get("force_causal", False))
    vtf_ops = []
    if not dims.ps_off:
        vtf_ops, _, _ = build_subnet(mb["ps_subnet"], "PS", dims.mel_channels, dims.n_ceps, 1, None,
                                     pad_to_valid=mb.get("ps_subnet_use_valid_padding", False),
                                     remove_inactive_pad_layers=mb.get("remove_inactive_pad_layers", False),
                                     use_prelu=use_prelu, alpha=dims.alpha, force_causal=mb.get("force_causal", False))
    return f0_ops, vtf_ops


def _fill_ops(dst, ops):
    if len(ops) > MBX_MAX_SUBNET_OPS:
        raise ValueError("sub-net too deep for the engine")
    for ii, op in enumerate(ops):
        cop = dst[ii]
        cop.kind = _OP_KIND[op["kind"]]
        cop.ks = op.get("ks", 0)
        cop.cin = op.get("cin", 0)
        cop.cout = op.get("cout", 0)
        cop.pad_l = op.get("pad_l", 0)
        cop.pad_r = op.get("pad_r", 0)
        cop.pad_mode = op.get("pad_mode", 0)
        cop.up = op.get("up", 1)
        if op["kind"] == "act":
            if op["fn"] not in _ACT:
                raise RuntimeError(f"ActivationLayer::error::unkown activation selected {op['fn']}")
            cop.act = _ACT[op["fn"]]
        cop.alpha = op.get("alpha", 0.0)
        cop.name = op.get("name", "").encode()
    return len(ops)


def experiment_overrides():
    """The MBX_* environment variables of the experiment scripts and fuzzers, mapped onto the mbx_config policy fields
    (the library itself reads no environment variable): MBX_WINOGRAD=0|2|4|44 -> conv_form direct / f23 / f43 / f43 +
    batch_invariant, MBX_FOLD_SKIP=0 / MBX_FOLD_START=0 -> keep_skip / keep_start, MBX_WG_SMALL=0|1 -> tune_gate_shape 1|2,
    MBX_RV_TILES=n -> tune_resskip_wave_tiles (0 = never), MBX_RV_SPLIT=1|2|3 -> tune_resskip_split.  Only consulted for
    policy arguments the caller left unset, and only when the process opts in with MBX_EXPERIMENT=1 (the experiment scripts
    do): a stray MBX_WINOGRAD in a user's shell must not change the numerics of a default engine.  Every variable in effect
    -- or ignored for want of the opt-in -- is named on stderr."""
    env, out = os.environ, {}
    knobs = ("MBX_WINOGRAD", "MBX_FOLD_SKIP", "MBX_FOLD_START", "MBX_WG_SMALL", "MBX_RV_TILES", "MBX_RV_SPLIT")
    if env.get("MBX_EXPERIMENT", "0") != "1":
        stray = [kk for kk in knobs if kk in env]
        if stray:
            import sys
            print(f"mbexwn_vocoder_amd: ignoring {' '.join(f'{kk}={env[kk]}' for kk in stray)} (experiment variables need "
                  f"MBX_EXPERIMENT=1)", file=sys.stderr)
        return out
    if "MBX_WINOGRAD" in env:
        mode = int(env["MBX_WINOGRAD"])
        out["conv_form"] = {0: "direct", 2: "f23", 4: "f43", 44: "f43"}[mode]
        if mode == 44:
            out["batch_invariant"] = True
    if env.get("MBX_FOLD_SKIP", "1") == "0":
        out["keep_skip"] = True
    if env.get("MBX_FOLD_START", "1") == "0":
        out["keep_start"] = True
    tune = {}
    if "MBX_WG_SMALL" in env:
        tune["gate_shape"] = 1 if int(env["MBX_WG_SMALL"]) == 0 else 2
    if "MBX_RV_TILES" in env:
        tune["resskip_wave_tiles"] = int(env["MBX_RV_TILES"]) or -1
    if "MBX_RV_SPLIT" in env:
        tune["resskip_split"] = int(env["MBX_RV_SPLIT"])
    if tune:
        out["tune"] = tune
    if out:
        import sys
        names = [kk for kk in knobs if kk in env]
        print(f"mbexwn_vocoder_amd: experiment variables in effect: {' '.join(f'{kk}={env[kk]}' for kk in names)}",
              file=sys.stderr)
    return out


PRECISIONS = {"f32": 0, "split_f16": 1}
F0_ACCUMULATE = {"f64": 0, "f32": 1}       # mbx_config.f0_accumulate: MBX_F0_ACC_F64 (default) / MBX_F0_ACC_F32


def make_config(config, wavetables, conv_form=None, batch_invariant=None, keep_skip=None, keep_start=None,
                calib_fraction=None, tune=None, precision="f32", f0_accumulate="f64"):
    """mbx_config of a model.  Policy arguments (None = default, or the experiment variable if one is set):
    conv_form "auto" | "direct" | "f23" | "f43" (mbx_config.wn_conv_form), batch_invariant, keep_skip, keep_start,
    calib_fraction, tune = {"gate_shape": 0|1|2, "resskip_wave_tiles": n, "resskip_split": 0..3}."""
    dims = ModelDims(config)
    mb = config["mbexwn_config"]
    cc = mbx_config()
    cc.struct_size = ctypes.sizeof(mbx_config)
    cc.abi_version = MBX_ABI_VERSION
    cc.sample_rate, cc.hop_size, cc.mel_channels = dims.sample_rate, dims.hop_size, dims.mel_channels
    cc.subbands = dims.subbands
    cc.pqmf_taps = int(mb["multi_band_config"]["taps"])
    cc.pulse_channels, cc.pulse_per_frame, cc.steps_per_frame = dims.pulse_channels, dims.pulse_per_frame, dims.steps_per_frame
    cc.pulse_rate = dims.pulse_rate
    cc.noise_sigma = dims.noise_sigma
    cc.f0_min, cc.f0_max = dims.f0_min, dims.f0_max
    cc.wn_channels, cc.wn_layers, cc.wn_kernel_size = dims.wn_channels, dims.wn_layers, dims.wn_kernel_size
    cc.wn_out_channels, cc.wn_in_channels = dims.wn_out_channels, dims.wn_in_channels
    if dims.wn_layers > MBX_MAX_WN_LAYERS:
        raise ValueError("too many WaveNet layers for the engine")
    for ll in range(dims.wn_layers):
        cc.wn_dilations[ll] = dims.wn_dilation(ll)
    cc.cond_kernel_size = dims.cond_kernel_size
    cc.cond_conv_upsampling = dims.cond_conv_upsampling
    cc.cond_lin_upsampling = dims.cond_lin_upsampling
    cc.stft_win, cc.fft_size, cc.n_ceps = dims.stft_win, dims.fft_size, dims.n_ceps
    use_windows = (bool(dims.ps_env_order_scale) and not mb.get("psns_use_cepstral_loss_constraint", False) and
                   not dims.no_envelope)
    cc.n_ceps_windows = 30 if use_windows else 0
    cc.filter_max_log_range = dims.filter_max_log_range
    cc.wt_n_period, cc.wt_n_tables = wavetables.n_period, wavetables.n_tables
    cc.wt_nominal_f0 = wavetables.nominalF0
    cc.wt_min_transposition = float(wavetables.min_transposition)
    cc.wt_max_transposition = float(wavetables.max_transposition)
    cc.wt_grid_norm = float(wavetables.grid_norm)
    cc.phase_chunk = 1000
    cc.wn_gate_activation = {"gtu": 0, "gfu": 1, "gsu": 2, "glu": 3}[dims.wn_activation]
    cc.wn_disable_conditioning = int(dims.wn_disable_conditioning)
    if len(dims.wn_pre_cond_channels) > MBX_MAX_PRECOND:
        raise ValueError("too many pre-conditioning layers for the engine")
    cc.n_precond = len(dims.wn_pre_cond_channels)
    for ii, chans in enumerate(dims.wn_pre_cond_channels):
        cc.precond_channels[ii] = chans
    cc.spect_preserve_energy = int(dims.preserve_energy)
    cc.wt_subharm_channels = dims.wt_subharm
    cc.wt_sinusoid_as_fun = int(dims.wt_sinusoid_as_fun)
    cc.ps_off, cc.no_pqmf = int(dims.ps_off), int(dims.no_pqmf)
    cc.pulse_pqmf_taps = int(dims.pulse_pqmf["taps"]) if dims.pulse_pqmf else 0
    cc.ps_subband_gain = int(dims.ps_subband_gain)
    cc.wn_causal = int(dims.wn_padding == "CAUSAL")
    if dims.wn_multi:                 # several WaveNet blocks / in-block upsampling: the generic path of the library
        if dims.n_wn_blocks > MBX_MAX_WN_BLOCKS:
            raise ValueError("too many WaveNet blocks for the engine")
        cc.n_wn_blocks = dims.n_wn_blocks
        for bb in range(dims.n_wn_blocks):
            cc.wn_block_channels[bb], cc.wn_block_ups[bb] = dims.wn_block_channels[bb], dims.wn_block_ups[bb]
    f0_ops, vtf_ops = subnet_ops(config)
    cc.n_f0_ops = _fill_ops(cc.f0_ops, f0_ops)
    cc.n_vtf_ops = _fill_ops(cc.vtf_ops, vtf_ops)
    if mb.get("normalize_rms_from_mell", False):                     # row A14 (reference wavegen_1d.py:93-104)
        from .norm_mel import NormMel
        nm = NormMel(config)
        if nm.win != dims.stft_win:
            raise RuntimeError("normalize_rms_from_mell: preprocess win_size must equal the generator's STFT window")
        cc.nm_iters, cc.nm_smooth_win = nm.iters, nm.smooth_win_size
        cc.nm_use_compressor = int(nm.compressor_exp is not None)
        cc.nm_use_max_limit = int(nm.use_max_limit)
        cc.nm_rms_norm_fact = float(nm.rms_norm_fact)
        cc.nm_rms_floor = float(1.0 / nm.max_norm_fact) if nm.max_norm_fact else 0.0
        cc.nm_compressor_exp = float(nm.compressor_exp) if nm.compressor_exp is not None else 1.0
        cc.nm_lin_amp_scale, cc.nm_lin_amp_off = float(nm.lin_amp_scale), float(nm.lin_amp_off)
        cc.nm_mel_amp_scale = float(nm.mel_amp_scale)
        cc.nm_use_pinv, cc.nm_win_norm = int(nm.use_pinv), float(nm.win_norm)
    over = experiment_overrides() if None in (conv_form, batch_invariant, keep_skip, keep_start, tune) else {}
    conv_form = over.get("conv_form", "auto") if conv_form is None else conv_form
    if conv_form not in CONV_FORMS:
        raise ValueError(f"conv_form must be one of {sorted(CONV_FORMS)}")
    cc.wn_conv_form = CONV_FORMS[conv_form]
    cc.batch_invariant = int(over.get("batch_invariant", False) if batch_invariant is None else batch_invariant)
    cc.wn_keep_skip = int(over.get("keep_skip", False) if keep_skip is None else keep_skip)
    cc.wn_keep_start = int(over.get("keep_start", False) if keep_start is None else keep_start)
    cc.calib_fraction = float(calib_fraction or 0.0)
    tune = over.get("tune", {}) if tune is None else tune
    cc.tune_gate_shape = int(tune.get("gate_shape", 0))
    cc.tune_resskip_wave_tiles = int(tune.get("resskip_wave_tiles", 0))
    cc.tune_resskip_split = int(tune.get("resskip_split", 0))
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
    cc.wn_precision = PRECISIONS[precision]
    if f0_accumulate not in F0_ACCUMULATE:
        raise ValueError(f"f0_accumulate must be one of {sorted(F0_ACCUMULATE)}")
    cc.f0_accumulate = F0_ACCUMULATE[f0_accumulate]
    return cc, dims


_WINOGRAD43_G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                          [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)


def pack_winograd4w_weights(w):
    """Winograd F(4,3) combinations U = G W of the three taps (3, C, 2C) of a dilated WaveNet convolution, packed for
    wn_gate_winograd4w_kernel (csrc/wn_winograd4w.hip, v_mfma_f32_16x16x4_f32); formed in float64, stored float32.

    Layout (ceil(C/32) column tiles, ceil(C/8) channel slices, 3072): the 12 KB image of one (tile, slice) is copied
    verbatim into LDS, ordered [product j][channel parity e][lane = 16*kq + n][tanh step 0, tanh step 1, sigmoid step 0,
    sigmoid step 1] with input channel 8*slice + 2*kq + step and output column (0 | C) + 32*tile + 2*n + e;
    out-of-range entries are zero.
    """
    w = np.asarray(w, dtype=np.float64)
    C = w.shape[1]
    assert w.shape == (3, C, 2 * C)
    u = np.einsum("ij,jcn->icn", _WINOGRAD43_G, w)
    nt, nk = (C + 31) // 32, (C + 7) // 8
    wp = np.zeros((6, nk * 8, 2, nt * 32))
    wp[:, :C, 0, :C] = u[:, :, :C]
    wp[:, :C, 1, :C] = u[:, :, C:]
    wp = wp.reshape(6, nk, 4, 2, 2, nt, 16, 2)                    # j, slice, kq, step, tanh|sigmoid, tile, n, e
    return np.ascontiguousarray(wp.transpose(5, 1, 0, 7, 2, 6, 4, 3).reshape(nt, nk, 3072), dtype=np.float32)


def pack_winograd2w_weights(w):
    """Winograd F(2,3) combinations W0, (W0+W1+W2)/2, (W0-W1+W2)/2, W2 of the three taps (3, C, 2C) of a dilated WaveNet
    convolution, packed for wn_gate_winograd2w_kernel (csrc/wn_winograd2w.hip, v_mfma_f32_16x16x4_f32); formed in
    float64, stored float32.

    Layout (ceil(C/32) column tiles, ceil(C/8) channel slices, 2048): the 8 KB image of one (tile, slice) is copied
    verbatim into LDS, ordered [product j][channel parity e][lane = 16*kq + n][tanh step 0, tanh step 1, sigmoid step 0,
    sigmoid step 1] with input channel 8*slice + 2*kq + step and output column (0 | C) + 32*tile + 2*n + e;
    out-of-range entries are zero.
    """
    w = np.asarray(w, dtype=np.float64)
    C = w.shape[1]
    assert w.shape == (3, C, 2 * C)
    u = np.stack((w[0], (w[0] + w[1] + w[2]) / 2, (w[0] - w[1] + w[2]) / 2, w[2]))
    nt, nk = (C + 31) // 32, (C + 7) // 8
    wp = np.zeros((4, nk * 8, 2, nt * 32))
    wp[:, :C, 0, :C] = u[:, :, :C]
    wp[:, :C, 1, :C] = u[:, :, C:]
    wp = wp.reshape(4, nk, 4, 2, 2, nt, 16, 2)                    # j, slice, kq, step, tanh|sigmoid, tile, n, e
    return np.ascontiguousarray(wp.transpose(5, 1, 0, 7, 2, 6, 4, 3).reshape(nt, nk, 2048), dtype=np.float32)


def pack_resskip_weights(w):
    """Weights (1, C, cout) of a WaveNet res/skip 1x1 convolution packed for wn_resskip_kernel (csrc/wn_resskip.hip).

    Layout (ceil(cout/128) column tiles, ceil(C/16) channel slices, 2048): the 8 KB image of one (tile, slice), ordered
    [channel half cc][column sub-tile jn][lane = 32*lk + n][k step st] with input channel 16*slice + 8*cc + 4*lk + st
    and output column 128*tile + 32*jn + n; out-of-range entries are zero.
    """
    w = np.asarray(w, dtype=np.float32)
    assert w.ndim == 3 and w.shape[0] == 1
    C, cout = w.shape[1], w.shape[2]
    nct, nk = (cout + 127) // 128, (C + 15) // 16
    wp = np.zeros((nk * 16, nct * 128), dtype=np.float32)
    wp[:C, :cout] = w[0]
    wp = wp.reshape(nk, 2, 2, 4, nct, 4, 32)                      # slice, cc, lk, st, tile, jn, n
    return np.ascontiguousarray(wp.transpose(4, 0, 1, 5, 2, 6, 3).reshape(nct, nk, 2048))


def pack_resskip_wide_weights(w):
    """Weights (1, K, cout) of a WaveNet res/skip 1x1 convolution packed for wn_resskip_wide_kernel
    (csrc/wn_resskip_wide.hip, v_mfma_f32_16x16x4_f32).

    Layout (ceil(K/8) channel slices, NP = ceil(cout/32) column tile pairs, 256): [pair p][lane = 16*kq + n][even tile
    step 0, even tile step 1, odd tile step 0, odd tile step 1] with input channel 8*slice + 2*kq + step and output
    column 32*p + 2*n + (0 even | 1 odd); out-of-range entries are zero.
    """
    w = np.asarray(w, dtype=np.float32)
    assert w.ndim == 3 and w.shape[0] == 1
    K, cout = w.shape[1], w.shape[2]
    nk, npair = (K + 7) // 8, (cout + 31) // 32
    wp = np.zeros((nk * 8, npair * 32), dtype=np.float32)
    wp[:K, :cout] = w[0]
    wp = wp.reshape(nk, 4, 2, npair, 16, 2)                        # slice, kq, step, pair, n, parity
    return np.ascontiguousarray(wp.transpose(0, 3, 1, 4, 5, 2).reshape(nk, npair, 256))


def pack_resskip_wave_weights(w):
    """Weights (1, K, cout <= 384) of a WaveNet res/skip 1x1 convolution packed for wn_resskip_wave_kernel
    (csrc/wn_resskip_wave.hip, v_mfma_f32_16x16x4_f32, 16-channel slices).

    Layout (ceil(K/16) channel slices, 12 column tile pairs, 512): [pair p][even tile | odd tile][lane = 16*kq + n][step]
    with input channel 16*slice + 4*kq + step and output column 32*p + 2*n + (0 even | 1 odd); out-of-range entries are
    zero (the image always has 12 pairs, so that every column split of the kernel finds its pairs).
    """
    w = np.asarray(w, dtype=np.float32)
    assert w.ndim == 3 and w.shape[0] == 1 and w.shape[2] <= 384
    K, cout = w.shape[1], w.shape[2]
    nk = (K + 15) // 16
    wp = np.zeros((nk * 16, 12 * 32), dtype=np.float32)
    wp[:K, :cout] = w[0]
    wp = wp.reshape(nk, 4, 4, 12, 16, 2)                           # slice, kq, step, pair, n, parity
    return np.ascontiguousarray(wp.transpose(0, 3, 5, 1, 4, 2).reshape(nk, 12, 512))


def pack_resskip_f16_weights(w):
    """Weights (1, K, cout <= 384) of a folded WaveNet res/skip layer for wn_resskip_f16_kernel (csrc/wn_resskip_f16.hip,
    v_mfma_f32_16x16x32_f16; opt-in split half precision): every float32 weight is split into hi = fp16(w) and
    lo' = fp16((w - hi) * 2^11).

    Layout (ceil(K/32) steps, 12 column tile pairs, 1024 float32 words = 4 images x 64 lanes x 8 halves): images [even tile
    hi | even tile lo' | odd tile hi | odd tile lo'], lane = 16 kq + n holds the input channels 32 step + 4 kq .. + 3 and
    32 step + 16 + 4 kq .. + 3 of output column 32 p + 2 n + (0 even | 1 odd); out-of-range entries are zero.  Returned as
    float32 words (two halves each) because the tensor table of the C ABI is float32; the bits are what counts.
    """
    w = np.asarray(w, dtype=np.float32)
    assert w.ndim == 3 and w.shape[0] == 1 and w.shape[2] <= 384
    K, cout = w.shape[1], w.shape[2]
    nk = (K + 31) // 32
    wp = np.zeros((nk * 32, 384), dtype=np.float32)
    wp[:K, :cout] = w[0]
    if not np.all(np.abs(wp) < 65000.0):
        raise ValueError("res/skip weights outside fp16's range: split half precision is not available for this model")
    hi = wp.astype(np.float16)
    lo = ((wp - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    both = np.stack((hi, lo))                                      # (part, channel, column)
    both = both.reshape(2, nk, 2, 4, 4, 12, 16, 2)                 # part, step, half (0 | +16), kq, v, pair, n, parity
    img = both.transpose(1, 5, 7, 0, 3, 6, 2, 4)                   # step, pair, parity, part, kq, n, half, v
    img = np.ascontiguousarray(img).reshape(nk, 12, 4, 64, 8)      # image = 2 parity + part ; lane = 16 kq + n ; 8 halves
    return np.ascontiguousarray(img).view(np.float32).reshape(nk, 12, 1024)


def pack_gate_f16_weights(w):
    """The three taps (3, C, 2C) of a dilated WaveNet convolution for wn_gate_f16_kernel (csrc/wn_gate_f16.hip,
    v_mfma_f32_16x16x32_f16; opt-in split half precision): hi = fp16(w), lo' = fp16((w - hi) * 2^11).

    Layout (ceil(C/32) column tiles, ceil(C/32) steps, 6144 float32 words): [tap][tile c = 2 e + (0 tanh | 1 sigmoid)]
    [hi | lo'][lane = 16 kq + n][8 halves], input channels 32 step + 8 kq .. + 7, output column (0 | C) + 32 tile block + 2 n
    + e; out-of-range entries are zero.  float32 words hold two halves each."""
    w = np.asarray(w, dtype=np.float32)
    C = w.shape[1]
    assert w.shape == (3, C, 2 * C)
    nt = nk = (C + 31) // 32
    wp = np.zeros((3, nk * 32, 2, nt * 32), dtype=np.float32)          # tap, channel, tanh|sigmoid, gate channel
    wp[:, :C, 0, :C] = w[:, :, :C]
    wp[:, :C, 1, :C] = w[:, :, C:]
    if not np.all(np.abs(wp) < 65000.0):
        raise ValueError("gate weights outside fp16's range: split half precision is not available for this model")
    hi = wp.astype(np.float16)
    lo = ((wp - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    both = np.stack((hi, lo))                                          # part, tap, channel, s, gate channel
    both = both.reshape(2, 3, nk, 4, 8, 2, nt, 16, 2)                  # part, tap, step, kq, v, s, block, n, e
    img = both.transpose(6, 2, 1, 8, 5, 0, 3, 7, 4)                    # block, step, tap, e, s, part, kq, n, v
    img = np.ascontiguousarray(img).reshape(nt, nk, 3 * 4 * 2 * 64 * 8)
    return np.ascontiguousarray(img).view(np.float32).reshape(nt, nk, 6144)


def pack_end_weights(w):
    """Weights (1, C, n_out <= 32) of the WaveNet end convolution packed for wn_tail_kernel (csrc/wn_tail.hip):
    (ceil(C/8), 2, 32, 4) = [channel group c][lane half lk][column n][k step st] with input channel 8c + 4lk + st,
    zero padded to 32 columns and to a multiple of 8 channels."""
    w = np.asarray(w, dtype=np.float32)
    assert w.ndim == 3 and w.shape[0] == 1 and w.shape[2] <= 32
    C, n_out = w.shape[1], w.shape[2]
    nc8 = (C + 7) // 8
    wp = np.zeros((nc8 * 8, 32), dtype=np.float32)
    wp[:C, :n_out] = w[0]
    return np.ascontiguousarray(wp.reshape(nc8, 2, 4, 32).transpose(0, 1, 3, 2))


def fold_skip_weights(folded, n_layers, channels, split_f16=False):
    """Fold the skip path of the WaveNet into its end convolution (both are linear, reference
    MBExWN_NVoc/vocoder/model/custom_AE_layers.py:322-341: output += res_skip[:, C:]; ...; self.end(output)):

        end(sum_l (a_l Ws_l + bs_l)) = sum_l a_l (Ws_l We) + (sum_l bs_l We + be)

    so layer l < L-1 needs the C x (C + n_out) matrix [Wr_l | Ws_l We] instead of C x 2C, the last layer C x n_out
    instead of C x C, and the (rows, C) skip tensor never exists.  Products are formed in float64.  Returns the extra
    tensors {name: array} (packed for wn_resskip_kernel / wn_tail_kernel) or {} if the layout does not allow it.
    """
    we = np.asarray(folded["wn.end.w"], dtype=np.float64)
    if we.shape[0] != 1 or we.shape[2] > 32:
        return {}
    C, n_out = channels, we.shape[2]
    we = we[0]
    const = np.asarray(folded["wn.end.b"], dtype=np.float64).copy()
    out = {}
    biases = []
    for ll in range(n_layers):
        w = np.asarray(folded[f"wn.res_skip_{ll}.w"], dtype=np.float64)
        b = np.asarray(folded[f"wn.res_skip_{ll}.b"], dtype=np.float64)
        if w.shape[0] != 1:
            return {}
        last = ll == n_layers - 1
        ws, bs = (w[0], b) if last else (w[0][:, C:], b[C:])
        proj = ws @ we
        const += bs @ we
        if last:
            out["wn.tail.fold"] = pack_end_weights(proj[None])
        else:
            out[f"wn.res_skip_{ll}.fold"] = pack_resskip_weights(np.concatenate((w[0][:, :C], proj), axis=1)[None])
            out[f"wn.res_skip_{ll}.fold_wide"] = pack_resskip_wide_weights(np.concatenate((w[0][:, :C], proj), axis=1)[None])
            if C + n_out <= 384:
                out[f"wn.res_skip_{ll}.fold_wave"] = pack_resskip_wave_weights(np.concatenate((w[0][:, :C], proj), axis=1)[None])
                if ll >= 1 and split_f16:
                    out[f"wn.res_skip_{ll}.fold_f16"] = pack_resskip_f16_weights(np.concatenate((w[0][:, :C], proj), axis=1)[None])
            if ll == 0:
                out["__proj_0"] = proj                 # for fold_start_weights; not a device tensor
            biases.append(np.concatenate((b[:C], np.zeros(n_out))))
    if biases:
        biases[0][C:] = const                 # layer 0 initialises the accumulator
        for ll, bb in enumerate(biases):
            out[f"wn.res_skip_{ll}.fold_b"] = bb
        out["wn.tail.fold_b"] = np.zeros(n_out)
    else:
        out["wn.tail.fold_b"] = const
    return out


def fold_start_weights(folded, dims, fold_skip, split_f16=False):
    """Fold the WaveNet's start convolution (1x1, reference custom_AE_layers.py:177-182,280) into layer 0.

    With x' = [x | 1 | 0] (8 channels; x = pulse channels (+ noise), the constant channel carries the start bias) and
    Ws' = [Ws ; bs ; 0], h0 = x' Ws' and conv0(h0) = sum_tau x'[t + (tau-1) d] (Ws' W0_tau): the dilated convolution of
    layer 0 becomes a K = 24 contraction of the excitation itself (csrc/wn_gate0.hip), and its res/skip layer obtains
    h0 by contracting [a0 | x'] with [Wr ; Ws'] (csrc/wn_resskip.hip, h_init).  Products are formed in float64.
    Returns {name: array}: the (ceil(C/32), 3, 2, 64, 4) image of the tap products -- [tap][channel parity e][lane =
    16*kq + n][tanh step 0, tanh step 1, sigmoid step 0, sigmoid step 1], input channel 2*kq + step, output column
    (0 | C) + 32*tile + 2*n + e -- and the K-extended res/skip image; {} if the layout does not allow it.
    """
    C, L = dims.wn_channels, dims.wn_layers
    ws = np.asarray(folded["wn.start.w"], dtype=np.float64)
    w0 = np.asarray(folded["wn.conv1D_0.w"], dtype=np.float64)
    cin = dims.wn_in_channels
    if ws.shape != (1, cin, C) or w0.shape != (3, C, 2 * C) or cin + 1 > 8 or dims.pulse_channels_eff + 2 > 8:
        return {}
    wsp = np.zeros((8, C))
    wsp[:cin] = ws[0]
    wsp[dims.pulse_channels_eff + 1] = np.asarray(folded["wn.start.b"], dtype=np.float64)   # the constant channel
    prod = np.einsum("kc,tcn->tkn", wsp, w0)                       # (3, 8, 2C)
    nt = (C + 31) // 32
    wp = np.zeros((3, 8, 2, nt * 32))
    wp[:, :, 0, :C] = prod[:, :, :C]
    wp[:, :, 1, :C] = prod[:, :, C:]
    wp = wp.reshape(3, 4, 2, 2, nt, 16, 2)                         # tap, kq, step, tanh|sigmoid, tile, n, e
    out = {"wn.conv1D_0.start_fold": np.ascontiguousarray(wp.transpose(4, 0, 6, 1, 5, 3, 2).reshape(nt, 3, 512))}
    if L > 1:
        if not fold_skip or "wn.res_skip_0.fold" not in fold_skip:
            return {}
        n_out = dims.wn_out_channels
        wr = np.asarray(folded["wn.res_skip_0.w"], dtype=np.float64)[0]
        proj = np.asarray(fold_skip["__proj_0"], dtype=np.float64)    # Ws_0 We (C, n_out)
        ext = np.zeros((C + 16, C + n_out))
        ext[:C, :C] = wr[:, :C]
        ext[:C, C:] = proj
        ext[C:C + 8, :C] = wsp
        out["wn.res_skip_0.fold_start"] = pack_resskip_weights(ext[None])
        out["wn.res_skip_0.fold_start_wide"] = pack_resskip_wide_weights(ext[None])
        if split_f16 and C + n_out <= 384:
            out["wn.res_skip_0.fold_start_f16"] = pack_resskip_f16_weights(ext[None])
        if C + n_out <= 384:
            out["wn.res_skip_0.fold_start_wave"] = pack_resskip_wave_weights(ext[None])
    return out


def tensor_table(config, raw_weights, wavetables, split_f16=False):
    """name -> float32 array of everything mbx_create needs: folded weights + constant tables (+ with ``split_f16`` the
    fp16-split images of the res/skip layers for the opt-in precision mode)."""
    dims = ModelDims(config)
    mb = config["mbexwn_config"]
    mbc = mb["multi_band_config"]
    wn_norm = mb.get("pp_mod_subnet", {}).get("use_weight_norm", None)
    out = dict(merge_channel_groups(fold_weights(raw_weights, wavenet_weight_norm=wn_norm,
                                                 wavenet_equalized_lr=dims.wn_equalized_lr), dims))
    # F0-net: the exact (float64) weight-norm fold of every layer, handed over as pairs of float32 words
    # (mbx_config.f0_accumulate = MBX_F0_ACC_F64: csrc/conv_mfma.hip::conv1d_f64_tile reads them as doubles)
    from .weights import fold_weight_norm_f64
    for key in raw_weights:
        if key.startswith("PulsPar_Layer_") and key.endswith(".v") and key[:-2] + ".g" in raw_weights:
            w64 = np.ascontiguousarray(fold_weight_norm_f64(raw_weights[key], raw_weights[key[:-2] + ".g"]))
            out[key[:-2] + ".w64"] = w64.view(np.float32).reshape(w64.shape + (2,))
    _, syn = tb.pqmf_filters(int(mbc["subbands"]), int(mbc["taps"]), float(mbc["cutoff_ratio"]), float(mbc["beta"]),
                             mbc.get("max_band", None))
    out["table.pqmf_syn"] = syn
    if dims.pulse_pqmf:                                            # analysis bank in front of the WaveNet
        pq = dims.pulse_pqmf
        out["table.pulse_ana"] = tb.pqmf_filters(int(pq["subbands"]), int(pq["taps"]), float(pq["cutoff_ratio"]),
                                                 float(pq["beta"]), pq.get("max_band", None))[0]
    # (several WaveNet blocks / in-block upsampling run the library's generic kernels: no operand-order images)
    if not dims.wn_multi and out["wn.end.w"].shape[0] == 1 and out["wn.end.w"].shape[2] <= 32:
        out["wn.end.packed"] = pack_end_weights(out["wn.end.w"])
        fs = fold_skip_weights(out, dims.wn_layers, dims.wn_channels, split_f16=split_f16)
        if dims.wn_kernel_size == 3:
            out.update(fold_start_weights(out, dims, fs, split_f16=split_f16))
        fs.pop("__proj_0", None)
        out.update(fs)
    for ll in range(0 if dims.wn_multi else dims.wn_layers):
        out[f"wn.res_skip_{ll}.packed"] = pack_resskip_weights(out[f"wn.res_skip_{ll}.w"])
    if dims.wn_kernel_size == 3 and not dims.wn_multi:
        for ll in range(dims.wn_layers):
            out[f"wn.conv1D_{ll}.wino4w"] = pack_winograd4w_weights(out[f"wn.conv1D_{ll}.w"])
            out[f"wn.conv1D_{ll}.wino2w"] = pack_winograd2w_weights(out[f"wn.conv1D_{ll}.w"])
            if split_f16 and ll >= 1:
                out[f"wn.conv1D_{ll}.gate_f16"] = pack_gate_f16_weights(out[f"wn.conv1D_{ll}.w"])
    if dims.wn_multi:
        # several blocks: the library's block runner takes the packed res/skip weights and the F(4,3) images per block
        from .weights import block_prefix
        for bb in range(dims.n_wn_blocks):
            pre = block_prefix(bb)
            for ll in range(dims.wn_layers):
                out[f"{pre}res_skip_{ll}.packed"] = pack_resskip_weights(out[f"{pre}res_skip_{ll}.w"])
                if dims.wn_kernel_size == 3 and dims.wn_padding == "SAME":
                    out[f"{pre}conv1D_{ll}.wino4w"] = pack_winograd4w_weights(out[f"{pre}conv1D_{ll}.w"])
    out["table.hann"] = tb.hann_periodic_f32(dims.stft_win)
    out["table.inv_win"] = tb.inverse_stft_window_f32(dims.stft_win, dims.hop_size)
    out["table.wavetables"] = np.ascontiguousarray(wavetables.tables, dtype=np.float32)
    if dims.ps_env_order_scale and not mb.get("psns_use_cepstral_loss_constraint", False) and not dims.no_envelope:
        logs, rows = tb.cepstral_windows(dims.ps_env_order_scale, dims.sample_rate, dims.f0_min, dims.f0_max, dims.n_ceps)
        out["table.ceps_windows"] = rows
        out["table.ceps_log10f0"] = logs
        out["table.f0_smooth"] = tb.f0_smoothing_kernel(dims.hop_size)
    if mb.get("normalize_rms_from_mell", False):
        from .norm_mel import NormMel
        nm = NormMel(config)
        out["table.nm_inv_enorm"] = nm.inv_enorm
        out["table.nm_gwin"] = nm.gwin
        out["table.nm_smooth_win"] = nm.smooth_syn_win
        if nm.use_pinv:
            out["table.nm_pinv"] = nm.pinv
    return {kk: np.ascontiguousarray(vv, dtype=np.float32) for kk, vv in out.items()}


# ------------------------------------------------------------------------------------------------
# the engine
# ------------------------------------------------------------------------------------------------
class MBExWNEngine:
    """Device-resident MBExWN generator. One instance per GPU (one process per GPU)."""

    def __init__(self, config, raw_weights, wavetables=None, device=None, weight_images=True, conv_form=None,
                 batch_invariant=None, keep_skip=None, keep_start=None, calib_fraction=None, tune=None, precision="f32",
                 f0_accumulate="f64"):
        """``weight_images=False`` hands mbx_create only the folded weights and the tables (what a minimal binding of the
        C ABI would do): the engine then runs its generic kernels instead of the specialised ones.

        ``conv_form`` = "auto" (default: Winograd F(4,3) when a calibration forward on this model's own weights stays
        within a quarter of the parity budget of the direct form, else F(2,3), else the direct form -- see
        :meth:`conv_form_info`, :meth:`calibrate`), "direct", "f23" or "f43"; ``batch_invariant=True`` pins the kernels so
        that an utterance's bits do not depend on the batch it ran in; ``keep_skip`` / ``keep_start`` keep the un-folded
        graph; ``tune`` holds measurement knobs (make_config).  ``precision="split_f16"`` is an opt-in experiment (never
        the default): the res/skip layers (and the gate layers behind the first one) contract on the 16-bit matrix pipe with fp16-split operands
        (three products, float32 accumulation; csrc/wn_resskip_f16.hip).  ``f0_accumulate`` = "f64" (default: the F0-net's
        contractions accumulate in float64 and round once -- its contour feeds the phase integrator) or "f32" (the float32
        kernels of the other mel-rate sub-nets: the behaviour up to ABI 8, kept for A/B measurements)."""
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("MBExWNEngine needs an AMD GPU (no CPU fallback for the mel-inversion path)")
        self._torch = torch
        self._lib = load_library()
        self.config = config
        if wavetables is None:
            dims = ModelDims(config)
            wavetables = tb.WaveTables(sample_rate=dims.pulse_rate, **config["mbexwn_config"]["wavetable_config"])
        self.wavetables = wavetables
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        cconf, self.dims = make_config(config, wavetables, conv_form=conv_form, batch_invariant=batch_invariant,
                                       keep_skip=keep_skip, keep_start=keep_start, calib_fraction=calib_fraction, tune=tune,
                                       precision=precision, f0_accumulate=f0_accumulate)
        self._tune_gate_shape = cconf.tune_gate_shape
        self.normalizes_rms = cconf.nm_iters > 0            # row A14: done on the device inside mbx_forward
        self._tensors = tensor_table(config, raw_weights, wavetables, split_f16=precision == "split_f16")   # keep the host arrays alive
        if not weight_images:
            self._tensors = {kk: vv for kk, vv in self._tensors.items()
                             if kk.startswith("table.") or kk.rsplit(".", 1)[-1] in ("w", "b", "alpha", "w64")}
        arr = (mbx_tensor * len(self._tensors))()
        for ii, (name, val) in enumerate(self._tensors.items()):
            arr[ii].name = name.encode()
            arr[ii].data = val.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
            arr[ii].ndim = val.ndim
            for dd in range(val.ndim):
                arr[ii].shape[dd] = val.shape[dd]
        handle = ctypes.c_void_p()
        _check(self._lib.mbx_create(ctypes.byref(cconf), arr, len(self._tensors), self.device.index,
                                    ctypes.byref(handle)))
        self._handle = handle
        self._workspace = None
        self._last_shape = None

    def close(self):
        if getattr(self, "_handle", None):
            self._lib.mbx_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- properties the reference's model object exposes to MELInverter / scripts
    @property
    def sample_rate(self):
        return self.dims.sample_rate

    @property
    def spect_hop_size(self):
        return self.dims.hop_size

    @property
    def mel_channels(self):
        return self.dims.mel_channels

    def _stream(self):
        return ctypes.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def workspace_bytes(self, batch, max_frames):
        return int(self._lib.mbx_workspace_size(self._handle, batch, max_frames))

    def _get_workspace(self, batch, max_frames):
        need = self.workspace_bytes(batch, max_frames)
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = self._torch.empty(need, dtype=self._torch.uint8, device=self.device)
        return self._workspace, need

    def layer_state_info(self):
        """(floats per slot of the per-layer state store, rows between the end of a WaveNet region and its last exact
        sub-band row, smallest number of new rows of a steady tick); floats == 0: this handle cannot carry layer state."""
        ff, rr, mm = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        _check(self._lib.mbx_layer_state_info(self._handle, ctypes.byref(ff), ctypes.byref(rr), ctypes.byref(mm)))
        return ff.value, rr.value, mm.value

    def forward(self, mel, n_frames=None, noise=None, out=None, stream_state=None, active=None, wavenet=None, carry=None,
                layers=None, state_out=None, frontend=None):
        """mel (B,T,80) float32 cuda tensor; n_frames int32 cuda tensor (B,) or None;
        noise (B, T*steps_per_frame) float32 cuda tensor (N(0,1) draw) -> audio (B, T*hop) cuda tensor.

        stream_state: optional int32 cuda tensor (B, 6) holding one ``mbx_stream_state`` per item (see
        streaming.pack_state); the call then returns (audio, state_out) with the carried phase state.
        active: optional (begin_frame, int32 cuda tensor (B,) of frames[, max of them]) with stream_state: the stages from the WaveNet
        on run on that region of the window only (``mbx_forward_options.active_begin / active_frames``); the audio
        outside the region is undefined.  wavenet: optional (begin_frame, int32 tensor (B,)) inner region of the WaveNet;
        carry: optional (store float32 tensor (slots, rows, subbands), int32 tensor (B, 5)) sub-band rows carried between
        ticks (``wn_begin / wn_frames / sub_store / sub_carry``); layers: optional (store float32 tensor (slots, floats),
        int32 tensor (B, 3), new rows) per-layer WaveNet state carried between ticks (``layer_store / layer_carry /
        layer_rows``)."""
        torch = self._torch
        if mel.dim() != 3 or mel.shape[2] != self.dims.mel_channels:
            raise ValueError(f"mel must be (batch, frames, {self.dims.mel_channels})")
        if mel.device != self.device or mel.dtype != torch.float32:
            raise ValueError("mel must be a float32 tensor on the engine's device")
        mel = mel.contiguous()
        B, T = int(mel.shape[0]), int(mel.shape[1])
        if B == 0 or T == 0:
            return torch.zeros((B, T * self.dims.hop_size), dtype=torch.float32, device=self.device)
        if T * self.dims.steps_per_frame >= 1 << 24:
            # mbx_forward's own limit (32-bit row / sample indices inside the kernels), checked before any allocation
            raise NotImplementedError(f"an item may have at most 2^24 - 1 sub-band rows ({(1 << 24) // self.dims.steps_per_frame} "
                                      f"frames); split longer recordings")
        steps = T * self.dims.wn_in_rows_per_frame            # one noise value per row of the (first) WaveNet block
        if self.dims.noise_sigma:
            if noise is None:
                raise ValueError("noise is required (the noise channel is an explicit input, see SURVEY.md F7)")
            if tuple(noise.shape) != (B, steps) or noise.dtype != torch.float32 or noise.device != self.device:
                raise ValueError(f"noise must be float32 ({B}, {steps}) on the engine's device")
            noise = noise.contiguous()
        if n_frames is not None:
            if n_frames.dtype != torch.int32 or tuple(n_frames.shape) != (B,) or n_frames.device != self.device:
                raise ValueError("n_frames must be an int32 tensor of shape (batch,) on the engine's device")
            n_frames = n_frames.contiguous()
        if out is None:
            out = torch.empty((B, T * self.dims.hop_size), dtype=torch.float32, device=self.device)
        ws, need = self._get_workspace(B, T)
        if stream_state is None and (active is not None or wavenet is not None or carry is not None or layers is not None):
            raise ValueError("active / wavenet / carry / layers describe a streaming window: pass stream_state as well")
        if stream_state is not None:
            if stream_state.dtype != torch.int32 or tuple(stream_state.shape) != (B, 6) or stream_state.device != self.device:
                raise ValueError("stream_state must be an int32 tensor of shape (batch, 6) on the engine's device")
            stream_state = stream_state.contiguous()
            if state_out is None:
                state_out = torch.empty_like(stream_state)
            elif (state_out.dtype != torch.int32 or tuple(state_out.shape) != (B, 6) or state_out.device != self.device or
                  not state_out.is_contiguous()):
                raise ValueError("state_out must be a contiguous int32 tensor of shape (batch, 6) on the engine's device")
            if active is None and (wavenet is not None or carry is not None or layers is not None or frontend is not None):
                raise ValueError("wavenet / carry / layers describe regions inside the active one: pass active as well")
            if active is not None:
                a0, act = int(active[0]), active[1]
                if act.dtype != torch.int32 or tuple(act.shape) != (B,) or act.device != self.device or not 0 <= a0 < T:
                    raise ValueError("active = (begin frame inside the window, int32 tensor of shape (batch,) on the device)")
                act = act.contiguous()
                opt = mbx_forward_options()
                opt.struct_size = ctypes.sizeof(mbx_forward_options)
                opt.transposition = 1.0
                opt.state_in, opt.state_out = stream_state.data_ptr(), state_out.data_ptr()
                opt.active_begin, opt.active_frames = a0, act.data_ptr()
                if len(active) > 2:
                    opt.active_max_frames = int(active[2])
                keep = [act]
                if wavenet is not None:
                    w0, wfr = int(wavenet[0]), wavenet[1]
                    if wfr.dtype != torch.int32 or tuple(wfr.shape) != (B,) or wfr.device != self.device or not a0 <= w0 < T:
                        raise ValueError("wavenet = (begin frame inside the active region, int32 tensor (batch,) on the device)")
                    wfr = wfr.contiguous()
                    keep.append(wfr)
                    opt.wn_begin, opt.wn_frames = w0, wfr.data_ptr()
                    if len(wavenet) > 2:
                        opt.wn_max_frames = int(wavenet[2])
                if carry is not None:
                    store, desc = carry
                    if (store.dtype != torch.float32 or store.dim() != 3 or store.shape[2] != self.dims.subbands or
                            store.device != self.device or not store.is_contiguous()):
                        raise ValueError("carry store must be a contiguous float32 tensor (slots, rows, subbands) on the device")
                    if desc.dtype != torch.int32 or tuple(desc.shape) != (B, 5) or desc.device != self.device:
                        raise ValueError("carry descriptors must be an int32 tensor of shape (batch, 5) on the device")
                    desc = desc.contiguous()
                    keep.append(desc)
                    opt.sub_store, opt.sub_store_rows, opt.sub_carry = store.data_ptr(), int(store.shape[1]), desc.data_ptr()
                if layers is not None:
                    lstore, ldesc, lrows = layers
                    if (lstore.dtype != torch.float32 or lstore.dim() != 2 or lstore.device != self.device or
                            not lstore.is_contiguous()):
                        raise ValueError("layer store must be a contiguous float32 tensor (slots, floats) on the device")
                    if ldesc.dtype != torch.int32 or tuple(ldesc.shape) != (B, 3) or ldesc.device != self.device:
                        raise ValueError("layer descriptors must be an int32 tensor of shape (batch, 3) on the device")
                    ldesc = ldesc.contiguous()
                    keep.append(ldesc)
                    opt.layer_store, opt.layer_store_floats = lstore.data_ptr(), int(lstore.shape[1])
                    opt.layer_carry, opt.layer_rows = ldesc.data_ptr(), int(lrows)
                if frontend is not None:
                    ring, fpos, fnew, fmargin = frontend[:4]
                    opt.fe_end_frames = int(frontend[4]) if len(frontend) > 4 else 0
                    if (ring.dtype != torch.float32 or ring.dim() != 3 or ring.device != self.device or not ring.is_contiguous() or
                            ring.shape[2] != self.frontend_frame_floats):
                        raise ValueError("front-end ring must be a contiguous float32 tensor (slots, ring frames, "
                                         f"{self.frontend_frame_floats}) on the device")
                    if fpos.dtype != torch.int32 or tuple(fpos.shape) != (B,) or fpos.device != self.device or carry is None:
                        raise ValueError("front-end positions must be an int32 tensor (batch,) on the device; carry is required")
                    fpos = fpos.contiguous()
                    keep.append(fpos)
                    opt.fe_store, opt.fe_ring_frames, opt.fe_pos = ring.data_ptr(), int(ring.shape[1]), fpos.data_ptr()
                    opt.fe_new_frames, opt.fe_margin_frames = int(fnew), int(fmargin)
                _check(self._lib.mbx_forward_ex(self._handle, mel.data_ptr(),
                                                n_frames.data_ptr() if n_frames is not None else None, B, T,
                                                noise.data_ptr() if noise is not None else None, out.data_ptr(),
                                                ws.data_ptr(), need, ctypes.byref(opt), self._stream()))
                self._last_shape = (B, T)
                return out, state_out
            _check(self._lib.mbx_forward_stream(self._handle, mel.data_ptr(),
                                                n_frames.data_ptr() if n_frames is not None else None, B, T,
                                                noise.data_ptr() if noise is not None else None, out.data_ptr(),
                                                ws.data_ptr(), need, stream_state.data_ptr(), state_out.data_ptr(),
                                                self._stream()))
            self._last_shape = (B, T)
            return out, state_out
        _check(self._lib.mbx_forward(self._handle, mel.data_ptr(),
                                     n_frames.data_ptr() if n_frames is not None else None, B, T,
                                     noise.data_ptr() if noise is not None else None, out.data_ptr(),
                                     ws.data_ptr(), need, self._stream()))
        self._last_shape = (B, T)
        return out

    def window_advance(self, mel_window, mel_new, noise_window=None, noise_new=None):
        """mbx_window_advance: shift the device-resident windows (B, T, mel_channels) / (B, T*steps_per_frame) left by the
        frames of mel_new (B, step, mel_channels) / noise_new (B, step*steps_per_frame) and append those, in place."""
        torch = self._torch
        B, T, step = int(mel_window.shape[0]), int(mel_window.shape[1]), int(mel_new.shape[1])
        for tt in (mel_window, mel_new, noise_window, noise_new):
            if tt is not None and (tt.dtype != torch.float32 or tt.device != self.device or not tt.is_contiguous()):
                raise ValueError("windows and new frames must be contiguous float32 tensors on the engine's device")
        if tuple(mel_new.shape) != (B, step, self.dims.mel_channels) or mel_window.shape[2] != self.dims.mel_channels:
            raise ValueError("mel_new must be (batch, step, mel_channels)")
        if noise_window is not None and (tuple(noise_window.shape) != (B, T * self.dims.steps_per_frame) or
                                         noise_new is None or tuple(noise_new.shape) != (B, step * self.dims.steps_per_frame)):
            raise ValueError("noise_window / noise_new must be (batch, frames * steps_per_frame) / (batch, step * steps_per_frame)")
        _check(self._lib.mbx_window_advance(self._handle, mel_window.data_ptr(), mel_new.data_ptr(),
                                            noise_window.data_ptr() if noise_window is not None else None,
                                            noise_new.data_ptr() if noise_window is not None else None, B, T, step,
                                            self._stream()))

    def shader_clock_under(self, work, seconds=0.02):
        """Shader clock (GHz) the part delivers while ``work()`` -- a callable that enqueues kernels on the engine's stream --
        runs: a one-wave probe (mbx_clock_probe) spins on a second stream for ``seconds`` and counts shader cycles against the
        constant 100 MHz clock.  ``work`` should enqueue at least that much device time.  Synchronises."""
        torch = self._torch
        out = torch.zeros(4, dtype=torch.int64, device=self.device)
        side = torch.cuda.Stream(device=self.device)
        work()                                                    # the chip is busy before the probe starts
        _check(self._lib.mbx_clock_probe(self._handle, out.data_ptr(), int(seconds * 1e8), ctypes.c_void_p(side.cuda_stream)))
        work()
        torch.cuda.synchronize(self.device)
        c0, c1, r0, r1 = (int(vv) for vv in out.cpu().numpy())
        return (c1 - c0) / max(r1 - r0, 1) * 0.1                    # cycles per 10 ns tick -> GHz

    def window_update(self, mel_window, mel_new, noise_window, noise_new, shift, keep):
        """mbx_window_update: frames [shift, shift + keep) of the device-resident windows move to the front, the frames of
        mel_new / noise_new land behind them; one launch."""
        B, T, step = int(mel_window.shape[0]), int(mel_window.shape[1]), int(mel_new.shape[1])
        _check(self._lib.mbx_window_update(self._handle, mel_window.data_ptr(), mel_new.data_ptr(),
                                           noise_window.data_ptr() if noise_window is not None else None,
                                           noise_new.data_ptr() if noise_window is not None else None, B, T, int(shift), int(keep),
                                           step, self._stream()))

    def emit_rows(self, audio, first, count, host_out):
        """mbx_emit_rows: samples [first, first + count) of every row of the device tensor `audio` (B, n) into the pinned
        host tensor host_out (B, count) as one strided copy on the engine's stream."""
        B, n = int(audio.shape[0]), int(audio.shape[1])
        if tuple(host_out.shape) != (B, count) or not host_out.is_contiguous() or not audio.is_contiguous():
            raise ValueError("emit_rows: host_out must be a contiguous (batch, count) tensor, audio contiguous")
        _check(self._lib.mbx_emit_rows(self._handle, audio.data_ptr(), n, B, int(first), int(count), host_out.data_ptr(),
                                       self._stream()))

    @property
    def frontend_carry_supported(self):
        """True when mbx_forward_ex can carry the mel-rate front end between streaming ticks (fe_store): no RMS
        normalisation, an F0-net that ends at the pulse rate, per-frame sizes that are multiples of 4 floats."""
        dd = self.dims
        f0_ops, _ = subnet_ops(self.config)
        factor = 1
        for op in f0_ops:
            factor *= op.get("up", 1) if op["kind"] in ("conv", "lin") else 1
        sizes = (2 * dd.wn_channels * dd.cond_conv_upsampling, dd.n_ceps, dd.pulse_per_frame)
        return not self.normalizes_rms and factor == dd.pulse_per_frame and all(vv % 4 == 0 for vv in sizes)

    @property
    def frontend_frame_floats(self):
        """Floats per mel frame of the carried front end: conditioning rows, cepstrum, F0 contour."""
        dd = self.dims
        return 2 * dd.wn_channels * dd.cond_conv_upsampling + dd.n_ceps + dd.pulse_per_frame

    def profile_enable(self, enabled=True):
        _check(self._lib.mbx_profile_enable(self._handle, 1 if enabled else 0))

    def profile_read(self, kernel):
        """(summed device ms, launches) of the WaveNet GEMM kernel 'gate' or 'res_skip' since the last read."""
        ms, cnt = ctypes.c_double(), ctypes.c_int64()
        _check(self._lib.mbx_profile_read(self._handle, kernel.encode(), ctypes.byref(ms), ctypes.byref(cnt)))
        return ms.value, cnt.value

    def profile_read_launches(self, kernel, capacity=4096):
        """Device ms of every bracketed launch group of a stage since the last read, in launch order (mbx_profile_read_launches);
        "gate": the layers of each forward in order (behind the folded first layer)."""
        buf = (ctypes.c_float * capacity)()
        cnt = ctypes.c_int64()
        _check(self._lib.mbx_profile_read_launches(self._handle, kernel.encode(), buf, capacity, ctypes.byref(cnt)))
        return [float(buf[ii]) for ii in range(min(capacity, cnt.value))]

    def conv_form_info(self):
        """What the handle decided about the dilated convolution (mbx_conv_form): dict with ``requested`` / ``form`` /
        ``stream_form`` ("auto" | "direct" | "f23" | "f43"), ``calibrated`` (0 no, 1 on the built-in synthetic mel at
        creation, 2 on caller data), ``batch_invariant``, ``fold_skip``, ``fold_start``, and the calibration's numbers
        ``err_f43`` / ``err_f23`` (max |audio(form) - audio(direct)|, None: form not available), ``ref_max``,
        ``threshold``; with ``precision="split_f16"`` also ``err_split`` (max |audio(this handle in split precision) -
        audio(float32 direct form)| of the calibration run at creation) and ``split_rejected`` (True: that error was above the
        threshold or not finite, the handle runs float32 after all: ``split_f16_layers`` is 0 then)."""
        info = mbx_conv_form_info()
        info.struct_size = ctypes.sizeof(mbx_conv_form_info)
        _check(self._lib.mbx_conv_form(self._handle, ctypes.byref(info)))
        return {"requested": _CONV_FORM_NAMES[info.requested], "form": _CONV_FORM_NAMES[info.form],
                "stream_form": _CONV_FORM_NAMES[info.stream_form], "calibrated": info.calibrated,
                "batch_invariant": bool(info.batch_invariant), "fold_skip": bool(info.fold_skip),
                "fold_start": bool(info.fold_start), "split_f16_layers": int(info.split_f16_layers),
                "split_f16_gate_layers": int(info.split_f16_gate_layers),
                "err_f43": None if info.err_f43 < 0 else float(info.err_f43),
                "err_f23": None if info.err_f23 < 0 else float(info.err_f23),
                "ref_max": float(info.ref_max), "threshold": float(info.threshold),
                "err_split": None if info.err_split < 0 else float(info.err_split),
                "split_rejected": bool(info.split_rejected), "f0_float64_chain": bool(info.f0_float64_chain),
                "gate_kernels": [GATE_KERNEL_NAMES[info.gate_kernel[ll]] for ll in range(info.n_gate_layers)]}

    def calibrate(self, mel, n_frames=None, noise=None):
        """mbx_calibrate: repeat the form calibration on the caller's own mel batch (device tensors as for
        :meth:`forward`) and adopt its decision; returns :meth:`conv_form_info`.  Synchronises."""
        torch = self._torch
        mel = mel.to(self.device, torch.float32).contiguous()
        B, T = int(mel.shape[0]), int(mel.shape[1])
        if self.dims.noise_sigma and noise is None:
            noise = torch.randn((B, T * self.dims.wn_in_rows_per_frame), device=self.device, dtype=torch.float32)
        if noise is not None:
            noise = noise.to(self.device, torch.float32).contiguous()
        ws, need = self._get_workspace(B, T)
        _check(self._lib.mbx_calibrate(self._handle, mel.data_ptr(), n_frames.data_ptr() if n_frames is not None else None,
                                       B, T, noise.data_ptr() if noise is not None else None, ws.data_ptr(), need,
                                       self._stream()))
        return self.conv_form_info()

    def gate_form(self, batch, max_frames):
        """Which implementation of the dilated convolution a forward of this size runs: the handle's form
        (mbx_conv_form) and, for F(4,3), the block shape the library's launch-size rule picks (csrc/mbx_api.hip: 256-row
        blocks, or 128-row blocks whose waves split the six products where those spread the work clearly more evenly over
        the SIMDs, or product-split blocks of half a column tile where even those load the CUs unevenly; all three give the same
        bits): "direct", "winograd_f23", "winograd_f43", "winograd_f43_psplit" or "winograd_f43_hsplit"."""
        info = self.conv_form_info()
        if info["form"] == "direct":
            return "direct"
        if info["form"] == "f23":
            return "winograd_f23"
        rows = max_frames * self.dims.steps_per_frame
        tiles = (self.dims.wn_channels + 31) // 32
        full_blocks = ((rows + 255) // 256) * batch * tiles
        half_blocks = ((rows + 127) // 128) * batch * tiles
        if info["batch_invariant"]:
            return "winograd_f43"
        if self._tune_gate_shape and full_blocks < 4 * 768:
            return {1: "winograd_f43", 2: "winograd_f43_psplit", 3: "winograd_f43_hsplit"}[self._tune_gate_shape]
        load_full, load_half = (full_blocks + 255) // 256, 0.5 * ((half_blocks + 255) // 256)
        if full_blocks <= 1024 and load_half <= load_full:
            return "winograd_f43_hsplit" if half_blocks <= 256 < 2 * half_blocks else "winograd_f43_psplit"
        return "winograd_f43"

    @property
    def folds_start(self):
        """True when layer 0 runs with the start convolution folded in (csrc/wn_gate0.hip): what mbx_create decided."""
        return self.conv_form_info()["fold_start"]

    def stage(self, name):
        """Intermediate tensor of the last forward (copy), shaped (B, count); see mbx_stage."""
        torch = self._torch
        ptr, cnt, stride = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64()
        status = self._lib.mbx_stage(self._handle, name.encode(), ctypes.byref(ptr), ctypes.byref(cnt), ctypes.byref(stride))
        if status != 0 and name == "wn_hidden":
            # split precision with the fp16 planes as the hidden state (csrc/mbx_api.hip: planes_only): the float32 tensor was
            # not written; rebuild hi + 2^-11 lo' from the planes -- per row [C8 hi halves | C8 lo' halves]
            planes = self.stage("wn_hidden_planes")
            C, c8 = self.dims.wn_channels, (self.dims.wn_channels + 7) // 8 * 8
            halves = planes.view(torch.float16).view(planes.shape[0], -1, 2, c8).float()
            return (halves[:, :, 0, :C] + halves[:, :, 1, :C] / 2048.0).reshape(planes.shape[0], -1)
        _check(status)
        B = self._last_shape[0]
        dtype = torch.int32 if name == "ceps_index" else torch.float32
        ws = self._workspace
        offset = ptr.value - ws.data_ptr()
        flat = ws[offset: offset + B * stride.value * 4].view(dtype)
        return flat.view(B, stride.value)[:, :cnt.value].clone()

    def _prepare(self, spect, synth_length, noise):
        """Common front of infer / infer_components: device mel (last frame repeated when the mel is shorter than
        synth_length, reference wavegen_1d.py:490-491, 537-538) and the noise draw."""
        torch = self._torch
        mel = torch.as_tensor(np.asarray(spect, dtype=np.float32) if not torch.is_tensor(spect) else spect)
        mel = mel.to(self.device, torch.float32)
        if mel.shape[1] * self.dims.hop_size < synth_length:
            mel = torch.cat((mel, mel[:, -1:]), dim=1)
        mel = mel.contiguous()
        if noise is None and self.dims.noise_sigma:
            # the reference draws tf.random.normal here (custom_pulsed_generator.py:905-906)
            noise = torch.randn((mel.shape[0], mel.shape[1] * self.dims.wn_in_rows_per_frame), device=self.device,
                                dtype=torch.float32)
        elif noise is not None:
            noise = torch.as_tensor(noise).to(self.device, torch.float32).contiguous()
        return mel, noise

    def _envelope(self, B, T):
        """Spectral envelope (B, T, fft/2+1) complex64 of the last forward, rebuilt on the host from its cepstrum stage
        (reference custom_pulsed_generator.py:801-855): lifter row, one-sided cepstrum -> rfft -> exp(R tanh(Re) + j Im)."""
        ceps = self.stage("cepstrum").cpu().numpy().reshape(B, T, self.dims.n_ceps)
        if "table.ceps_windows" in self._tensors:
            idx = self.stage("ceps_index").cpu().numpy()
            ceps = ceps * self._tensors["table.ceps_windows"][idx]
        full = np.zeros((B, T, self.dims.fft_size), dtype=np.float32)
        full[:, :, 1:self.dims.n_ceps] = ceps[:, :, 1:]
        spec = np.fft.rfft(full, axis=-1)
        rng = self.dims.filter_max_log_range
        env = np.exp(rng * np.tanh(spec.real) + 1j * spec.imag) if rng else np.exp(spec)
        return env.astype(np.complex64)

    def infer(self, spect, sigma=None, z_in=None, synth_length=0, F0=None, return_F0=False, return_components=False,
              training=False, test_grad=None, noise=None, **_):
        """Keras-model look-alike of PaNWaveNet.infer (reference wavegen_1d.py:483-526):
        spect numpy/torch (B,T,80) -> tensor with .numpy() of shape (B, synth_length).

        As in the reference: ``sigma`` and ``z_in`` are unused; ``F0`` is only used by the training branch of
        ``MBExWN.call`` (custom_pulsed_generator.py:640-663: at inference ``pulse_frequency_ = pulse_frequency``), so it is
        ignored here -- :meth:`infer_components` is the F0-injection path; ``synth_length = 0`` means the model's
        ``segment_length`` (wavegen_1d.py:489).  ``return_F0`` adds the parameter list
        ``[["F0", .], ["PSig", excitation], ["PS", |envelope|]]`` (custom_pulsed_generator.py:756-767, every entry cut to
        ``[:, :synth_length]`` as wavegen_1d.py:512-515 does), ``return_components`` returns the list of signals.
        ``noise`` (B, T*steps_per_frame) is this build's explicit N(0,1) draw of the noise channel."""
        if training or test_grad is not None:
            raise NotImplementedError("infer(): training / test_grad belong to the training graph, not to the "
                                      "mel-inversion path")
        synth_length = int(synth_length) if synth_length else int(self.dims.segment_length)
        if synth_length <= 0:
            raise ValueError("infer(): synth_length is 0 and the model configuration has no segment_length to fall back "
                             "to (reference wavegen_1d.py:489)")
        mel, noise = self._prepare(spect, synth_length, noise)
        # the optional RMS normalisation of the mel input and the matching output gain (reference
        # wavegen_1d.py:493-495, 506-507, row A14) run inside mbx_forward
        audio = self.forward(mel, noise=noise)
        signals = [_HostTensor(audio[:, :synth_length])]
        if not return_F0:
            return signals if return_components else signals[0]
        B, T = int(mel.shape[0]), int(mel.shape[1])
        rate = int(self.dims.sample_rate // self.dims.pulse_rate)
        f0 = self.stage("f0")[:, :audio.shape[1]:rate]               # the reference's own slice (:757)
        params = [["F0", _HostTensor(f0[:, :synth_length])]]
        if not self.dims.no_envelope:     # ps_off / sub-band gains: neither excitation_signal nor source_filter_stft exist (reference :756-767)
            params += [["PSig", _HostTensor(self.stage("excitation")[:, :audio.shape[1]][:, :synth_length])],
                       ["PS", _HostTensor(np.abs(self._envelope(B, T))[:, :synth_length])]]
        return (signals, params) if return_components else (signals[0], params)

    def infer_components(self, spect, synth_length=0, F0=None, transposition_factor=None, noise=None):
        """PaNWaveNet.infer_components (reference wavegen_1d.py:528-557): returns
        (F0 (B, T*pulse_per_frame), excitation (B, T*hop), spectral envelope complex (B, T, fft/2+1), upsampled_rms or
        None) as numpy arrays; ``F0`` may be given (Hz at the pulse rate), ``transposition_factor`` scales it.  As in the
        reference, ``synth_length`` is replaced by ``F0.shape[1]`` when a contour is given and only decides whether the
        last mel frame is repeated and how long ``upsampled_rms`` is.
        Additionally the (transposed) synthesis itself is available as ``self.last_audio`` (device tensor)."""
        torch = self._torch
        if self.dims.no_envelope:
            raise NotImplementedError("infer_components(): a ps_off / ps_use_stft: false model has no spectral envelope "
                                      "(the reference's generate_specenv has no cepstral VTF-net to run)")
        synth_length = int(synth_length) if F0 is None else int(np.asarray(F0).shape[1])
        mel, noise = self._prepare(spect, synth_length, noise)
        hop, ppf = self.dims.hop_size, self.dims.pulse_per_frame
        B, T = int(mel.shape[0]), int(mel.shape[1])
        gain = None
        if self.normalizes_rms:                                      # 4th output: upsampled_rms (reference :539-542)
            gain = self.norm_mel_stage(mel)[1][:, :synth_length].cpu().numpy()
        f0_dev = None
        if F0 is not None:
            f0_np = np.zeros((B, T * ppf), dtype=np.float32)
            src = np.asarray(F0, dtype=np.float32).reshape(B, -1)
            nn = min(src.shape[1], f0_np.shape[1])
            f0_np[:, :nn] = src[:, :nn]
            f0_np[:, nn:] = src[:, -1:]
            f0_dev = torch.as_tensor(f0_np, device=self.device)
        out = torch.empty((B, T * hop), dtype=torch.float32, device=self.device)
        ws, need = self._get_workspace(B, T)
        opt = mbx_forward_options()
        opt.struct_size = ctypes.sizeof(mbx_forward_options)
        opt.transposition = float(transposition_factor) if transposition_factor else 1.0
        opt.f0 = f0_dev.data_ptr() if f0_dev is not None else None
        _check(self._lib.mbx_forward_ex(self._handle, mel.data_ptr(), None, B, T,
                                        noise.data_ptr() if noise is not None else None, out.data_ptr(), ws.data_ptr(),
                                        need, ctypes.byref(opt), self._stream()))
        self._last_shape = (B, T)
        self.last_audio = out
        f0 = self.stage("f0").cpu().numpy()
        exc = self.stage("excitation").cpu().numpy()
        return f0, exc, self._envelope(B, T), gain

    # -- stage entry points (unit parity tests)
    def pqmf_synthesis(self, x):
        torch = self._torch
        x = x.contiguous()
        B, S, M = x.shape
        y = torch.empty((B, S * M), dtype=torch.float32, device=self.device)
        _check(self._lib.mbx_pqmf_synthesis(self._handle, x.data_ptr(), B, S, y.data_ptr(), self._stream()))
        return y

    def conv1d(self, x, w, b=None, alpha=None, dilation=1, pad_l=0, pad_mode=0, f64_accumulate=False):
        """mbx_conv1d, or mbx_conv1d_f64acc with ``f64_accumulate`` (float64 sums, one rounding per output)."""
        torch = self._torch
        x, w = x.contiguous(), w.contiguous()
        B, R, cin = x.shape
        ks, _, cout = w.shape
        y = torch.empty((B, R, cout), dtype=torch.float32, device=self.device)
        fn = self._lib.mbx_conv1d_f64acc if f64_accumulate else self._lib.mbx_conv1d
        _check(fn(self._handle, x.data_ptr(), B, R, cin, w.data_ptr(),
                  b.data_ptr() if b is not None else None,
                  alpha.data_ptr() if alpha is not None else None, ks, cout, dilation, pad_l,
                  pad_mode, y.data_ptr(), self._stream()))
        return y

    def lin_interp(self, x, up):
        torch = self._torch
        x = x.contiguous()
        B, R, C = x.shape
        y = torch.empty((B, R * up, C), dtype=torch.float32, device=self.device)
        _check(self._lib.mbx_lin_interp(self._handle, x.data_ptr(), B, R, C, up, y.data_ptr(), self._stream()))
        return y

    def wavetable(self, f0):
        torch = self._torch
        f0 = f0.contiguous()
        B, N = f0.shape
        nch = 1 + self.dims.wt_subharm                       # pulse + sub-harmonic sinusoid channels per sample
        pulse = torch.empty((B, N, nch) if nch > 1 else (B, N), dtype=torch.float32, device=self.device)
        phase = torch.empty_like(f0)
        scratch = torch.empty(B * (N + N // 1000 + 3), dtype=torch.float32, device=self.device)
        _check(self._lib.mbx_wavetable(self._handle, f0.data_ptr(), B, N, pulse.data_ptr(), phase.data_ptr(),
                                       scratch.data_ptr(), self._stream()))
        return pulse, phase

    def norm_mel_stage(self, mel, n_frames=None):
        """NormMelComponents.normalize_inputs_by_rms on the device: mel (B,T,80) -> (mel' (B,T,80), gain (B,T*hop))."""
        torch = self._torch
        mel = mel.contiguous()
        B, T = int(mel.shape[0]), int(mel.shape[1])
        out = torch.empty_like(mel)
        gain = torch.zeros((B, T * self.dims.hop_size), dtype=torch.float32, device=self.device)
        scratch = torch.empty(2 * B * T, dtype=torch.float32, device=self.device)
        _check(self._lib.mbx_norm_mel(self._handle, mel.data_ptr(), n_frames.data_ptr() if n_frames is not None else None,
                                      B, T, out.data_ptr(), gain.data_ptr(), scratch.data_ptr(), self._stream()))
        return out, gain

    def stft_filter(self, excitation, cepstrum, ceps_index=None):
        torch = self._torch
        excitation, cepstrum = excitation.contiguous(), cepstrum.contiguous()
        B, T = cepstrum.shape[:2]
        audio = torch.empty((B, T * self.dims.hop_size), dtype=torch.float32, device=self.device)
        scratch = torch.empty(B * T * self.dims.stft_win, dtype=torch.float32, device=self.device)
        _check(self._lib.mbx_stft_filter(self._handle, excitation.data_ptr(), cepstrum.data_ptr(),
                                         ceps_index.contiguous().data_ptr() if ceps_index is not None else None,
                                         B, T, audio.data_ptr(), scratch.data_ptr(), self._stream()))
        return audio


class _HostTensor:
    """What ``model.infer(...)`` returns: something with ``.numpy()`` (reference mel_inverter.py:152)."""

    def __init__(self, tensor):
        self.tensor = tensor

    def numpy(self):
        if isinstance(self.tensor, np.ndarray):
            return self.tensor
        return self.tensor.detach().cpu().numpy()

    @property
    def shape(self):
        return tuple(self.tensor.shape)
