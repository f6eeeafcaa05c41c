#!/usr/bin/env python3
"""Capture constants and host-side I/O pairs from the importable numpy parts of the reference
(build container only; needs /root/reference).  SURVEY.md section 8(c) groups G1-G6.

Writes tests/golden/reference_constants.npz (+ reference_scale_mel.npz with the scale_mel pairs).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import tf_numpy_shim as shim  # noqa: E402


def main():
    shim.install("/root/reference")
    shim.set_float(np.float32)
    from MBExWN_NVoc.vocoder.model import tf_preprocess, tf_wavetable
    from MBExWN_NVoc.glottis.FglotspecLF import FglotspecLF
    from MBExWN_NVoc.glottis.FglotLFsynthparams import FglotLFsynthparams
    from MBExWN_NVoc.sig_proc.Mwindows import window
    from MBExWN_NVoc.utils import nextpow2_val
    from MBExWN_NVoc import mel_inverter

    out = {}
    # G1 PQMF prototype + banks
    for tag, (sub, taps, cut, beta) in {"mbmelgan4": (4, 62, 0.15, 9.0), "canon15": (15, 120, 0.0421, 9.0)}.items():
        out[f"pqmf/{tag}/params"] = np.asarray([sub, taps, cut, beta], dtype=np.float64)
        out[f"pqmf/{tag}/proto"] = tf_preprocess._design_prototype_filter(taps, cut, beta)
        bank = tf_preprocess.TFPQMF(subbands=sub, taps=taps, cutoff_ratio=cut, beta=beta)
        out[f"pqmf/{tag}/analysis"] = bank.analysis_filter
        out[f"pqmf/{tag}/synthesis"] = bank.synthesis_filter
    # G2 windows
    out["window/hann1200"] = window("hann", 1200)
    out["window/hann9"] = window("hann", 9)
    out["window/tf_hann_periodic_1200"] = np.asarray(shim.tf.signal.hann_window(1200))
    # G3 LF model
    freqs = np.arange(0, 40) * 0.37
    for ii, (oq, am, ta) in enumerate([(0.5, 0.8, 0.025), (0.7, 0.66, 0.09), (0.35, 0.9, 0.0)]):
        alpha, epar, ta_out = FglotLFsynthparams(oq, am, ta)
        out[f"lf/{ii}/params"] = np.asarray([oq, am, ta, alpha, epar, ta_out])
        for deriv in (True, False):
            spec = FglotspecLF(freqs, oq=oq, am=am, ta=ta, get_derivative=deriv)[0]
            out[f"lf/{ii}/spec_{'d' if deriv else 'f'}"] = np.asarray(spec)
    out["lf/freqs"] = freqs
    # G4 wavetable entries
    for ii, (rs, rad) in enumerate([(1.0, True), (1.25 ** 3, True), (1.25 ** 7, False)]):
        tab, f0 = tf_wavetable.PulseWaveTable.create_normed_pulse(
            0.5, target_nominalF0=31.25, nominalBandWidth=0.5, sample_rate=8000.0, am=0.8, rta=0.05,
            use_radiation=rad, bandWidthReductionFactor=rs, wt_oversampling=2, return_nominal_f0=True, quiet=True)
        out[f"wt/{ii}/params"] = np.asarray([rs, float(rad), f0])
        out[f"wt/{ii}/table"] = tab
    tab, f0 = tf_wavetable.PulseWaveTable.create_normed_pulse(
        0.5, target_nominalF0=40.0, nominalBandWidth=0.4, sample_rate=8000.0, am=0.8, rta=0.05, use_radiation=True,
        bandWidthReductionFactor=15.0, wt_oversampling=2, return_nominal_f0=True, quiet=True)
    out["wt/adapted_nominalF0"] = np.asarray(f0)
    # whitened LF pulse (get_LFpulse white_pulse branch, tf_wavetable.py:110-120), single entries and the whole grid
    for ii, (rs, rad) in enumerate([(1.0, True), (1.25 ** 4, False)]):
        out[f"wt/white/{ii}/params"] = np.asarray([rs, float(rad)])
        out[f"wt/white/{ii}/table"] = tf_wavetable.PulseWaveTable.create_normed_pulse(
            0.5, target_nominalF0=31.25, nominalBandWidth=0.5, sample_rate=8000.0, am=0.8, rta=0.05,
            use_radiation=rad, bandWidthReductionFactor=rs, wt_oversampling=2, quiet=True, use_white_pulse=True)
    white = tf_wavetable.PulseWaveTable(sample_rate=8000.0, nominalF0=40.0, maxF0=600.0, F0GridFactor=1.25,
                                        wt_oversampling=2, Oq=0.5, am=0.8, rta=0.05, use_radiation=True, quiet=True,
                                        use_white_pulse=True)
    out["wt/white/full/tables"] = np.asarray(white.wavetables)
    out["wt/lowpass_0p2"] = tf_wavetable.get_pulse_lowpass_kaiser(0.2, stop_att_db=70, trans_width_normed=0.05)
    full = tf_wavetable.PulseWaveTable(sample_rate=8000.0, nominalF0=40.0, maxF0=600.0, F0GridFactor=1.25,
                                       wt_oversampling=2, Oq=0.5, am=0.8, rta=0.05, use_radiation=True, quiet=True)
    out["wt/full/tables"] = np.asarray(full.wavetables)
    out["wt/full/consts"] = np.asarray([full.nominalF0, float(full.minTranspositionFactorInGrid),
                                        float(full.maxTranspositionFactorInGrid), float(full.grid_f0_diff_norm_factor),
                                        full.n_period])
    # analysis side: magnitude STFT of the reference's numpy code (sig_proc/spec/stft.py:14-96)
    from MBExWN_NVoc.sig_proc.spec.stft import calc_stft
    rng2 = np.random.default_rng(11)
    snd = rng2.normal(size=(2, 3000)).astype(np.float32)
    out["stft/snd"] = snd
    out["stft/mag_1200_300_2048"] = calc_stft(snd, win_len=1200, hop_len=300, fft_size=2048, win_type="hann",
                                              center=True, pad_mode="reflect", do_mag=True, axis=-1, dtype=np.float32)
    # G6
    out["nextpow2_val"] = np.asarray([[nn, nextpow2_val(nn)] for nn in (1, 2, 3, 1200, 2048, 2049)])
    np.savez_compressed(os.path.join(HERE, "reference_constants.npz"), **out)

    # G5 scale_mel I/O pairs (host-side numpy; reference mel_inverter.py:48-148)
    inv = mel_inverter.MELInverter(None)
    inv.hop_size, inv._srate, inv.fft_size, inv.fmin, inv.fmax = 300, 24000, 2048, 0.0, 12000.0
    rng = np.random.default_rng(7)
    base = {"nfft": 2048, "hoplen": 300, "winlen": 1200, "nmels": 80, "sr": 24000, "fmin": 0.0, "fmax": 12000.0,
            "lin_spec_offset": 1e-5, "lin_spec_scale": 1, "log_spec_offset": 0.0, "log_spec_scale": 1, "time_axis": 1}
    sm = {}

    def run(tag, cfg_updates, inv_updates):
        dd = dict(base)
        dd.update(cfg_updates)
        dd["mell"] = rng.normal(-5, 2, size=(80, 13)).astype(np.float32)
        for kk, vv in {"lin_amp_scale": 1, "lin_amp_off": 1e-5, "mel_amp_scale": 1, "use_max_limit": False}.items():
            setattr(inv, kk, vv)
        for kk, vv in inv_updates.items():
            setattr(inv, kk, vv)
        sm[f"{tag}/in_mell"] = dd["mell"].copy()
        sm[f"{tag}/out"] = inv.scale_mel(dd)

    run("plain", {}, {})
    run("max_limit", {}, {"use_max_limit": True, "lin_amp_off": 1e-4})
    run("nfft1024", {"nfft": 1024}, {})
    run("hop256", {"hoplen": 256}, {})
    run("scaled", {"lin_spec_scale": 2.0, "log_spec_scale": 0.5, "log_spec_offset": 0.3},
        {"lin_amp_scale": 1.5, "mel_amp_scale": 0.25})
    try:
        dd = dict(base, fmin=50.0)
        dd["mell"] = np.zeros((80, 3), np.float32)
        inv.scale_mel(dd)
        sm["fmin_mismatch_raises"] = np.asarray(0)
    except RuntimeError:
        sm["fmin_mismatch_raises"] = np.asarray(1)
    np.savez_compressed(os.path.join(HERE, "reference_scale_mel.npz"), **sm)
    print("wrote reference_constants.npz / reference_scale_mel.npz")


if __name__ == "__main__":
    main()
