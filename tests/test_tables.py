"""Host-side constant tables against constants captured from the importable reference code
(tests/golden/make_reference_constants.py)."""
import os

import numpy as np
import pytest

from mbexwn_vocoder_amd import lf_pulse, tables
from mbexwn_vocoder_amd.config import ModelDims, canonical_config


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "reference_constants.npz"))


@pytest.mark.parametrize("tag", ["mbmelgan4", "canon15"])
def test_pqmf_banks(gold, tag):
    sub, taps, cut, beta = gold[f"pqmf/{tag}/params"]
    proto = tables.pqmf_prototype(int(taps), float(cut), float(beta))
    np.testing.assert_allclose(proto, gold[f"pqmf/{tag}/proto"], rtol=0, atol=1e-15)
    ana, syn = tables.pqmf_filters(int(sub), int(taps), float(cut), float(beta))
    assert np.array_equal(ana, gold[f"pqmf/{tag}/analysis"][:, 0, :])
    assert np.array_equal(syn, gold[f"pqmf/{tag}/synthesis"][:, :, 0])


def test_pqmf_polyphase_equals_zero_stuffing(gold):
    """The polyphase table used by the HIP kernel reproduces zero-stuff + cross-correlation."""
    _, syn = tables.pqmf_filters(15, 120, 0.0421, 9.0)
    G, dm_min = tables.pqmf_polyphase(syn, 15)
    rng = np.random.default_rng(0)
    S, M, taps = 37, 15, 120
    x = rng.normal(size=(S, M))
    up = np.zeros((S * M + taps, M))
    up[taps // 2: taps // 2 + S * M: M] = M * x
    ref = sum(np.correlate(up[:, k], syn[:, k].astype(np.float64), mode="valid") for k in range(M))
    xp = np.zeros((S + 2 * G.shape[1], M))
    off = G.shape[1]
    xp[off:off + S] = x
    got = np.zeros(S * M)
    for q in range(S):
        for p in range(M):
            acc = 0.0
            for ii in range(G.shape[1]):
                acc += np.dot(M * xp[off + q + dm_min + ii], G[p, ii].astype(np.float64))
            got[q * M + p] = acc
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)


def test_pqmf_near_perfect_reconstruction():
    """analysis -> synthesis of the canonical bank is a delay with small ripple (property test)."""
    ana, syn = tables.pqmf_filters(15, 120, 0.0421, 9.0)
    rng = np.random.default_rng(1)
    M, taps, n = 15, 120, 15 * 200
    x = rng.normal(size=n)
    xp = np.pad(x, (taps // 2, taps // 2))
    sub = np.stack([np.correlate(xp, ana[:, k].astype(np.float64), mode="valid")[::M] for k in range(M)], axis=1)
    up = np.zeros((n + taps, M))
    up[taps // 2: taps // 2 + n: M] = M * sub
    y = sum(np.correlate(up[:, k], syn[:, k].astype(np.float64), mode="valid") for k in range(M))
    err = y[300:-300] - x[300:-300]
    assert np.sqrt(np.mean(err ** 2)) / np.sqrt(np.mean(x ** 2)) < 0.05


def test_windows(gold):
    hann = tables.hann_periodic_f32(1200)
    np.testing.assert_allclose(hann, gold["window/tf_hann_periodic_1200"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(hann, 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(1200) / 1200), rtol=0, atol=5e-7)
    inv = tables.inverse_stft_window_f32(1200, 300)
    # a frame that passes the forward and the inverse window overlap-adds to one (4 overlaps)
    tot = (hann * inv).reshape(4, 300).sum(axis=0)
    np.testing.assert_allclose(tot, 1.0, rtol=0, atol=1e-6)


def test_lf_model(gold):
    freqs = gold["lf/freqs"]
    for ii in range(3):
        oq, am, ta, alpha, epar, ta_out = gold[f"lf/{ii}/params"]
        got = lf_pulse.lf_synthesis_params(oq, am, ta)
        np.testing.assert_allclose(got, (alpha, epar, ta_out), rtol=1e-12, atol=1e-14)
        for deriv, key in ((True, "d"), (False, "f")):
            spec = lf_pulse.lf_spectrum(freqs, oq, am, ta, derivative=deriv)
            np.testing.assert_allclose(spec, gold[f"lf/{ii}/spec_{key}"], rtol=1e-11, atol=1e-13)


def test_wavetable_entries(gold):
    for ii in range(3):
        rs, rad, f0 = gold[f"wt/{ii}/params"]
        tab, got_f0 = lf_pulse.normed_pulse(0.5, 31.25, 0.5, 8000.0, am=0.8, rta=0.05, use_radiation=bool(rad),
                                            bandWidthReductionFactor=rs, wt_oversampling=2)
        assert got_f0 == f0
        np.testing.assert_allclose(tab, gold[f"wt/{ii}/table"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(lf_pulse.pulse_lowpass(0.2, 70, 0.05), gold["wt/lowpass_0p2"], rtol=0, atol=1e-16)


def test_full_wavetable_grid(gold):
    cfg = canonical_config("SPEECH")
    dims = ModelDims(cfg)
    wt = tables.WaveTables(sample_rate=dims.pulse_rate, **cfg["mbexwn_config"]["wavetable_config"])
    assert np.array_equal(wt.tables, gold["wt/full/tables"])
    nominal, tmin, tmax, gnorm, n_period = gold["wt/full/consts"]
    assert wt.nominalF0 == nominal == float(gold["wt/adapted_nominalF0"])
    assert float(wt.min_transposition) == tmin and float(wt.max_transposition) == tmax
    assert float(wt.grid_norm) == gnorm and wt.n_period == int(n_period)
    assert wt.tables.shape == (513, 15) and wt.tables.min() == -1.0
    # last row repeats the first (interpolation across the period boundary)
    assert np.array_equal(wt.tables[-1], wt.tables[0])


def test_white_pulse_tables(gold):
    """use_white_pulse (get_LFpulse's white_pulse branch, reference tf_wavetable.py:110-120): single entries and the whole
    table grid against the reference's own output."""
    for ii in range(2):
        rs, rad = gold[f"wt/white/{ii}/params"]
        tab, _ = lf_pulse.normed_pulse(0.5, 31.25, 0.5, 8000.0, am=0.8, rta=0.05, use_radiation=bool(rad),
                                       bandWidthReductionFactor=rs, wt_oversampling=2, use_white_pulse=True)
        np.testing.assert_allclose(tab, gold[f"wt/white/{ii}/table"], rtol=0, atol=1e-14)
        plain, _ = lf_pulse.normed_pulse(0.5, 31.25, 0.5, 8000.0, am=0.8, rta=0.05, use_radiation=bool(rad),
                                         bandWidthReductionFactor=rs, wt_oversampling=2)
        assert np.max(np.abs(plain - tab)) > 1e-3                  # it is a different pulse
    cfg = canonical_config("SPEECH")
    dims = ModelDims(cfg)
    wt = tables.WaveTables(sample_rate=dims.pulse_rate, **dict(cfg["mbexwn_config"]["wavetable_config"], use_white_pulse=True))
    assert np.array_equal(wt.tables, gold["wt/white/full/tables"])


def test_wavetable_options():
    """use_sinusoid: one table = one sine period under a periodic Hann window (reference tf_wavetable.py:254-259,387-390;
    unpinned: that branch calls scipy.signal.hanning, which the scipy of this image no longer has, so the reference cannot
    run it here).  add_subharm_chans / use_sinusoid_as_fun leave the tables alone; the unbuilt options still raise."""
    cfg = canonical_config("SPEECH")
    dims = ModelDims(cfg)
    base = dict(cfg["mbexwn_config"]["wavetable_config"])
    wt = tables.WaveTables(sample_rate=dims.pulse_rate, **dict(base, use_sinusoid=True))
    period = 2 * int(np.floor(dims.pulse_rate / base["nominalF0"]))
    assert wt.tables.shape == (period + 1, 1) and wt.n_tables == 1 and wt.nominalF0 == 2 * dims.pulse_rate / period
    assert float(wt.min_transposition) == float(wt.max_transposition) == 1.0 and wt.tables.min() == -1.0
    want = np.sin(2 * np.pi * np.arange(period) / period) * (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(period) / period))
    np.testing.assert_allclose(wt.tables[:-1, 0], want / -want.min(), rtol=0, atol=1e-6)
    ref = tables.WaveTables(sample_rate=dims.pulse_rate, **base)
    for extra in ({"add_subharm_chans": 2}, {"use_sinusoid_as_fun": True}):
        assert np.array_equal(tables.WaveTables(sample_rate=dims.pulse_rate, **dict(base, **extra)).tables, ref.tables)
    over = canonical_config("SPEECH", **{"mbexwn_config:wavetable_config:add_subharm_chans": 2})
    assert ModelDims(over).wn_in_channels == 5 * 3 + 1 and ModelDims(over).pulse_channels_eff == 15
    for kk in ("no_interp", "pulse_sync_gain_avg"):
        with pytest.raises(NotImplementedError):
            tables.WaveTables(sample_rate=dims.pulse_rate, **dict(base, **{kk: True}))


def test_cepstral_windows_and_smoother():
    logs, rows = tables.cepstral_windows(1.0, 24000, 40.0, 600.0, 240)
    assert rows.shape == (30, 240) and logs.shape == (30,)
    assert np.all(rows[:, 0] == 1.0)            # asserted by the reference at custom_pulsed_generator.py:807
    assert np.all(np.diff(logs) > 0)
    ker = tables.f0_smoothing_kernel(300)
    assert ker.shape == (601,) and abs(float(ker.sum()) - 1.0) < 1e-6 and ker[0] > 0


def test_lin_interp_weights():
    w0, w1 = tables.lin_interp_weights(10)
    np.testing.assert_allclose(w0 + w1, 1.0, atol=1e-7)
    assert w0[0] == 1.0 and w1[0] == 0.0


def test_nextpow2(gold):
    from mbexwn_vocoder_amd.config import ModelDims
    for nn, vv in gold["nextpow2_val"]:
        v = 2
        while v < nn:
            v *= 2
        assert v == vv
    assert ModelDims(canonical_config()).fft_size == 2048


# ---------------------------------------------------------------- analysis side (next row: audio -> mel)
def test_stft_magnitude_matches_reference(gold):
    from mbexwn_vocoder_amd import analysis
    np.testing.assert_allclose(analysis.hann_symmetric(1200), gold["window/hann1200"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(analysis.hann_symmetric(9), gold["window/hann9"], rtol=0, atol=1e-15)
    mag = analysis.stft_magnitude(gold["stft/snd"], 1200, 300, 2048)
    ref = gold["stft/mag_1200_300_2048"]
    assert mag.shape == ref.shape == (2, 11, 1025)
    np.testing.assert_allclose(mag, ref, rtol=0, atol=2e-4)          # float32 FFT of amplitude-40 spectra


def test_slaney_mel_basis_properties():
    from mbexwn_vocoder_amd import analysis
    basis = analysis.mel_basis_slaney(24000, 2048, 80, 0.0, 12000.0)
    assert basis.shape == (80, 1025) and basis.dtype == np.float32 and np.all(basis >= 0)
    freqs = analysis.mel_frequencies(82, 0.0, 12000.0)
    assert freqs[0] == 0.0 and abs(freqs[-1] - 12000.0) < 1e-6 and np.all(np.diff(freqs) > 0)
    # linear below 1 kHz (200/3 Hz per mel step ratio), logarithmic above
    low = freqs[freqs < 900]
    assert np.allclose(np.diff(low), np.diff(low)[0])
    high = freqs[freqs > 1100]
    assert np.allclose(high[1:] / high[:-1], high[1] / high[0])
    # slaney norm: every triangle has (almost) unit area in Hz
    df = 24000 / 2048
    area = basis.sum(axis=1) * df
    assert np.all(np.abs(area[5:] - 1.0) < 0.05)
    peaks = np.argmax(basis, axis=1)
    assert np.all(np.diff(peaks) > 0)


def test_compute_log_mel_shapes():
    from mbexwn_vocoder_amd import analysis
    cfg = canonical_config()["preprocess_config"]
    rng = np.random.default_rng(0)
    mell, rate = analysis.compute_log_mel(rng.normal(size=(7200,)).astype(np.float32), cfg)
    assert mell.shape == (1, 25, 80) and rate == 80.0 and np.all(np.isfinite(mell))
