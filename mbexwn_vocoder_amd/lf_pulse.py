"""Band-limited Liljencrants-Fant (LF) glottal pulse wavetables (init-time, host, float64).

SURVEY.md section 8(f) rank 3 / A4 table construction.  Restates, in this package's own
structure, what the reference computes at model construction time:

* LF synthesis parameters (alpha, epsilon*ta) by two scalar root solves
  -- reference MBExWN_NVoc/glottis/FglotLFsynthparams.py:13-190
* closed-form spectrum of the LF pulse (flow derivative, or flow)
  -- reference MBExWN_NVoc/glottis/FglotspecLF.py:15-217
* Kaiser low-pass band limitation applied as a magnitude response, inverse rFFT
  -- reference MBExWN_NVoc/vocoder/model/tf_wavetable.py:37-80 (filter), :93-162 (get_LFpulse)
* the log-spaced grid of progressively band-limited tables and their joint normalisation
  -- reference tf_wavetable.py:244-292,309-410

The result is checked against tables captured from the importable reference functions
(tests/golden/reference_constants.npz, groups G3/G4).
"""
import numpy as np
import scipy.optimize as sopt
import scipy.signal as ss

_EPS32 = float(np.finfo(np.float32).eps)
_EPS64 = float(np.finfo(np.float64).eps)


def _bracket_and_solve(fun):
    """Expanding bracket around 0 followed by Brent (reference FglotLFsynthparams.py:91-109,166-183)."""
    lo, hi = 0.0, 0.1
    f0 = fun(0.0)
    if abs(f0) > _EPS64:
        while (f0 * fun(hi) > 0) and (f0 * fun(-hi) > 0):
            lo = hi
            hi += 1
        if fun(-hi) * f0 < 0:
            lo, hi = -lo, -hi
    else:
        lo, hi = -0.1, 0.1
    root = sopt.brentq(fun, lo, hi)
    if root > max(lo, hi):
        raise RuntimeError("LF model: alpha estimate did not converge")
    return root


def lf_synthesis_params(oq, am, ta):
    """Return (alpha, epar, ta) of the LF model for a unit period.

    oq open quotient in ]0,1[, am asymmetry in [0.5,1[, ta return-phase time constant in [0,1-oq].
    """
    if not (_EPS64 < oq < 1 - _EPS64):
        raise RuntimeError("open quotient out of range")
    if not (0.5 <= am < 1 - _EPS64):
        raise RuntimeError("asymetry is out of range")
    if ta < 0 or ta > (1 - oq):
        raise RuntimeError("return phase length(ta) is out of range")
    te = oq
    wg = np.pi / (oq * am)
    cos_wgte = np.cos(wg * te)
    sin_wgte = np.sin(wg * te)

    if ta <= _EPS32:
        # abrupt closure
        alpha = _bracket_and_solve(lambda a: np.exp(a * oq) * (wg * cos_wgte - a * sin_wgte) - wg)
        return alpha, 0.0, 0.0

    if oq > 0.999:
        epar, ta = 0.5, 0.5 * (1 - oq)
    elif ta > 0.99 * (1 - oq):
        epar, ta = 0.0, 1 - oq
    else:
        slope = (te - 1) / ta
        e_left = -np.log(-slope) / slope
        epar = sopt.brentq(lambda e: e - 1 + np.exp(e * slope), e_left, 1.1)

    if epar == 0:
        ret_integral = -ta / 2
    else:
        xx = np.exp(epar / ta * (te - 1))
        ret_integral = (-xx * (ta + epar - te * epar) + ta) / (epar * (-1 + xx))

    def eq_alpha(a):
        return -(-wg * cos_wgte + a * sin_wgte + wg * np.exp(-a * te)) / (a ** 2 + wg ** 2) / sin_wgte + ret_integral

    alpha = _bracket_and_solve(eq_alpha)
    return alpha, epar, ta


def _cis(x):
    return np.cos(x) + 1j * np.sin(x)


def lf_spectrum(f, oq, am, ta, derivative=True, Ee=1.0):
    """Spectrum of one LF pulse period sampled at the normalised frequencies ``f``
    (``f == k`` is harmonic k).  ``derivative`` selects flow derivative (lip radiation included)
    versus plain glottal flow."""
    alpha, epar, ta = lf_synthesis_params(oq, am, ta)
    te = float(oq)
    wg = np.pi / (oq * am)
    w = np.asarray(f, dtype=np.float64) * 2 * np.pi
    e0_half = -0.5 * Ee / (np.exp(alpha * te) * np.sin(wg * te))
    amp = np.exp(alpha * te + np.log(e0_half))
    wg_eps = _EPS64 if (abs(alpha) < _EPS64 and np.min(np.abs(w - wg)) < _EPS64) else 0.0
    # open phase
    spec = ((amp * _cis(te * (wg - w)) - e0_half) / (1j * alpha + (w - wg + wg_eps))
            - (amp * _cis(-te * (w + wg)) - e0_half) / (1j * alpha + (w + wg)))
    # return phase
    if ta != 0:
        nz = np.flatnonzero(w > _EPS64)
        if epar > 0:
            xx = np.exp(epar * (te - 1) / ta)
            shift_te = _cis(-te * w)
            hh = np.ones(w.shape, dtype=np.complex128) * (-1j * (te - 1))
            hh[nz] = (shift_te[nz] - _cis(-w[nz])) / w[nz]
            ret = ((Ee * ta * (1 - xx)) * shift_te + (1j * Ee * epar * xx) * hh) \
                / (w * (1j * ta * (xx - 1)) + epar * (xx - 1))
        else:
            ret = Ee * ta * 0.5 * np.ones(w.shape) + 0j
            ret[nz] = Ee * (1j * ta * w[nz] - 1 + np.exp(-1j * w[nz] * ta)) / (ta * w[nz] ** 2)
            ret = ret * np.exp(-1j * oq * w)
        spec = spec + ret

    if derivative:
        if w[0] == 0:
            spec[0] = 0
    else:
        if w[0] != 0:
            spec = spec / (1j * w)
        else:
            spec[1:] = spec[1:] / (1j * w[1:])
            e0 = -Ee / (np.exp(alpha * oq) * np.sin(wg * oq))
            ex = np.exp(alpha * te)
            opening = e0 * (-2 * alpha * ex * wg * np.cos(wg * te) + alpha ** 2 * ex * np.sin(wg * te)
                            - wg ** 2 * ex * np.sin(wg * te)
                            + wg * te * alpha ** 2 + wg ** 3 * te + 2 * alpha * wg) / (alpha ** 2 + wg ** 2) ** 2
            if ta > 0:
                eps_ = epar / ta
                xe = np.exp(eps_ * (-1 + te))
                closing = -0.5 * Ee * ta ** 2 * (xe * (2 + eps_ ** 2 + 2 * eps_ + (eps_ * te) ** 2
                                                        - 2 * eps_ * te - 2 * eps_ ** 2 * te) - 2) / (epar ** 3)
            else:
                closing = 0
            spec[0] = opening + closing
    return spec


def pulse_lowpass(pass_band_edge, stop_att_db=70.0, trans_width_normed=0.1):
    """Kaiser-window FIR low-pass; edges relative to the sample rate (Nyquist = 0.5).
    reference tf_wavetable.py:37-80."""
    if stop_att_db >= 50:
        beta = 0.1102 * (stop_att_db - 8.7)
    elif stop_att_db >= 21:
        beta = 0.5842 * (stop_att_db - 21.0) ** 0.4 + 0.07886 * (stop_att_db - 21.0)
    else:
        beta = 0.0
    trans_width = 2 * np.pi * trans_width_normed
    while True:
        radius = int(np.ceil((stop_att_db - 8.0) / 2.285 / trans_width / 2))
        if 2 * radius > 8000 and stop_att_db > 10:
            stop_att_db -= 6
        else:
            break
    return ss.firwin(2 * radius + 1, cutoff=[pass_band_edge - 0.5 * trans_width_normed],
                     window=("kaiser", beta), pass_zero=True, fs=1.0)


def min_phase_spectrum(log_magnitude):
    """Minimum-phase spectrum of a log magnitude through the folded real cepstrum (reference tf_wavetable.py:82-89)."""
    fft_size = log_magnitude.shape[-1] * 2 - 2
    real_cepst = np.fft.irfft(np.fmax(log_magnitude, np.finfo(log_magnitude.dtype).eps), n=fft_size)
    mask = np.concatenate(([1.0], 2 * np.ones(fft_size // 2 - 1), [1.0]), axis=0)
    return np.exp(np.fft.rfft(real_cepst[:mask.shape[0]] * mask, n=fft_size))


def lf_pulse(n_wavetable, oq=0.5, am=0.7, rta=0.1, pul_bw=0.1, use_deriv=False, transition_width=0.1, white_pulse=False):
    """One band-limited LF pulse period of power-of-two length >= n_wavetable.
    reference tf_wavetable.py:93-162; white_pulse (:110-120): the spectrum above its maximum is flattened up to the
    band edge by a minimum-phase filter (the norm option is not used by the model)."""
    fft_size = 16
    while fft_size < n_wavetable:
        fft_size *= 2
    freqs = np.arange(fft_size // 2 + 1) / fft_size
    spec = lf_spectrum(freqs * n_wavetable, oq=oq, am=am, ta=rta * (1 - oq), derivative=use_deriv)
    if white_pulse:
        n_max = int(np.argmax(spec))                              # of the complex spectrum, as the reference takes it
        n_white = int(np.fmax(n_max, int(fft_size * (pul_bw - 0.5 * transition_width))))
        if n_max < n_white:
            wfilt = np.ones(spec.shape)
            wfilt[n_max:n_white] = np.abs(spec[n_max]) / np.abs(spec[n_max:n_white])
            wfilt[n_white:] = np.abs(spec[n_max]) / np.abs(spec[n_white])
            spec = spec * min_phase_spectrum(np.log(wfilt))
    fcoef = pulse_lowpass(pul_bw, stop_att_db=70, trans_width_normed=min(pul_bw / 2.0, transition_width))
    over = 1
    while fcoef.shape[0] > fft_size * over:
        over *= 2
    filt = np.fft.rfft(fcoef, fft_size * over)[::over]
    spec = spec * np.abs(filt)
    return np.fft.irfft(spec, fft_size)


def normed_pulse(Oq, target_nominalF0, nominalBandWidth, sample_rate, am=0.8, rta=0.1, use_radiation=False,
                 bandWidthReductionFactor=1.0, wt_oversampling=1, use_sinusoid=False, use_white_pulse=False):
    """reference tf_wavetable.py:309-410 (create_normed_pulse: LF branch, or one Hann-weighted sine period with
    use_sinusoid, :387-390). Returns (table, realised F0)."""
    if use_sinusoid:
        period = int(wt_oversampling * np.floor(sample_rate / target_nominalF0))
        res = np.sin(np.arange(period) / period * np.pi * 2) * ss.get_window("hann", period, fftbins=True)
        return res, wt_oversampling * sample_rate / period
    res = lf_pulse(int(np.ceil(wt_oversampling * sample_rate / target_nominalF0)), oq=Oq, am=am, rta=rta,
                   pul_bw=nominalBandWidth / (bandWidthReductionFactor * wt_oversampling),
                   transition_width=0.1 / wt_oversampling, use_deriv=use_radiation, white_pulse=use_white_pulse)
    return res, wt_oversampling * sample_rate / res.shape[0]


class WaveTables:
    """The runtime constants of the wavetable oscillator.

    reference tf_wavetable.py:181-306 (PulseWaveTable.__init__): the LF-pulse tables, or the single sine table of
    use_sinusoid / use_sinusoid_as_fun (:239,254-259).  add_subharm_chans and use_sinusoid_as_fun change what the
    oscillator emits, not the tables (config.ModelDims carries them to the kernel); use_white_pulse whitens the LF
    tables (:110-120).  Not built: no_interp, pulse-synchronous gains.
    """

    def __init__(self, sample_rate, nominalF0, Oq=0.5, am=0.8, rta=0.05, use_radiation=False, F0GridFactor=1.25,
                 numF0InGrid=5, maxF0=None, wt_oversampling=2, nominalBandWidth=None, use_sinusoid=False,
                 use_sinusoid_as_fun=False, add_subharm_chans=0, use_white_pulse=False, **unsupported):
        # no_interp: the reference's own branch (tf_wavetable.py:623-629) gathers with batch_dims=1 from the un-batched
        # tables and cannot run; pulse_sync_gain_avg needs the gain inputs of another model family
        for kk in ("no_interp", "pulse_sync_gain_avg"):
            if unsupported.get(kk, False):
                raise NotImplementedError(f"wavetable_config option {kk} is not supported")
        self.add_subharm_chans = int(add_subharm_chans or 0)
        self.use_sinusoid_as_fun = bool(use_sinusoid_as_fun)
        # reference :239,250,275: the tables are built with the constructor argument use_sinusoid; use_sinusoid_as_fun
        # alone keeps the LF tables (they are then not looked up)
        use_sinusoid = bool(use_sinusoid)
        self.sample_rate = float(sample_rate)
        grid = float(F0GridFactor)
        # first pass only to learn which nominal F0 a power-of-two table realises (reference :244-254)
        band = 0.5 / grid
        ref_f0 = maxF0 if maxF0 is not None else nominalF0 * grid ** numF0InGrid
        _, nominal = normed_pulse(Oq, nominalF0, band, sample_rate, am=am, rta=rta, use_radiation=use_radiation,
                                  bandWidthReductionFactor=ref_f0 / nominalF0, wt_oversampling=wt_oversampling,
                                  use_sinusoid=use_sinusoid, use_white_pulse=use_white_pulse)
        self.nominalF0 = float(nominal)
        n_grid = int(numF0InGrid)
        if maxF0 is not None:
            n_grid = int(np.ceil(np.log(maxF0 / self.nominalF0) / np.log(grid)))
        if use_sinusoid:                                           # reference :254-259: one table
            n_grid = 0
        tables = []
        self.F0_list = []
        for ir in range(n_grid + 1):
            rs = grid ** ir if ir > 0 else 1
            tab, _ = normed_pulse(Oq, self.nominalF0, 0.5, sample_rate, am=am, rta=rta, use_radiation=use_radiation,
                                  bandWidthReductionFactor=rs, wt_oversampling=wt_oversampling, use_sinusoid=use_sinusoid,
                                  use_white_pulse=use_white_pulse)
            tab = tab.astype(np.float32)
            self.F0_list.append(self.nominalF0 * rs)
            # first sample appended for the interpolation across the period boundary (reference :278-280)
            tables.append(np.concatenate([tab, tab[0:1]], axis=0)[:, np.newaxis])
        norm_factor = -np.min([tables])
        # (n_period + 1, R) float32, normalised to a minimum of -1 (reference :286-291)
        self.tables = np.concatenate([tt / norm_factor for tt in tables], axis=1).astype(np.float32)
        self.n_period = int(self.tables.shape[0] - 1)
        self.n_tables = int(self.tables.shape[1])
        self.min_transposition = np.float32(np.min(self.F0_list) / self.nominalF0)
        self.max_transposition = np.float32(np.max(self.F0_list) / self.nominalF0)
        self.grid_norm = np.float32(1.0 / np.log(np.float32(grid)))
