"""Init-time constant tables of the mel-inversion path (host side, numpy).

Each function cites the reference construction it restates; values are checked against
constants captured from the importable parts of the reference (tests/golden).
"""
import numpy as np
import scipy.signal as ss

from .lf_pulse import WaveTables  # noqa: F401  (re-export)


# ---------------------------------------------------------------- PQMF ------------------
def pqmf_prototype(taps=62, cutoff_ratio=0.15, beta=9.0):
    """Kaiser-windowed sinc prototype, length taps+1 (float64).
    reference MBExWN_NVoc/vocoder/model/tf_preprocess.py:30-80 (numpy branch)."""
    if taps % 2 != 0:
        raise AssertionError("The number of taps mush be even number.")
    if not (0.0 < cutoff_ratio < 1.0):
        raise AssertionError("Cutoff ratio must be > 0.0 and < 1.0.")
    nn = np.arange(taps + 1) - 0.5 * taps
    omega_c = np.pi * cutoff_ratio
    with np.errstate(invalid="ignore", divide="ignore"):
        h_i = np.sin(omega_c * nn) / (np.pi * nn)
    h_i[taps // 2] = cutoff_ratio
    return h_i * ss.windows.kaiser(taps + 1, beta)


def pqmf_filters(subbands, taps, cutoff_ratio, beta, max_band=None):
    """Cosine-modulated analysis / synthesis banks, each (taps+1, bands) float32
    (tap index first = the reference's conv kernel layout with the unit channel axis dropped).
    reference tf_preprocess.py:119-161."""
    proto = pqmf_prototype(taps, cutoff_ratio, beta)
    used = max_band if max_band else subbands
    nn = np.arange(taps + 1) - (taps / 2)
    ana = np.zeros((subbands, taps + 1))
    syn = np.zeros((used, taps + 1))
    for kk in range(subbands):
        arg = (2 * kk + 1) * (np.pi / (2 * subbands)) * nn
        ana[kk] = 2 * proto * np.cos(arg + (-1) ** kk * np.pi / 4)
        if kk < used:
            syn[kk] = 2 * proto * np.cos(arg - (-1) ** kk * np.pi / 4)
    return ana.T.astype(np.float32), syn.T.astype(np.float32)


def pqmf_polyphase(syn, subbands):
    """Polyphase view of the synthesis bank for the HIP kernel.

    y[q*M + p] = sum_{dm} sum_k (M * x[q + dm, k]) * G[p, dm - dm_min, k]
    with G[p, i, k] = g_k[(dm)*M + taps/2 - p]  (zero where the tap index leaves [0, taps]).
    Derived from the zero-stuff + cross-correlation form of reference tf_preprocess.py:208-226.
    Returns (G float32 (M, n_dm, bands), dm_min).
    """
    taps = syn.shape[0] - 1
    M = subbands
    half = taps // 2
    dm_min = -((half + M - 1) // M)          # ceil((0 - half)/M) for p = 0 ... conservative bound
    dm_max = (half + M - 1) // M
    n_dm = dm_max - dm_min + 1
    G = np.zeros((M, n_dm, syn.shape[1]), dtype=np.float32)
    for p in range(M):
        for ii in range(n_dm):
            jj = (dm_min + ii) * M + half - p
            if 0 <= jj <= taps:
                G[p, ii, :] = syn[jj, :]
    return G, dm_min


# ---------------------------------------------------------------- STFT windows ----------
def hann_periodic_f32(n):
    """tf.signal.hann_window(n, periodic=True, float32): 0.5 - 0.5 cos(2 pi k / n'), all float32 arithmetic.
    Used by reference custom_pulsed_generator.py:388,692 through tf.signal.stft (third party, TensorFlow)."""
    even = 1 - n % 2
    denom = np.float32(n + even - 1)
    arg = np.float32(2 * np.pi) * np.arange(n, dtype=np.float32) / denom
    return (np.float32(0.5) - np.float32(0.5) * np.cos(arg, dtype=np.float32)).astype(np.float32)


def inverse_stft_window_f32(frame_length, frame_step):
    """tf.signal.inverse_stft_window_fn(frame_step, hann): w / sum_i w^2[(n mod step) + i step].
    Call site reference custom_pulsed_generator.py:716-720."""
    win = hann_periodic_f32(frame_length)
    den = np.square(win)
    overlaps = -(-frame_length // frame_step)
    den = np.pad(den, (0, overlaps * frame_step - frame_length)).astype(np.float32)
    den = den.reshape(overlaps, frame_step).sum(axis=0, keepdims=True, dtype=np.float32)
    den = np.tile(den, (overlaps, 1)).reshape(overlaps * frame_step)
    return (win / den[:frame_length]).astype(np.float32)


# ---------------------------------------------------------------- cepstral lifters ------
def cepstral_windows(scale, sample_rate, f_min, f_max, n_ceps, n_windows=30):
    """(log10 f0 grid float32 (n_windows,), half-Hamming lifter rows float32 (n_windows, n_ceps)).
    reference custom_pulsed_generator.py:434-450."""
    rows, logs = [], []
    for f0 in np.logspace(np.log10(f_min), np.log10(f_max), n_windows):
        win_len = int(scale * 0.5 * sample_rate / f0)
        if (win_len // 2) * 2 == win_len:
            win_len += 1
        logs.append(np.log10(f0))
        half = np.hamming(win_len)[win_len // 2:]
        if win_len // 2 + 1 > n_ceps:
            rows.append(half[:n_ceps])
        else:
            rows.append(np.concatenate((half, np.zeros(n_ceps - 1 - (win_len // 2)))))
    return np.asarray(logs, dtype=np.float32), np.asarray(rows, dtype=np.float32)


def f0_smoothing_kernel(hop_size):
    """Bartlett window without its zero end points, unit sum, float32 (2*hop+1 taps).
    reference custom_pulsed_generator.py:403-406."""
    win = np.bartlett(2 * hop_size + 3)[1:-1]
    return (win / np.sum(win)).astype(np.float32)


# ---------------------------------------------------------------- misc -------------------
def lin_interp_weights(up):
    """(w_cur, w_next) float32 (up,) : (U-u)/U and u/U. reference support_layers.py:19-27."""
    uu = np.arange(up)
    return ((up - uu) / up).astype(np.float32), (uu / up).astype(np.float32)


def fft_twiddles(n):
    """exp(-2 pi i k / n), k < n/2, as float32 (n/2, 2) computed in float64."""
    kk = np.arange(n // 2)
    ang = -2.0 * np.pi * kk / n
    return np.stack((np.cos(ang), np.sin(ang)), axis=1).astype(np.float32)
