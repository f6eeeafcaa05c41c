"""Independent check of the TensorFlow-op stand-in that the reference-generated goldens run on.

tests/golden/tf_numpy_shim.py restates the TF kernels the reference calls (Keras Conv1D / tf.nn.conv1d with SAME and
dilation, tf.nn.depthwise_conv2d, tf.nn.conv1d_transpose, tf.signal.stft / inverse_stft / inverse_stft_window_fn /
overlap_and_add / hann_window) from their documentation; the oracle restates the same ops from the same reading.
This file compares the stand-in with a third implementation written by other people: torch-CPU (F.conv1d,
F.conv_transpose1d, torch.stft / torch.istft, torch.hann_window) on random inputs in float64, <= 1e-6 (most agree to
1e-12).  It does not pin TensorFlow, but it removes the single-author risk on the arithmetic.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

import tf_numpy_shim as shim  # noqa: E402

TOL = 1e-6


@pytest.fixture(autouse=True)
def float64_shim():
    shim.set_float(np.float64)
    yield
    shim.set_float(np.float32)


def _maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))))


@pytest.mark.parametrize("T,cin,cout,K,dil", [(37, 5, 7, 3, 1), (64, 8, 4, 3, 4), (50, 3, 6, 3, 16), (21, 4, 4, 1, 1),
                                              (30, 6, 5, 5, 2), (9, 2, 3, 4, 1)])
def test_conv1d_same_dilated(T, cin, cout, K, dil):
    """Keras Conv1D(padding="same", dilation_rate=d) / tf.nn.conv1d(SAME): cross-correlation, pad total = (K-1)d split
    floor/ceil (the extra sample goes to the right for even K)."""
    rng = np.random.default_rng(T + K)
    x = rng.normal(size=(2, T, cin))
    w = rng.normal(size=(K, cin, cout))
    got = shim.conv1d_nwc(x, w, padding="SAME", dilation=dil)
    total = (K - 1) * dil
    xt = F.pad(torch.as_tensor(x).permute(0, 2, 1), (total // 2, total - total // 2))
    ref = F.conv1d(xt, torch.as_tensor(w).permute(2, 1, 0), dilation=dil).permute(0, 2, 1).numpy()
    assert got.shape == ref.shape == (2, T, cout)
    assert _maxdiff(got, ref) <= TOL
    # VALID and strided VALID (the NormMelComponents smoothing uses stride = hop)
    got = shim.conv1d_nwc(x, w, stride=3, padding="VALID", dilation=dil)
    ref = F.conv1d(torch.as_tensor(x).permute(0, 2, 1), torch.as_tensor(w).permute(2, 1, 0), stride=3,
                   dilation=dil).permute(0, 2, 1).numpy()
    assert got.shape == ref.shape and _maxdiff(got, ref) <= TOL
    # Keras padding="causal" (the golden cases "causal" / "causal1"): (K - 1) d zeros in front, none behind -- the function
    # and the Keras-like layer the reference's TF2C_Conv1DWeightNorm wraps
    xt = F.pad(torch.as_tensor(x).permute(0, 2, 1), (total, 0))
    ref = F.conv1d(xt, torch.as_tensor(w).permute(2, 1, 0), dilation=dil).permute(0, 2, 1).numpy()
    got = shim.conv1d_nwc(x, w, padding="CAUSAL", dilation=dil)
    assert got.shape == ref.shape == (2, T, cout) and _maxdiff(got, ref) <= TOL
    layer = shim.Conv1D(cout, K, padding="causal", dilation_rate=dil, use_bias=False)
    layer.build(shim.Shape((2, T, cin)))
    layer.kernel.assign(w)
    assert _maxdiff(np.asarray(layer(shim.Tensor(x))), ref) <= TOL


@pytest.mark.parametrize("W,C,kw,mult", [(12, 3, 2, 10), (7, 5, 2, 4), (20, 1, 3, 2)])
def test_depthwise_conv2d(W, C, kw, mult):
    """tf.nn.depthwise_conv2d(SAME) as the reference's interpolation layer uses it (support_layers.py:105): output
    channel c*mult + m = sum_j x[.., w + j - pad_l, c] f[0, j, c, m]."""
    rng = np.random.default_rng(W)
    x = rng.normal(size=(2, 1, W, C))
    f = rng.normal(size=(1, kw, C, mult))
    got = np.asarray(shim._depthwise_conv2d(x, f, [1, 1, 1, 1], "SAME"))
    total = kw - 1
    xt = F.pad(torch.as_tensor(x[:, 0]).permute(0, 2, 1), (total // 2, total - total // 2))     # (B, C, W)
    wt = torch.as_tensor(f[0]).permute(1, 2, 0).reshape(C * mult, 1, kw)                          # group c: rows c*mult..+mult
    ref = F.conv1d(xt, wt, groups=C).permute(0, 2, 1).numpy()[:, None]
    assert got.shape == ref.shape and _maxdiff(got, ref) <= TOL


@pytest.mark.parametrize("S,cin,cout,K,stride", [(11, 4, 1, 31, 5), (8, 15, 1, 121, 15), (6, 3, 2, 7, 3)])
def test_conv1d_transpose(S, cin, cout, K, stride):
    """tf.nn.conv1d_transpose(SAME) as TFPQMF.synthesis uses it (tf_preprocess.py:215-222): full transposed convolution
    cut at (K - stride) // 2, length S * stride."""
    rng = np.random.default_rng(S)
    x = rng.normal(size=(2, S, cin))
    f = rng.normal(size=(K, cout, cin))
    out_len = S * stride
    got = np.asarray(shim._conv1d_transpose(x, f, (2, out_len, cout), stride, "SAME"))
    full = F.conv_transpose1d(torch.as_tensor(x).permute(0, 2, 1), torch.as_tensor(f).permute(2, 1, 0),
                              stride=stride).permute(0, 2, 1).numpy()
    begin = max(K - stride, 0) // 2
    ref = full[:, begin:begin + out_len]
    if ref.shape[1] < out_len:
        ref = np.pad(ref, ((0, 0), (0, out_len - ref.shape[1]), (0, 0)))
    assert got.shape == ref.shape and _maxdiff(got, ref) <= TOL


def test_hann_window_periodic_and_symmetric():
    """Even lengths (the only ones the path uses: 1200) agree with torch.  For ODD lengths TensorFlow's
    window_ops._raised_cosine_window ignores `periodic` (n = window_length + periodic * even - 1): the stand-in follows
    TensorFlow there, torch does not -- pinned here so that the difference is a documented one."""
    for n in (1200, 16, 8):
        assert _maxdiff(shim._hann_window(n, periodic=True, dtype=np.float64),
                        torch.hann_window(n, periodic=True, dtype=torch.float64).numpy()) <= 1e-12
    for n in (1200, 16, 7):
        assert _maxdiff(shim._hann_window(n, periodic=False, dtype=np.float64),
                        torch.hann_window(n, periodic=False, dtype=torch.float64).numpy()) <= 1e-12
    assert _maxdiff(shim._hann_window(7, periodic=True, dtype=np.float64),
                    torch.hann_window(7, periodic=False, dtype=torch.float64).numpy()) <= 1e-12


@pytest.mark.parametrize("win,hop,nfft,frames", [(1200, 300, 2048, 9), (16, 4, 32, 13), (12, 3, 16, 5)])
def test_stft_matches_torch(win, hop, nfft, frames):
    """tf.signal.stft(pad_end=False): frames of `win` every `hop`, periodic Hann, zero extension to fft_length, rfft."""
    rng = np.random.default_rng(win)
    n = win + (frames - 1) * hop + 2
    x = rng.normal(size=(2, n))
    got = np.asarray(shim._stft(x, win, hop, fft_length=nfft))
    window = torch.hann_window(win, periodic=True, dtype=torch.float64)
    window = F.pad(window, (0, nfft - win))            # torch centres a short window: extend it explicitly instead
    ref = torch.stft(F.pad(torch.as_tensor(x), (0, nfft - win)), n_fft=nfft, hop_length=hop, win_length=nfft,
                     window=window, center=False, onesided=True, return_complex=True).permute(0, 2, 1).numpy()
    assert got.shape[1] == frames and got.shape == ref.shape
    assert _maxdiff(got, ref) <= TOL


@pytest.mark.parametrize("win,hop,nfft,frames", [(1200, 300, 2048, 9), (16, 4, 32, 13)])
def test_inverse_stft_matches_torch(win, hop, nfft, frames):
    """tf.signal.inverse_stft with window_fn = inverse_stft_window_fn(hop): irfft, first `win` samples, times
    w / sum_k w^2[n + k hop], overlap-add; in the interior (every sample under win/hop frames) it inverts the STFT."""
    rng = np.random.default_rng(frames)
    spec = rng.normal(size=(2, frames, nfft // 2 + 1)) + 1j * rng.normal(size=(2, frames, nfft // 2 + 1))
    spec[..., 0] = spec[..., 0].real
    spec[..., -1] = spec[..., -1].real
    inv_win = shim._inverse_stft_window_fn(hop)
    got = np.asarray(shim._inverse_stft(spec, win, hop, fft_length=nfft, window_fn=inv_win))
    assert got.shape == (2, (frames - 1) * hop + win)
    # independent evaluation: frames through torch.fft.irfft, the normalised window and overlap-add by F.fold
    fr = torch.fft.irfft(torch.as_tensor(spec), n=nfft, dim=-1)[..., :win]
    w = torch.hann_window(win, periodic=True, dtype=torch.float64)
    den = (w ** 2).reshape(win // hop, hop).sum(0).repeat(win // hop)
    fr = fr * (w / den)
    ola = F.fold(fr.permute(0, 2, 1), output_size=(1, (frames - 1) * hop + win), kernel_size=(1, win),
                 stride=(1, hop))[:, 0, 0].numpy()
    assert _maxdiff(got, ola) <= TOL
    # perfect reconstruction in the interior (torch.istft refuses this window: its zero first sample fails the NOLA
    # check at the signal edge): torch.stft forward, stand-in inverse, every sample under win/hop frames comes back
    x = rng.normal(size=(2, (frames - 1) * hop + win))
    wpad = F.pad(w, (0, nfft - win))
    fwd = torch.stft(F.pad(torch.as_tensor(x), (0, nfft - win)), n_fft=nfft, hop_length=hop, win_length=nfft, window=wpad,
                     center=False, onesided=True, return_complex=True).permute(0, 2, 1).numpy()
    back = np.asarray(shim._inverse_stft(fwd, win, hop, fft_length=nfft, window_fn=inv_win))
    lo, hi = win - hop, (frames - 1) * hop
    assert _maxdiff(back[:, lo:hi], x[:, lo:hi]) <= TOL


def test_overlap_and_add_and_cumsum_gather():
    rng = np.random.default_rng(0)
    fr = rng.normal(size=(3, 6, 8))
    got = np.asarray(shim._overlap_and_add(fr, 2))
    ref = F.fold(torch.as_tensor(fr).permute(0, 2, 1), output_size=(1, 5 * 2 + 8), kernel_size=(1, 8),
                 stride=(1, 2))[:, 0, 0].numpy()
    assert _maxdiff(got, ref) <= 1e-12
    x = rng.normal(size=(2, 50))
    assert _maxdiff(shim._cumsum(x, axis=1), torch.cumsum(torch.as_tensor(x), dim=1).numpy()) <= 1e-12
    table = rng.normal(size=(17, 4))
    idx = rng.integers(0, 17, size=(2, 9))
    assert np.array_equal(np.asarray(shim._gather(table, idx)), table[idx])
