"""Audio -> log-mel analysis (the step in front of the hot path).  Host side, numpy.

Restates, for the configuration the CLI uses (no band limiting, ``do_post=False``):
  compute_mel_spectrogram_internal   reference MBExWN_NVoc/vocoder/model/preprocess.py:417-572
  calc_stft (magnitude, centred)     reference MBExWN_NVoc/sig_proc/spec/stft.py:14-96
  window("hann", N)                  reference MBExWN_NVoc/sig_proc/Mwindows.py:60-67,176-185 (symmetric, zero end points)
  get_mel_filter                     reference preprocess.py:51-74 -> librosa.filters.mel(htk=False, norm="slaney")

librosa (requirements.txt: librosa >= 0.8, unpinned) is a third-party dependency that is neither in the reference
tree nor installable here; its mel basis is restated from the published Slaney Auditory-Toolbox formulas that the
librosa documentation gives (linear below 1 kHz with 200/3 Hz per mel, logarithmic above with step ln(6.4)/27,
triangles normalised by 2 / bandwidth).  The STFT part is pinned by golden vectors captured from the reference's
importable numpy code (tests/golden/reference_constants.npz); the mel basis is pinned only by its defining properties.
"""
import numpy as np


def hann_symmetric(n):
    """reference Mwindows.window("hann", n): 0.5 - 0.5 cos(2 pi k / (n-1)), mirrored around the centre."""
    win = np.zeros((n,))
    mid = (n - 1) // 2
    xx = np.arange(mid + 1)
    half = 0.5 - 0.5 * np.cos(2.0 * np.pi * xx / (n - 1))
    win[:mid + 1] = half
    win[n - 1:n - 2 - mid:-1] = half
    return win


def stft_magnitude(x, win_len, hop_len, fft_size, dtype=np.float32, pad_mode="reflect"):
    """|STFT| of x (batch, time): frames centred on multiples of hop_len, reference calc_stft(center=True, do_mag=True).
    Returns (batch, n_frames, fft_size//2+1)."""
    x = np.atleast_2d(np.asarray(x))
    win = hann_symmetric(win_len).astype(dtype)
    n_frames = x.shape[-1] // hop_len + 1
    xp = np.pad(x.astype(dtype, copy=False), ((0, 0), (win_len // 2, win_len)), mode=pad_mode)
    out = np.empty((x.shape[0], n_frames, fft_size // 2 + 1), dtype=dtype)
    for ii in range(n_frames):
        seg = xp[:, ii * hop_len: ii * hop_len + win_len]
        out[:, ii] = np.abs(np.fft.rfft(win * seg, fft_size))
    return out


def _hz_to_mel_slaney(freq):
    freq = np.asarray(freq, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = freq / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(freq >= min_log_hz, min_log_mel + np.log(np.maximum(freq, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(mels):
    mels = np.asarray(mels, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(mels >= min_log_mel, min_log_hz * np.exp(logstep * (mels - min_log_mel)), f_sp * mels)


def mel_frequencies(n_mels, fmin, fmax):
    """n_mels frequencies uniformly spaced on the Slaney mel scale between fmin and fmax."""
    return _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels))


def mel_basis_slaney(sr, n_fft, n_mels, fmin, fmax, dtype=np.float32):
    """(n_mels, n_fft//2+1) triangular filters, area-normalised ("slaney" norm)."""
    if fmax is None:
        fmax = sr / 2.0
    fft_freqs = np.linspace(0, sr / 2.0, n_fft // 2 + 1)
    mel_f = mel_frequencies(n_mels + 2, fmin, fmax)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_freqs[None, :]
    weights = np.zeros((n_mels, n_fft // 2 + 1))
    for ii in range(n_mels):
        lower = -ramps[ii] / fdiff[ii]
        upper = ramps[ii + 2] / fdiff[ii + 1]
        weights[ii] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (weights * enorm[:, None]).astype(dtype)


def compute_log_mel(sound, preprocess_config, dtype=np.float32):
    """reference compute_mel_spectrogram_internal(sound, cfg, band_limit=None, do_post=False):
    (batch, time) audio -> (batch, frames, mel_channels) natural-log mel amplitudes, and the mel frame rate."""
    sound = np.atleast_2d(np.asarray(sound))
    win_len = preprocess_config.get("win_size", preprocess_config["fft_size"])
    spec = stft_magnitude(sound, win_len, preprocess_config["hop_size"], preprocess_config["fft_size"], dtype=dtype)
    basis = mel_basis_slaney(preprocess_config["sample_rate"], preprocess_config["fft_size"],
                             preprocess_config["mel_channels"], preprocess_config["fmin"], preprocess_config["fmax"], dtype=dtype)
    mel = np.dot(spec, basis.T)
    mell = np.log(np.fmax(mel, np.finfo(mel.dtype).eps))
    return mell, preprocess_config["sample_rate"] / preprocess_config["hop_size"]


def compute_log_mel_device(sound, preprocess_config, n_samples=None):
    """:func:`compute_log_mel` on the GPU (csrc/mel_analysis.hip through ``mbx_mel_analysis``): sound is a float32 cuda
    tensor (batch, time), ``n_samples`` an optional int32 cuda tensor (batch,) of item lengths.  Returns a cuda tensor
    (batch, time // hop + 1, mel_channels) -- rows of item b beyond ``n_samples[b] // hop + 1`` are not written -- and the
    mel frame rate.  The window and the mel basis are the tables of this module; the transform runs in float32 (the
    host path transforms in float64 and rounds: the two agree to float32 rounding of the magnitudes)."""
    import ctypes
    import torch
    from .engine import _check, load_library
    if sound.dim() != 2 or sound.dtype != torch.float32 or not sound.is_cuda:
        raise ValueError("sound must be a float32 cuda tensor of shape (batch, time)")
    cfg = preprocess_config
    win_len = int(cfg.get("win_size", cfg["fft_size"]))
    hop, fft_size, n_mels = int(cfg["hop_size"]), int(cfg["fft_size"]), int(cfg["mel_channels"])
    dev = sound.device
    basis = mel_basis_slaney(cfg["sample_rate"], fft_size, n_mels, cfg["fmin"], cfg["fmax"], dtype=np.float32)
    nz = basis != 0
    lo = np.where(nz.any(axis=1), nz.argmax(axis=1), 1).astype(np.int32)
    hi = np.where(nz.any(axis=1), basis.shape[1] - 1 - nz[:, ::-1].argmax(axis=1), 0).astype(np.int32)
    ang = -2.0 * np.pi * np.arange(fft_size // 2) / fft_size
    tables = [torch.as_tensor(np.ascontiguousarray(tt), device=dev) for tt in
              (hann_symmetric(win_len).astype(np.float32), np.stack((np.cos(ang), np.sin(ang)), axis=1).astype(np.float32),
               basis, lo, hi)]
    sound = sound.contiguous()
    B, N = int(sound.shape[0]), int(sound.shape[1])
    frames = N // hop + 1
    out = torch.zeros((B, frames, n_mels), dtype=torch.float32, device=dev)
    if n_samples is not None:
        if n_samples.dtype != torch.int32 or tuple(n_samples.shape) != (B,) or n_samples.device != dev:
            raise ValueError("n_samples must be an int32 tensor of shape (batch,) on the device of sound")
        n_samples = n_samples.contiguous()
    # mbx_mel_analysis has no handle (hence no device of its own): it launches on the CURRENT device, which must be the one
    # the buffers and the stream belong to
    with torch.cuda.device(dev):
        _check(load_library().mbx_mel_analysis(sound.data_ptr(), n_samples.data_ptr() if n_samples is not None else None, B, N,
                                               win_len, hop, fft_size, n_mels, tables[0].data_ptr(), tables[1].data_ptr(),
                                               tables[2].data_ptr(), tables[3].data_ptr(), tables[4].data_ptr(),
                                               ctypes.c_float(float(np.finfo(np.float32).eps)), out.data_ptr(), frames,
                                               torch.cuda.current_stream(dev).cuda_stream))
    return out, cfg["sample_rate"] / hop
