"""Optional RMS normalisation of the mel input and de-normalisation of the output (row A14).

Restates ``NormMelComponents`` (reference MBExWN_NVoc/vocoder/model/wavegen_1d.py:578-769) as it is used at
inference by ``PaNWaveNet.infer`` (wavegen_1d.py:493-507): when ``mbexwn_config.normalize_rms_from_mell`` is set, the
mel spectrogram is divided by an RMS contour estimated from the mel bands, and the synthesised audio is multiplied by
the up-sampled contour.  Both steps are cheap pre/post-processing around the HIP forward pass (mel rate / one multiply
per output sample) and run on the host in float32 numpy next to ``scale_mel``.

``librosa.mel_frequencies`` / ``librosa.filters.mel`` (third party, absent here) are restated by
``analysis.mel_frequencies`` / ``analysis.mel_basis_slaney`` (Slaney scale, area-normalised triangles).
Both RMS estimates are built: the band-width weighted one and ``normalize_use_pinv`` (the energy of the minimum-energy
spectrum that explains the mel frame: pseudo inverse of the mel filters, reference :603-608, 683-685).  Only the smoothing
variant (``normalize_rms_num_smooth_iters > 0``) is supported: the reference's other branch reduces over the wrong axis.
"""
import numpy as np

from .analysis import hann_symmetric, mel_basis_slaney, mel_frequencies

EPS = np.float32(1e-7)   # tf.keras.backend.epsilon()


def _overlap_add(frames, hop):
    n_frames, flen = frames.shape[-2:]
    out = np.zeros(frames.shape[:-2] + ((n_frames - 1) * hop + flen,), dtype=frames.dtype)
    for tt in range(n_frames):
        out[..., tt * hop: tt * hop + flen] += frames[..., tt, :]
    return out


class NormMel:
    def __init__(self, config):
        pp = config["preprocess_config"]
        mb = config["mbexwn_config"]
        self.hop = int(pp["hop_size"])
        self.win = int(pp.get("win_size", pp["fft_size"]))
        if 4 * self.hop != self.win:
            raise RuntimeError("NormMelComponents:error: this module currently supports only the case where "
                               f"win_size {self.win} = 4 * hop_size {self.hop}")          # reference :592-594
        self.use_pinv = bool(mb.get("normalize_use_pinv", False))
        self.iters = int(mb.get("normalize_rms_num_smooth_iters", 0))
        if self.iters <= 0:
            raise NotImplementedError("normalize_rms_from_mell is supported with normalize_rms_num_smooth_iters > 0 only")
        self.n_mels = int(pp["mel_channels"])
        self.rms_norm_fact = np.float32(pp["fft_size"] * self.win * 0.5)
        mel_f = mel_frequencies(self.n_mels + 2, pp["fmin"], pp["fmax"])
        self.inv_enorm = ((mel_f[2:self.n_mels + 2] - mel_f[:self.n_mels]) / 2.0).astype(np.float32)
        self.win_norm, self.pinv = np.float32(1.0), None
        if self.use_pinv:                                     # reference :603-608
            self.win_norm = np.sqrt(np.sum(hann_symmetric(self.win).astype(np.float32) ** 2)).astype(np.float32)
            basis = mel_basis_slaney(pp["sample_rate"], pp["fft_size"], self.n_mels, pp["fmin"], pp["fmax"], dtype=np.float32)
            self.pinv = np.ascontiguousarray(np.linalg.pinv(basis).T, dtype=np.float32)      # (n_mels, fft_size / 2 + 1)
        self.max_norm_fact = mb.get("max_norm_fact", None)
        self.compressor_exp = mb.get("normalize_compressor_exp", None)
        self.lin_amp_scale = np.float32(mb.get("lin_amp_scale", 1.0))
        self.lin_amp_off = np.float32(mb.get("lin_amp_off", 1.0e-5))
        self.mel_amp_scale = np.float32(mb.get("mel_amp_scale", 1.0))
        self.use_max_limit = bool(mb.get("use_max_limit", False))
        win = hann_symmetric(self.win).astype(np.float32)
        self.gwin = (win / np.sum(win)).astype(np.float32)
        self.smooth_win_size = int(self.win * mb.get("normalize_smooth_win_scale", 1))
        sw = hann_symmetric(self.smooth_win_size).astype(np.float32)
        if mb.get("normalize_smooth_with_squared_win", True):
            sw = sw ** 2
        self.smooth_syn_win = sw.astype(np.float32)

    def normalize(self, mell, synth_length):
        """mell (B,T,n_mels) float32 -> (mell' (B,T,n_mels) float32, gain (B, synth_length) float32)."""
        mell = np.asarray(mell, dtype=np.float32)
        T = mell.shape[1]
        mel = np.exp(mell)
        if self.use_pinv:                                     # reference :684-685
            spec = (np.tensordot(mel, self.pinv, axes=1) / self.win_norm).astype(np.float32)
            rms = np.sqrt(np.sum(np.square(spec), axis=-1, dtype=np.float32) / self.rms_norm_fact).astype(np.float32)
        else:
            rms = np.sqrt(np.sum(np.square(mel * self.inv_enorm), axis=-1) / self.rms_norm_fact).astype(np.float32)
        if self.max_norm_fact:
            rms = np.maximum(rms, np.float32(1.0 / self.max_norm_fact))
        if self.compressor_exp is not None:
            rms = np.power(rms, np.float32(self.compressor_exp)).astype(np.float32)
        cut = self.smooth_win_size // 2 + 2 * self.hop - self.win // 2
        norm_gain = _overlap_add(np.ones((1, T + 4, 1), np.float32) * self.smooth_syn_win[None, None, :], self.hop)[:, cut:]
        gain = None
        for _ in range(self.iters):
            ext = np.concatenate((rms[:, :1], rms[:, :1], rms, rms[:, -1:], rms[:, -1:]), axis=1)
            gain = _overlap_add(ext[:, :, None] * self.smooth_syn_win[None, None, :], self.hop)[:, cut:]
            gain = (gain / np.maximum(EPS, norm_gain)).astype(np.float32)
            n_out = (gain.shape[1] - self.win) // self.hop + 1
            idx = np.arange(self.win)[None, :] + self.hop * np.arange(n_out)[:, None]
            rms = np.sum(gain[:, idx] * self.gwin[None, None, :], axis=-1, dtype=np.float32)[:, :T]
        mel = mel / np.maximum(EPS, rms[:, :, None]) * self.lin_amp_scale
        if self.use_max_limit:
            out = self.mel_amp_scale * np.log(np.maximum(mel, self.lin_amp_off))
        else:
            out = self.mel_amp_scale * np.log(mel + self.lin_amp_off)
        off = self.win // 2
        up = np.maximum(gain[:, off:off + synth_length], EPS)
        if up.shape[1] < synth_length:                         # reference :760-765
            up = np.concatenate((up, np.repeat(up[:, -1:], synth_length - up.shape[1], axis=1)), axis=1)
        return out.astype(np.float32), up.astype(np.float32)
