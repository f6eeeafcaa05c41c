"""``MELInverter``: the reference's inference object, backed by the MI355X HIP engine.

Mirrors reference MBExWN_NVoc/mel_inverter.py:21-239 (same constructor, attributes, methods and error
behaviour).  What differs is what ``self.model`` is: an ``MBExWNEngine`` (HIP kernels behind the C ABI of
include/mbexwn.h) instead of a Keras model on TensorFlow.
"""
import os
import sys
from typing import Dict, Union

import numpy as np
from scipy.interpolate import interp1d

log_to_db = 20 * np.log10(np.exp(1))   # reference vocoder/model/preprocess.py:78


class MELInverter(object):
    def __init__(self, model_id_or_path: Union[str, None] = None, verbose: bool = False, calibrate: bool = False):
        """As the reference's constructor (mel_inverter.py:22-41); ``calibrate`` (this build) is handed to
        :meth:`load_model`."""
        self.model = None
        self._calibrate_pending = False
        self._verbose = verbose
        self.model_id_or_path = model_id_or_path
        self.config_file = None
        self.preprocess_config = None
        self.mel_channels = None
        self.hop_size = None
        self.fft_size = None
        self.fmin = None
        self.fmax = None
        self._srate = None
        self.win_len = None

        self.lin_amp_scale = 1
        self.lin_amp_off = 1.e-5
        self.mel_amp_scale = 1
        self.use_max_limit = False

        if model_id_or_path:
            self.load_model(model_id_or_path=model_id_or_path, verbose=verbose, calibrate=calibrate)

    @property
    def srate(self):
        return self._srate

    # ------------------------------------------------------------------------------------------
    def scale_mel(self, mel_config: Dict, verbose=False):
        """Convert the content of a ``.mell`` dictionary into the (1, T, mel_channels) float32 log-mel
        conditioning of the model (reference mel_inverter.py:48-148; host side, numpy).

        Arithmetic is carried out in the dtype of the stored spectrogram, in the reference's order, so the
        result is bit-identical to the reference's (tests/golden/reference_scale_mel.npz).  Unlike the
        reference the caller's array is not modified in place.
        """
        hop_ratio = (mel_config['hoplen'] / mel_config['sr']) / (self.hop_size / self.srate)
        resample_hop = np.abs(hop_ratio - 1) > 0.001
        if resample_hop and verbose:
            print(f"compensate change in analysis hop size. mel analysis has {mel_config['hoplen'] / mel_config['sr']}"
                  f" while the model expects {self.hop_size / self.srate}.", file=sys.stderr)
        if mel_config['sr'] != self.srate and verbose:
            print(f"    WARNING::sample rate of mel analysis is  {mel_config['sr']} model expects {self.srate}.",
                  file=sys.stderr)

        if mel_config['fmin'] != self.fmin:
            raise RuntimeError(f"mell fmin {mel_config['fmin']} does not match model fmin {self.fmin}")
        if ((mel_config['fmax'] is None) and self.fmax != mel_config['sr'] / 2) \
                or ((mel_config['fmax'] is not None) and mel_config['fmax'] != self.fmax):
            raise RuntimeError(f"mell fmax {mel_config['fmax']} does not match model fmax {self.fmax}")

        if "mell" in mel_config:
            log_mel = np.array(mel_config['mell'].T[np.newaxis])
            if mel_config.get("log_spec_offset", 0) != 0:
                log_mel -= mel_config["log_spec_offset"]
            if mel_config.get("log_spec_scale", 1) != 1:
                log_mel /= mel_config["log_spec_scale"]
            mel = np.exp(log_mel)
        elif "mel" in mel_config:
            mel = np.array(mel_config['mel'].T[np.newaxis])
        else:
            raise RuntimeError("error::no supported mel spectrum (keys:mell or mel) in the mel dictionary")

        n_fft = mel_config.get("nfft", None)
        if n_fft is None:
            n_fft = mel_config.get("n_fft", None)
        if n_fft is None:
            n_fft = mel_config.get("fft_size", None)
        fft_scale_factor = self.fft_size // n_fft
        if fft_scale_factor != 1:
            mel *= fft_scale_factor
        if mel_config.get("lin_spec_offset", None) is not None and mel_config["lin_spec_offset"] != 0:
            mel -= mel_config["lin_spec_offset"]
        if mel_config.get("lin_spec_scale", 1) != 1:
            mel /= mel_config["lin_spec_scale"]
        if self.lin_amp_scale != 1:
            mel *= self.lin_amp_scale

        if self.use_max_limit:
            mell = np.log(np.fmax(mel, self.lin_amp_off)).astype(np.float32)
        else:
            mell = np.log(mel + self.lin_amp_off).astype(np.float32)

        if verbose:
            # diagnostics only (the reference prints comparable statistics at mel_inverter.py:110-116)
            db = log_to_db * mell
            print(f"    scaled log-mel {mell.shape}: level in dB -- min {db.min():.3f}, median {np.median(db):.3f}, "
                  f"mean {db.mean():.3f}, max {db.max():.3f}", file=sys.stderr)
            print(f"    analysis of the input mel: sample rate {mel_config['sr']}, hop {mel_config['hoplen']}, "
                  f"window {mel_config.get('winlen')}, FFT {n_fft}", file=sys.stderr)

        if resample_hop:
            # same expression order as the reference (mel_inverter.py:133-146): the grids decide the rounding
            mell = interp1d(np.arange(mell.shape[1]) * mel_config['hoplen'] / mel_config['sr'], mell, axis=1,
                            bounds_error=False, fill_value="extrapolate")(
                np.arange(0, (mell.shape[1] - 1 + 0.1) * mel_config['hoplen'] / mel_config['sr'],
                          self.hop_size / self.srate)).astype(np.float32)

        return mell * self.mel_amp_scale

    # ------------------------------------------------------------------------------------------
    def synth_from_mel(self, scaled_mell, noise=None):
        """(1, T, mel_channels) log-mel -> float32 audio of T*hop_size samples
        (reference mel_inverter.py:151-154; like there, a batch is flattened by ``ravel``).

        ``noise`` optionally injects the N(0,1) draw of the noise channel (shape (B, T*steps_per_frame));
        by default it is drawn on the device, as the reference draws tf.random.normal."""
        if self._calibrate_pending:
            # load_model(..., calibrate=True): the FIRST mel synthesised decides the form for every later call (at most 400 of
            # its frames are used) -- results therefore depend on which utterance came first; calibrate([...]) on a fixed set
            # (resynth_mel.py --calibrate N) is the reproducible way.  A calibration that cannot run (e.g. a form pinned by the
            # configuration) must not fail the synthesis: keep the creation-time form and say so.
            self._calibrate_pending = False
            try:
                self.calibrate([scaled_mell], verbose=self._verbose)
            except Exception as exc:                # noqa: BLE001 -- any engine error: the form of mbx_create stays
                import warnings
                warnings.warn(f"MELInverter: calibration on the first mel failed ({exc}); keeping the convolution form "
                              f"chosen at creation ({self.model.conv_form_info()['form']})", RuntimeWarning)
        syn_audio = self.model.infer(scaled_mell, sigma=None, synth_length=scaled_mell.shape[1] * self.hop_size,
                                     noise=noise).numpy()
        return syn_audio.ravel()

    def calibrate(self, scaled_mells, verbose=False, max_frames=400, seed=42):
        """Decide the form of the WaveNet's dilated convolution on REAL data (this build; C ABI mbx_calibrate).

        The engine runs Winograd F(4,3) only where its own rounding stays within a quarter of the float32 parity budget of
        the direct form; at creation that is measured on a seeded synthetic mel (mbx_create), which a louder or otherwise
        unusual corpus need not resemble.  ``scaled_mells``: a list of ``scale_mel`` outputs (1, T, mel_channels), e.g. the
        first utterances of the job; at most ``max_frames`` frames of each are used.  The same procedure (direct form,
        F(4,3), F(2,3) on these inputs, the fastest form within the threshold is kept) then binds every later
        ``synth_from_mel``.  Returns ``self.model.conv_form_info()``; ``verbose`` prints the decision."""
        import torch
        mels = [np.asarray(mm, dtype=np.float32).reshape(-1, mm.shape[-2], mm.shape[-1])[0][:max_frames] for mm in scaled_mells]
        if not mels:
            raise ValueError("calibrate() needs at least one mel spectrogram")
        lengths = [int(mm.shape[0]) for mm in mels]
        batch = np.zeros((len(mels), max(lengths), mels[0].shape[1]), dtype=np.float32)
        for ii, mm in enumerate(mels):
            batch[ii, :lengths[ii]] = mm
        dims = self.model.dims
        noise = np.random.default_rng(seed).normal(size=(len(mels), max(lengths) * dims.wn_in_rows_per_frame)).astype(np.float32)
        dev = self.model.device
        info = self.model.calibrate(torch.as_tensor(batch, device=dev),
                                    n_frames=torch.as_tensor(lengths, dtype=torch.int32, device=dev),
                                    noise=torch.as_tensor(noise, device=dev) if dims.noise_sigma else None)
        self._calibrate_pending = False
        if verbose:
            e43 = "n/a" if info["err_f43"] is None else f"{info['err_f43']:.2e}"
            e23 = "n/a" if info["err_f23"] is None else f"{info['err_f23']:.2e}"
            print(f"    calibrated the convolution form on {len(mels)} mel spectrogram(s), {sum(lengths)} frames: {info['form']} "
                  f"(|audio(F(4,3)) - audio(direct)| {e43}, F(2,3) {e23}, threshold {info['threshold']:.2e} on |audio| <= "
                  f"{info['ref_max']:.2f})", file=sys.stderr)
        return info

    def generate_mel_from_snd(self, snd, srate, on_device=False):
        """Audio -> ``.mell`` dictionary (reference mel_inverter.py:156-182); host side (analysis.py) or, with
        ``on_device=True``, the HIP kernel of csrc/mel_analysis.hip (same tables, float32 transform).
        The reference resamples when ``srate`` differs from the model rate through a function it never imports
        (mel_inverter.py:173, a NameError there); here the sound is resampled with a polyphase FIR
        (scipy.signal.resample_poly, Kaiser window) along the last axis."""
        from .analysis import compute_log_mel
        if srate != self.srate:
            from math import gcd
            from scipy.signal import resample_poly
            srate, target = int(round(srate)), int(round(self.srate))
            if srate <= 0:
                raise ValueError(f"generate_mel_from_snd: invalid sample rate {srate}")
            gg = gcd(srate, target)
            snd = resample_poly(np.asarray(snd, dtype=np.float64), target // gg, srate // gg, axis=-1)
        data_dict = {'nfft': self.fft_size,
                     'hoplen': self.hop_size,
                     'winlen': self.win_len,
                     'nmels': self.mel_channels,
                     'sr': self.srate,
                     'fmin': self.fmin,
                     'fmax': self.fmax,
                     'lin_spec_offset': self.lin_amp_off,
                     'lin_spec_scale': self.lin_amp_scale,
                     'log_spec_offset': 0.,
                     'log_spec_scale': self.mel_amp_scale,
                     "time_axis": 1}
        snd = np.asarray(snd)
        if snd.ndim == 1:
            snd = snd[np.newaxis]
        if on_device:
            import torch
            from .analysis import compute_log_mel_device
            mel_dev, _ = compute_log_mel_device(torch.as_tensor(np.ascontiguousarray(snd, dtype=np.float32)).cuda(),
                                                self.preprocess_config)
            mel_ref = mel_dev.cpu().numpy()
        else:
            mel_ref, _ = compute_log_mel(snd, self.preprocess_config, dtype=np.float32)
        data_dict['mell'] = mel_ref[0].T
        return data_dict

    # ------------------------------------------------------------------------------------------
    def load_model(self, model_id_or_path, verbose=False, calibrate=False):
        """reference mel_inverter.py:184-239: resolve the model directory, read ``config.yaml``, build the
        generator, restore the weights and copy the pre-processing parameters onto the instance.

        ``calibrate=True`` (this build; ORDER DEPENDENT: the decision is taken on at most 400 frames of the first utterance and
        binds every later one, a failing calibration keeps the creation-time form with a warning): the first mel handed to
        :meth:`synth_from_mel` goes through :meth:`calibrate`
        before it is synthesised -- the form of the WaveNet's convolution is then decided on the job's own data instead
        of on the synthetic mel of ``mbx_create`` (``resynth_mel.py --calibrate N`` does the same on the first N files)."""
        from . import get_config_file
        from .config import read_config
        from .engine import MBExWNEngine
        from .weights import load_weights

        config_file = get_config_file(model_id_or_path=model_id_or_path)
        model_dir = os.path.dirname(config_file)
        hparams = read_config(config_file=config_file)
        if "mbexwn_config" not in hparams:
            raise NotImplementedError(f"create_model::error::unkown config requested {list(hparams.keys())}. "
                                      "Only mbexwn_config is currently supported.")   # reference models.py:22-31
        self.config_file = config_file
        self.preprocess_config = hparams["preprocess_config"]

        weights_npz = os.path.join(model_dir, "weights.npz")
        weights_tf = os.path.join(model_dir, "weights.tf")            # reference mel_inverter.py:206
        if os.path.exists(weights_npz):
            if verbose:
                print(f"restore from {weights_npz}", file=sys.stderr)
            raw = load_weights(weights_npz)
        elif os.path.exists(weights_tf + ".index"):
            # the pretrained models of the reference ship as TensorFlow checkpoints; read without TensorFlow
            from .tf_checkpoint import load_reference_checkpoint
            if verbose:
                print(f"restore from {weights_tf}", file=sys.stderr)
            raw = load_reference_checkpoint(weights_tf, hparams)
        else:
            raise FileNotFoundError(f"error::no weights found under {model_dir} (expected weights.npz or weights.tf.index)")
        self.model = MBExWNEngine(hparams, raw)
        self._calibrate_pending = bool(calibrate)
        self._verbose = bool(verbose)
        if verbose:
            info = self.model.conv_form_info()
            print(f"convolution form {info['form']} (requested {info['requested']}, calibrated at creation on a synthetic mel: "
                  f"{'yes' if info['calibrated'] == 1 else 'no'})", file=sys.stderr)

        self.mel_channels = self.preprocess_config["mel_channels"]
        self.hop_size = self.preprocess_config["hop_size"]
        self.fft_size = self.preprocess_config["fft_size"]
        self.fmin = self.preprocess_config["fmin"]
        self.fmax = self.preprocess_config["fmax"]
        self._srate = self.preprocess_config['sample_rate']
        # the reference falls back to an undefined name here (mel_inverter.py:221); the fft size is what it means
        self.win_len = self.preprocess_config.get('win_size', self.fft_size)

        self.lin_amp_scale = 1
        if self.preprocess_config.get("lin_amp_scale", 1) != 1:
            self.lin_amp_scale = self.preprocess_config["lin_amp_scale"]
        self.lin_amp_off = 1.e-5
        if self.preprocess_config.get("lin_amp_off", None) is not None:
            self.lin_amp_off = self.preprocess_config["lin_amp_off"]
        self.mel_amp_scale = 1
        if self.preprocess_config.get("mel_amp_scale", 1) != 1:
            self.mel_amp_scale = self.preprocess_config["mel_amp_scale"]
        self.use_max_limit = False
        if self.preprocess_config.get("use_max_limit", False):
            self.use_max_limit = self.preprocess_config["use_max_limit"]
        return


def create_synthetic_model_dir(path, voice_type="SPEECH", seed=1234, weights_format="npz", **config_overrides):
    """Write a model directory (config.yaml + weights.npz, or weights.tf.* in the TensorFlow checkpoint format of the
    reference's model zips) with the canonical architecture and seeded synthetic weights -- the stand-in for the
    pretrained model zip that is not part of the reference tree."""
    from .config import canonical_config, dump_config
    from .weights import save_weights, synthetic_weights
    os.makedirs(path, exist_ok=True)
    cfg = canonical_config(voice_type, **config_overrides)
    dump_config(os.path.join(path, "config.yaml"), cfg)
    raw = synthetic_weights(cfg, seed=seed)
    if weights_format == "tf":
        from .tf_checkpoint import to_reference_variables, write_checkpoint
        write_checkpoint(os.path.join(path, "weights.tf"), to_reference_variables(raw, config=cfg))
    else:
        save_weights(os.path.join(path, "weights.npz"), raw)
    return path
