#!/usr/bin/env python3
"""Mel-spectrogram inversion CLI -- same flags and file naming as the reference's bin/resynth_mel.py
(reference bin/resynth_mel.py:34-135), running on the MI355X HIP path.

Deviations (documented in INTEGRATION.md):
  * this build has only the GPU path: ``-g`` is accepted and implied; without a GPU the script fails loudly
  * ``-nt`` (TensorFlow CPU threads) is accepted and ignored
  * audio files are written with ``soundfile`` if it is installed, otherwise as float32 ``.wav`` through scipy
    (the reference uses pysndfile, default format flac)
"""
import os
import sys
import time

import numpy as np

test_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'mbexwn_vocoder_amd')
if os.path.exists(test_path):
    sys.path.insert(0, os.path.dirname(os.path.abspath(test_path)))

from mbexwn_vocoder_amd import list_models, mel_inverter  # noqa: E402
from mbexwn_vocoder_amd.fileio import load_var  # noqa: E402


def write_audio(outfile, data, rate, format):
    """reference bin/resynth_mel.py:104-105 (sndio.write): libsndfile through soundfile where it is installed, else the
    built-in writers -- flac (mbexwn_vocoder_amd/flac.py: 16-bit, uncompressed sub-frames) and wav (float32)."""
    try:
        import soundfile
        soundfile.write(outfile, data, rate, format=format.upper())
        return outfile
    except ImportError:
        pass
    if format.lower() == "flac":
        from mbexwn_vocoder_amd import flac
        return flac.write(outfile, data, rate)
    if format.lower() == "wav":
        from scipy.io import wavfile
        wavfile.write(outfile, rate, np.asarray(data, dtype=np.float32))
        return outfile
    raise RuntimeError(f"cannot write format {format}: soundfile is not installed, only flac and wav are built in")


def main(model_id, input_mell_files, output_dir, use_gpu=False, sigma=None, format=None, verbose=False, seed=42,
         num_threads=2, quiet=False, calibrate=0):
    import torch
    if not torch.cuda.is_available():
        print("resynth_mel::error:: no GPU available; this build has no CPU path", file=sys.stderr)
        sys.exit(1)
    if not use_gpu and not quiet:
        print("resynth_mel::note:: running on the MI355X HIP path (this build has no CPU path, -g is implied)",
              file=sys.stderr)
    format = format or "flac"                                   # the reference's default (bin/resynth_mel.py:119)
    if num_threads:                                           # -nt: host threads (numpy / torch CPU work around the HIP path)
        torch.set_num_threads(max(1, int(num_threads)))
        try:
            from threadpoolctl import threadpool_limits
            threadpool_limits(limits=max(1, int(num_threads)))
        except ImportError:
            pass
    if seed >= 0:                                  # reference :65-67
        np.random.seed(seed)
        torch.manual_seed(seed)

    MelInv = mel_inverter.MELInverter(model_id_or_path=model_id, verbose=verbose)
    if output_dir and not os.path.exists(output_dir):
        os.makedirs(output_dir)
    if calibrate and input_mell_files:
        # --calibrate N (this build): the form of the WaveNet's convolution is decided on the first N mels of the job
        # (MELInverter.calibrate -> mbx_calibrate) instead of on the synthetic mel of the engine's creation
        first = [MelInv.scale_mel(load_var(ff)) for ff in input_mell_files[:int(calibrate)]]
        info = MelInv.calibrate(first, verbose=verbose)
        if not quiet and not verbose:
            print(f"calibrated on {len(first)} file(s): convolution form {info['form']}", file=sys.stderr)

    for mell_file in input_mell_files:
        outfile = os.path.join(output_dir or "", "syn_" + os.path.splitext(os.path.basename(mell_file))[0] + "." + format)
        if not quiet:
            print(f"synthesize {mell_file} into {outfile}", file=sys.stderr)
        if verbose:
            print(f"load mell  from {mell_file}", file=sys.stderr)
        dd = load_var(mell_file)
        log_mel_spectrogram = MelInv.scale_mel(dd, verbose=verbose)

        start_time = time.time()
        syn_audio = MelInv.synth_from_mel(log_mel_spectrogram)
        end_time = time.time()

        if verbose:                                  # reference :90-96
            mel_resyn = MelInv.generate_mel_from_snd(syn_audio, srate=MelInv.srate)['mell'].T[np.newaxis]
            mell_err = mel_inverter.log_to_db * np.mean(np.abs(log_mel_spectrogram
                                                               - mel_resyn[:, :log_mel_spectrogram.shape[1]]))
            print(f"    synthesized audio with {syn_audio.size} samples in {end_time - start_time:.3f}s "
                  f"({syn_audio.size / (end_time - start_time):.2f}Hz), mel_error: {mell_err:.3f}dB", file=sys.stderr)
        if np.max(np.abs(syn_audio)) > 1:
            norm = 0.99 / np.max(np.abs(syn_audio))
            print(f'    to prevent clipping you would need to normalize {outfile} by {norm:.3f}', file=sys.stderr)
        if verbose:
            print(f"    save audio under {outfile}", file=sys.stderr)
        write_audio(outfile, syn_audio, MelInv.srate, format)


if __name__ == "__main__":
    from argparse import ArgumentParser
    parser = ArgumentParser(description="invert mel spectrograms into audio with an MBExWN model (MI355X HIP path)")
    parser.add_argument("model_id", default=None, nargs="?", const=None,
                        help="model identifier or path to a model directory. If not given the script lists all known "
                             "model names; the first model whose DOMAIN/name contains the identifier is used.")
    parser.add_argument("-i", "--input_mell_files", nargs="+", help="list of mell spectra stored in pickle files")
    parser.add_argument("-o", "--output_dir", help="output directory where synthetic sounds will be stored")
    parser.add_argument("--format", default="flac", help="file format for generated audio files (Def: %(default)s)")
    parser.add_argument("-nt", "--num_threads", default=2, type=int,
                        help="number of cpu threads of the host-side work (Def: %(default)s)")
    parser.add_argument("-g", "--use_gpu", action="store_true", help="run on gpu (implied)")
    parser.add_argument("-v", "--verbose", action="store_true", help="display verbose progress info")
    parser.add_argument("-q", "--quiet", action="store_true", help="dont display progress")
    parser.add_argument("--calibrate", default=0, type=int, metavar="N",
                        help="decide the form of the WaveNet's convolution on the first N input files before synthesis "
                             "(Def: %(default)s = keep the decision made at model load on a synthetic mel); the decision then "
                             "binds every file of the job, so the output depends on which N files come first")
    args = parser.parse_args()

    if not args.model_id:
        print("Please select one of the following models for mel inversion.\nYou don't need to select with a full ID. "
              "The first model containing the model_id you provide will be selected.\nFor example just specifying SPEECH "
              "will select the default SPEECH model.")
        for kk, ll in list_models().items():
            for md in ll:
                print(f" - {kk}/{md}")
    else:
        main(**vars(args))
