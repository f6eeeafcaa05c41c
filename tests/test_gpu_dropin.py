"""Drop-in surface on the GPU: MELInverter + resynth_mel.py CLI + utterance sharding through the real engine."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import synthetic_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3}


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    from mbexwn_vocoder_amd.mel_inverter import create_synthetic_model_dir
    return create_synthetic_model_dir(str(tmp_path_factory.mktemp("model") / "speech_small"), "SPEECH", **SMALL)


def mell_dict(frames, seed=5):
    rng = np.random.default_rng(seed)
    return {"nfft": 2048, "hoplen": 300, "winlen": 1200, "nmels": 80, "sr": 24000, "fmin": 0.0, "fmax": 12000.0,
            "lin_spec_offset": 1e-5, "lin_spec_scale": 1, "log_spec_offset": 0.0, "log_spec_scale": 1, "time_axis": 1,
            "mell": rng.normal(-5, 2, size=(80, frames)).astype(np.float32)}


def test_mel_inverter_end_to_end(model_dir):
    import torch
    from mbexwn_vocoder_amd.config import read_config
    from mbexwn_vocoder_amd.mel_inverter import MELInverter
    from mbexwn_vocoder_amd.tables import WaveTables
    from mbexwn_vocoder_amd.weights import load_weights
    from oracle.mbexwn_oracle import OracleModel
    inv = MELInverter(model_dir)
    assert (inv.srate, inv.hop_size, inv.mel_channels, inv.fft_size, inv.win_len) == (24000, 300, 80, 2048, 1200)
    dd = mell_dict(19)
    mell = inv.scale_mel(dd)
    assert mell.shape == (1, 19, 80) and mell.dtype == np.float32
    rng = np.random.default_rng(0)
    noise = rng.normal(size=(1, 19 * 20)).astype(np.float32)
    audio = inv.synth_from_mel(mell, noise=noise)
    assert audio.shape == (19 * 300,) and audio.dtype == np.float32
    cfg = read_config(os.path.join(model_dir, "config.yaml"))
    wt = WaveTables(sample_rate=8000.0, **cfg["mbexwn_config"]["wavetable_config"])
    ref = OracleModel(cfg, load_weights(os.path.join(model_dir, "weights.npz")), wt).forward(mell, noise)[0]
    assert np.max(np.abs(audio - ref)) <= 1e-4 * max(1.0, np.abs(ref).max())
    # default path draws the noise on the device: same seed, same audio
    torch.manual_seed(7)
    a1 = inv.synth_from_mel(mell)
    torch.manual_seed(7)
    a2 = inv.synth_from_mel(mell)
    assert np.array_equal(a1, a2) and not np.array_equal(a1, audio)


def test_tf_checkpoint_model_dir_gives_the_same_audio(model_dir, tmp_path):
    """A model directory in the layout of the reference's model zips (config.yaml + weights.tf.index/.data-*) is read
    without TensorFlow (tf_checkpoint.py) and synthesises exactly what the weights.npz directory does."""
    from mbexwn_vocoder_amd.mel_inverter import MELInverter, create_synthetic_model_dir
    tf_dir = create_synthetic_model_dir(str(tmp_path / "speech_small_tf"), "SPEECH", weights_format="tf", **SMALL)
    assert os.path.exists(os.path.join(tf_dir, "weights.tf.index")) and not os.path.exists(os.path.join(tf_dir, "weights.npz"))
    inv_npz, inv_tf = MELInverter(model_dir), MELInverter(tf_dir)
    mell = inv_npz.scale_mel(mell_dict(11))
    noise = np.random.default_rng(1).normal(size=(1, 11 * 20)).astype(np.float32)
    assert np.array_equal(inv_tf.synth_from_mel(mell, noise=noise), inv_npz.synth_from_mel(mell, noise=noise))


def test_cli_round_trip(model_dir, tmp_path):
    from mbexwn_vocoder_amd.fileio import save_var
    from scipy.io import wavfile
    files = []
    for ii, frames in enumerate((12, 31)):
        path = str(tmp_path / f"utt{ii}.mell")
        save_var(path, mell_dict(frames, seed=ii))
        files.append(path)
    cli = os.path.join(ROOT, "mbexwn_vocoder_amd", "bin", "resynth_mel.py")
    out_dir = str(tmp_path / "out")
    res = subprocess.run([sys.executable, cli, model_dir, "-i", *files, "-o", out_dir, "--format", "wav", "-g", "-v"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    for ii, frames in enumerate((12, 31)):
        rate, data = wavfile.read(os.path.join(out_dir, f"syn_utt{ii}.wav"))
        assert rate == 24000 and data.shape == (frames * 300,) and np.all(np.isfinite(data))
    # the reference's defaults: --format flac (built-in writer when libsndfile is absent), -nt host threads
    res = subprocess.run([sys.executable, cli, model_dir, "-i", files[0], "-o", out_dir, "-nt", "1", "-q"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    stream = open(os.path.join(out_dir, "syn_utt0.flac"), "rb").read()
    assert stream[:4] == b"fLaC" and int.from_bytes(stream[18:26], "big") & ((1 << 36) - 1) == 12 * 300
    listing = subprocess.run([sys.executable, cli], capture_output=True, text=True, timeout=120)
    assert " - SPEECH/MBExWN_SIIConv_V71g_SPEECH" in listing.stdout


def test_cli_config1_canonical_3s_against_the_oracle(tmp_path):
    """BASELINE config 1 as written: the MW-SP-FD model (canonical C = 320, L = 5), one 3 s utterance (80 x 240 mel) through
    resynth_mel.py (reference protocol bin/resynth_mel.py:74-104: load the .mell pickle, scale_mel, synth_from_mel, write
    the audio).  The file the CLI wrote is held to the float64 oracle run on the same scaled mel and the same noise draw
    (the CLI seeds torch with 42 and draws the noise channel on the device, reference :65-67)."""
    import torch
    from scipy.io import wavfile
    from mbexwn_vocoder_amd.config import read_config
    from mbexwn_vocoder_amd.fileio import save_var
    from mbexwn_vocoder_amd.mel_inverter import MELInverter, create_synthetic_model_dir
    from mbexwn_vocoder_amd.tables import WaveTables
    from mbexwn_vocoder_amd.weights import load_weights
    from oracle.mbexwn_oracle import OracleModel
    canon_dir = create_synthetic_model_dir(str(tmp_path / "MW-SP-FD_canonical"), "SPEECH")
    frames = 240
    dd = mell_dict(frames, seed=21)
    path = str(tmp_path / "utt3s.mell")
    save_var(path, dd)
    cli = os.path.join(ROOT, "mbexwn_vocoder_amd", "bin", "resynth_mel.py")
    out_dir = str(tmp_path / "out")
    res = subprocess.run([sys.executable, cli, canon_dir, "-i", path, "-o", out_dir, "--format", "wav", "-g"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    rate, data = wavfile.read(os.path.join(out_dir, "syn_utt3s.wav"))
    assert rate == 24000 and data.dtype == np.float32 and data.shape == (frames * 300,)
    inv = MELInverter(canon_dir)
    assert inv.model.dims.wn_channels == 320 and inv.model.dims.wn_layers == 5
    mell = inv.scale_mel(dd)
    torch.manual_seed(42)                                   # the CLI's seed; its first draw is the noise channel
    noise = torch.randn((1, frames * 20), device="cuda", dtype=torch.float32).cpu().numpy()
    cfg = read_config(os.path.join(canon_dir, "config.yaml"))
    wt = WaveTables(sample_rate=8000.0, **cfg["mbexwn_config"]["wavetable_config"])
    ref = OracleModel(cfg, load_weights(os.path.join(canon_dir, "weights.npz")), wt).forward(mell, noise)[0]
    err = float(np.max(np.abs(data.astype(np.float64) - ref)))
    assert err <= 1e-4 * max(1.0, float(np.abs(ref).max())), err


def test_calibrate_on_real_data_through_the_drop_in_surface(tmp_path):
    """MELInverter.load_model(calibrate=True) / resynth_mel.py --calibrate N: the form of the WaveNet's convolution is
    decided on the job's own mels (mbx_calibrate) instead of on the synthetic mel of the engine's creation.  Stressed
    weights (the WaveNet's gains x ~5, shifted biases: |h| ~ 25) for which the creation-time calibration keeps a Winograd
    form while a loud mel -- quieter audio, hence a tighter threshold -- does not: after the calibration on that mel the
    handle reports the safer form, and its audio is as exact as the direct form's."""
    import torch
    from mbexwn_vocoder_amd.config import read_config
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    from mbexwn_vocoder_amd.fileio import save_var
    from mbexwn_vocoder_amd.mel_inverter import MELInverter, create_synthetic_model_dir
    from mbexwn_vocoder_amd.tables import WaveTables
    from mbexwn_vocoder_amd.weights import load_weights, save_weights
    from oracle.mbexwn_oracle import OracleModel
    from test_gpu_forms import stressed_weights
    rank = {"direct": 0, "f23": 1, "f43": 2}
    frames = 60
    rng = np.random.default_rng(4242)
    loud = np.clip(np.log(np.exp(rng.normal(-5.0, 2.0, size=(1, frames, 80))) + 1e-5) + 4.0, -11.5, 2.0).astype(np.float32)
    noise = rng.normal(size=(1, frames * 20)).astype(np.float32)
    found = None
    for gain in (5.0, 4.7, 5.3, 4.4, 5.6):
        mdir = create_synthetic_model_dir(str(tmp_path / f"stressed_{gain}"), "SING")
        wpath = os.path.join(mdir, "weights.npz")
        save_weights(wpath, stressed_weights(load_weights(wpath), gain, 0.3))
        plain = MELInverter(mdir)
        at_creation = plain.model.conv_form_info()
        assert at_creation["calibrated"] == 1
        inv = MELInverter(mdir, calibrate=True)
        assert inv.model.conv_form_info()["form"] == at_creation["form"]          # nothing decided before the first mel
        audio = inv.synth_from_mel(loud, noise=noise)
        after = inv.model.conv_form_info()
        assert after["calibrated"] == 2 and not inv._calibrate_pending
        # the decision follows the numbers measured on the caller's data
        want = "f43" if after["err_f43"] is not None and after["err_f43"] <= after["threshold"] else \
            "f23" if after["err_f23"] is not None and after["err_f23"] <= after["threshold"] else "direct"
        assert after["form"] == want
        if rank[after["form"]] < rank[at_creation["form"]]:
            found = (gain, mdir, at_creation, after, audio, plain.synth_from_mel(loud, noise=noise))
            break
    assert found, "no stress level at which the loud mel changes the creation-time decision"
    gain, mdir, at_creation, after, audio, audio_uncal = found
    cfg = read_config(os.path.join(mdir, "config.yaml"))
    raw = load_weights(os.path.join(mdir, "weights.npz"))
    wt = WaveTables(sample_rate=8000.0, **cfg["mbexwn_config"]["wavetable_config"])
    ref = OracleModel(cfg, raw, wt).forward(loud, noise)[0]
    direct = MBExWNEngine(cfg, raw, wt, conv_form="direct").forward(torch.as_tensor(loud).cuda(), noise=torch.as_tensor(noise).cuda()).cpu().numpy()[0]
    err, err_uncal, err_direct = (float(np.abs(xx.astype(np.float64) - ref).max()) for xx in (audio, audio_uncal, direct))
    print(f"gain {gain}: creation {at_creation['form']} -> calibrated on the loud mel {after['form']}; |audio - oracle| calibrated "
          f"{err:.2e}, un-calibrated {err_uncal:.2e}, direct form {err_direct:.2e} on |audio| <= {np.abs(ref).max():.1f}")
    assert err <= 1.25 * err_direct + 1e-7
    # the CLI: --calibrate N decides on the first N files and says so under -v
    dd = mell_dict(40, seed=3)
    path = str(tmp_path / "utt.mell")
    save_var(path, dd)
    cli = os.path.join(ROOT, "mbexwn_vocoder_amd", "bin", "resynth_mel.py")
    res = subprocess.run([sys.executable, cli, mdir, "-i", path, "-o", str(tmp_path / "out"), "--format", "wav", "-g", "-v",
                          "--calibrate", "1"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    assert "calibrated the convolution form on 1 mel spectrogram(s)" in res.stderr and "convolution form" in res.stderr


def test_sharded_synthesis_matches_single_runs(model_dir):
    """config 4 in miniature: ragged utterances, LPT shards, padded micro-batches, per-item parity."""
    import torch
    from mbexwn_vocoder_amd.mel_inverter import MELInverter
    from mbexwn_vocoder_amd.sharding import ShardedSynthesizer, lpt_partition
    eng = MELInverter(model_dir).model
    rng = np.random.default_rng(9)
    lengths = [int(vv) for vv in rng.integers(2, 40, size=11)]
    mels, noises = [], []
    for ll in lengths:
        mm, nn = synthetic_inputs(ll, 1, ll)
        mels.append(mm[0])
        noises.append(nn[0])

    def forward(mel, n_frames, noise):
        return eng.forward(torch.as_tensor(mel).cuda(), n_frames=torch.as_tensor(n_frames).cuda(),
                           noise=torch.as_tensor(noise).cuda())

    singles = [forward(mm[None], np.asarray([mm.shape[0]], np.int32), nn[None]).cpu().numpy()[0]
               for mm, nn in zip(mels, noises)]
    got = ShardedSynthesizer(forward, 300, 20, max_batch=4).run(mels, noises)
    for ii in range(len(mels)):
        assert np.array_equal(got[ii], singles[ii])
    # a 4-rank partition covers everything; each simulated rank reproduces its items
    for rank in range(4):
        local = ShardedSynthesizer(forward, 300, 20, rank=rank, world_size=4, max_batch=3).run(mels, noises, gather=None)
        assert sorted(local) == sorted(lpt_partition(lengths, 4)[rank])
        for ii, audio in local.items():
            assert np.array_equal(audio, singles[ii])


def test_rms_normalised_model_end_to_end(tmp_path):
    """row A14: a model with normalize_rms_from_mell runs through infer() = normalise -> HIP forward -> de-normalise."""
    import torch
    from mbexwn_vocoder_amd.config import read_config
    from mbexwn_vocoder_amd.mel_inverter import MELInverter, create_synthetic_model_dir
    from mbexwn_vocoder_amd.tables import WaveTables
    from mbexwn_vocoder_amd.weights import load_weights
    from oracle.mbexwn_oracle import OracleModel, normalize_inputs_by_rms
    over = dict(SMALL)
    over.update({"mbexwn_config:normalize_rms_from_mell": True, "mbexwn_config:normalize_rms_num_smooth_iters": 1})
    mdir = create_synthetic_model_dir(str(tmp_path / "norm_model"), "SPEECH", **over)
    inv = MELInverter(mdir)
    mell = inv.scale_mel(mell_dict(21))
    rng = np.random.default_rng(1)
    noise = rng.normal(size=(1, 21 * 20)).astype(np.float32)
    audio = inv.synth_from_mel(mell, noise=noise)
    cfg = read_config(os.path.join(mdir, "config.yaml"))
    mel_n, gain = normalize_inputs_by_rms(mell, cfg, 21 * 300)
    wt = WaveTables(sample_rate=8000.0, **cfg["mbexwn_config"]["wavetable_config"])
    ref = OracleModel(cfg, load_weights(os.path.join(mdir, "weights.npz")), wt).forward(mel_n.astype(np.float32), noise)[0] * gain[0]
    assert audio.shape == ref.shape
    assert np.max(np.abs(audio - ref)) <= 1e-4 * max(1.0, np.abs(ref).max())


def test_mel_analysis_on_the_device_matches_the_host_analysis():
    """csrc/mel_analysis.hip (mbx_mel_analysis, the step in front of the hot path) against analysis.compute_log_mel, whose
    STFT is pinned to the reference's own numpy code (tests/test_mel_inverter.py, reference_constants.npz): ragged batch,
    a frame count that is not a multiple of anything, edges reflected."""
    import torch
    from mbexwn_vocoder_amd.analysis import compute_log_mel, compute_log_mel_device
    from mbexwn_vocoder_amd.config import canonical_config
    pre = canonical_config("SPEECH")["preprocess_config"]
    rng = np.random.default_rng(17)
    lengths = [7231, 3000, 1201]
    snd = np.zeros((3, max(lengths)), dtype=np.float32)
    for ii, ll in enumerate(lengths):
        tt = np.arange(ll) / pre["sample_rate"]
        snd[ii, :ll] = (0.3 * np.sin(2 * np.pi * (110.0 * (ii + 1)) * tt) + 0.05 * rng.normal(size=ll)).astype(np.float32)
    got, rate = compute_log_mel_device(torch.as_tensor(snd).cuda(), pre,
                                       n_samples=torch.as_tensor(lengths, dtype=torch.int32).cuda())
    got = got.cpu().numpy()
    assert rate == pre["sample_rate"] / pre["hop_size"]
    for ii, ll in enumerate(lengths):
        ref, _ = compute_log_mel(snd[ii:ii + 1, :ll], pre, dtype=np.float32)
        nfr = ll // pre["hop_size"] + 1
        assert ref.shape == (1, nfr, pre["mel_channels"])
        # float32 transform against the float64 transform of the host path, compared on the amplitudes
        err = np.abs(np.exp(got[ii, :nfr]) - np.exp(ref[0]))
        assert np.max(err) <= 2e-5 * np.max(np.exp(ref[0])), f"item {ii}: {np.max(err)}"
        assert np.max(np.abs(got[ii, :nfr] - ref[0])) <= 2e-3          # log domain, including the quiet channels


def test_mel_analysis_clamps_item_lengths():
    """n_samples comes from the device and must not address outside the item's row -- an empty item is one
    frame of silence (log eps), an over-long entry is clamped to the row."""
    import torch
    from mbexwn_vocoder_amd.analysis import compute_log_mel_device
    from mbexwn_vocoder_amd.config import canonical_config
    pre = canonical_config("SPEECH")["preprocess_config"]
    rng = np.random.default_rng(3)
    snd = (0.1 * rng.normal(size=(3, 2400))).astype(np.float32)
    good, _ = compute_log_mel_device(torch.as_tensor(snd).cuda(), pre)
    got, _ = compute_log_mel_device(torch.as_tensor(snd).cuda(), pre,
                                    n_samples=torch.as_tensor([0, 10 ** 6, 2400], dtype=torch.int32).cuda())
    got, good = got.cpu().numpy(), good.cpu().numpy()
    assert np.allclose(got[0, 0], np.log(np.finfo(np.float32).eps))
    assert np.array_equal(got[1], good[1]) and np.array_equal(got[2], good[2])


@pytest.mark.timeout(900)
def test_rccl_single_rank_collectives(tmp_path):
    """The N > 1 code path on the one GPU this box has, over the real `nccl` backend (= RCCL): a FRESH child process (the
    parent of a rank must not have touched the GPU) runs bench.py with --force-dist on the sharded workload -- process group
    on a device, barrier, max-over-ranks all_reduce, ShardedSynthesizer's device-resident all_gather with
    force_collective -- and must print the bench line with an in-tolerance max|delta| of the gathered output."""
    import json
    import socket
    with socket.socket() as ss:
        ss.bind(("127.0.0.1", 0))
        port = ss.getsockname()[1]
    env = {kk: vv for kk, vv in os.environ.items() if kk not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--workload",
                          "config4_vo_256utt", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env=env,
                         capture_output=True, text=True, timeout=840)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["config"]["max_abs_delta_ok"] is True, line["config"]


def test_sharded_gather_over_rccl_equals_local_shard(model_dir):
    """ShardedSynthesizer(force_collective=True) on device tensors with world_size 1 over `nccl`: all_gather and gather of
    the flat shard must return exactly the local shard."""
    code = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from helpers import synthetic_inputs
from mbexwn_vocoder_amd.mel_inverter import MELInverter
from mbexwn_vocoder_amd.sharding import ShardedSynthesizer
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
eng = MELInverter(sys.argv[2]).model
lengths = [5, 33, 12, 7, 21]
mels, noises = zip(*[(mm[0], nn[0]) for mm, nn in (synthetic_inputs(ll, 1, ll) for ll in lengths)])
fwd = lambda mel, nfr, noise: eng.forward(mel, n_frames=nfr, noise=noise)
dev = torch.device("cuda", 0)
local = ShardedSynthesizer(fwd, 300, 20, max_batch=3, device=dev).run(list(mels), list(noises))
for mode in ("all", "rank0"):
    syn = ShardedSynthesizer(fwd, 300, 20, max_batch=3, device=dev, force_collective=True)
    plan = syn.stage(list(mels), list(noises))
    res = syn.run_staged(plan, gather=mode)
    assert plan["parts"] is not None and res.parts[0].is_cuda          # the collective ran, on device tensors
    got = res.to_list()
    assert all(np.array_equal(aa, bb) for aa, bb in zip(got, local)), mode
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
"""
    import socket
    with socket.socket() as ss:
        ss.bind(("127.0.0.1", 0))
        port = ss.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    res = subprocess.run([sys.executable, "-c", code, ROOT, model_dir], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "RCCL_OK" in res.stdout, res.stderr[-2000:]

@pytest.mark.timeout(600)
def test_two_ranks_share_the_one_gpu_sharded_synthesis(model_dir, tmp_path):
    """world_size 2 with REAL kernels on the one GPU this box has: two fresh processes, each with its own handle on cuda:0,
    synthesise their LPT shards of 13 ragged utterances at the same time and gather them on rank 0 (gloo on host tensors:
    RCCL refuses two ranks on one device, so the collective itself is the CPU one here; the device-resident RCCL gather runs
    with one rank in the tests above).  Rank 0 must end up with every utterance, bit-equal to its single run."""
    code = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from helpers import synthetic_inputs
from mbexwn_vocoder_amd.mel_inverter import MELInverter
from mbexwn_vocoder_amd.sharding import ShardedSynthesizer, lpt_partition
rank, world = int(sys.argv[3]), 2
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
eng = MELInverter(sys.argv[2]).model
lengths = [int(vv) for vv in np.random.default_rng(4).integers(2, 60, size=13)]
mels, noises = zip(*[(mm[0], nn[0]) for mm, nn in (synthetic_inputs(100 + ll, 1, ll) for ll in lengths)])
def fwd(mel, nfr, noise):
    return eng.forward(torch.as_tensor(mel).cuda(), n_frames=torch.as_tensor(nfr).cuda(), noise=torch.as_tensor(noise).cuda()).cpu().numpy()
syn = ShardedSynthesizer(fwd, 300, 20, rank=rank, world_size=world, max_batch=3)
dist.barrier()                                   # both ranks compute at the same time
got = syn.run(list(mels), list(noises))          # default: gathered on rank 0
mine = lpt_partition(lengths, world)[rank]
assert len(mine) > 0
if rank == 0:
    assert got is not None and len(got) == len(lengths)
    for ii, (mm, nn) in enumerate(zip(mels, noises)):
        single = fwd(mm[None], np.asarray([mm.shape[0]], np.int32), nn[None])[0]
        assert np.array_equal(got[ii], single), ii
else:
    assert got is None
dist.barrier()
dist.destroy_process_group()
print("TWO_RANKS_OK", rank)
"""
    import socket
    with socket.socket() as ss:
        ss.bind(("127.0.0.1", 0))
        port = ss.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    procs = [subprocess.Popen([sys.executable, "-c", code, ROOT, model_dir, str(rank)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for rank in range(2)]
    outs = [pp.communicate(timeout=540) for pp in procs]
    for rank, (pp, (so, se)) in enumerate(zip(procs, outs)):
        assert pp.returncode == 0 and f"TWO_RANKS_OK {rank}" in so, se[-2000:]


@pytest.mark.parametrize("case", ["plain", "rmsnorm"])
def test_infer_and_infer_components_against_the_reference_methods(case):
    """engine.infer / infer_components against the reference's own PaNWaveNet.infer / infer_components bodies
    (tests/golden/make_reference_infer.py): synth_length shorter than, equal to (0 = segment_length) and longer than the
    mel (the last frame is repeated once), the parameter list of return_F0 with the reference's slices, a transposition
    factor, an external F0 contour, with and without the RMS normalisation."""
    import torch
    from helpers import build_case
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    gold = np.load(os.path.join(ROOT, "tests", "golden", "reference_infer.npz"))
    over = dict(SMALL)
    if case == "rmsnorm":
        over.update({"mbexwn_config:normalize_rms_from_mell": True, "mbexwn_config:normalize_rms_num_smooth_iters": 1})
    cfg, raw, wt = build_case("SPEECH", over)
    cfg["preprocess_config"]["segment_length"] = 14 * 300          # what synth_length = 0 falls back to
    eng = MBExWNEngine(cfg, raw, wt)
    mell = gold[f"{case}/mell"]

    def close(a, b, tol=1e-4):
        a, b = np.asarray(a), np.asarray(b)
        assert a.shape == b.shape, (a.shape, b.shape)
        assert float(np.max(np.abs(a - b))) <= tol * max(1.0, float(np.abs(b).max())), float(np.max(np.abs(a - b)))

    for tag, synth_length in (("short", 14 * 300 - 123), ("exact", 0), ("long", 14 * 300 + 150)):
        audio, params = eng.infer(mell, synth_length=synth_length, return_F0=True, noise=gold[f"{case}/{tag}/noise"])
        close(audio.numpy(), gold[f"{case}/{tag}/audio"])
        assert [pp[0] for pp in params] == ["F0", "PSig", "PS"]
        got = {pp[0]: pp[1].numpy() for pp in params}
        close(got["F0"], gold[f"{case}/{tag}/F0"], tol=1e-5)       # Hz up to 600: 6e-3 Hz
        close(got["PSig"], gold[f"{case}/{tag}/PSig"])
        assert tuple(got["PS"].shape) == tuple(gold[f"{case}/{tag}/PS_shape"])
        close(got["PS"][:, :, ::8], gold[f"{case}/{tag}/PS"], tol=1e-4)
    f0, exc, env, gain = eng.infer_components(mell, synth_length=14 * 300, transposition_factor=1.25, noise=gold[f"{case}/comp/noise"])
    close(f0, gold[f"{case}/comp/f0"], tol=1e-5)
    close(exc, gold[f"{case}/comp/excitation"])
    close(np.abs(env)[:, :, ::8], gold[f"{case}/comp/env_abs"])
    if case == "rmsnorm":
        close(gain, gold[f"{case}/comp/gain"])
    else:
        assert gain is None
    _, exc, env, _ = eng.infer_components(mell, F0=gold[f"{case}/ext/f0_in"], noise=gold[f"{case}/comp/noise"])
    close(exc, gold[f"{case}/ext/excitation"])
    close(np.abs(env)[:, :, ::8], gold[f"{case}/ext/env_abs"])
