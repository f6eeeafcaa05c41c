#!/usr/bin/env python3
"""Benchmark of the MBExWN mel-inversion forward pass on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (mel -> 24 kHz audio) over one batch of synthetic mels that is already
resident in HBM.  Default workload at every N: BASELINE.json configs[2] -- MW-SI-FD (C=320), batch = 16 x 10 s
(80 x 800 mels) per GPU, the largest single-GPU configuration; utterances are independent, so N ranks run N such
batches (weak scaling, no data-path collective; RCCL only carries the barrier and the max-over-ranks of the timing).
Started without a launcher and with --gpus N > 1 the script starts the N ranks itself (one process per GPU, RCCL).
Rank 0 prints ONE JSON line: the default workload's throughput, max|delta| against the float64 oracle, the roofline
of the dominant kernel and of the bandwidth-type stages, the CPU baseline (best leg of a thread sweep), and as secondary
fields the other four BASELINE configurations: configs[0]'s 3 s utterance and configs[1] (1 x 10 s) on the GPU,
configs[3] (256 utterances sharded over the ranks, strong scaling, with its own max|delta|) and configs[4] (64 streams).
max|delta| is computed from the output buffer the timed steps wrote (prefix property), not from a separate forward.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (voice type, batch per GPU, frames)
    "config1_sp_b1_3s": ("SPEECH", 1, 240),
    "config2_sp_b1_10s": ("SPEECH", 1, 800),
    "config3_si_b16_10s": ("SING", 16, 800),
    # BASELINE.json configs[3]: 256 variable-length utterances (U[2 s, 15 s]) sharded over the ranks (strong scaling)
    "config4_vo_256utt": ("VOICE", 256, None),
    # BASELINE.json configs[4]: 64 concurrent streams, one tick = 8 mel frames (100 ms; "80 ms" = 6.4 frames is not
    # frame aligned) per stream
    "config5_sp_stream64": ("SPEECH", 64, -8),
    # the same at the stated 80 ms: 6.4 frames per tick = the cyclic schedule 6 / 6 / 7 / 6 / 7 frames (400 ms per period)
    "config5_sp_stream64_80ms": ("SPEECH", 64, (6, 6, 7, 6, 7)),
    # builder-run secondary (not a BASELINE config): the generic path -- two WaveNet blocks with in-block upsampling (the
    # geometry of the golden case "blocks": C = 320 at 800 Hz, then C = 160 at 1600 Hz), the block runner's kernels
    "variant_blocks2": ("SING", 16, 800),
    # builder-run secondary, NOT float32 end to end and never the headline: the config-3 workload with the opt-in split half
    # precision of the gate and res/skip layers behind the first one (mbx_config.wn_precision: fp16-split operands, three
    # products on the 16-bit matrix pipe, float32 accumulation)
    "config3_split_f16": ("SING", 16, 800),
    # builder-run secondary: GEOMETRY_SWEEP below (each geometry at the config-3 shape)
    "geometry_sweep": ("SING", 16, 800),
}
ENGINE_KW = {"config3_split_f16": {"precision": "split_f16"}}
# builder-run secondary: the canonical L = 5 / k = 3 / d = 2^l is an inference of SURVEY.md; the
# reference's own default is 12 layers without a dilation cycle (custom_AE_layers.py:120-123, 229-233).  The config-3 shape
# (16 x 10 s) over the geometries a shipped model might have: per geometry ms per step, the kernel that ran every gate layer,
# per-layer launch times and executed fraction of the fp32 MFMA peak.
GEOMETRY_SWEEP = {
    "C320_L5_canonical": ("SING", {}),
    "C320_L12_d2048": ("SING", {"mbexwn_config:pp_mod_subnet:n_layers": 12}),
    "C320_L12_cycle8": ("SING", {"mbexwn_config:pp_mod_subnet:n_layers": 12, "mbexwn_config:pp_mod_subnet:max_log2_dilation_rate": 4}),
    "C340_L8_d128": ("VOICE", {"mbexwn_config:pp_mod_subnet:n_layers": 8}),
    "C320_L5_k5": ("SING", {"mbexwn_config:pp_mod_subnet:kernel_size": 5}),
    "C320_L5_groups2": ("SING", {"mbexwn_config:pp_mod_subnet:n_ch_groups": 2}),
    "C512_L5": ("SING", {"mbexwn_config:pp_mod_subnet:n_channels": 512}),
}
GATE_EXECUTED = {"direct": 1.0, "f23": 2.0 / 3.0, "f43": 0.5, "f43_psplit": 0.5, "f43_hsplit": 0.5, "f43_strided": 0.5,
                 "f43_strided_psplit": 0.5}
FP16_MATRIX_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16 / FP16 MFMA dense peak (spec)
WORKLOAD_OVERRIDES = {
    "variant_blocks2": {"mbexwn_config:pp_mod_subnet_upsampling_factors": [2, 1],
                        "mbexwn_config:pp_mod_subnet_channel_factors": [1, 0.5],
                        "mbexwn_config:pulse_channels": 10, "mbexwn_config:pp_mod_subnet:cond_lin_upsampling": 5},
}
DEFAULT_WORKLOAD = "config3_si_b16_10s"
FP32_MATRIX_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
NOMINAL_CLOCK_GHZ = 2.4           # the clock the guide's peak figures are priced at
HBM_PEAK_GBS = 8000.0             # same guide: HBM3E 8 TB/s (spec; 6.3 TB/s measured for a float4 copy)
DELTA_FRAMES = 80                 # prefix of the benchmark input the oracle is run on for max|delta|
DELTA_MARGIN = 12                 # frames at the end of a prefix that reach into what follows (receptive field of the path)
DELTA_TOL = 1e-4                  # tolerance of max|delta|, relative to max(1, max|oracle audio|) (tests/test_gpu_parity.py)
PORT_IN_USE_EXIT = 98             # exit status of a rank that found the rendezvous port taken (EADDRINUSE)
REJECTED_ENV = ("MBX_WG_ABLATE",)  # switches of timing experiments that produce wrong audio: never measured


def mbx_env():
    """Every MBX_* variable of the environment: they change which kernels run, so they belong to the measurement."""
    return {kk: vv for kk, vv in sorted(os.environ.items()) if kk.startswith("MBX_")}


_ENGINES = {}


def build_engine(voice, overrides=None, **engine_kw):
    """(cfg, raw weights, wavetables, dims, engine) of the canonical model of a voice type; engines are shared
    between voice types whose configuration is identical (SING == SPEECH: C = 320)."""
    from mbexwn_vocoder_amd.config import ModelDims, canonical_config
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    from mbexwn_vocoder_amd.tables import WaveTables
    from mbexwn_vocoder_amd.weights import synthetic_weights
    cfg = canonical_config(voice, **(overrides or {}))
    key = json.dumps([cfg, engine_kw], sort_keys=True, default=str)
    if key not in _ENGINES:
        dims = ModelDims(cfg)
        raw = synthetic_weights(cfg, seed=1234)          # bias 0, PReLU alpha 0.2
        wt = WaveTables(sample_rate=dims.pulse_rate, **cfg["mbexwn_config"]["wavetable_config"])
        _ENGINES[key] = (cfg, raw, wt, dims, MBExWNEngine(cfg, raw, wt, **engine_kw))
    return _ENGINES[key]


def synthetic_batch(rng, batch, frames, steps_per_frame):
    mell = np.log(np.exp(rng.normal(-5.0, 2.0, size=(batch, frames, 80))) + 1e-5)
    mell = np.clip(mell, -11.5, 2.0).astype(np.float32)
    noise = rng.normal(size=(batch, frames * steps_per_frame)).astype(np.float32)
    return mell, noise


def pmc_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); null when no profile of
    this workload has been committed.  Regenerate with scripts/profile_round.sh + scripts/summarize_profiles.py."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as fi:
            data = json.load(fi)
        return data[workload]["gate"]["hbm_bytes_per_launch"], os.path.basename(files[-1])
    except (KeyError, ValueError):
        return None, None


def cpu_baseline(cfg, raw, wt, seconds=3.0):
    """The CPU port of the reference graph timed on the host cores on a bounded sample: one 3 s utterance (the reference's
    own CPU-runnable case, configs[0]).  Two variants of the same float32 graph: the numpy port
    (oracle/mbexwn_oracle.py, matrix products on the host BLAS; 2 threads -- the reference CLI's default `-nt 2`,
    bin/resynth_mel.py:120 -- and 8) and the torch-CPU port (oracle/mbexwn_oracle_torch.py: the WaveNet, 98 % of the FLOPs,
    on torch's MKL / oneDNN ops; 2, 8 and 16 threads).  `value` is the BEST leg; every leg is listed.  No leg uses more than 16
    threads: a one-GPU box of the pool has a CPU share of 16 cores whatever os.cpu_count() says (measured there: 32 / 64 /
    256 torch threads ran at 14 / 10 / 0.2 x real time against 26-29 x at 8).  Timing protocol of the reference CLI
    (bin/resynth_mel.py:86-88): wall clock around the synthesis call only, one warm-up call first.  TensorFlow itself cannot
    run here (not installable, no network)."""
    import torch
    from oracle.mbexwn_oracle import OracleModel
    from oracle.mbexwn_oracle_torch import TorchOracleModel
    from threadpoolctl import threadpool_limits
    frames = int(round(seconds * 80))
    rng = np.random.default_rng(42)
    mel, noise = synthetic_batch(rng, 1, frames, 20)
    legs = {}
    all_cores = os.cpu_count() or 1

    def leg(om, budget_s):
        om.forward(mel[:, :16], noise[:, :320])             # warm-up (weight folding, thread start)
        times = []
        budget = time.time() + budget_s                     # bounded: a few seconds of CPU work per leg
        while len(times) < 7 and (len(times) < 2 or time.time() < budget):
            t0 = time.time()
            om.forward(mel, noise)
            times.append(time.time() - t0)
        return float(np.median(times)), len(times)

    om_np = OracleModel(cfg, raw, wt, dtype=np.float32)
    for threads in sorted({tt for tt in (2, 8, 16) if tt <= all_cores} | {min(2, all_cores)}):
        with threadpool_limits(limits=threads):
            med, runs = leg(om_np, 4.0)
        legs[f"numpy_threads_{threads}"] = {"value": frames * 300 / med, "x_realtime": frames * 300 / med / 24000.0,
                                            "cores": threads, "runs": runs, "port": "numpy"}
    om_t = TorchOracleModel(cfg, raw, wt)
    before = torch.get_num_threads()
    for threads in sorted({tt for tt in (2, 8, 16) if tt <= all_cores} | {min(2, all_cores)}):   # 2 = the reference CLI's -nt default
        torch.set_num_threads(threads)
        with threadpool_limits(limits=threads):
            med, runs = leg(om_t, 3.0)
        legs[f"torch_threads_{threads}"] = {"value": frames * 300 / med, "x_realtime": frames * 300 / med / 24000.0,
                                            "cores": threads, "runs": runs, "port": "torch"}
    torch.set_num_threads(before)
    top_name, top = max(legs.items(), key=lambda kv: kv[1]["value"])
    return {"value": top["value"], "unit": "audio samples/s", "cores": top["cores"], "kind": "port",
            "sample": f"1 utterance x {seconds:g} s (80x{frames} mel), float32 CPU port of the reference graph, best leg "
                      f"({top_name}) of a sweep over the numpy port and the torch-CPU port (2, 8, 16 threads each), "
                      f"median of {top['runs']} runs after 1 warm-up per leg, time.time() around the synthesis call only "
                      f"(reference bin/resynth_mel.py:86-88)",
            "x_realtime": top["x_realtime"], "legs": legs, "host_cores": all_cores,
            # the reference CLI's default thread count (-nt 2, bin/resynth_mel.py:120): the better of the two ports at 2 threads
            "x_realtime_nt2": max((vv["x_realtime"] for vv in legs.values() if vv["cores"] == min(2, all_cores)), default=None),
            "reference_claim": "README.md:222-223: about 2x real time on one laptop core (TF-CPU)"}


def _delta_vs_oracle(got, cfg, raw, wt, mel, noise, keep_frames):
    """max|got - oracle| over the first keep_frames frames; the oracle (float64) runs on the given mel / noise."""
    from oracle.mbexwn_oracle import OracleModel
    ref = OracleModel(cfg, raw, wt).forward(mel, noise)[:, :keep_frames * 300]
    got = np.asarray(got, dtype=np.float64)[:, :keep_frames * 300]
    return float(np.max(np.abs(got - ref))), float(np.max(np.abs(ref)))


def max_abs_delta_timed(samples, cfg, raw, wt, what):
    """Second half of BASELINE.json's metric, measured ON THE OUTPUT OF THE TIMED REGION.  samples = [(label, audio the
    timed steps wrote for one utterance (>= DELTA_FRAMES frames of it), the utterance's mel (T, 80), its noise (T*spf,))]:
    the first DELTA_FRAMES - DELTA_MARGIN frames of that audio are compared with the float64 oracle run on the
    utterance's DELTA_FRAMES-frame prefix.  Prefix property (finite receptive field + causal phase,
    tests/test_gpu_parity.py::test_full_size_prefix_property): the audio of a prefix equals the prefix of the audio except
    for the last DELTA_MARGIN frames, which reach into what follows."""
    worst, peak, labels, keep, nf = 0.0, 0.0, [], 0, 0
    for label, got, mel, noise in samples:
        nf = min(DELTA_FRAMES, mel.shape[0])
        keep = nf - DELTA_MARGIN if nf < mel.shape[0] else nf
        spf = noise.shape[0] // mel.shape[0]
        dd, pk = _delta_vs_oracle(got[None], cfg, raw, wt, mel[None, :nf], noise[None, :nf * spf], keep)
        worst, peak = max(worst, dd), max(peak, pk)
        labels.append(label)
    scale = max(1.0, peak)
    return {"max_abs_delta": worst, "max_abs_delta_tolerance": DELTA_TOL * scale, "max_abs_ref": peak,
            "max_abs_delta_ok": bool(worst <= DELTA_TOL * scale),
            "max_abs_delta_sample": f"output buffer of the timed steps ({what}): first {keep} frames ({keep / 80:g} s) of "
                                    f"{', '.join(labels)}, float32 HIP vs float64 numpy oracle (oracle/mbexwn_oracle.py) run "
                                    f"on the {nf}-frame prefixes (prefix property, margin {nf - keep} frames)"}


def max_abs_delta_full(ctx, what):
    """max|delta| of a single utterance over its WHOLE length: the buffer the timed steps wrote against the float64 oracle run
    on the whole utterance (no prefix property involved)."""
    got, mel, noise = ctx["timed_full"], ctx["mel_h"][:1], ctx["noise_h"][:1]
    dd, pk = _delta_vs_oracle(got[None], ctx["cfg"], ctx["raw"], ctx["wt"], mel, noise, mel.shape[1])
    return {"max_abs_delta_full": dd, "max_abs_delta_full_tolerance": DELTA_TOL * max(1.0, pk),
            "max_abs_delta_full_ok": bool(dd <= DELTA_TOL * max(1.0, pk)),
            "max_abs_delta_full_sample": f"output buffer of the timed steps ({what}): all {mel.shape[1]} frames "
                                         f"({mel.shape[1] / 80:g} s) of the utterance, float32 HIP vs float64 numpy oracle"}


def max_abs_delta_small(eng, cfg, raw, wt, mel_h, noise_h, torch):
    """The same comparison on a separate small launch (1 x DELTA_FRAMES frames: the small-launch kernels -- channel-split
    F(4,3) gate, narrow res/skip, split-K mel-rate convolutions), outside the timed region."""
    nf = min(DELTA_FRAMES, mel_h.shape[1])
    spf = noise_h.shape[1] // mel_h.shape[1]
    mel, noise = mel_h[:1, :nf], noise_h[:1, :nf * spf]
    got = eng.forward(torch.as_tensor(mel).cuda(), noise=torch.as_tensor(noise).cuda()).cpu().numpy()
    dd, pk = _delta_vs_oracle(got, cfg, raw, wt, mel, noise, nf)
    return {"max_abs_delta_small": dd, "max_abs_delta_small_tolerance": DELTA_TOL * max(1.0, pk),
            "max_abs_delta_small_ok": bool(dd <= DELTA_TOL * max(1.0, pk)),
            "max_abs_delta_small_sample": f"separate forward of the first {nf} frames of item 0 (small-launch kernels: "
                                          f"{eng.gate_form(1, nf)})"}


def delta_vs_reference_f32(torch):
    """max|HIP - reference float32 run| over a whole 10 s utterance (and the 3 s one): the default handle on the inputs of
    the golden cases speech800 / speech240 (tests/golden/make_reference_long.py: the reference's own MBExWN.call executed in
    float32).  The float32 phase integrator of the reference makes this distance grow with the utterance (DESIGN.md
    section 5), so it is stated per length; tolerance 1e-4 max(1, |audio|)."""
    path = os.path.join(ROOT, "tests", "golden", "reference_long_f32.npz")
    if not os.path.exists(path):
        return {}
    gold = np.load(path)
    # the golden cases' own variables (tests/helpers.py::build_case: seed 1234 with bias and PReLU jitter), not the bench's
    from mbexwn_vocoder_amd.config import ModelDims, canonical_config
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    from mbexwn_vocoder_amd.tables import WaveTables
    from mbexwn_vocoder_amd.weights import synthetic_weights
    cfg = canonical_config("SPEECH")
    dims = ModelDims(cfg)
    raw = synthetic_weights(cfg, seed=1234, bias_std=0.05, alpha_jitter=0.05)
    eng = MBExWNEngine(cfg, raw, WaveTables(sample_rate=dims.pulse_rate, **cfg["mbexwn_config"]["wavetable_config"]))
    out = {}
    for case, key in (("speech800", "max_abs_delta_vs_ref_f32"), ("speech240", "max_abs_delta_vs_ref_f32_3s")):
        mel, noise, ref = gold[f"{case}/mell"], gold[f"{case}/noise"], gold[f"{case}/audio"]
        got = eng.forward(torch.as_tensor(mel).cuda(), noise=torch.as_tensor(noise).cuda()).cpu().numpy()
        amp = float(np.abs(ref).max())
        out[key] = float(np.max(np.abs(got.astype(np.float64) - ref)))
        out[key + "_tolerance"] = DELTA_TOL * max(1.0, amp)
        out[key + "_ok"] = bool(out[key] <= out[key + "_tolerance"])
    out["max_abs_delta_vs_ref_f32_what"] = ("default handle vs the float32 run of the reference's own graph (numpy stand-in for the TF "
                                            "kernels), whole utterance: 800 frames / 240 frames, C = 320")
    return out


class Fence:
    def __init__(self, torch, dist):
        self.torch, self.dist = torch, dist

    def __call__(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, elapsed):
        if self.dist is None:
            return elapsed
        tt = self.torch.tensor([elapsed], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())


def time_steps(step, steps, warmup, fence):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize; max over ranks."""
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    return fence.max_over_ranks(time.perf_counter() - t0)


def run_batch(args, name, rank, world, fence, torch, profile, steps=None, warmup=None):
    """configs[0..2]: one padded batch per GPU.  Returns (result dict, context for the roofline / delta legs)."""
    voice, batch, frames = WORKLOADS[name]
    cfg, raw, wt, dims, eng = build_engine(voice, WORKLOAD_OVERRIDES.get(name), **ENGINE_KW.get(name, {}))
    rng = np.random.default_rng(42 + rank)
    mel_h, noise_h = synthetic_batch(rng, batch, frames, dims.wn_in_rows_per_frame)
    mel, noise = torch.as_tensor(mel_h).cuda(), torch.as_tensor(noise_h).cuda()
    out = torch.empty((batch, frames * dims.hop_size), dtype=torch.float32, device=mel.device)
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    elapsed = time_steps(lambda: eng.forward(mel, noise=noise, out=out), steps, warmup, fence)
    # what the timed steps left in `out` (the max|delta| leg looks at these rows, not at a separate forward)
    delta_items = sorted({0, batch - 1})
    keep = min(DELTA_FRAMES, frames) * dims.hop_size
    timed_out = {ii: out[ii, :keep].cpu().numpy() for ii in delta_items} if rank == 0 else {}
    # single utterances: the whole item as well (max|delta| over the full length: the phase integrator makes a contour error
    # grow with the length of the utterance, which a prefix check cannot see)
    timed_full = out[0].cpu().numpy() if rank == 0 and batch == 1 else None
    samples = world * batch * frames * dims.hop_size * steps
    res = {"workload": f"{name}: MW-{voice[:2]}-FD canonical (C={dims.wn_channels}, L={dims.wn_layers}), batch {batch} x "
                       f"{frames / 80:g} s per GPU, 80x{frames} synthetic mel, seeded synthetic weights",
           "batch_per_gpu": batch, "frames": frames, "value": samples / elapsed, "x_realtime": samples / elapsed / 24000.0,
           "ms_per_step": elapsed / steps * 1e3, "steps": steps, "scaling": "weak",
           "gate_form": eng.gate_form(batch, frames), "conv_form": eng.conv_form_info()}
    ctx = {"cfg": cfg, "raw": raw, "wt": wt, "dims": dims, "eng": eng, "mel_h": mel_h, "noise_h": noise_h, "batch": batch,
           "frames": frames, "timed_out": timed_out, "timed_full": timed_full, "delta_items": delta_items, "stages": None}
    if profile:
        # per-stage device times from HIP events on the launch stream, in a separate pass so that the events do not sit
        # inside the throughput measurement
        eng.profile_enable(True)
        for _ in range(max(3, min(steps, 10))):
            eng.forward(mel, noise=noise, out=out)
        torch.cuda.synchronize()
        stages = {kk: eng.profile_read(kk) for kk in ("gate", "gate0", "res_skip", "res_skip_f16", "frontend", "wavetable",
                                                       "start", "tail", "pqmf", "stft_filter", "overlap_add")}
        eng.profile_enable(False)
        ctx["stages"] = stages
        # the shader clock the part delivers under this workload (one-wave probe on a second stream, mbx_clock_probe): the
        # guide's fp32 MFMA peak is priced at 2.4 GHz
        reps = max(2, int(0.03 / max(elapsed / steps, 1e-4)) + 1)

        def busy():
            for _ in range(reps):
                eng.forward(mel, noise=noise, out=out)
        try:
            ctx["shader_clock_ghz_no_load"] = eng.shader_clock_under(lambda: None, seconds=0.005)   # the probe alone, for scale
            ctx["shader_clock_ghz"] = eng.shader_clock_under(busy, seconds=0.02)
        except Exception as exc:                              # noqa: BLE001 -- evidence only: never fail the measurement
            ctx["shader_clock_ghz"] = None
            ctx["shader_clock_error"] = str(exc)

    return res, ctx


def run_geometry_sweep(args, rank, world, fence, torch, steps, warmup, batch=16, frames=800):
    """GEOMETRY_SWEEP at the config-3 shape.  Per geometry: wall-clock ms per step like every other workload, then a
    separate pass with HIP events around every gate launch (mbx_profile_read_launches: per-layer device times), the kernel
    each layer ran (mbx_conv_form_info.gate_kernel) and the executed FLOPs of that form against the fp32 MFMA peak."""
    out = {}
    for name, (voice, overrides) in GEOMETRY_SWEEP.items():
        cached = set(_ENGINES)
        cfg, raw, wt, dims, eng = build_engine(voice, overrides)
        rng = np.random.default_rng(42 + rank)
        mel_h, noise_h = synthetic_batch(rng, batch, frames, dims.wn_in_rows_per_frame)
        mel, noise = torch.as_tensor(mel_h).cuda(), torch.as_tensor(noise_h).cuda()
        audio = torch.empty((batch, frames * dims.hop_size), dtype=torch.float32, device=mel.device)
        elapsed = time_steps(lambda: eng.forward(mel, noise=noise, out=audio), steps, warmup, fence)
        finite = bool(torch.isfinite(audio).all().item())
        eng.profile_enable(True)
        n_prof = 3
        for _ in range(n_prof):
            eng.forward(mel, noise=noise, out=audio)
        torch.cuda.synchronize()
        per_launch = eng.profile_read_launches("gate")
        rs_ms, rs_n = eng.profile_read("res_skip")
        g0_ms, g0_n = eng.profile_read("gate0")
        fe_ms, fe_n = eng.profile_read("frontend")
        eng.profile_enable(False)
        info = eng.conv_form_info()
        kernels = info["gate_kernels"]
        L, C, ks = dims.wn_layers, dims.wn_channels, dims.wn_kernel_size
        first = 1 if kernels and kernels[0] == "folded_start" else 0
        n_gate = L - first
        rows = batch * frames * dims.steps_per_frame
        flop_alg = 2.0 * rows * (ks * C) * (2 * C)
        layers = []
        for ll in range(first, L):
            ts = [per_launch[ff * n_gate + ll - first] for ff in range(n_prof) if ff * n_gate + ll - first < len(per_launch)]
            ms = float(np.mean(ts)) if ts else None
            ex = GATE_EXECUTED.get(kernels[ll], 1.0)
            layers.append({"layer": ll, "dilation": dims.wn_dilation(ll), "kernel": kernels[ll], "ms": round(ms, 4) if ms else None,
                           "frac": round(flop_alg * ex / (ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4) if ms else None})
        small = [ly["ms"] for ly in layers if ly["dilation"] <= 16 and ly["kernel"].startswith("f43") and ly["ms"]]
        base = float(np.mean(small)) if small else None
        gate_total = sum(ly["ms"] for ly in layers if ly["ms"])
        exec_total = sum(flop_alg * GATE_EXECUTED.get(ly["kernel"], 1.0) for ly in layers)
        samples = world * batch * frames * dims.hop_size * steps
        out[name] = {"channels": C, "layers": L, "kernel_size": ks, "groups": dims.wn_groups, "form": info["form"],
                     "ms_per_step": round(elapsed / steps * 1e3, 4), "x_realtime": round(samples / elapsed / 24000.0, 1),
                     "steps": steps, "finite": finite,
                     "gate_ms_per_step": round(gate_total, 4), "res_skip_ms_per_step": round(rs_ms / max(n_prof, 1), 4),
                     "gate0_ms": round(g0_ms / max(g0_n, 1), 4) if g0_n else None,
                     "frontend_ms_per_step": round(fe_ms / max(n_prof, 1), 4),
                     "gate_frac_executed": round(exec_total / (gate_total * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4) if gate_total else None,
                     "gate_frac_algorithmic": round(flop_alg * n_gate / (gate_total * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4) if gate_total else None,
                     "direct_form_share_of_gate_time": round(sum(ly["ms"] for ly in layers if ly["kernel"] == "direct" and ly["ms"]) / gate_total, 4) if gate_total else None,
                     "slowest_layer_over_d_le_16_layer": round(max(ly["ms"] for ly in layers if ly["ms"]) / base, 4) if base else None,
                     "gate_layers": layers}
        del eng, mel, noise, audio
        for kk in set(_ENGINES) - cached:          # the sweep's own engines (weights + images of up to 12 layers) go again
            del _ENGINES[kk]
        torch.cuda.empty_cache()
    return out


def split_report(ress, ctxs):
    """Fields of a run with the opt-in split half precision (mbx_config.wn_precision): max|delta| of the timed output, launch
    times of the split kernels and their rooflines (res/skip: HBM; gate: executed fp16 FLOPs against the fp16 matrix peak)."""
    ress.update(max_abs_delta_timed(
        [(f"item {ii}", ctxs["timed_out"][ii], ctxs["mel_h"][ii], ctxs["noise_h"][ii]) for ii in ctxs["delta_items"]],
        ctxs["cfg"], ctxs["raw"], ctxs["wt"], "config-3 batch, gate and res/skip layers in split half precision"))
    rs_ms, rs_n = ctxs["stages"]["res_skip"]              # float32 res/skip launches, if any (layers whose image is missing)
    sp_ms, sp_n = ctxs["stages"]["res_skip_f16"]          # layers 0 .. L-2: three fp16 products each
    dd = ctxs["dims"]
    rows = ctxs["batch"] * ctxs["frames"] * dd.steps_per_frame
    L, C, n_out = dd.wn_layers, dd.wn_channels, dd.wn_out_channels
    g0_n = ctxs["stages"]["gate0"][1]                     # one launch of the folded first layer per forward
    n_fwd = max(1, g0_n if g0_n else (sp_n + rs_n) // max(1, L - 1))
    split_ms = sp_ms / sp_n if sp_n else None
    flop = 3 * 2.0 * rows * C * (C + n_out)
    hbm = rows * (3 * C + 2 * n_out) * 4.0
    g_ms, g_n = ctxs["stages"]["gate"]
    ress["precision"] = ("gate layers 1..L-1 and res/skip layers 0..L-2: fp16 x 3 (hi hi + 2^-11 (hi lo' + lo' hi)) on "
                         "v_mfma_f32_16x16x32_f16, float32 accumulation; everything else float32")
    ress["gate_split_launch_ms"] = g_ms / g_n if g_n else None
    if g_n:
        gflop = 3 * 2.0 * rows * (3 * C) * (2 * C)
        ress["gate_roofline"] = {"kernel": "wn_gate_f16_kernel (direct form, three fp16 products)", "bound": "mfma",
                                 "avg_launch_ms": g_ms / g_n, "flop_executed": gflop,
                                 "achieved": gflop / (g_ms / g_n * 1e-3) / 1e12, "peak": FP16_MATRIX_PEAK_TFLOPS,
                                 "unit": "TFLOP/s", "frac": gflop / (g_ms / g_n * 1e-3) / 1e12 / FP16_MATRIX_PEAK_TFLOPS,
                                 "note": "against the fp16 matrix peak; the kernel is bound by its LDS operand reads "
                                         "(256 B per clock and CU), not by the matrix pipe"}
    ress["res_skip_ms_per_forward"] = rs_ms / n_fwd + (sp_ms / n_fwd if sp_n else 0.0)
    ress["res_skip_split_launch_ms"] = split_ms
    if split_ms:
        ress["roofline"] = {"kernel": "wn_resskip_f16_kernel", "bound": "hbm", "avg_launch_ms": split_ms,
                            "achieved": hbm / (split_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": hbm / (split_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "mfma_frac_of_fp16_peak": flop / (split_ms * 1e-3) / 1e12 / FP16_MATRIX_PEAK_TFLOPS,
                            "bytes": hbm, "flop_executed": flop,
                            "note": "algorithmic bytes: a read, h read + written, output accumulator read + written"}


def roofline_blocks(ctx):
    """The generic path (several WaveNet blocks, block runner of csrc/mbx_api.hip): all gate launches of a step against the
    fp32 MFMA peak -- block b runs L full gate layers on T * spf_b rows with C_b channels (no folded first layer) -- and
    the per-stage device times of a step."""
    dims, eng, batch, frames, stages = ctx["dims"], ctx["eng"], ctx["batch"], ctx["frames"], ctx["stages"]
    info = eng.conv_form_info()
    executed = {"direct": 1.0, "f23": 2.0 / 3.0, "f43": 0.5}[info["form"]]
    if info["form"] == "f43" and (256 + dims.cond_lin_upsampling - 2) // dims.cond_lin_upsampling + 2 > 56:
        executed = 1.0          # the F(4,3) kernel's conditioning tile holds 56 rows: finer conditioning runs the direct form
    L, ks = dims.wn_layers, dims.wn_kernel_size
    rpf = dims.wn_in_rows_per_frame
    flop_alg, rs_flop, geometry = 0.0, 0.0, []
    for C, ups in zip(dims.wn_block_channels, dims.wn_block_ups):
        rows = batch * frames * rpf
        flop_alg += L * 2.0 * rows * (ks * C) * (2 * C)
        rs_flop += 2.0 * rows * C * ((L - 1) * 2 * C + C)            # un-folded res/skip layers: C -> 2C, last one C -> C
        geometry.append({"channels": C, "rows_per_frame": rpf, "upsampling_behind": ups})
        rpf *= ups
    gate_ms, gate_n = stages["gate"]
    rs_ms, rs_n = stages["res_skip"]
    n_fwd = max(1, gate_n // (L * len(dims.wn_block_channels)))
    gate_s, rs_s = gate_ms / n_fwd * 1e-3, rs_ms / n_fwd * 1e-3
    per_step = {kk: stages[kk][0] / n_fwd for kk in stages if stages[kk][1]}
    return {"bound": "mfma", "kernel": "block runner: wn_gate_winograd4w_kernel per block and layer where its image fits "
                                       "(conv1d_mfma_dma_kernel<EPI_GATE> otherwise), wn_resskip_kernel<un-folded>",
            "blocks": geometry, "gate_launches_per_step": L * len(dims.wn_block_channels),
            "achieved": flop_alg * executed / gate_s / 1e12, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": flop_alg * executed / gate_s / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
            "frac_algorithmic": flop_alg / gate_s / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
            "gate_ms_per_step": gate_s * 1e3, "res_skip_ms_per_step": rs_s * 1e3,
            "res_skip_frac": rs_flop / rs_s / 1e12 / FP32_MATRIX_PEAK_TFLOPS if rs_s else None,
            "stage_ms_per_step": per_step, "traffic": None}


def roofline(ctx, workload):
    """Dominant kernel (dilated conv + gate: 75 % of the WaveNet FLOPs) against the fp32 MFMA peak, and the
    bandwidth-type stages against the HBM peak.  Algorithmic work per launch: DESIGN.md section 4."""
    dims, eng, batch, frames, stages = ctx["dims"], ctx["eng"], ctx["batch"], ctx["frames"], ctx["stages"]
    form = eng.gate_form(batch, frames)
    executed = {"direct": 1.0, "winograd_f23": 2.0 / 3.0, "winograd_f43": 0.5, "winograd_f43_psplit": 0.5, "winograd_f43_hsplit": 0.5}[form]
    kernel = {"direct": "conv1d_mfma_dma_kernel<EPI_GATE> (dilated conv k=3 C->2C + cond + tanh*sigmoid)",
              "winograd_f23": "wn_gate_winograd2w_kernel (dilated conv k=3 C->2C in Winograd F(2,3) form on v_mfma_f32_16x16x4_f32, wave-granular tiles + cond + tanh*sigmoid)",
              "winograd_f43": "wn_gate_winograd4w_kernel (dilated conv k=3 C->2C in Winograd F(4,3) form on v_mfma_f32_16x16x4_f32 + cond + tanh*sigmoid)",
              "winograd_f43_psplit": "wn_gate_winograd4p_kernel (same, 128-row blocks whose waves split the six products: same bits, finer units)",
              "winograd_f43_hsplit": "wn_gate_winograd4h_kernel (same, product-split blocks of half a column tile: same bits, finer still)"}[form]
    C, ks, L = dims.wn_channels, dims.wn_kernel_size, dims.wn_layers
    rows = batch * frames * dims.steps_per_frame
    gate_ms, gate_n = stages["gate"]
    gate_s = gate_ms / max(gate_n, 1) * 1e-3
    flop_alg = 2.0 * rows * (ks * C) * (2 * C)               # direct convolution: the algorithmic FLOPs of one launch
    flop_exec = flop_alg * executed                           # what the matrix cores execute in this form
    rs_ms, rs_n = stages["res_skip"]
    rs_flop = 2.0 * rows * C * (C + dims.wn_out_channels)     # folded res/skip layer
    traffic, traffic_src = pmc_traffic(workload)
    T, B = frames, batch
    n_out, M, hop, ppf = dims.wn_out_channels, dims.subbands, dims.hop_size, dims.pulse_per_frame
    spf, nceps, win = dims.steps_per_frame, dims.n_ceps, dims.stft_win
    stage_bytes = {            # algorithmic bytes of one launch (inputs read once + outputs written once)
        # layer 0 with the start convolution folded in (wn_gate0.hip): pulse + noise + conditioning -> a0 (+ 16 channels)
        "gate0": B * T * (ppf * 4 + spf * 4 + 2 * 2 * C * 4 + spf * (C + 16) * 4),
        "start": B * T * (ppf * 4 + spf * 4 + spf * C * 4),                       # pulse + noise -> h (un-folded graph only)
        "tail": B * T * spf * (C * 4 + 2 * n_out * 4 + M * 4),                    # a + output accumulator r/w -> sub-bands
        "pqmf": B * T * (spf * M * 4 + hop * 4),                                   # sub-bands -> excitation
        "stft_filter": B * T * (hop * 4 + nceps * 4 + ppf * 4 + win * 4),          # excitation + cepstrum + f0 -> frames
        "overlap_add": B * T * (win * 4 + hop * 4),                                # frames -> audio
        "wavetable": B * T * (ppf * 4 * 2),                                        # f0 -> pulse (phase + lookup)
    }
    # what the measurements say limits each of these stages (NOTEBOOK.md R4 section 9, profiles/README.md); the GB/s figure is
    # their algorithmic traffic over the launch time whatever the limiter is
    limiter = {
        "gate0": "vector + transcendental issue of the gate activation, which shares the SIMD with its K = 24 fp32 MFMAs "
                 "(ablation: arithmetic 81 us, block prologue 49, stores 29 of 158 us at 16 x 10 s)",
        "tail": "HBM latency under 3 blocks per CU: a wave owns 16 rows over all channels (16x16x4 MFMAs), the 40 KB weight image is "
                "staged once per 64-row block by LDS-DMA, all activations of the rows are requested at once (round 5: 129 -> 87-91 us; "
                "one kernel and one summation order at every launch size)",
        "stft_filter": "vector instructions of three FFT-1024 per frame: one wave per frame, radix 16 / 16 / 4 in registers, "
                       "2 900 instructions per frame at two waves per SIMD (round 3: block per frame, 4 x 1 785, 196 us)",
        "wavetable": "the float32 phase chain: 1 000 dependent adds per chunk, kept in the reference's order",
        "pqmf": "latency of a rows x 135 x 15 product per launch",
        "overlap_add": "HBM",
        "start": "HBM",
    }
    stage_list = []
    for name, nbytes in stage_bytes.items():
        ms, cnt = stages[name]
        if not cnt:
            continue                                  # stage not on this graph (start: folded into layer 0; gate0: un-folded)
        avg_s = ms / max(cnt, 1) * 1e-3
        gbs = nbytes / avg_s / 1e9 if avg_s > 0 else None
        stage_list.append({"stage": name, "bound": "hbm", "avg_launch_ms": avg_s * 1e3, "bytes": nbytes,
                           "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": gbs / HBM_PEAK_GBS if gbs else None, "limited_by": limiter.get(name, "HBM")})
    fe_ms, fe_n = stages["frontend"]
    return {"bound": "mfma", "kernel": kernel,
            "achieved": flop_exec / gate_s / 1e12, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": flop_exec / gate_s / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
            "achieved_algorithmic": flop_alg / gate_s / 1e12,
            "frac_algorithmic": flop_alg / gate_s / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
            "flop_per_launch": flop_alg, "mfma_flop_executed_per_launch": flop_exec,
            "note": "achieved / frac = FLOPs the matrix cores execute (Winograd F(4,3): 1/2, F(2,3): 2/3 of the direct "
                    "convolution's 2*rows*3C*2C) / launch time: <= 1, comparable with MfmaUtil in profiles/; "
                    "*_algorithmic = the direct convolution's FLOPs / launch time (can exceed 1)",
            "avg_launch_ms": gate_s * 1e3, "launches_timed": gate_n,
            "shader_clock_ghz_under_load": ctx.get("shader_clock_ghz"),
            "shader_clock_ghz_probe_alone": ctx.get("shader_clock_ghz_no_load"),
            "frac_at_delivered_clock": (flop_exec / gate_s / 1e12 / (FP32_MATRIX_PEAK_TFLOPS * ctx["shader_clock_ghz"] / NOMINAL_CLOCK_GHZ)
                                        if ctx.get("shader_clock_ghz") else None),
            "clock_note": "peak = 157.3 TFLOP/s at the nominal 2.4 GHz; shader_clock_ghz_under_load = s_memtime cycles per s_memrealtime "
                          "tick of a one-wave probe that runs beside the timed forward passes on a second stream",
            "launches_per_step": L - 1 if eng.folds_start else L,
            "first_layer": ("start convolution folded into layer 0: a K=24 contraction of the excitation (wn_gate0_kernel, "
                            "listed under stages), not one of these launches") if eng.folds_start else "same kernel",
            "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)",
            "traffic_source": traffic_src,
            "res_skip": {"avg_launch_ms": rs_ms / max(rs_n, 1), "flop_per_launch": rs_flop,
                         "achieved": rs_flop / (rs_ms / max(rs_n, 1) * 1e-3) / 1e12 if rs_ms else None,
                         "frac": rs_flop / (rs_ms / max(rs_n, 1) * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS if rs_ms else None,
                         "unit": "TFLOP/s", "launches_per_step": L - 1},
            "frontend_ms_per_step": fe_ms / max(fe_n, 1),
            "stages": stage_list}


def run_sharded(args, name, rank, world, dist, fence, torch, steps=None, warmup=None, check_delta=False):
    """configs[3]: every rank sees the same seeded list of utterance lengths, ShardedSynthesizer takes its LPT shard,
    stages the padded micro-batches in HBM once, and every step runs them and gathers the audio on rank 0 (RCCL, device
    tensors: no host copy between the forward pass and the collective; chunks of the shard are handed to the collective
    as soon as their micro-batches are packed, so the transfer runs under the next forward pass).  The line carries the
    step's compute and exposed-gather time of every rank."""
    from mbexwn_vocoder_amd.sharding import ShardedSynthesizer
    voice, n_utt, _ = WORKLOADS[name]
    cfg, raw, wt, dims, eng = build_engine(voice)
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    rng = np.random.default_rng(4242)
    lengths = [int(vv) for vv in rng.integers(160, 1201, size=n_utt)]          # 2 s .. 15 s in frames
    syn = ShardedSynthesizer(lambda mel, nfr, noise: eng.forward(mel, n_frames=nfr, noise=noise),
                             dims.hop_size, dims.steps_per_frame, rank=rank, world_size=world, max_batch=16,
                             max_padded_frames=16 * 1200, device=torch.device("cuda", torch.cuda.current_device()),
                             force_collective=dist is not None)
    from mbexwn_vocoder_amd.sharding import lpt_partition, plan_stats
    mine = set(lpt_partition(lengths, world)[rank])
    mels, noises = [], []
    for ii, ll in enumerate(lengths):       # only this rank's utterances are generated; the others are placeholders
        if ii in mine:
            mm, nn = synthetic_batch(np.random.default_rng(1000 + ii), 1, ll, dims.steps_per_frame)
            mels.append(mm[0])
            noises.append(nn[0])
        else:
            mels.append(np.zeros((ll, 80), dtype=np.float32))
            noises.append(np.zeros((ll * dims.steps_per_frame,), dtype=np.float32))
    plan = syn.stage(mels, noises)
    last = {}

    def step():
        last["res"] = syn.run_staged(plan, gather=args.gather)

    elapsed = time_steps(step, steps, warmup, fence)
    # compute / exposed gather time of the last timed step, per rank (events on the launch stream, sharding.py)
    timing = last["res"].timing
    mine_ms = [float(timing.get("compute_ms", 0.0)), float(timing.get("gather_ms", 0.0))]
    per_rank = [mine_ms]
    if dist is not None and world > 1:
        tt = torch.tensor(mine_ms, dtype=torch.float64, device="cuda")
        every = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        per_rank = [[float(vv) for vv in ee.cpu()] for ee in every]
    samples = sum(lengths) * dims.hop_size * steps
    padded = sum(int(bb[1].shape[0]) * int(bb[1].shape[1]) for bb in plan["batches"])
    delta = {}
    if rank == 0 and check_delta:
        # one utterance of this rank's shard, as the timed steps (forward + gather) left it, against the oracle on its prefix;
        # the comparison itself (seconds of CPU work) runs when the caller asks for it -- behind every timed region of the
        # run, so that no GPU measurement starts on a chip that idled through an oracle run
        pick = max(mine, key=lambda ii: (lengths[ii], -ii))                 # the longest one: first micro-batch, large launch
        got = last["res"].item(pick)[:DELTA_FRAMES * dims.hop_size].cpu().numpy()
        delta = {"_delta_later": lambda: max_abs_delta_timed(
            [(f"utterance {pick} ({lengths[pick]} frames)", got, mels[pick], noises[pick])],
            cfg, raw, wt, f"forward + gather ({args.gather}) of the sharded run")}
    return {**delta, "workload": f"{name}: MW-{voice[:2]}-FD canonical (C={dims.wn_channels}), {n_utt} utterances U[2 s,15 s] = "
                        f"{sum(lengths) / 80:.0f} s of audio, LPT-sharded over {world} ranks (ShardedSynthesizer), padded "
                        f"micro-batches <= 16 items, device-resident chunked asynchronous gather ({args.gather}) of the audio each step",
            "gather": args.gather, "gather_chunks": int(timing.get("chunks", 0)),
            "compute_ms_per_rank": [round(vv[0], 3) for vv in per_rank],
            "gather_exposed_ms_per_rank": [round(vv[1], 3) for vv in per_rank],
            "value": samples / elapsed, "x_realtime": samples / elapsed / 24000.0, "ms_per_step": elapsed / steps * 1e3,
            "steps": steps, "scaling": "strong", "parallelism": f"utterance-sharded x{world}",
            "padding_overhead": padded / max(1, sum(lengths[ii] for ii in mine)),
            # what the partition allows at 1 .. 8 ranks (computed, not measured: every rank derives the same plan):
            # LPT makespan / mean load, and the bytes of the gather left behind a rank's last forward
            "partition": {str(nn): {"lpt_makespan_over_mean_load": round(st["imbalance"], 5),
                                    "micro_batches_per_rank": st["micro_batches"],
                                    "exposed_gather_bytes_per_rank": st["exposed_gather_bytes"]}
                          for nn, st in ((nn, plan_stats(lengths, nn, dims.hop_size, 16, 16 * 1200)) for nn in (1, 2, 4, 8))}}


def run_streaming(args, name, rank, world, fence, torch, steps=None, warmup=None):
    """configs[4]: steady-state tick of the streaming driver -- every stream advances by `chunk` frames.  `value`
    counts the emitted audio only, inputs resident in HBM; the tick is timed on the device with HIP events (events on
    the launch stream around each tick) and, separately, host-inclusive through the Python driver."""
    from mbexwn_vocoder_amd.streaming import StreamingSynthesizer
    voice, n_streams, chunk = WORKLOADS[name]
    schedule = list(chunk) if isinstance(chunk, tuple) else [-chunk]
    cfg, raw, wt, dims, eng = build_engine(voice)
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    if len(schedule) > 1:                            # whole periods of the schedule, so that the mean tick is the stated one
        steps = max(len(schedule), steps // len(schedule) * len(schedule))
    syn = StreamingSynthesizer(eng, chunk_frames=schedule if len(schedule) > 1 else schedule[0])
    syn.time_device = True
    # ticks before the timed ones: the growing left context, then (schedules) one period in which every phase's steady tick
    # is recorded and one in which its graph is captured
    # (+ three periods more: a one-off stall of tens of ms has been seen within a few ticks behind the captures -- runtime
    # housekeeping after graph instantiation, not a property of the steady state)
    lead_ticks = 4 if len(schedule) == 1 else 6 * len(schedule)
    n_ticks = lead_ticks + warmup + steps
    emitted = [schedule[ii % len(schedule)] for ii in range(n_ticks + 1)]
    chunk = float(np.mean(emitted[lead_ticks + warmup:lead_ticks + warmup + steps]))     # frames per timed tick (mean)
    total = int(sum(emitted)) + syn.right + 8
    for sid in range(n_streams):
        syn.open(sid)
        mm, nn = synthetic_batch(np.random.default_rng(1000 * rank + sid), 1, total, dims.steps_per_frame)
        syn.push(sid, mm[0], nn[0])
    dev_ms, host_ms, frames, act_frames = [], [], [], []
    for tick in range(n_ticks):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        res = syn.tick()
        torch.cuda.synchronize()
        assert len(res) == n_streams
        if tick >= lead_ticks + warmup:
            host_ms.append((time.perf_counter() - t1) * 1e3)
            dev_ms.append(syn.last_tick_device_ms)
            frames.append(syn.last_tick_frames)
            act_frames.append(syn.last_tick_wavenet_frames)
    fence()
    elapsed = fence.max_over_ranks(float(np.sum(dev_ms)) * 1e-3)       # device time of the timed ticks
    samples = world * n_streams * chunk * dims.hop_size * steps
    how = ("steady ticks are one replayed hipGraph (upload of the new frames + window advance + forward + read-back of the "
           "chunk)") if len(schedule) == 1 else (f"cyclic tick schedule {schedule} frames (mean {chunk:g}): the window geometry "
           "repeats with the schedule's period, every phase has its own captured hipGraph, all working on one device-resident "
           "window")
    return {"workload": f"{name}: MW-SP-FD canonical, {n_streams} streams per GPU, tick = {chunk:g} frames "
                        f"({chunk * 12.5:g} ms) per stream, look-ahead {syn.right * 12.5:g} ms, carried phase state, "
                        f"bit-equal to offline synthesis; {how}; value = emitted audio / device time of the ticks "
                        f"(HIP events around each tick's launches, copies included), host-inclusive latency beside it",
            "tick_ms": chunk * 12.5, "conv_form": eng.conv_form_info()["stream_form"],
            "value": samples / elapsed, "x_realtime": samples / elapsed / 24000.0, "ms_per_step": elapsed / steps * 1e3,
            "steps": steps, "scaling": "weak",
            "ticks_replayed_as_graph": int(syn.graph_ticks), "ticks_total": int(n_ticks),
            "tick_ms_device_p50": float(np.percentile(dev_ms, 50)), "tick_ms_device_p99": float(np.percentile(dev_ms, 99)),
            "tick_ms_host_inclusive_p50": float(np.percentile(host_ms, 50)),
            "tick_ms_host_inclusive_p99": float(np.percentile(host_ms, 99)),
            # window frames (mel-rate stages, phase) and frames of the active region (WaveNet ... overlap-add) per emitted frame
            "frames_computed_per_emitted_frame": float(np.mean(frames)) / (n_streams * chunk),
            "wavenet_frames_per_emitted_frame": float(np.mean(act_frames)) / (n_streams * chunk)}


def visible_gpu_count():
    """GPUs this process may use, WITHOUT initialising HIP in this process (the parent of the ranks must never touch the
    GPU): the visibility variables when they are set, else torch.cuda.device_count() evaluated in a child process."""
    counts = []
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = os.environ.get(var)
        if val is not None:
            counts.append(len([tok for tok in val.split(",") if tok.strip() != ""]))
    if counts:
        return min(counts)
    res = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                         text=True, timeout=600)
    try:
        return int(res.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        raise SystemExit(f"bench.py: cannot count the GPUs: {res.stderr[-400:]}")


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (one per GPU, RCCL).
    This parent never touches the GPU (visible_gpu_count) and never re-execs; it polls the children, and when one of
    them fails it ends the others (a rank that died in front of a collective would otherwise leave them waiting for
    RCCL's timeout) and exits non-zero.  The rendezvous port is --master-port / MASTER_PORT when given; else a free port
    is probed, and a rank 0 that cannot bind it (taken in between) makes the whole attempt be repeated on another one."""
    have = visible_gpu_count()
    if have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible")
    fixed = args.master_port or int(os.environ.get("MASTER_PORT", "0"))
    for attempt in range(3):
        port = fixed
        if not port:
            with socket.socket() as ss:
                ss.bind(("127.0.0.1", 0))
                port = ss.getsockname()[1]
        procs = []
        for rank in range(args.gpus):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                       LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        codes = [None] * len(procs)
        while any(cc is None for cc in codes):
            for ii, pp in enumerate(procs):
                if codes[ii] is None:
                    codes[ii] = pp.poll()
            if any(cc not in (None, 0) for cc in codes):
                for ii, pp in enumerate(procs):                 # exactly the children started above, by PID
                    if codes[ii] is None:
                        pp.terminate()
                for ii, pp in enumerate(procs):
                    if codes[ii] is None:
                        try:
                            codes[ii] = pp.wait(timeout=30)
                        except subprocess.TimeoutExpired:
                            pp.kill()
                            codes[ii] = pp.wait()
                break
            time.sleep(0.2)
        worst = max(abs(cc) for cc in codes)
        if worst == PORT_IN_USE_EXIT and not fixed and attempt < 2:
            continue                                            # rank 0 lost the race for the probed port
        raise SystemExit(worst)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary workloads (the other four configs) and the max|delta| legs: profiling runs")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-spawned ranks (default: probe a free one)")
    ap.add_argument("--gather", default="rank0", choices=["rank0", "all", "none"],
                    help="result gather of the sharded workload: on rank 0 (default), on every rank, or none (local shards)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even for one rank: exercises the N>1 code path on a 1-GPU box")
    args = ap.parse_args()
    args.gather = None if args.gather == "none" else args.gather
    if args.gpus < 1 or args.steps < 1:
        raise SystemExit("--gpus and --steps must be >= 1")
    for kk in REJECTED_ENV:
        if kk in os.environ:
            raise SystemExit(f"{kk} is set: it selects a timing experiment that produces wrong audio; refusing to measure")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product path has no CPU fallback)")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: local GPU {local_rank} does not exist ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        except Exception as exc:                              # noqa: BLE001 -- the store's errors have no common type
            if "address already in use" in str(exc).lower() or "EADDRINUSE" in str(exc):
                raise SystemExit(PORT_IN_USE_EXIT)
            raise
    fence = Fence(torch, dist)

    voice, batch, frames = WORKLOADS[args.workload]
    line = {"metric": "24 kHz audio samples/sec (whole job; x real-time = value / 24000) + max|delta| vs the CPU oracle"}
    ctx = None
    if args.workload == "geometry_sweep":
        sweep = run_geometry_sweep(args, rank, world, fence, torch, steps=args.steps, warmup=args.warmup)
        if rank == 0:
            worst = min(sweep, key=lambda kk: sweep[kk]["x_realtime"])
            print(json.dumps({"metric": "geometry sweep at the config-3 shape (16 x 10 s per GPU): 24 kHz audio samples/s of the slowest geometry",
                              "value": sweep[worst]["x_realtime"] * 24000.0, "unit": "audio samples/s", "x_realtime": sweep[worst]["x_realtime"],
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sweep[worst]["ms_per_step"],
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": f"geometry_sweep, slowest: {worst}", "parallelism": f"utterance-sharded x{world}"},
                              "geometry_sweep": sweep, "env": mbx_env()}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if frames is None:
        main_res = run_sharded(args, args.workload, rank, world, dist, fence, torch, check_delta=not args.no_secondary)
        later = main_res.pop("_delta_later", None)
        if later is not None:
            main_res.update(later())
    elif isinstance(frames, tuple) or frames < 0:
        main_res = run_streaming(args, args.workload, rank, world, fence, torch)
    else:
        main_res, ctx = run_batch(args, args.workload, rank, world, fence, torch, profile=True)

    secondary = {}
    if args.workload == DEFAULT_WORKLOAD and not args.no_secondary:
        # the other four BASELINE configurations with bounded step counts (the whole default run stays within a minute or
        # two): configs[1] at the same step count; configs[0]'s utterance size on the GPU; configs[3] (one step = 2 181 s of
        # audio) with its own max|delta| on a C = 340 utterance of the timed, gathered output; configs[4] streaming ticks
        # (single utterances: at least 100 steps, so that the wall-clock bracket of a step that lasts 0.4-0.7 ms is not
        # dominated by the two synchronisations at its ends -- 20 steps read 5 % long against event timing)
        short_steps = max(args.steps, 100)
        short_warm = max(args.warmup, 20)
        # every timed region first, every comparison with the oracle (seconds of CPU work each) afterwards: a measurement
        # that starts right behind an oracle run starts on an idle chip whose clocks have dropped (seen: the 3 s utterance at
        # 0.72 instead of 0.36 ms per step behind the 10 s oracle run)
        res2, ctx2 = run_batch(args, "config2_sp_b1_10s", rank, world, fence, torch, profile=False, steps=short_steps, warmup=short_warm)
        secondary["config2_sp_b1_10s"] = res2
        res1, ctx1 = run_batch(args, "config1_sp_b1_3s", rank, world, fence, torch, profile=False, steps=short_steps, warmup=short_warm)
        secondary["config1_sp_b1_3s"] = res1
        # the generic path (two WaveNet blocks, in-block upsampling): builder-run secondary, not a BASELINE config
        resb, ctxb = run_batch(args, "variant_blocks2", rank, world, fence, torch, profile=True, steps=min(args.steps, 5), warmup=1)
        secondary["variant_blocks2"] = resb
        # opt-in split half precision of the res/skip layers: NOT the float32 path, reported beside it with its own max|delta|
        ress, ctxs = run_batch(args, "config3_split_f16", rank, world, fence, torch, profile=True, steps=min(args.steps, 5), warmup=1)
        secondary["config3_split_f16"] = ress
        secondary["config4_vo_256utt"] = run_sharded(args, "config4_vo_256utt", rank, world, dist, fence, torch,
                                                     steps=min(args.steps, 3), warmup=1, check_delta=True)
        secondary["config5_sp_stream64"] = run_streaming(args, "config5_sp_stream64", rank, world, fence, torch,
                                                         steps=max(20, min(args.steps, 50)), warmup=3)
        secondary["config5_sp_stream64_80ms"] = run_streaming(args, "config5_sp_stream64_80ms", rank, world, fence, torch,
                                                              steps=max(20, min(args.steps, 50)), warmup=5)
        # the geometry sweep (builder-run secondary): the config-3 shape over WaveNet geometries other than the inferred one
        sweep = run_geometry_sweep(args, rank, world, fence, torch, steps=min(args.steps, 5), warmup=2)
        if rank == 0:
            secondary["geometry_sweep"] = sweep
        # config 2's input from the reference's own float32 run (tests/golden/reference_long_f32.npz: 80 x 800 mel, MW-SP-FD):
        # the distance to what a user of the TF-CPU path gets for a 10 s utterance (float32 emulation of the reference graph)
        ref32 = delta_vs_reference_f32(torch) if rank == 0 else {}
        if rank == 0:
            res2.update(ref32)
            res2.update(max_abs_delta_timed([("item 0", ctx2["timed_out"][0], ctx2["mel_h"][0], ctx2["noise_h"][0])],
                                            ctx2["cfg"], ctx2["raw"], ctx2["wt"], "config 2 batch"))
            res2.update(max_abs_delta_full(ctx2, "config 2 batch"))
            res1.update(max_abs_delta_timed([("item 0", ctx1["timed_out"][0], ctx1["mel_h"][0], ctx1["noise_h"][0])],
                                            ctx1["cfg"], ctx1["raw"], ctx1["wt"], "config 1 batch"))
            res1.update(max_abs_delta_full(ctx1, "config 1 batch"))
            resb.update(max_abs_delta_timed([("item 0", ctxb["timed_out"][0], ctxb["mel_h"][0], ctxb["noise_h"][0])],
                                            ctxb["cfg"], ctxb["raw"], ctxb["wt"], "two-block variant batch"))
            resb["roofline"] = roofline_blocks(ctxb)
            split_report(ress, ctxs)
        later = secondary["config4_vo_256utt"].pop("_delta_later", None)
        if later is not None:
            secondary["config4_vo_256utt"].update(later())

    if rank == 0:
        line.update({
            "value": main_res["value"], "unit": "audio samples/s", "x_realtime": main_res["x_realtime"],
            "x_realtime_per_gpu": main_res["x_realtime"] / world, "n_gpus": world, "steps": main_res["steps"],
            "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"], "higher_is_better": True,
            "scaling": main_res["scaling"], "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {kk: vv for kk, vv in main_res.items()
                       if kk not in ("value", "x_realtime", "ms_per_step", "steps", "scaling") and not kk.startswith("_")},
            "env": mbx_env()})
        line["config"]["parallelism"] = main_res.get("parallelism", f"utterance-sharded x{world}")
        if ctx is not None:
            if args.workload in ENGINE_KW:                  # the opt-in split precision as the main workload (profiling runs)
                main_res_extra = {}
                split_report(main_res_extra, ctx)
                line["config"].update({kk: vv for kk, vv in main_res_extra.items() if not kk.startswith("max_abs_delta")})
                line["roofline"] = main_res_extra.get("gate_roofline")
            else:
                line["roofline"] = roofline_blocks(ctx) if args.workload in WORKLOAD_OVERRIDES else roofline(ctx, args.workload)
            if not args.no_secondary:
                line.update(max_abs_delta_timed(
                    [(f"item {ii}", ctx["timed_out"][ii], ctx["mel_h"][ii], ctx["noise_h"][ii]) for ii in ctx["delta_items"]],
                    ctx["cfg"], ctx["raw"], ctx["wt"], f"{args.workload} batch, kernels: {main_res['gate_form']}"))
                line.update(max_abs_delta_small(ctx["eng"], ctx["cfg"], ctx["raw"], ctx["wt"], ctx["mel_h"], ctx["noise_h"], torch))
        if secondary:
            line["secondary"] = secondary
            # the five BASELINE configs (+ the two builder secondaries) at a glance, as scalars of `config` -- the part of the
            # line a record keeper that drops nested objects and long strings still holds
            summary = compact_summary(line, main_res, secondary)
            line["config"].update(summary)
            line["config"]["all"] = dict(summary)
        if not args.no_cpu_baseline:
            cfg, raw, wt = build_engine(voice)[:3]
            line["cpu_baseline"] = cpu_baseline(cfg, raw, wt)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # a parity miss anywhere in the line fails the run: every max_abs_delta*_ok, main and secondary
        bad = [path for path, ok in _delta_flags(line) if ok is False]
        if bad:
            print(f"bench.py: max|delta| out of tolerance: {', '.join(bad)}", file=sys.stderr)
            raise SystemExit(3)


def _delta_ratios(node, path="line"):
    """(path, max_abs_delta / its tolerance) of every max|delta| figure anywhere in the line."""
    found = []
    if isinstance(node, dict):
        for kk, vv in node.items():
            if kk in ("max_abs_delta", "max_abs_delta_small", "max_abs_delta_full") and isinstance(node.get(kk + "_tolerance"), float):
                found.append((f"{path}.{kk}", vv / node[kk + "_tolerance"]))
            elif isinstance(vv, dict):
                found.extend(_delta_ratios(vv, f"{path}.{kk}"))
    return found


def compact_summary(line, main_res, secondary):
    """Short numeric keys: ms per step (per tick for config 5, device p50) of every workload of the default run, the
    max|delta| of each against the float64 oracle, and whether all of them are inside the tolerance."""
    def ms(name):
        return round(secondary[name]["ms_per_step"], 4) if name in secondary else None

    def dd(name):
        vv = secondary.get(name, {}).get("max_abs_delta")
        return float(f"{vv:.3e}") if vv is not None else None
    ratios = _delta_ratios(line)
    flags = [ok for _, ok in _delta_flags(line)]
    c5a, c5b = secondary.get("config5_sp_stream64", {}), secondary.get("config5_sp_stream64_80ms", {})
    return {"c1_ms": ms("config1_sp_b1_3s"), "c2_ms": ms("config2_sp_b1_10s"), "c3_ms": round(main_res["ms_per_step"], 4),
            "c4_ms": ms("config4_vo_256utt"),
            "c5_80ms_tick_ms": round(c5b["tick_ms_device_p50"], 4) if c5b else None,
            "c5_80ms_tick_host_ms": round(c5b["tick_ms_host_inclusive_p50"], 4) if c5b else None,
            "c5_100ms_tick_ms": round(c5a["tick_ms_device_p50"], 4) if c5a else None,
            "c5_100ms_tick_host_ms": round(c5a["tick_ms_host_inclusive_p50"], 4) if c5a else None,
            "split_f16_ms": ms("config3_split_f16"), "blocks2_ms": ms("variant_blocks2"),
            "c1_delta": dd("config1_sp_b1_3s"), "c2_delta": dd("config2_sp_b1_10s"),
            "c1_delta_full_3s": float(f"{secondary['config1_sp_b1_3s']['max_abs_delta_full']:.3e}") if "max_abs_delta_full" in secondary.get("config1_sp_b1_3s", {}) else None,
            "c2_delta_full_10s": float(f"{secondary['config2_sp_b1_10s']['max_abs_delta_full']:.3e}") if "max_abs_delta_full" in secondary.get("config2_sp_b1_10s", {}) else None,
            "c3_delta": float(f"{line['max_abs_delta']:.3e}") if "max_abs_delta" in line else None,
            "c4_delta": dd("config4_vo_256utt"), "split_f16_delta": dd("config3_split_f16"),
            "c2_delta_vs_ref_f32_10s": float(f"{secondary['config2_sp_b1_10s']['max_abs_delta_vs_ref_f32']:.3e}") if "max_abs_delta_vs_ref_f32" in secondary.get("config2_sp_b1_10s", {}) else None,
            "c1_delta_vs_ref_f32_3s": float(f"{secondary['config2_sp_b1_10s']['max_abs_delta_vs_ref_f32_3s']:.3e}") if "max_abs_delta_vs_ref_f32_3s" in secondary.get("config2_sp_b1_10s", {}) else None,
            **sweep_summary(secondary.get("geometry_sweep")),
            "deltas_ok": bool(flags) and all(ok is True for ok in flags), "deltas_checked": len(flags),
            "worst_delta_over_tol": round(max(rr for _, rr in ratios), 4) if ratios else None}


def sweep_summary(sweep):
    """Short scalars of the geometry sweep: ms per step of the reference's default depth (12 layers, d <= 2048) and of the
    slowest geometry, the lowest executed fraction of the fp32 MFMA peak over the gate launches of a geometry, and the worst
    ratio of a gate layer's time to the d <= 16 F(4,3) layers of the same model."""
    if not sweep:
        return {}
    worst = min(sweep, key=lambda kk: sweep[kk]["x_realtime"])
    fracs = [vv["gate_frac_executed"] for vv in sweep.values() if vv.get("gate_frac_executed")]
    ratios = [vv["slowest_layer_over_d_le_16_layer"] for kk, vv in sweep.items() if vv.get("slowest_layer_over_d_le_16_layer") and vv["kernel_size"] == 3]
    return {"sweep_L12_d2048_ms": sweep.get("C320_L12_d2048", {}).get("ms_per_step"),
            "sweep_L12_d2048_xrt": sweep.get("C320_L12_d2048", {}).get("x_realtime"),
            "sweep_slowest_ms": sweep[worst]["ms_per_step"], "sweep_slowest_xrt": sweep[worst]["x_realtime"],
            "sweep_min_gate_frac": min(fracs) if fracs else None,
            "sweep_worst_layer_ratio_k3": max(ratios) if ratios else None}


def _delta_flags(node, path="line"):
    """(path, value) of every key that starts with max_abs_delta and ends with _ok, anywhere in the line."""
    found = []
    if isinstance(node, dict):
        for kk, vv in node.items():
            if kk.startswith("max_abs_delta") and kk.endswith("_ok"):
                found.append((f"{path}.{kk}", vv))
            else:
                found.extend(_delta_flags(vv, f"{path}.{kk}"))
    return found


if __name__ == "__main__":
    main()
