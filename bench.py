#!/usr/bin/env python3
"""Benchmark of the MBExWN mel-inversion forward pass on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (mel -> 24 kHz audio) over one batch of synthetic mels that is already
resident in HBM.  Workload at every N: BASELINE.json configs[1] -- MW-SP-FD (C=320), batch = 1, 10 s
(80 x 800 mel) per GPU; utterances are independent, so N ranks run N utterances (weak scaling, no
data-path collective; RCCL only carries the barrier and the max-over-ranks of the timing).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
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
    # frame aligned) per stream with 10 frames of left context and 11 frames (137.5 ms) of look-ahead
    "config5_sp_stream64": ("SPEECH", 64, -8),
}
FP32_MATRIX_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def build_engine(voice):
    from mbexwn_vocoder_amd.config import ModelDims, canonical_config
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    from mbexwn_vocoder_amd.tables import WaveTables
    from mbexwn_vocoder_amd.weights import synthetic_weights
    cfg = canonical_config(voice)
    dims = ModelDims(cfg)
    raw = synthetic_weights(cfg, seed=1234)          # bias 0, PReLU alpha 0.2
    wt = WaveTables(sample_rate=dims.pulse_rate, **cfg["mbexwn_config"]["wavetable_config"])
    return cfg, raw, wt, dims, MBExWNEngine(cfg, raw, wt)


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
        return None
    try:
        with open(files[-1]) as fi:
            data = json.load(fi)
        return data[workload]["gate"]["hbm_bytes_per_launch"]
    except (KeyError, ValueError):
        return None


def cpu_baseline(cfg, raw, wt, seconds=3.0):
    """The oracle (numpy float32 port of the reference graph) timed on the host cores, on a bounded sample
    (one 3 s utterance, the reference's own CPU-runnable case configs[0]); timing protocol of the reference
    CLI (bin/resynth_mel.py:86-88): wall clock around the synthesis call only, one warm-up call."""
    from oracle.mbexwn_oracle import OracleModel
    try:
        from threadpoolctl import threadpool_info
        threads = max([pp.get("num_threads", 1) for pp in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    frames = int(round(seconds * 80))
    om = OracleModel(cfg, raw, wt, dtype=np.float32)
    rng = np.random.default_rng(42)
    mel, noise = synthetic_batch(rng, 1, frames, 20)
    om.forward(mel[:, :16], noise[:, :320])         # warm-up (weight folding, BLAS thread start)
    times = []
    budget = time.time() + 20.0                     # bounded: at most ~20 s of CPU work
    while len(times) < 15 and (len(times) < 3 or time.time() < budget):
        t0 = time.time()
        om.forward(mel, noise)
        times.append(time.time() - t0)
    best = float(np.median(times))
    return {"value": frames * 300 / best, "unit": "audio samples/s", "cores": int(threads), "kind": "port",
            "sample": f"1 utterance x {seconds:g} s (80x{frames} mel), numpy float32 oracle, median of {len(times)} runs after 1 warm-up",
            "x_realtime": frames * 300 / best / 24000.0}


def run_sharded(args, eng, dims, voice, n_utt, rank, world, dist, torch):
    """config 4: every rank sees the same seeded list of utterance lengths, takes its LPT shard, runs padded
    micro-batches (inputs staged in HBM before the timed region) and all-gathers the audio (RCCL) every step."""
    from mbexwn_vocoder_amd.sharding import lpt_partition, plan_batches
    rng = np.random.default_rng(4242)
    lengths = [int(vv) for vv in rng.integers(160, 1201, size=n_utt)]          # 2 s .. 15 s in frames
    shards = lpt_partition(lengths, world)
    mine = shards[rank]
    batches = []
    for group in plan_batches(mine, lengths, max_batch=16, max_padded_frames=16 * 1200):
        tmax = max(lengths[ii] for ii in group)
        mel_h, noise_h = synthetic_batch(np.random.default_rng(1000 + group[0]), len(group), tmax, dims.steps_per_frame)
        nfr = torch.as_tensor([lengths[ii] for ii in group], dtype=torch.int32).cuda()
        batches.append((torch.as_tensor(mel_h).cuda(), nfr, torch.as_tensor(noise_h).cuda(),
                        torch.empty((len(group), tmax * dims.hop_size), dtype=torch.float32, device="cuda"), group))
    totals = [sum(lengths[ii] for ii in ss) * dims.hop_size for ss in shards]
    flat = torch.zeros(max(totals), dtype=torch.float32, device="cuda")
    parts = [torch.empty_like(flat) for _ in range(world)] if world > 1 else None

    def step():
        pos = 0
        for mel, nfr, noise, out, group in batches:
            eng.forward(mel, n_frames=nfr, noise=noise, out=out)
            for jj, ii in enumerate(group):                                  # pack this rank's shard
                nn = lengths[ii] * dims.hop_size
                flat[pos:pos + nn] = out[jj, :nn]
                pos += nn
        if world > 1:
            dist.all_gather(parts, flat)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        samples = sum(lengths) * dims.hop_size * args.steps
        value = samples / elapsed
        print(json.dumps({
            "metric": "24 kHz audio samples/sec (whole job; x real-time = value / 24000)", "value": value,
            "unit": "audio samples/s", "x_realtime": value / 24000.0, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: MW-{voice[:2]}-FD canonical (C={dims.wn_channels}), {n_utt} utterances "
                                   f"U[2 s,15 s] = {sum(lengths) / 80:.0f} s of audio, LPT-sharded over {world} ranks, padded "
                                   f"micro-batches <= 16 items, result all_gather each step",
                       "parallelism": f"utterance-sharded x{world}", "padding_overhead":
                       sum(bb[0].shape[0] * bb[0].shape[1] for bb in batches) / max(1, sum(lengths[ii] for ii in mine))}}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_streaming(args, eng, dims, cfg, n_streams, chunk, rank, world, dist, torch):
    """config 5: steady-state tick of the streaming driver -- every stream advances by `chunk` frames; the engine
    sees a batch of windows (left context + chunk + look-ahead) with the carried phase state.  `value` counts the
    emitted audio only (the context frames are overhead of chunked operation), inputs resident in HBM; the
    host-inclusive tick latency of the Python driver (numpy staging + H2D + D2H) is reported next to it."""
    from mbexwn_vocoder_amd.streaming import StreamingSynthesizer, pack_state, stream_margins
    left, right, lead = stream_margins(dims, cfg)
    left = -(-left // 8) * 8            # window starts are aligned to 8 frames (see streaming.py: bit-exact pairing)
    win = left + chunk + right
    rng = np.random.default_rng(7 + rank)
    mel_h, noise_h = synthetic_batch(rng, n_streams, win, dims.steps_per_frame)
    mel, noise = torch.as_tensor(mel_h).cuda(), torch.as_tensor(noise_h).cuda()
    st = np.stack([pack_state(0.1, 0.3, 137, lead * dims.pulse_per_frame, (lead + chunk) * dims.pulse_per_frame)
                   for _ in range(n_streams)])
    state = torch.as_tensor(st).cuda()
    out = torch.empty((n_streams, win * dims.hop_size), dtype=torch.float32, device="cuda")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.forward(mel, noise=noise, out=out, stream_state=state)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.forward(mel, noise=noise, out=out, stream_state=state)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # host-inclusive latency through the real driver
    syn = StreamingSynthesizer(eng, chunk_frames=chunk)
    total = 40 * chunk + right
    for sid in range(n_streams):
        syn.open(sid)
        mm, nn = synthetic_batch(np.random.default_rng(sid), 1, total, dims.steps_per_frame)
        syn.push(sid, mm[0], nn[0])
    lat = []
    for _ in range(40):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        res = syn.tick()
        torch.cuda.synchronize()
        if len(res) == n_streams:
            lat.append(time.perf_counter() - t1)
    if rank == 0:
        samples = world * n_streams * chunk * dims.hop_size * args.steps
        value = samples / elapsed
        print(json.dumps({
            "metric": "24 kHz audio samples/sec (whole job; x real-time = value / 24000)", "value": value,
            "unit": "audio samples/s", "x_realtime": value / 24000.0, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: MW-SP-FD canonical, {n_streams} streams per GPU, tick = {chunk} frames "
                                   f"({chunk * 12.5:g} ms) per stream, window {left}+{chunk}+{right} frames, look-ahead "
                                   f"{right * 12.5:g} ms, carried phase state (bit-exact with offline synthesis)",
                       "tick_ms_device": elapsed / args.steps * 1e3,
                       "tick_ms_host_inclusive_p50": float(np.percentile(lat, 50) * 1e3) if lat else None,
                       "tick_ms_host_inclusive_p99": float(np.percentile(lat, 99) * 1e3) if lat else None,
                       "recompute_overhead": win / chunk}}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config2_sp_b1_10s", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even for one rank: exercises the N>1 code path on a 1-GPU box")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    voice, batch, frames = WORKLOADS[args.workload]
    cfg, raw, wt, dims, eng = build_engine(voice)
    if frames is None:
        run_sharded(args, eng, dims, voice, batch, rank, world, dist, torch)
        return
    if frames < 0:
        run_streaming(args, eng, dims, cfg, batch, -frames, rank, world, dist, torch)
        return
    rng = np.random.default_rng(42 + rank)
    mel_h, noise_h = synthetic_batch(rng, batch, frames, dims.steps_per_frame)
    mel = torch.as_tensor(mel_h).cuda()
    noise = torch.as_tensor(noise_h).cuda()
    out = torch.empty((batch, frames * dims.hop_size), dtype=torch.float32, device=mel.device)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.forward(mel, noise=noise, out=out)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.forward(mel, noise=noise, out=out)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=mel.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- dominant kernel: dilated conv + gate (75 % of the WaveNet FLOPs), timed with HIP events on the
    #      launch stream in a separate pass so that the events do not sit inside the throughput measurement
    eng.profile_enable(True)
    for _ in range(max(3, min(args.steps, 10))):
        eng.forward(mel, noise=noise, out=out)
    torch.cuda.synchronize()
    gate_ms, gate_n = eng.profile_read("gate")
    rs_ms, rs_n = eng.profile_read("res_skip")
    eng.profile_enable(False)

    if rank == 0:
        form = eng.gate_form(batch, frames)
        executed = {"direct": 1.0, "winograd_f23": 2.0 / 3.0, "winograd_f43": 0.5, "winograd_f43_small": 0.5}[form]
        kernel = {"direct": "conv1d_mfma_dma_kernel<EPI_GATE> (dilated conv k=3 C->2C + cond + tanh*sigmoid)",
                  "winograd_f23": "wn_gate_winograd_kernel (dilated conv k=3 C->2C in Winograd F(2,3) form + cond + tanh*sigmoid)",
                  "winograd_f43": "wn_gate_winograd4_kernel (dilated conv k=3 C->2C in Winograd F(4,3) form + cond + tanh*sigmoid)",
                  "winograd_f43_small": "wn_gate_winograd4k_kernel (same, 128-row blocks whose waves split the input channels)"}[form]
        C, L, ks = dims.wn_channels, dims.wn_layers, dims.wn_kernel_size
        rows = batch * frames * dims.steps_per_frame
        gate_flop = 2.0 * rows * (ks * C) * (2 * C)                   # algorithmic FLOPs of one launch
        gate_avg_s = gate_ms / max(gate_n, 1) * 1e-3
        achieved = gate_flop / gate_avg_s / 1e12
        samples = world * batch * frames * dims.hop_size * args.steps
        value = samples / elapsed
        line = {
            "metric": "24 kHz audio samples/sec (whole job; x real-time = value / 24000)",
            "value": value,
            "unit": "audio samples/s",
            "x_realtime": value / 24000.0,
            "x_realtime_per_gpu": value / 24000.0 / world,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: MW-{voice[:2]}-FD canonical (C={C}, L={L}), batch {batch} x "
                                   f"{frames / 80:g} s per GPU, 80x{frames} synthetic mel, seeded synthetic weights",
                       "batch_per_gpu": batch, "frames": frames, "parallelism": f"utterance-sharded x{world}"},
            "roofline": {"bound": "mfma",
                         "kernel": kernel,
                         "achieved": achieved, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_MATRIX_PEAK_TFLOPS, "traffic": pmc_traffic(args.workload),
                         "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/)",
                         "flop_per_launch": gate_flop,
                         "note": "achieved = algorithmic FLOPs of the direct convolution (2*rows*3C*2C) / launch time; a "
                                 "Winograd form executes 2/3 (F(2,3)) or 1/2 (F(4,3)) of them on the matrix cores, so "
                                 "frac can exceed 1; frac_executed = executed FLOPs / launch time / peak (MfmaUtil in profiles/)",
                         "mfma_flop_executed_per_launch": gate_flop * executed,
                         "frac_executed": achieved * executed / FP32_MATRIX_PEAK_TFLOPS,
                         "avg_launch_ms": gate_avg_s * 1e3, "launches_timed": gate_n,
                         "res_skip_avg_launch_ms": rs_ms / max(rs_n, 1)},
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, raw, wt)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
