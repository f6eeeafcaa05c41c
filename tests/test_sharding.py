"""Utterance sharding + result gather: partition properties and a world_size-2 gloo run on CPU.

The forward function used here is a deterministic test double (NOT the oracle and not a product fallback):
what is under test is the partition / padding / gather logic around the engine."""
import os
import socket
import sys

import numpy as np
import pytest

from mbexwn_vocoder_amd.sharding import ShardedSynthesizer, lpt_partition, plan_batches

HOP, SPF = 300, 20


def fake_forward(mel, n_frames, noise):
    """audio[b, t*HOP + k] = sum(mel[b,t]) + 0.001*k + noise[b, t*SPF]; zero behind the item's length."""
    B, T, _ = mel.shape
    out = np.zeros((B, T * HOP), dtype=np.float32)
    for b in range(B):
        n = int(n_frames[b])
        base = mel[b, :n].sum(axis=1)
        if noise is not None:
            base = base + noise[b, :n * SPF:SPF]
        out[b, :n * HOP] = (base[:, None] + 0.001 * np.arange(HOP)[None, :]).reshape(-1)
    return out


def make_utterances(n, seed=0):
    rng = np.random.default_rng(seed)
    lengths = rng.integers(1, 60, size=n)
    mels = [rng.normal(size=(int(ll), 80)).astype(np.float32) for ll in lengths]
    noises = [rng.normal(size=(int(ll) * SPF,)).astype(np.float32) for ll in lengths]
    return mels, noises


def test_lpt_partition_properties():
    rng = np.random.default_rng(1)
    lengths = rng.integers(160, 1200, size=256)           # BASELINE config 4: 256 utterances, 2..15 s
    for world in (1, 2, 4, 8):
        shards = lpt_partition(lengths, world)
        flat = sorted(ii for ss in shards for ii in ss)
        assert flat == list(range(256))                    # every utterance exactly once
        loads = [sum(int(lengths[ii]) for ii in ss) for ss in shards]
        assert max(loads) - min(loads) <= int(lengths.max())
        assert max(loads) <= 1.02 * sum(loads) / world     # near-perfect balance => near-linear scaling
    assert lpt_partition([], 4) == [[], [], [], []]
    assert lpt_partition([5], 2) == [[0], []]


def test_plan_batches_limits():
    lengths = [100, 1200, 5, 700, 700, 3, 1200, 1]
    batches = plan_batches(list(range(8)), lengths, max_batch=3, max_padded_frames=2500)
    assert sorted(ii for bb in batches for ii in bb) == list(range(8))
    for bb in batches:
        assert len(bb) <= 3
        assert len(bb) * max(lengths[ii] for ii in bb) <= 2500 or len(bb) == 1


def test_single_process_equals_one_at_a_time():
    mels, noises = make_utterances(13)
    syn = ShardedSynthesizer(fake_forward, HOP, SPF, max_batch=4, max_padded_frames=200)
    got = syn.run(mels, noises)
    for ii, (mm, nn) in enumerate(zip(mels, noises)):
        ref = fake_forward(mm[None], np.asarray([mm.shape[0]], np.int32), nn[None])[0]
        assert np.array_equal(got[ii], ref)
    assert syn.run([], []) == []


def _worker(rank, world, port, tmpdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mels, noises = make_utterances(21, seed=3)
        syn = ShardedSynthesizer(fake_forward, HOP, SPF, rank=rank, world_size=world, max_batch=4)
        everywhere = syn.run(mels, noises, gather="all")
        on_root = syn.run(mels, noises, gather="rank0")
        ok = all(np.array_equal(everywhere[ii], fake_forward(mm[None], np.asarray([mm.shape[0]], np.int32),
                                                             noises[ii][None])[0]) for ii, mm in enumerate(mels))
        if rank == 0:
            ok = ok and all(np.array_equal(aa, bb) for aa, bb in zip(everywhere, on_root))
        else:
            ok = ok and on_root is None
        local = syn.run(mels, noises, gather=None)
        ok = ok and sorted(local) == sorted(lpt_partition([mm.shape[0] for mm in mels], world)[rank])
        with open(os.path.join(tmpdir, f"ok{rank}"), "w") as fo:
            fo.write("1" if ok else "0")
    finally:
        dist.destroy_process_group()


def _worker4(rank, world, port, tmpdir):
    """world_size 4: uneven shards (one long utterance dominates a rank), one rank without any utterance, the shard buffer
    cut into many chunks that are gathered asynchronously while later micro-batches still run."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for case, n_utt in (("uneven", 11), ("empty_rank", 3)):
            mels, noises = make_utterances(n_utt, seed=17 + n_utt)
            if case == "uneven":
                rng = np.random.default_rng(5)
                mels[4] = rng.normal(size=(400, 80)).astype(np.float32)            # one utterance as long as all others together
                noises[4] = rng.normal(size=(400 * SPF,)).astype(np.float32)
            lengths = [mm.shape[0] for mm in mels]
            shards = lpt_partition(lengths, world)
            if case == "empty_rank":
                ok = ok and shards[3] == []
            calls = []

            def counted(mel, n_frames, noise, calls=calls):
                calls.append(mel.shape[0])
                return fake_forward(mel, n_frames, noise)
            syn = ShardedSynthesizer(counted, HOP, SPF, rank=rank, world_size=world, max_batch=2, gather_chunk_floats=7001)
            plan = syn.stage(mels, noises)
            for mode in ("rank0", "all"):
                res = syn.run_staged(plan, gather=mode)
                ok = ok and res.timing["chunks"] == -(-int(plan["flat"].numel()) // 7001) and res.timing["chunks"] >= 2
                ok = ok and res.timing["compute_ms"] >= 0.0 and res.timing["gather_ms"] >= 0.0
                if mode == "all" or rank == 0:
                    got = res.to_list()
                    for ii, mm in enumerate(mels):
                        ref = fake_forward(mm[None], np.asarray([mm.shape[0]], np.int32), noises[ii][None])[0]
                        ok = ok and np.array_equal(got[ii], ref)
                else:
                    ok = ok and all(pp is None for pp in res.parts)
            ok = ok and len(calls) == 2 * len(plan["batches"])
            default = syn.run(mels, noises)                               # the default: gathered on rank 0 only
            ok = ok and ((default is None) if rank else len(default) == n_utt)
        with open(os.path.join(tmpdir, f"ok{rank}"), "w") as fo:
            fo.write("1" if ok else "0")
    finally:
        dist.destroy_process_group()


def _config4_lengths():
    """BASELINE config 4 as bench.py draws it: 256 utterances, U[2 s, 15 s] in frames."""
    return [int(vv) for vv in np.random.default_rng(4242).integers(160, 1201, size=256)]


def _utterance(ii, ll, channels=8):
    rng = np.random.default_rng(1000 + ii)
    return rng.normal(size=(ll, channels)).astype(np.float32), rng.normal(size=(ll * SPF,)).astype(np.float32)


def _worker8(rank, world, port, tmpdir):
    """world_size 8 on the real length distribution of config 4: every rank stages only its own utterances, the audio is
    gathered on rank 0 (chunked, asynchronous), which checks every one of the 256 utterances against its own forward."""
    import torch.distributed as dist
    from mbexwn_vocoder_amd.sharding import plan_stats
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lengths = _config4_lengths()
        mine = set(lpt_partition(lengths, world)[rank])
        mels, noises = [], []
        for ii, ll in enumerate(lengths):
            if ii in mine:
                mm, nn = _utterance(ii, ll)
            else:                                            # placeholders: only the lengths of the others matter
                mm, nn = np.zeros((ll, 8), dtype=np.float32), np.zeros((ll * SPF,), dtype=np.float32)
            mels.append(mm)
            noises.append(nn)
        order = []

        def counted(mel, n_frames, noise):
            order.append(int(np.asarray(n_frames).sum()))
            return fake_forward(mel, n_frames, noise)
        syn = ShardedSynthesizer(counted, HOP, SPF, rank=rank, world_size=world, max_batch=16, max_padded_frames=16 * 1200)
        plan = syn.stage(mels, noises)
        res = syn.run_staged(plan, gather="rank0")
        st = plan_stats(lengths, world, HOP, 16, 16 * 1200)
        ok = st["imbalance"] <= 1.03 and st["micro_batches"][rank] == len(plan["batches"]) == len(order)
        ok = ok and order == sorted(order, reverse=True)                     # the smallest micro-batch runs last
        ok = ok and st["exposed_gather_bytes"][rank] <= st["shard_buffer_bytes"]
        ok = ok and res.timing["chunks"] == -(-int(plan["flat"].numel()) // (1 << 22))
        if rank == 0:
            got = res.to_list()
            seen = 0
            for ii, ll in enumerate(lengths):
                mm, nn = _utterance(ii, ll)
                ref = fake_forward(mm[None], np.asarray([ll], np.int32), nn[None])[0]
                ok = ok and got[ii] is not None and got[ii].shape == ref.shape and np.array_equal(got[ii], ref)
                seen += 1
            ok = ok and seen == 256 and sum(len(ss) for ss in res.shards) == 256
        else:
            ok = ok and all(pp is None for pp in res.parts)
        with open(os.path.join(tmpdir, f"ok{rank}"), "w") as fo:
            fo.write("1" if ok else "0")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_world_size_8_gloo_config4_length_distribution(tmp_path):
    """The shape of the 8-GPU job of BASELINE config 4, on CPU ranks: 256 utterances U[2 s, 15 s], LPT imbalance <= 3 %,
    two micro-batches per rank of which the smaller runs last, every utterance gathered exactly once on rank 0."""
    import torch.multiprocessing as mp
    from mbexwn_vocoder_amd.sharding import plan_stats
    st = plan_stats(_config4_lengths(), 8, HOP, 16, 16 * 1200)
    assert st["imbalance"] <= 1.03 and st["micro_batches"] == [2] * 8
    assert max(st["exposed_gather_bytes"]) <= 0.5 * st["shard_buffer_bytes"]     # at most the last micro-batch's audio is left
    with socket.socket() as ss:
        ss.bind(("127.0.0.1", 0))
        port = ss.getsockname()[1]
    mp.spawn(_worker8, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    for rank in range(8):
        assert (tmp_path / f"ok{rank}").read_text() == "1", f"rank {rank}"


@pytest.mark.timeout(300)
def test_world_size_4_gloo_uneven_shards_and_an_empty_rank(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as ss:
        ss.bind(("127.0.0.1", 0))
        port = ss.getsockname()[1]
    mp.spawn(_worker4, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    for rank in range(4):
        assert (tmp_path / f"ok{rank}").read_text() == "1", f"rank {rank}"


@pytest.mark.timeout(300)
def test_random_shardings_cover_every_utterance_exactly_once():
    """Random utterance sets x world sizes x micro-batch limits: the ranks' local shards (no collective: every rank is
    simulated in this process) together hold every utterance once, each equal to its one-at-a-time result, and no
    micro-batch exceeds its limits."""
    for seed in range(40):
        rng = np.random.default_rng(seed)
        n, world = int(rng.integers(1, 60)), int(rng.integers(1, 9))
        max_batch, max_padded = int(rng.integers(1, 20)), int(rng.integers(60, 1500))
        mels, noises = make_utterances(n, seed=100 + seed)
        use_noise = bool(rng.integers(0, 2))
        seen = {}
        for rank in range(world):
            def checked(mel, n_frames, noise, max_batch=max_batch, max_padded=max_padded):
                assert mel.shape[0] <= max_batch and (mel.shape[0] == 1 or mel.shape[0] * mel.shape[1] <= max_padded)
                return fake_forward(mel, n_frames, noise)
            syn = ShardedSynthesizer(checked, HOP, SPF, rank=rank, world_size=world, max_batch=max_batch,
                                     max_padded_frames=max_padded)
            local = syn.run(mels, noises if use_noise else None, gather=None)
            assert not set(local) & set(seen)
            seen.update(local)
        assert sorted(seen) == list(range(n))
        for ii in range(n):
            one = fake_forward(mels[ii][None], np.asarray([mels[ii].shape[0]]), noises[ii][None] if use_noise else None)[0]
            assert np.array_equal(np.asarray(seen[ii]), one), (seed, ii)


def test_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as ss:
        ss.bind(("127.0.0.1", 0))
        port = ss.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        assert (tmp_path / f"ok{rank}").read_text() == "1"


def test_force_collective_runs_the_gather_with_one_rank():
    """world_size 1 normally skips the collective; force_collective executes all_gather / gather anyway (this is how a
    1-GPU box exercises the RCCL path, tests/test_gpu_dropin.py) -- here over gloo on CPU tensors."""
    import torch.distributed as dist
    with socket.socket() as ss:
        ss.bind(("127.0.0.1", 0))
        port = ss.getsockname()[1]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        mels, noises = make_utterances(9, seed=5)
        plain = ShardedSynthesizer(fake_forward, HOP, SPF, max_batch=4).run(mels, noises)
        for mode in ("all", "rank0"):
            syn = ShardedSynthesizer(fake_forward, HOP, SPF, max_batch=4, force_collective=True)
            plan = syn.stage(mels, noises)
            res = syn.run_staged(plan, gather=mode)
            assert plan["parts"] is not None                       # the collective's receive buffers were used
            assert all(np.array_equal(aa, bb) for aa, bb in zip(res.to_list(), plain))
    finally:
        dist.destroy_process_group()


def test_bench_counts_gpus_without_touching_them(monkeypatch):
    """bench.spawn_ranks' parent must not initialise HIP: the count comes from the visibility variables when they are
    set (else from a child process)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2,3")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0,1")
    assert bench.visible_gpu_count() == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == 0


@pytest.mark.timeout(180)
def test_bench_parent_ends_the_other_ranks_when_one_fails(monkeypatch):
    """`bench.py --gpus 2` without a launcher: the parent starts the ranks as fresh child processes and polls them.  Here
    (no GPU) every rank exits with an error right after start: the parent must come back with a non-zero status instead of
    waiting on the survivors, without having touched a GPU itself."""
    import argparse
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: 2)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    for kk in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(kk, raising=False)
    t0 = time.time()
    with pytest.raises(SystemExit) as exc:
        bench.spawn_ranks(argparse.Namespace(gpus=2, master_port=0))
    assert exc.value.code not in (0, None)
    assert time.time() - t0 < 120
