"""Utterance-level data parallelism: one process per GPU, utterances sharded across ranks, result gather.

The reference has no parallelism of any kind: its CLI walks the files one by one with batch = 1
(reference bin/resynth_mel.py:74).  Utterances are independent forward passes, so the path shards without
any exchange step inside the forward; the only collective is the result gather (RCCL over xGMI when the
tensors live on the GPUs, gloo in the CPU tests).  SURVEY.md section 8(e).

Partitioning: longest-processing-time-first greedy assignment (cost = frames), then per rank padded
micro-batches of similar length; every boundary op of the engine honours the item's own length, so padded
batches give exactly the one-at-a-time results.
"""
import numpy as np


def lpt_partition(lengths, world_size):
    """Greedy longest-processing-time assignment. Returns ``world_size`` lists of utterance indices;
    deterministic (ties by index), identical on every rank."""
    order = sorted(range(len(lengths)), key=lambda ii: (-int(lengths[ii]), ii))
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for ii in order:
        rr = min(range(world_size), key=lambda r: (loads[r], r))
        shards[rr].append(ii)
        loads[rr] += int(lengths[ii])
    return shards


def plan_batches(indices, lengths, max_batch=16, max_padded_frames=16 * 1200):
    """Group a rank's utterances into padded micro-batches: sorted by length (little padding), at most
    ``max_batch`` items and ``max_padded_frames`` = batch * longest item per batch."""
    order = sorted(indices, key=lambda ii: (-int(lengths[ii]), ii))
    batches, cur = [], []
    for ii in order:
        longest = int(lengths[cur[0]]) if cur else int(lengths[ii])
        if cur and (len(cur) + 1 > max_batch or (len(cur) + 1) * longest > max_padded_frames):
            batches.append(cur)
            cur = []
        cur.append(ii)
    if cur:
        batches.append(cur)
    return batches


def plan_stats(lengths, world_size, hop_size, max_batch=16, max_padded_frames=16 * 1200, gather_chunk_floats=1 << 22):
    """What the partition of a job looks like before it runs (every rank computes the same numbers):

    ``imbalance``       LPT makespan / mean load (frames): 1.0 = perfectly even; the scaling efficiency the partition allows
    ``padding``         padded frames / real frames of the micro-batches, per rank
    ``micro_batches``   micro-batches per rank (sorted by falling length: the SMALLEST runs last)
    ``exposed_gather_bytes``  per rank, the bytes of the result gather that cannot be issued before the rank's last forward
                        has been packed (the gather is chunked and asynchronous: a piece goes to the collective as soon as
                        the micro-batches that fill it are packed, so only the pieces the last micro-batch touches -- and the
                        zero padding behind the rank's own audio up to the longest shard -- are left for the end)"""
    lengths = [int(ll) for ll in lengths]
    shards = lpt_partition(lengths, world_size)
    loads = [sum(lengths[ii] for ii in ss) for ss in shards]
    mean = sum(loads) / max(1, world_size)
    flat = max(loads) * hop_size if loads else 0
    chunk = max(1, int(gather_chunk_floats))
    padding, n_batches, exposed = [], [], []
    for rr, ss in enumerate(shards):
        batches = plan_batches(ss, lengths, max_batch, max_padded_frames)
        padded = sum(len(bb) * max(lengths[ii] for ii in bb) for bb in batches)
        padding.append(padded / max(1, loads[rr]))
        n_batches.append(len(batches))
        last_audio = sum(lengths[ii] for ii in batches[-1]) * hop_size if batches else 0
        before_last = loads[rr] * hop_size - last_audio                       # floats packed before the last forward
        exposed.append(4 * (flat - (before_last // chunk) * chunk))
    return {"imbalance": (max(loads) / mean) if mean else 1.0, "loads_frames": loads, "padding": padding,
            "micro_batches": n_batches, "exposed_gather_bytes": exposed, "shard_buffer_bytes": 4 * flat}


class ShardResult:
    """Audio of a sharded run, still where the forward pass left it (device tensors when the engine ran on a GPU).

    ``parts[r]`` is rank r's shard: its utterances (LPT order) packed back to back in one flat float32 tensor
    (``None`` for ranks whose shard this process does not hold)."""

    def __init__(self, parts, shards, lengths, hop, timing=None):
        self.parts, self.shards, self.lengths, self.hop = parts, shards, lengths, hop
        # {"compute_ms", "gather_ms", "chunks"} of the run that produced it (this rank): forward passes + packing, and the
        # part of the gather that was NOT hidden behind them (time between the last forward and the last chunk's arrival)
        self.timing = timing or {}

    def item(self, index):
        """Audio of utterance ``index`` as a view of the flat shard (tensor), or None if its shard is not held here."""
        for rr, ss in enumerate(self.shards):
            if index in ss:
                if self.parts[rr] is None:
                    return None
                pos = sum(self.lengths[ii] for ii in ss[:ss.index(index)]) * self.hop
                return self.parts[rr][pos:pos + self.lengths[index] * self.hop]
        raise KeyError(index)

    def to_list(self):
        """List of numpy arrays in input order (None where the shard is not held): ONE device->host copy per shard."""
        result = [None] * len(self.lengths)
        for rr, ss in enumerate(self.shards):
            if self.parts[rr] is None:
                continue
            data = self.parts[rr].detach().cpu().numpy()
            pos = 0
            for ii in ss:
                nn = self.lengths[ii] * self.hop
                result[ii] = data[pos:pos + nn].copy()
                pos += nn
        return result


class ShardedSynthesizer:
    """Runs ``forward_fn`` over this rank's share of a list of utterances and gathers the audio.

    forward_fn(mel (B,Tmax,C) float32, n_frames (B,) int32, noise (B,Tmax*spf) float32 or None) -> audio (B, Tmax*hop)

    ``force_collective`` executes the gather with a single rank too (needs an initialised process group).
    With ``device`` set (a torch device) the padded micro-batches are staged there once (:meth:`stage`), ``forward_fn``
    receives and returns tensors on that device, the shard is packed on the device and handed to the collective as it
    is -- RCCL over xGMI for CUDA tensors, no host copy between the forward pass and the gather.  Without ``device``
    the arguments are numpy arrays (CPU test doubles) and the collective runs on CPU tensors (gloo).
    """

    def __init__(self, forward_fn, hop_size, steps_per_frame, rank=0, world_size=1, max_batch=16,
                 max_padded_frames=16 * 1200, device=None, force_collective=False, gather_chunk_floats=1 << 22):
        self.forward_fn = forward_fn
        self.hop = int(hop_size)
        self.spf = int(steps_per_frame)
        self.rank, self.world = int(rank), int(world_size)
        self.max_batch, self.max_padded_frames = max_batch, max_padded_frames
        self.device = device
        # run the gather collective even with one rank (an initialised process group is then required): lets a 1-GPU
        # box execute the RCCL path that N > 1 takes
        self.force_collective = bool(force_collective)
        # the shard buffer is gathered in chunks of this many floats (16 MB), each as soon as the micro-batches that fill it
        # have been packed, asynchronously: the transfer of micro-batch i runs under the forward pass of micro-batch i + 1
        self.gather_chunk_floats = max(1, int(gather_chunk_floats))

    def stage(self, mels, noises=None):
        """Partition, pad and (with a device) upload this rank's micro-batches.  Returns the plan for run_staged."""
        import torch
        lengths = [int(mm.shape[0]) for mm in mels]
        shards = lpt_partition(lengths, self.world)
        batches = []
        for group in plan_batches(shards[self.rank], lengths, self.max_batch, self.max_padded_frames):
            tmax = max(lengths[ii] for ii in group)
            mel = np.zeros((len(group), tmax, mels[group[0]].shape[1]), dtype=np.float32)
            noise = None if noises is None else np.zeros((len(group), tmax * self.spf), dtype=np.float32)
            nfr = np.asarray([lengths[ii] for ii in group], dtype=np.int32)
            for jj, ii in enumerate(group):
                mel[jj, :lengths[ii]] = mels[ii]
                if noises is not None:
                    noise[jj, :lengths[ii] * self.spf] = noises[ii]
            if self.device is not None:
                mel, nfr = torch.as_tensor(mel, device=self.device), torch.as_tensor(nfr, device=self.device)
                noise = None if noise is None else torch.as_tensor(noise, device=self.device)
            batches.append((group, mel, nfr, noise))
        totals = [sum(lengths[ii] for ii in ss) * self.hop for ss in shards]
        dev = self.device if self.device is not None else torch.device("cpu")
        # one padded flat buffer per rank (all_gather needs equal sizes); reused by every run_staged of this plan
        flat = torch.zeros(max(totals) if totals else 0, dtype=torch.float32, device=dev)   # (equal sizes: collectives)
        return {"lengths": lengths, "shards": shards, "batches": batches, "totals": totals, "flat": flat, "parts": None}

    def _now(self):
        """Time stamp of this point of the launch sequence: an event on the device's current stream, wall clock on the CPU."""
        import time
        import torch
        if self.device is not None and torch.device(self.device).type == "cuda":
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            return ev
        return time.perf_counter()

    @staticmethod
    def _elapsed_ms(t0, t1):
        if isinstance(t0, float):
            return (t1 - t0) * 1e3
        t1.synchronize()
        return float(t0.elapsed_time(t1))

    def run_staged(self, plan, gather="rank0"):
        """Forward passes of this rank's micro-batches + the result gather; everything stays on the plan's device.
        gather: "rank0" (default: rank 0 gets every shard -- the shape of the CLI job, whose audio one process writes out),
        "all" (every rank gets every shard) or None (local shard only: zero collectives, e.g. every rank writes its own
        files).  The gather is chunked and asynchronous: the shard buffer (micro-batches packed back to back, the same
        size on every rank) is cut into ``gather_chunk_floats`` pieces, and a piece is handed to the collective as soon as
        the micro-batches that fill it have been packed -- on NCCL/RCCL the transfer then runs on the communicator's own
        stream under the next micro-batch's forward pass.  Every rank issues the pieces in the same order."""
        import torch
        lengths, shards, flat = plan["lengths"], plan["shards"], plan["flat"]
        collective = gather is not None and (self.world > 1 or self.force_collective)
        chunk = self.gather_chunk_floats
        n_chunks = (int(flat.numel()) + chunk - 1) // chunk if collective else 0
        if collective:
            import torch.distributed as dist
            if plan["parts"] is None and (gather == "all" or self.rank == 0):
                plan["parts"] = [torch.empty_like(flat) for _ in range(self.world)]
        works, issued = [], 0

        def issue_upto(filled):
            """Hand every piece that lies completely in front of position ``filled`` to the collective."""
            nonlocal issued
            while issued < n_chunks and min((issued + 1) * chunk, int(flat.numel())) <= filled:
                lo, hi = issued * chunk, min((issued + 1) * chunk, int(flat.numel()))
                src = flat[lo:hi]
                if gather == "rank0":
                    dst = [pp[lo:hi] for pp in plan["parts"]] if self.rank == 0 else None
                    works.append(dist.gather(src, dst, dst=0, async_op=True))
                else:
                    works.append(dist.all_gather([pp[lo:hi] for pp in plan["parts"]], src, async_op=True))
                issued += 1

        t_start = self._now()
        pos, order = 0, list(shards[self.rank])
        done = 0                                           # utterances of the shard packed so far (shard order)
        where = {}
        for group, mel, nfr, noise in plan["batches"]:
            audio = self.forward_fn(mel, nfr, noise)
            audio = audio if torch.is_tensor(audio) else torch.as_tensor(np.asarray(audio))
            for jj, ii in enumerate(group):
                where[ii] = (audio, jj)
            # pack what is complete, in shard order (device-side copies, no sync): micro-batches and shards are both sorted
            # by falling length, so a micro-batch fills the next contiguous stretch of the shard buffer
            while done < len(order) and order[done] in where:
                audio_i, jj = where.pop(order[done])
                nn = lengths[order[done]] * self.hop
                flat[pos:pos + nn] = audio_i[jj, :nn]
                pos += nn
                done += 1
            if collective:
                issue_upto(pos)
        assert pos == plan["totals"][self.rank] and done == len(order) and not where
        t_compute = self._now()
        parts = [None] * self.world
        if not collective:
            parts[self.rank] = flat[:pos]
        else:
            issue_upto(int(flat.numel()))                  # the pieces behind this rank's own audio (zeros) and the last one
            for ww in works:
                ww.wait()
            if gather == "all" or self.rank == 0:
                parts = [pp[:tt] for pp, tt in zip(plan["parts"], plan["totals"])]
        t_end = self._now()
        timing = {"compute_ms": self._elapsed_ms(t_start, t_compute), "gather_ms": self._elapsed_ms(t_compute, t_end),
                  "chunks": n_chunks}
        return ShardResult(parts, shards, lengths, self.hop, timing)

    def run(self, mels, noises=None, gather="rank0"):
        """Returns the list of audio arrays in input order on rank 0 only (gather="rank0", the default; other ranks get
        None), on every rank (gather="all") or the local dict {index: audio} (gather=None)."""
        plan = self.stage(mels, noises)
        res = self.run_staged(plan, gather)
        if gather is None:
            listed = res.to_list()
            return {ii: listed[ii] for ii in plan["shards"][self.rank]}
        if gather == "rank0" and self.rank != 0:
            return None
        return res.to_list()
