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


class ShardResult:
    """Audio of a sharded run, still where the forward pass left it (device tensors when the engine ran on a GPU).

    ``parts[r]`` is rank r's shard: its utterances (LPT order) packed back to back in one flat float32 tensor
    (``None`` for ranks whose shard this process does not hold)."""

    def __init__(self, parts, shards, lengths, hop):
        self.parts, self.shards, self.lengths, self.hop = parts, shards, lengths, hop

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
                 max_padded_frames=16 * 1200, device=None, force_collective=False):
        self.forward_fn = forward_fn
        self.hop = int(hop_size)
        self.spf = int(steps_per_frame)
        self.rank, self.world = int(rank), int(world_size)
        self.max_batch, self.max_padded_frames = max_batch, max_padded_frames
        self.device = device
        # run the gather collective even with one rank (an initialised process group is then required): lets a 1-GPU
        # box execute the RCCL path that N > 1 takes
        self.force_collective = bool(force_collective)

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
        flat = torch.zeros(max(totals) if totals else 0, dtype=torch.float32, device=dev)
        return {"lengths": lengths, "shards": shards, "batches": batches, "totals": totals, "flat": flat, "parts": None}

    def run_staged(self, plan, gather="all"):
        """Forward passes of this rank's micro-batches + the result gather; everything stays on the plan's device.
        gather: "all" (every rank gets every shard), "rank0" (rank 0 only) or None (local shard only)."""
        import torch
        lengths, shards, flat = plan["lengths"], plan["shards"], plan["flat"]
        where = {}
        for group, mel, nfr, noise in plan["batches"]:
            audio = self.forward_fn(mel, nfr, noise)
            audio = audio if torch.is_tensor(audio) else torch.as_tensor(np.asarray(audio))
            for jj, ii in enumerate(group):
                where[ii] = (audio, jj)
        pos = 0
        for ii in shards[self.rank]:                       # pack in shard order (device-side copies, no sync)
            audio, jj = where[ii]
            nn = lengths[ii] * self.hop
            flat[pos:pos + nn] = audio[jj, :nn]
            pos += nn
        assert pos == plan["totals"][self.rank]
        parts = [None] * self.world
        if (self.world == 1 and not self.force_collective) or gather is None:
            parts[self.rank] = flat[:pos]
        else:
            import torch.distributed as dist
            if plan["parts"] is None:
                plan["parts"] = [torch.empty_like(flat) for _ in range(self.world)]
            if gather == "rank0":
                dist.gather(flat, plan["parts"] if self.rank == 0 else None, dst=0)
                if self.rank == 0:
                    parts = [pp[:tt] for pp, tt in zip(plan["parts"], plan["totals"])]
            else:
                dist.all_gather(plan["parts"], flat)
                parts = [pp[:tt] for pp, tt in zip(plan["parts"], plan["totals"])]
        return ShardResult(parts, shards, lengths, self.hop)

    def run(self, mels, noises=None, gather="all"):
        """Returns the list of audio arrays in input order on every rank (gather="all"), on rank 0 only
        (gather="rank0", other ranks get None) or the local dict {index: audio} (gather=None)."""
        plan = self.stage(mels, noises)
        res = self.run_staged(plan, gather)
        if gather is None:
            listed = res.to_list()
            return {ii: listed[ii] for ii in plan["shards"][self.rank]}
        if gather == "rank0" and self.rank != 0:
            return None
        return res.to_list()
