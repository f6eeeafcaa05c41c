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


class ShardedSynthesizer:
    """Runs ``forward_fn`` over this rank's share of a list of utterances and gathers the audio.

    forward_fn(mel (B,Tmax,C) float32 array, n_frames (B,) int32 array, noise (B,Tmax*spf) float32 array or None)
        -> audio (B, Tmax*hop) array-like (torch tensor on the device or numpy)
    """

    def __init__(self, forward_fn, hop_size, steps_per_frame, rank=0, world_size=1, max_batch=16,
                 max_padded_frames=16 * 1200):
        self.forward_fn = forward_fn
        self.hop = int(hop_size)
        self.spf = int(steps_per_frame)
        self.rank, self.world = int(rank), int(world_size)
        self.max_batch, self.max_padded_frames = max_batch, max_padded_frames

    def local_run(self, mels, noises=None):
        """Process this rank's shard. Returns ({index: audio np.ndarray}, shard index list)."""
        lengths = [int(mm.shape[0]) for mm in mels]
        shard = lpt_partition(lengths, self.world)[self.rank]
        out = {}
        for batch in plan_batches(shard, lengths, self.max_batch, self.max_padded_frames):
            tmax = max(lengths[ii] for ii in batch)
            mel = np.zeros((len(batch), tmax, mels[batch[0]].shape[1]), dtype=np.float32)
            noise = None if noises is None else np.zeros((len(batch), tmax * self.spf), dtype=np.float32)
            nfr = np.asarray([lengths[ii] for ii in batch], dtype=np.int32)
            for jj, ii in enumerate(batch):
                mel[jj, :lengths[ii]] = mels[ii]
                if noises is not None:
                    noise[jj, :lengths[ii] * self.spf] = noises[ii]
            audio = self.forward_fn(mel, nfr, noise)
            audio = audio.detach().cpu().numpy() if hasattr(audio, "detach") else np.asarray(audio)
            for jj, ii in enumerate(batch):
                out[ii] = np.array(audio[jj, :lengths[ii] * self.hop], dtype=np.float32)
        return out, shard

    def run(self, mels, noises=None, gather="all", device=None):
        """Returns the list of audio arrays in input order on every rank (gather="all"), on rank 0 only
        (gather="rank0", other ranks get None) or only the local dict (gather=None)."""
        local, shard = self.local_run(mels, noises)
        if gather is None:
            return local
        if self.world == 1:
            return [local[ii] for ii in range(len(mels))]
        import torch
        import torch.distributed as dist
        lengths = [int(mm.shape[0]) for mm in mels]
        shards = lpt_partition(lengths, self.world)
        totals = [sum(lengths[ii] for ii in ss) * self.hop for ss in shards]
        flat = np.concatenate([local[ii] for ii in shard]) if shard else np.zeros((0,), np.float32)
        assert flat.shape[0] == totals[self.rank]
        dev = device if device is not None else torch.device("cpu")
        buf = torch.zeros(max(totals), dtype=torch.float32, device=dev)   # one padded shard per rank
        buf[:flat.shape[0]] = torch.as_tensor(flat, device=dev)
        if gather == "rank0":
            parts = [torch.empty_like(buf) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(buf, parts, dst=0)
            if self.rank != 0:
                return None
        else:
            parts = [torch.empty_like(buf) for _ in range(self.world)]
            dist.all_gather(parts, buf)
        result = [None] * len(mels)
        for rr, ss in enumerate(shards):
            data = parts[rr].cpu().numpy()
            pos = 0
            for ii in ss:
                nn = lengths[ii] * self.hop
                result[ii] = data[pos:pos + nn].copy()
                pos += nn
        return result
