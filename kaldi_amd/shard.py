"""Utterance sharding across the GPUs of one node.

The reference scales decoding by process-level job splitting (utils/split_data.sh +
`$cmd JOB=1:$nj`, egs/wsj/s5/steps/nnet3/decode.sh:96,123): utterances are independent, the
HCLG and the model are replicated, no data-path exchange exists.  Here: one process per GPU
(torchrun), longest-processing-time-first partition by duration, each rank decodes its
shard with its own pipeline; the only collectives are the result gather and the timing
max-reduce (RCCL when launched with backend nccl, gloo in the CPU tests).
"""
import numpy as np


def lpt_shards(durations, n_shards):
    """Longest-processing-time-first: returns a list of index lists, one per shard, with
    near-equal total duration; every index appears exactly once."""
    order = np.argsort(-np.asarray(durations, np.float64), kind="stable")
    load = np.zeros(n_shards)
    shards = [[] for _ in range(n_shards)]
    for i in order:
        k = int(np.argmin(load))
        shards[k].append(int(i))
        load[k] += float(durations[i])
    return shards


def decode_sharded(utt_ids, durations, decode_fn, rank, world, dist=None):
    """Every rank decodes shard `rank` with decode_fn(list_of_indices) -> list of results
    (same order).  Rank 0 returns {utt_id: result} for ALL utterances (order restored by
    key, like the per-JOB lat.JOB.gz archives merged by key); other ranks return None."""
    shards = lpt_shards(durations, world)
    mine = shards[rank]
    res = decode_fn(mine) if mine else []
    assert len(res) == len(mine)
    local = {utt_ids[i]: r for i, r in zip(mine, res)}
    if world == 1 or dist is None:
        return local
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(local, gathered, dst=0)
    if rank != 0:
        return None
    out = {}
    for part in gathered:
        out.update(part)
    assert len(out) == len(utt_ids)
    return out


def aggregate_rtf(audio_seconds, wall_seconds, dist=None):
    """Whole-job RTF = sum of audio over ranks / max wall time over ranks."""
    if dist is None:
        return audio_seconds / wall_seconds
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    a = torch.tensor([audio_seconds], dtype=torch.float64, device=dev)
    t = torch.tensor([wall_seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(a, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(a.item() / t.item())
