"""N>1 path on CPU: world_size-2 gloo processes shard a test set, decode their shards (the
oracle stands in for the GPU pipeline -- test infrastructure only), gather by key and
aggregate the RTF; the result must equal the single-process decode."""
import os

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from kaldi_amd import abi, shard, synth
from oracle import orc


def make_set():
    g = synth.make_hclg(num_units=24, vocab=60, n_hist=12, seed=3)
    utts = [synth.sample_utterance(g, n_words=2 + (i % 5), seed=30 + i, peak=7.0) for i in range(9)]
    return g, utts


def decode_indices(g, utts, idx):
    out = []
    for i in idx:
        d = orc.Decoder(g, abi.decoder_config_recipe(), 1)
        d.Decode(utts[i][0])
        out.append(d.GetRawLattice().best_path()["words"].tolist())
    return out


def worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g, utts = make_set()
    ids = ["utt%02d" % i for i in range(len(utts))]
    durs = [u[0].shape[0] * 0.03 for u in utts]
    res = shard.decode_sharded(ids, durs, lambda idx: decode_indices(g, utts, idx), rank, world, dist)
    rtf = shard.aggregate_rtf(sum(durs[i] for i in shard.lpt_shards(durs, world)[rank]), 1.0 + rank, dist)
    if rank == 0:
        q.put((res, rtf))
    dist.barrier()
    dist.destroy_process_group()


def test_lpt_shards_partition_and_balance():
    d = synth.utterance_durations(200, seed=3)
    for n in (1, 2, 4, 8):
        sh = shard.lpt_shards(d, n)
        flat = sorted(i for s in sh for i in s)
        assert flat == list(range(200))
        loads = [d[s].sum() for s in sh]
        assert max(loads) - min(loads) <= d.max() + 1e-9
    assert shard.lpt_shards([], 4) == [[], [], [], []]
    assert sorted(map(len, shard.lpt_shards([3.0, 1.0], 4))) == [0, 0, 1, 1]     # ragged: empty shards


def test_two_rank_gloo_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res, rtf = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g, utts = make_set()
    single = decode_indices(g, utts, range(len(utts)))
    assert [res["utt%02d" % i] for i in range(len(utts))] == single
    assert [res["utt%02d" % i] for i in range(len(utts))] == [u[1] for u in utts]    # truth recovered
    total = sum(u[0].shape[0] * 0.03 for u in utts)
    assert abs(rtf - total / 2.0) < 1e-9          # sum(audio) / max(wall) with walls 1 s and 2 s
