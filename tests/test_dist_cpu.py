"""CPU: the N>1 path (utterance sharding + the single all_gather of hypotheses) with gloo, world_size 2."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from omni_avsr_amd.dist import gather_hypotheses, make_batches, pack_records, reduce_wer_counters, shard_utterances


def test_shard_covers_everything_once_and_balances():
    dur = [1.0 + (i * 7 % 16) for i in range(37)]
    seen = []
    loads = []
    for r in range(4):
        own = shard_utterances(dur, 4, r)
        seen += own
        loads.append(sum(dur[i] for i in own))
    assert sorted(seen) == list(range(37))
    assert max(loads) - min(loads) <= max(dur)
    assert make_batches(list(range(10)), 4) == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dur = [float(3 + (i * 5) % 11) for i in range(13)]
    own = shard_utterances(dur, world, rank)
    # fake "decode": utterance i produces tokens [i, i+1, ...] of length 1 + i % 5
    width = 8
    ids = torch.full((len(own), width), 99, dtype=torch.int32)
    lens = torch.zeros(len(own), dtype=torch.int32)
    for j, u in enumerate(own):
        n = 1 + u % 5
        ids[j, :n] = torch.arange(u, u + n, dtype=torch.int32)
        lens[j] = n
    hyp = gather_hypotheses(pack_records(own, ids, lens, width), len(dur))
    e, w = reduce_wer_counters(rank + 1, 10)
    q.put((rank, hyp, e, w))
    dist.destroy_process_group()


def test_gather_hypotheses_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, hyp, e, w in res:
        assert sorted(hyp) == list(range(13))
        for u, toks in hyp.items():
            assert toks == list(range(u, u + 1 + u % 5))
        assert (e, w) == (3, 20)
