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


class _StubModel:
    """Stands in for AVSR_LLMs in the evaluation loop: `forward` returns ids that are a function of the utterance alone (its id rides in the
    first sample of its waveform, its duration decides the hypothesis length), so any utterance evaluated twice, dropped or paired with
    another utterance's ids shows."""
    is_task_specific = True
    num_beams = 1

    def __init__(self):
        self.modality = None
        self.calls = []

    @staticmethod
    def ids_of(u, n):
        return [(7 * u + k) % 50 + 1 for k in range(n)]

    def __call__(self, batch, is_trainval=False, modality=None, **kw):
        assert not is_trainval and modality == "audio" and self.modality == "audio"
        assert kw == dict(test_ratio_matry_audio=4, test_ratio_matry_video=4)
        B = batch["audio"].shape[0]
        uids = [int(round(float(batch["audio"][b, 0, 0]))) for b in range(B)]
        ns = [1 + int(batch["lengths"][b]) // 640 % 5 for b in range(B)]
        self.calls.append(uids)
        out = torch.zeros(B, max(ns), dtype=torch.long)          # right-padded with 0 like the engine's pad id
        for b in range(B):
            out[b, : ns[b]] = torch.tensor(self.ids_of(uids[b], ns[b]))
        return out


def _eval_samples():
    """19 utterances, durations 3..13 'frames' (640 samples each); the waveform's first sample carries the utterance id"""
    samples = []
    for u in range(19):
        n = 640 * (3 + (u * 5) % 11)
        a = torch.zeros(n, 1)
        a[0, 0] = float(u)
        n_tok = 1 + (n // 640) % 5
        gold = " ".join(f"w{t}" for t in _StubModel.ids_of(u, n_tok))
        if u % 4 == 0:
            gold += " extra"                                      # one deletion error for every fourth utterance
        samples.append(({"audio": a}, gold))
    return samples


def _eval_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from omni_avsr_amd import harness as H
    samples = _eval_samples()
    model = _StubModel()
    rows = []

    def to_text(ids):
        out = []
        for r in ids.tolist():
            r = [t for t in r if t != 0]
            rows.append(r)
            out.append(" ".join(f"w{t}" for t in r))
        return out
    meter = H.evaluate(model, samples, "audio", to_text, batch_size=3, bos_id=None, rank=rank, world=world, in_flight=1,
                       test_ratio_matry_audio=4, test_ratio_matry_video=4)
    W = 8
    ids_t = torch.full((len(rows), W), -1, dtype=torch.int32)
    lens_t = torch.zeros(len(rows), dtype=torch.int32)
    for k, r in enumerate(rows):
        ids_t[k, : len(r)] = torch.tensor(r, dtype=torch.int32)
        lens_t[k] = len(r)
    hyp = gather_hypotheses(pack_records(meter.utt_ids, ids_t, lens_t, W), len(samples))
    edits, words, failed = reduce_wer_counters(meter.total_edit_distance, meter.total_length, failed=len(meter.failed))
    q.put((rank, hyp, edits, words, failed, sorted(meter.utt_ids), model.calls))
    dist.destroy_process_group()


def test_evaluate_world2_every_utterance_once_with_its_own_ids():
    """The reference's evaluation loop (eval_OmniAVSR.py:273-357) as this repo shards it: harness.evaluate(rank, world = 2) -> pack_records ->
    gather_hypotheses + reduce_wer_counters, two gloo processes, a stub model.  Every utterance is decoded by exactly one rank, exactly
    once, in batches of <= batch_size taken longest first; after the one all_gather both ranks hold every hypothesis, each the ids of ITS
    utterance; the summed WER counters are those of the whole set."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 1000
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    samples = _eval_samples()
    N = len(samples)
    dur = [float(s["audio"].shape[0]) / 640.0 for s, _ in samples]
    owned = [res[0][5], res[1][5]]
    assert sorted(owned[0] + owned[1]) == list(range(N)), "an utterance was evaluated twice or not at all"
    for r in range(2):
        assert owned[r] == sorted(shard_utterances(dur, 2, r))
        calls = res[r][6]
        assert all(len(c) <= 3 for c in calls) and [u for c in calls for u in c] == shard_utterances(dur, 2, r), "batches must follow the rank's shard, longest first"
    words = sum(len(g.split()) for _, g in samples)
    for rank, hyp, edits, w, failed, _, _ in res:
        assert sorted(hyp) == list(range(N))
        for u in range(N):
            n = 1 + int(dur[u]) % 5
            assert hyp[u] == _StubModel.ids_of(u, n), f"rank {rank}: utterance {u} carries another utterance's ids"
        assert (edits, w, failed) == (len(range(0, N, 4)), words, 0)
