"""Utterance-sharded data parallelism.  Utterances are independent units, so the hot path shards
with NO data-path collective: every rank holds a full weight replica and decodes its own utterances; the single
exchange step is one all_gather of fixed-width hypothesis records at the end (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  The reference has no inference-time parallelism at all (batch 1, one GPU,
eval_OmniAVSR.py:18-24)."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_utterances(durations: Sequence[float], world: int, rank: int) -> List[int]:
    """Sort by duration (longest first) and deal round-robin so every rank gets a similar length mix.
    Returns the utterance indices owned by `rank`, longest first (batches of similar length pad least)."""
    order = sorted(range(len(durations)), key=lambda i: (-durations[i], i))
    return order[rank::world]


def make_batches(owned: Sequence[int], batch: int) -> List[List[int]]:
    return [list(owned[i:i + batch]) for i in range(0, len(owned), batch)]


def pack_records(utt_ids: Sequence[int], ids: torch.Tensor, lens: torch.Tensor, width: int) -> torch.Tensor:
    """{utt_id:int32, n:int32, ids:int32[width]} per utterance -> int32 [n_utt, width + 2] (on ids.device)."""
    n = len(utt_ids)
    rec = torch.full((n, width + 2), -1, dtype=torch.int32, device=ids.device)
    rec[:, 0] = torch.tensor(list(utt_ids), dtype=torch.int32, device=ids.device)
    rec[:, 1] = lens.to(torch.int32)
    w = min(width, ids.shape[1])
    rec[:, 2:2 + w] = ids[:, :w].to(torch.int32)
    return rec


def gather_hypotheses(rec: torch.Tensor, n_total: int, group=None, return_failed: bool = False):
    """All-gathers the per-rank records (ranks may own different counts: padded to the max) and returns
    {utt_id: token ids} on every rank.  A rank reports an utterance it could not decode as a record with n = -1
    (`failed_record`); with `return_failed=True` the result is (hyp, failed utterance ids) -- failed = reported as failed by
    their rank, or owned by nobody -- instead of an exception, so one bad clip does not lose a whole evaluation."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        allrec = rec.cpu()
    else:
        cnt = torch.tensor([rec.shape[0]], dtype=torch.int64, device=rec.device)
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt, group=group)
        mx = int(max(int(c.item()) for c in cnts))
        pad = torch.full((mx, rec.shape[1]), -1, dtype=torch.int32, device=rec.device)
        pad[: rec.shape[0]] = rec
        outs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(outs, pad, group=group)
        allrec = torch.cat([o[: int(c.item())] for o, c in zip(outs, cnts)], 0).cpu()
    hyp: Dict[int, List[int]] = {}
    failed: List[int] = []
    for row in allrec.tolist():
        uid, n = row[0], row[1]
        if uid < 0:
            continue
        if n < 0:
            failed.append(uid)
            continue
        hyp[uid] = row[2:2 + n]
    if return_failed:
        seen = set(hyp) | set(failed)
        failed += [u for u in range(n_total) if u not in seen]
        return hyp, sorted(set(failed))
    if failed or len(hyp) != n_total:
        raise RuntimeError(f"gathered {len(hyp)} hypotheses, expected {n_total}; failed utterance ids: {sorted(failed)}")
    return hyp


def failed_record(utt_ids: Sequence[int], width: int, device="cpu") -> torch.Tensor:
    """Records for utterances this rank could not decode (n = -1), same shape as `pack_records` rows."""
    rec = torch.full((len(utt_ids), width + 2), -1, dtype=torch.int32, device=device)
    if len(utt_ids):
        rec[:, 0] = torch.tensor(list(utt_ids), dtype=torch.int32, device=device)
    return rec


def reduce_wer_counters(edits: int, words: int, device="cpu", group=None, failed: int = None):
    """all_reduce(sum) of {edits, words} when WER is computed distributed (lightning_OmniAVSR.py:206-219 semantics).  With `failed`
    (utterances this rank could not decode, harness.WerMeter.failed) the sum of those travels too and (edits, words, failed) comes back:
    a WER over fewer utterances than the test set holds must be reported as such."""
    t = torch.tensor([edits, words, 0 if failed is None else failed], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
    return (int(t[0]), int(t[1])) if failed is None else (int(t[0]), int(t[1]), int(t[2]))
