"""CPU: bench.py's multi-GPU launcher (no GPU is touched by the parent) and the dist helpers it uses inside the timed region."""
import json
import os
import subprocess
import sys

import torch

from omni_avsr_amd.dist import failed_record, gather_hypotheses, pack_records
from omni_avsr_amd.modeling import whisper_keep_tokens

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = dict(os.environ, **env)
    e.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=300)


def test_gpus_flag_fails_loudly_without_enough_devices():
    # more GPUs than any node has: the answer must not depend on the host this CPU test runs on
    r = _run(["--gpus", "1024", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and "needs 1024 GPUs" in r.stderr and r.stdout.strip() == ""


def test_gpus_flag_builds_a_torchrun_child_command(monkeypatch):
    """With devices present the parent would start `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>`."""
    code = ("import sys, json, torch; torch.cuda.device_count = lambda: 8; sys.argv = ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1'];"
            "import runpy; runpy.run_path(%r, run_name='__main__')" % os.path.join(ROOT, "bench.py"))
    e = dict(os.environ, OAVSR_BENCH_PRINT_LAUNCH="1")
    e.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=300)
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip().splitlines()[-1])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]


def test_gather_reports_failed_utterances_instead_of_asserting():
    ids = torch.tensor([[5, 6, 7, 0], [9, 0, 0, 0]], dtype=torch.int32)
    rec = torch.cat([pack_records([0, 2], ids, torch.tensor([3, 1]), 4), failed_record([3], 4)], 0)
    hyp, failed = gather_hypotheses(rec, 5, return_failed=True)
    assert hyp == {0: [5, 6, 7], 2: [9]} and failed == [1, 3, 4]          # 3 reported by its rank, 1 and 4 owned by nobody
    try:
        gather_hypotheses(rec, 5)
        raise AssertionError("strict mode must raise")
    except RuntimeError as e:
        assert "failed utterance ids: [3]" in str(e)


def test_whisper_keep_clamps_at_the_encoder_length():
    assert whisper_keep_tokens(480000) == 1500 and whisper_keep_tokens(600000) == 1500 and whisper_keep_tokens(16000) == 50
    assert whisper_keep_tokens(100) == 25


def test_rank_pinning_gives_each_local_rank_its_own_cpus():
    """bench.py: pin_rank (multi-GPU runs pin every rank before it touches the GPU).  In a child process each -- the affinity sticks to the caller."""
    code = ("import json, os, sys; sys.path.insert(0, %r); import bench; r = bench.pin_rank(int(sys.argv[1]), 2); "
            "print(json.dumps({'ret': r, 'aff': sorted(os.sched_getaffinity(0))}))") % ROOT
    before = sorted(os.sched_getaffinity(0))
    outs = []
    for local in (0, 1):
        p = subprocess.run([sys.executable, "-c", code, str(local)], capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    for o in outs:
        assert o["ret"] is not None and o["ret"]["cpus"] == len(o["aff"]) >= 1
        assert set(o["aff"]) <= set(before)
    if len(before) >= 2:                       # without NUMA information about GPUs (this container): disjoint even slices
        assert not (set(outs[0]["aff"]) & set(outs[1]["aff"]))
    assert sorted(os.sched_getaffinity(0)) == before
