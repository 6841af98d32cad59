"""-m gpu: the `nccl` (= RCCL on ROCm) branch of bench.py on the one GPU this box has: bench.py as a FRESH child process under
`python -m torch.distributed.run --nproc-per-node 1` with OAVSR_BENCH_FORCE_DIST=1, i.e. init_process_group("nccl"), the barrier / max
over ranks bracket of the timed region and dist.gather_hypotheses (one all_gather of {utt_id, n, ids} records) all run
for real.  The 8-GPU curve is the driver's to measure; this is what can be executed before an 8-GPU node is available."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_under_torchrun_with_the_rccl_backend():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OAVSR_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--calls-per-step", "4", "--frames", "100",
           "--no-cpu-baseline", "--no-roofline", "--no-grouped"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, f"rc={r.returncode}\n{r.stderr[-3000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["gathered_hypotheses"] == d["config"]["batch_per_call"]          # the all_gather returned every utterance of the last call
    assert d["config"]["dist_backend"] == "nccl"
    assert d["config"]["ids_verified"] is True


def test_testset_shaped_run_under_torchrun_with_the_rccl_backend():
    """`bench.py --testset N` (SURVEY.md 8d C5 / BASELINE.json configs[4] shape: a fixed set of utterances with lengths U[1 s, T], duration-sorted
    sharding -> batches -> harness.evaluate(in_flight=3) -> all_gather of the hypotheses + all_reduce of the WER counters; strong scaling) on
    the one GPU this box has, with the RCCL branch forced: every utterance comes back exactly once, the counters are reduced, and the
    sample of batches re-decoded in isolation matches (the run exits non-zero otherwise)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OAVSR_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--testset", "70", "--frames", "100"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, f"rc={r.returncode}\n{r.stderr[-3000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    c = d["config"]
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["value"] > 0
    assert c["utterances"] == 70 and c["gathered_hypotheses"] == 70 and c["utterances_per_rank"] == [70]
    assert c["wer_counters"]["words"] == sum(5 + u % 9 for u in range(70)) and c["wer_counters"]["failed"] == 0
    assert c["dist_backend"] == "nccl" and c["ids_verified"] is True
