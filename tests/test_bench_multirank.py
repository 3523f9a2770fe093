"""The multi-rank control flow of bench.py (barriers, collectives inside the sharded step, which rank prints) rehearsed
with two ranks on ONE GPU: gloo instead of RCCL (OKGE_BENCH_ONE_GPU=1), real HIP kernels.  Guards against rank-asymmetric
code around collectives (a rank-0-only timing pass once deadlocked every N > 1 run).  Not a measurement."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_bench_two_ranks_one_gpu_rehearsal(okge_lib):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OKGE_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                      # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["global_batch"] == 1024 and d["roofline"]["kernels_us"]["fused_tile_train"] > 0
    assert d["cpu_baseline"] is None and d["vs_baseline"] is None


@pytest.mark.gpu
def test_bench_four_ranks_one_gpu_rehearsal(okge_lib):
    """the control flow of a wider run -- shard ranges with a short last shard (14 543 rows over 4 ranks), exchange plans that
    fall back to None under Zipf ids (the `olp` leg), which rank prints -- with FOUR gloo ranks on one GPU (the GPU boxes
    allow six processes on a card; this test process and the launcher count: five ranks were killed by that guard, and the
    eight-rank run is the driver's)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OKGE_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["config"]["global_batch"] == 2048 and d["value"] > 0
    assert d["olp"]["n_gpus"] == 4 and d["olp"]["scaling"] == "strong" and d["olp"]["value"] > 0
    assert d["configs"] is None and d["cpu_baseline"] is None
