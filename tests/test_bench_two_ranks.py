"""The N > 1 flow of bench.py as the driver launches it (python -m torch.distributed.run, one rank per GPU), on this 1-GPU box: two
ranks share cuda:0 over gloo.  No scaling claim -- the plumbing must be right by construction before the first node
runs it: the RCCL-shaped broadcast of the chunk (shard.py), barrier + max-over-ranks timing, the ordered gather of scores / offsets /
CIGAR blob on rank 0, the per-rank clocks, and the one-process leg of the C ABI (two contexts behind gnx_init_devices)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_share_one_gpu(gpu_lib):
    pairs = 4096
    gpu_lib.lib().gnx_shutdown()  # this process gives its workspace back while the ranks run
    try:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("GNX_FP_SMALL", None)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--dist-backend", "gloo", "--pairs", str(pairs), "--steps", "2", "--warmup", "1",
               "--no-extras", "--no-cpu", "--one-process", "--ws-gb", "16"]
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        out = json.loads(line)
    finally:
        gpu_lib.check(gpu_lib.lib().gnx_init(0, 8 << 30))
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["bit_exact_sample"] is True
    assert out["gathered_pairs"] == 2 * pairs                     # rank 0 holds every rank's results ...
    assert out["gather_order_ok"] is True                         # ... in input order: each rank's first / last score, run counts and run-length sum sit at its block's places
    pr = out["per_rank_ms_per_step"]
    assert len(pr["all"]) == 2 and 0 < pr["min"] <= pr["max"] <= out["ms_per_step"] * 1.05
    op = out["one_process"]
    assert "error" not in op, op
    assert op["contexts"] == 2 and op["pairs"] == 2 * pairs
    assert op["windows"]["transport"] == "peer copies" and op["windows"]["equals_one_gpu"] and op["by_offset"]["equals_one_gpu"]


def test_bench_two_ranks_long_series(gpu_lib):
    """config 5's multi-GPU leg as bench.py runs it (--series long --gpus N): one shared list of ConstGap 20 kb x 100 kb pairs cut into contiguous
    blocks of equal cells by shard.partition_by_cells, one block per rank, the ordered gather on rank 0 -- two ranks on this box's one GPU over gloo"""
    pairs = 48
    gpu_lib.lib().gnx_shutdown()
    try:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--dist-backend", "gloo", "--series", "long", "--pairs", str(pairs), "--steps", "1", "--warmup", "1",
               "--no-extras", "--no-cpu", "--verify", "1", "--ws-gb", "16"]
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    finally:
        gpu_lib.check(gpu_lib.lib().gnx_init(0, 8 << 30))
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["bit_exact_sample"] is True
    assert out["gathered_pairs"] == 2 * pairs and out["gather_order_ok"] is True
