"""The N > 1 path on CPU: two gloo processes shard a batch, broadcast the reference chunk, align their
blocks (the CPU oracle stands in for the device call -- test infrastructure only) and gather scores and
CIGARs on rank 0, which must equal the unsharded result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import common
    import oracle
    from gonomics_amd import align, shard
    reads, chunk = common.c2_workload(5, 37, read_len=60, chunk_len=400)  # every rank can regenerate the reads
    ref = torch.from_numpy(chunk.copy() if rank == 0 else np.zeros_like(chunk))
    shard.broadcast_reference(ref, src=0)
    chunk_r = ref.numpy()
    assert np.array_equal(chunk_r, chunk)
    n = reads.shape[0]
    b, e = shard.partition(n, world, rank)
    k = e - b
    res = oracle.align_batch_windows(oracle.MODE_AFFINE, align.HumanChimpTwoScoreMatrix, -600, -150,
                                     reads[b:e].reshape(-1), np.arange(k) * 60, np.full(k, 60), chunk_r,
                                     np.zeros(k, np.int64), np.full(k, 400))
    got = shard.gather_results(torch.from_numpy(res[0].copy()), torch.from_numpy(np.frombuffer(res[1].tobytes(), dtype=np.uint8).copy()),
                               torch.from_numpy(res[2].copy()), dst=0)
    if rank == 0:
        full = oracle.align_batch_windows(oracle.MODE_AFFINE, align.HumanChimpTwoScoreMatrix, -600, -150,
                                          reads.reshape(-1), np.arange(n) * 60, np.full(n, 60), chunk,
                                          np.zeros(n, np.int64), np.full(n, 400))
        gops = got[1].numpy().view(oracle.CIGAR_DTYPE)  # compare fields: the 7 pad bytes of a record are unspecified
        ok = (np.array_equal(got[0].numpy(), full[0]) and np.array_equal(got[2].numpy(), full[2])
              and np.array_equal(gops["run_length"], full[1]["run_length"]) and np.array_equal(gops["op"], full[1]["op"]))
        open(os.path.join(outdir, "ok"), "w").write("1" if ok else "0")
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert open(os.path.join(str(tmp_path), "ok")).read() == "1"


def test_partitions():
    from gonomics_amd import shard
    assert [shard.partition(10, 3, r) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert [shard.partition(2, 4, r) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    a = np.array([100, 100, 100, 100, 400, 400])
    b = np.array([10, 10, 10, 10, 10, 10])
    bd = shard.partition_by_cells(a, b, 2)
    assert bd[0] == 0 and bd[-1] == 6 and 4 <= bd[1] <= 5
    assert (np.diff(shard.partition_by_cells(a, b, 8)) >= 0).all()
