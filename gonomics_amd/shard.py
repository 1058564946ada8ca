"""Single-node multi-GPU sharding of a batch of independent pairs.

The path has no exchange step: every pair is an independent DP.  One process per GPU
(torch.distributed, backend "nccl" == RCCL over xGMI on MI355X, "gloo" on CPU in the tests):
  * the shared reference chunk is broadcast once from rank 0           (broadcast_reference)
  * pairs are split into contiguous blocks of (nearly) equal DP cells (partition / partition_by_cells)
  * each rank aligns its block with the C-ABI batch entry point
  * scores, CIGAR offsets and the CIGAR blob are gathered on rank 0 in input order   (gather_results)
There is no data-path collective inside the timed region; the reference has nothing comparable
(goroutine pools only, /root/reference/genomeGraph/routines.go:12-65).
"""
import numpy as np
import torch
import torch.distributed as dist


def partition(n_pairs, world, rank):
    """Contiguous block [begin, end) of rank `rank`; sizes differ by at most one."""
    base, rem = divmod(int(n_pairs), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def partition_by_cells(a_len, b_len, world):
    """Boundaries (world+1,) of contiguous blocks with balanced sum(n*m)."""
    cells = np.asarray(a_len, dtype=np.float64) * np.asarray(b_len, dtype=np.float64)
    cum = np.concatenate([[0.0], np.cumsum(cells)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world):
        bounds.append(int(np.searchsorted(cum, total * r / world, side="left")))
    bounds.append(len(cells))
    for r in range(1, len(bounds)):
        bounds[r] = max(bounds[r], bounds[r - 1])
    return np.asarray(bounds, dtype=np.int64)


def broadcast_reference(t, src=0):
    """Broadcast the packed reference / chunk tensor from `src` to every rank (no-op for world 1)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return t


def _gather_var(t, dst, device):
    """Gather 1-D tensors of different lengths on dst (lengths first, then padded payloads)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=t.dtype, device=device)
    pad[: t.numel()] = t
    # all_gather rather than gather: the plainest ring collective of RCCL (gather is grouped send/recv); the payloads are a few
    # MB per rank (scores, offsets, 16-byte CIGAR records), so the extra copies on the other ranks do not matter
    bufs = [torch.zeros(mx, dtype=t.dtype, device=device) for _ in range(world)]
    dist.all_gather(bufs, pad)
    if rank != dst:
        return None
    return [b[:s] for b, s in zip(bufs, sizes)]


def gather_results(scores, ops_bytes, off, dst=0):
    """scores int64[n_local], ops_bytes uint8[16*total_local] (gnx_cigar records), off int64[n_local+1].
    Returns on dst the concatenation in rank (== input) order with rebased offsets, None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return scores, ops_bytes, off
    device = scores.device
    s = _gather_var(scores, dst, device)
    o = _gather_var(ops_bytes, dst, device)
    f = _gather_var(off, dst, device)
    if dist.get_rank() != dst:
        return None
    out_off = [torch.zeros(1, dtype=torch.int64, device=device)]
    base = 0
    for fr in f:
        out_off.append(fr[1:] + base)
        base += int(fr[-1].item())
    return torch.cat(s), torch.cat(o), torch.cat(out_off)
