"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests).  The hot path shards by target chains with no data-path collective; the
only exchange is the gather of per-rank hit buffers."""
import numpy as np


def shard_by_residues(lengths, world_size):
    """Contiguous target ranges [(lo, hi), ...] balanced by the sum of chain lengths (the cost of a
    target against a fixed query set is proportional to its length)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    csum = np.concatenate([[0], np.cumsum(lengths)])
    total = csum[-1]
    bounds = [0]
    for r in range(1, world_size):
        bounds.append(int(np.searchsorted(csum, total * r / world_size, side="left")))
    bounds.append(n)
    for r in range(1, len(bounds)):
        bounds[r] = max(bounds[r], bounds[r - 1])
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def gather_rows(rows, dst=0, group=None, device=None):
    """Variable-length gather of fixed-width records (int32 [n_r, k] per rank) onto rank `dst`:
    all_gather of the counts, then an all_gather of buffers padded to the largest count (hit buffers
    are tiny compared with the pair space, so padding costs nothing).  Returns the concatenation in
    rank order on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    if rows.ndim != 2:
        raise ValueError("rows must be [n, k]")
    k = rows.shape[1]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    cnt = torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c.item()) for c in cnts]
    m = max(max(counts), 1)
    buf = torch.zeros((m, k), dtype=torch.int32, device=dev)
    if rows.shape[0]:
        buf[:rows.shape[0]] = torch.from_numpy(rows).to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    if rank != dst:
        return None
    return np.concatenate([b[:c].cpu().numpy() for b, c in zip(bufs, counts)], axis=0)
