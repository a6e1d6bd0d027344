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


def gather_text(text, dst=0, group=None):
    """Concatenation (rank order) of every rank's text on rank `dst`, None elsewhere: the hit tables of the
    target shards (one exchange at the end of a search, tens of MB at most)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    parts = [None] * world if rank == dst else None
    dist.gather_object(text, parts, dst=dst, group=group)
    return "".join(parts) if rank == dst else None


def search_sharded(ctx, query, out_tsv, mode, db=None, group=None, **kw):
    """One process per GPU: every rank runs its shard of the search (rsk_search with shard_index = rank,
    shard_count = world size; -db mode: a contiguous target range balanced by residues with the queries
    replicated; self search: a target range of the triangle balanced by DP cells), then the hit tables are
    gathered on rank 0, which writes `out_tsv`.  Returns (hits of all ranks, per-rank stats) on rank 0,
    (local hits, stats) elsewhere.  No collective on the data path."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    part = "%s.rank%d" % (out_tsv, rank)
    nhits, stats = ctx.search(query, part, mode, db=db, shard_index=rank, shard_count=world, **kw)
    with open(part) as f:
        text = f.read()
    import os
    os.remove(part)
    if world == 1:
        with open(out_tsv, "w") as f:
            f.write(text)
        return nhits, stats
    merged = gather_text(text, dst=0, group=group)
    if rank == 0:
        with open(out_tsv, "w") as f:
            f.write(merged)
        return merged.count("\n"), stats
    return nhits, stats
