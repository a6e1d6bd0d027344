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


def _all_gather_padded(local, group=None, device=None):
    """all_gather of variable-length 1-D/2-D arrays of one dtype: the counts first, then buffers padded to the largest
    count.  -> list of per-rank arrays (every rank gets all of them)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    tail = local.shape[1:]
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c.item()) for c in cnts]
    m = max(max(counts), 1)
    t = torch.from_numpy(np.array(local, order="C", copy=True))      # a writable copy: frombuffer views are read-only
    buf = torch.zeros((m,) + tuple(tail), dtype=t.dtype, device=dev)
    if local.shape[0]:
        buf[:local.shape[0]] = t.to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    return [b[:c].cpu().numpy() for b, c in zip(bufs, counts)]


def gather_records_device(rec, group=None):
    """Variable-length gather of fixed-width DEVICE records (torch tensor [n_r, k] on this rank's GPU, e.g. the hit records
    rsk_mu_gapless_hits_dev appended): counts, then one all_gather of buffers padded to the largest count -- device to
    device over RCCL / xGMI, nothing crosses the host.  -> the concatenation in rank order, a device tensor, on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    cnt = torch.tensor([rec.shape[0]], dtype=torch.int64, device=rec.device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c.item()) for c in cnts]
    m = max(max(counts), 1)
    buf = torch.zeros((m,) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
    buf[:rec.shape[0]] = rec
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def gather_rows(rows, dst=0, group=None, device=None, all_ranks=False):
    """Variable-length gather of fixed-width records (int32 [n_r, k] per rank): all_gather of the counts, then an
    all_gather of buffers padded to the largest count (hit buffers are tiny compared with the pair space, so padding
    costs nothing).  Returns the concatenation in rank order on `dst` (None elsewhere), or on every rank with all_ranks."""
    import torch.distributed as dist
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    if rows.ndim != 2:
        raise ValueError("rows must be [n, k]")
    parts = _all_gather_padded(rows, group=group, device=device)
    if not all_ranks and dist.get_rank(group) != dst:
        return None
    return np.concatenate(parts, axis=0)


def gather_text(text, dst=0, group=None, device=None):
    """Concatenation (rank order) of every rank's text on rank `dst`, None elsewhere: the hit tables of the target
    shards (one exchange at the end of a search, tens of MB at most) as byte counts + one padded uint8
    all_gather -- tensors only, so the same call runs over RCCL (device = the rank's GPU) and gloo."""
    import torch.distributed as dist
    raw = np.frombuffer(text.encode(), dtype=np.uint8)
    parts = _all_gather_padded(raw, group=group, device=device)
    if dist.get_rank(group) != dst:
        return None
    return b"".join(p.tobytes() for p in parts).decode()


def merge_rskdb(parts):
    """RSKDB1 containers (bytes / uint8 arrays: 8-byte magic, chain count, feature count, records) of consecutive chain slices -> one
    container with the slices' chains in order"""
    import struct
    bodies, n, nfeat = [], 0, None
    for p in parts:
        b = p.tobytes() if hasattr(p, "tobytes") else bytes(p)
        if len(b) < 16 or b[:8] != b"RSKDB1\0\0":
            raise ValueError("merge_rskdb: not an RSKDB1 container")
        k, f = struct.unpack("<II", b[8:16])
        if nfeat is not None and f != nfeat:
            raise ValueError("merge_rskdb: feature counts differ")
        nfeat = f
        n += k
        bodies.append(b[16:])
    return b"RSKDB1\0\0" + struct.pack("<II", n, nfeat if nfeat is not None else 8) + b"".join(bodies)


def featurise_sharded(ctx, bca, out_rskdb, mode, group=None, device=None, **kw):
    """One process per GPU, a self search from a .bca file: rank r featurises slice r of the chains (DSS profiles, Mu letters,
    self-rev scores: rsk_bca_to_rskdb, contiguous slices balanced by residues), the slices' containers are all-gathered (~26 bytes
    per residue: the exchange step of this path) and every rank writes the whole set as `out_rskdb`, the prepared form rsk_search
    loads without featurising.  (Measured r06 on one GPU, 11,211 chains: every shard of 8 paid 0.27 s of load + featurisation +
    self-rev for ALL chains against 0.24 s for its share of the pairs.)"""
    import os
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    part = "%s.part%d" % (out_rskdb, rank)
    ctx.bca_to_rskdb(bca, part, mode, shard_index=rank, shard_count=world, **{k: v for k, v in kw.items() if k in ("selfrev0",)})
    raw = np.fromfile(part, dtype=np.uint8)
    os.remove(part)
    parts = _all_gather_padded(raw, group=group, device=device)
    with open(out_rskdb, "wb") as f:
        f.write(merge_rskdb(parts))
    return sum(len(p) for p in parts)


def search_sharded(ctx, query, out_tsv, mode, db=None, group=None, device=None, **kw):
    """One process per GPU: every rank runs its shard of the search, then the hit tables are gathered on rank 0, which
    writes `out_tsv`.  -db mode: a contiguous target range balanced by residues with the queries replicated; self
    search: a window of the set's length order balanced by DP cells + one world-th of the long-chain pair list (rsk_search with
    shard_index = rank, shard_count = world size): no collective on the pair path; from a .bca file the ranks first featurise
    one slice of the chains each and all-gather the prepared containers (featurise_sharded).  -fast -db: the prefilter's per-query top-B is a reduction over all targets,
    so the ranks exchange their prefilter triples once (all_gather) between the prefilter and the alignment stage
    (rsk_fast_shard_*): all of them by default -- every rank then replays the reference's bags over the union and the
    table is the single-GPU one --, or with exchange="topb" only the local top-B lists (own tie rule at the cut).
    Returns (hits of all ranks, stats) on rank 0, (local hits, stats) elsewhere."""
    import os
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    part = "%s.rank%d" % (out_tsv, rank)
    if mode == "fast" and db and dist.is_initialized():
        keeptmp = kw.pop("keeptmp", 0)
        exact = kw.pop("exchange", "all") != "topb"
        sh = ctx.fast_shard_open(query, db, shard_index=rank, shard_count=world, **kw)
        try:
            allrows = gather_rows(sh.triples() if exact else sh.candidates(), group=group, device=device, all_ranks=True)
            nhits, stats = sh.finish(allrows, part, tmp_tsv=(out_tsv + ".prefilter.tmp") if keeptmp and rank == 0 else None, exact=exact)
        finally:
            sh.close()
    else:
        prepared = None
        if db is None and query.endswith(".bca") and dist.is_initialized() and world > 1:
            prepared = "%s.rank%d.all.rskdb" % (out_tsv, rank)
            featurise_sharded(ctx, query, prepared, mode, group=group, device=device, **kw)
        try:
            nhits, stats = ctx.search(prepared or query, part, mode, db=db, shard_index=rank, shard_count=world, **kw)
        finally:
            if prepared and os.path.exists(prepared):
                os.remove(prepared)
    with open(part) as f:
        text = f.read()
    os.remove(part)
    if not dist.is_initialized():
        with open(out_tsv, "w") as f:
            f.write(text)
        return nhits, stats
    # (a process group of ONE rank still goes through the collective: that is how a single-GPU box exercises RCCL)
    merged = gather_text(text, dst=0, group=group, device=device)
    if rank == 0:
        with open(out_tsv, "w") as f:
            f.write(merged)
        return merged.count("\n"), stats
    return nhits, stats
