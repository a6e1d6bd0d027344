#!/usr/bin/env python3
"""Multi-GPU search smoke test: `python -m torch.distributed.run --nproc-per-node N tools/search_dist_demo.py`
runs reseek_amd.dist.search_sharded on the q100 fixture (self search, -db mode and the two-stage -fast -db path) with one process per GPU
and checks the gathered hit table against the reference's golden table on rank 0.
RSK_BENCH_ONE_DEVICE=1: all ranks share cuda:0 and use gloo (single-GPU boxes)."""
import gzip
import os
import sys
import tempfile

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import reseek_amd  # noqa: E402
from reseek_amd import dist as rdist  # noqa: E402

COLS = "query+target+qlo+qhi+ql+tlo+thi+tl+pctid+pvalue+evalue+cigar+dpscore+lddt+newts+ids+gaps+aq"


def main():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    one = os.environ.get("RSK_BENCH_ONE_DEVICE", "") == "1"
    if one:
        local = 0
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # RSK_DIST_FORCE=1: a process group even for ONE rank, so that a single-GPU box runs the RCCL collectives of this path
    use_dist = world > 1 or os.environ.get("RSK_DIST_FORCE", "") == "1"
    if use_dist:
        if one:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    coll_dev = torch.device("cpu") if (one or not use_dist) else torch.device("cuda", local)      # RCCL gathers device tensors
    ctx = reseek_amd.Ctx(local, stream=torch.cuda.current_stream().cuda_stream)
    golden = os.path.join(ROOT, "tests", "golden")
    with tempfile.TemporaryDirectory() as td:
        bca = os.path.join(td, "q100_rank%d.bca" % rank)
        with gzip.open(os.path.join(golden, "q100.bca.gz"), "rb") as f, open(bca, "wb") as g:
            g.write(f.read())
        palms = os.path.join(td, "palms_rank%d.bca" % rank)
        with gzip.open(os.path.join(golden, "palms.bca.gz"), "rb") as f, open(palms, "wb") as g:
            g.write(f.read())
        ok = True
        # the last leg: bags of 5 overflow for nearly every query, the exchange has to reproduce the reference's cut
        # (self searches from a .bca file with more than one rank: every rank featurises a slice of the chains, the prepared
        # containers are all-gathered -- palms: every chain is long, self-rev scores through the long-chain device batch)
        for q, mode, db, gold, kw in ((bca, "sensitive", None, "hits_q100_sensitive.tsv.gz", {}), (palms, "sensitive", None, "hits_palms_sensitive.tsv.gz", {}),
                                      (bca, "sensitive", bca, "hits_q100_db_q100_sensitive.tsv.gz", {}),
                                      (bca, "fast", bca, "hits_q100_db_q100_fast.tsv.gz", {}), (bca, "fast", bca, "hits_q100_db_q100_fast_rsb5.tsv.gz", {"rsb_size": 5})):
            out = os.path.join(td, "hits_rank%d.tsv" % rank)
            n, st = rdist.search_sharded(ctx, q, out, mode, db=db, columns=COLS, device=coll_dev, **kw)
            if rank == 0:
                got = sorted(open(out).read().splitlines())
                want = sorted(gzip.open(os.path.join(golden, gold)).read().decode().splitlines())
                ok = ok and got == want
                print("world %d %s %s %s: %d hits, %s" % (world, os.path.basename(q).split("_")[0], mode, "db" if db else "self", len(got),
                                                       "identical to the reference" if got == want else "MISMATCH"))
    if use_dist:
        if not one:
            # the device-record gather of bench.py's N > 1 leg on this backend
            rec = torch.arange(3 * (5 + rank), dtype=torch.int32, device=coll_dev).reshape(-1, 3)
            allrec = rdist.gather_records_device(rec)
            assert allrec.is_cuda and allrec.shape[0] == sum(5 + r for r in range(world))
            print("world %d gather_records_device over %s: %d records" % (world, dist.get_backend(), allrec.shape[0]))
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
