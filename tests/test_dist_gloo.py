"""N > 1 plumbing on CPU (gloo, world_size 2): target sharding + the hit-buffer gather used by the
multi-GPU path (bench.py --gpus N, one process per GPU over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from reseek_amd import dist as rdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lengths = np.random.default_rng(0).integers(5, 1400, 1000)
    lo, hi = rdist.shard_by_residues(lengths, world)[rank]
    # each rank "finds" a deterministic, rank-dependent number of hits inside its target shard
    rng = np.random.default_rng(100 + rank)
    n = 37 + 100 * rank
    rows = np.stack([rng.integers(0, 256, n), rng.integers(lo, hi, n), rng.integers(0, 5000, n)], axis=1).astype(np.int32)
    got = rdist.gather_rows(rows, dst=0)
    # empty contribution from one rank must also work
    got2 = rdist.gather_rows(rows[:0] if rank == 1 else rows, dst=0)
    dist.barrier()
    if rank == 0:
        q.put((got, got2))
    dist.destroy_process_group()


def test_shards_cover_and_balance():
    lengths = np.random.default_rng(1).integers(5, 1400, 11211)
    for world in (1, 2, 4, 8):
        sh = rdist.shard_by_residues(lengths, world)
        assert sh[0][0] == 0 and sh[-1][1] == len(lengths)
        assert all(sh[i][1] == sh[i + 1][0] for i in range(world - 1))
        sums = [lengths[a:b].sum() for a, b in sh]
        assert max(sums) - min(sums) <= 2 * lengths.max()


def test_gather_rows_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, got2 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lengths = np.random.default_rng(0).integers(5, 1400, 1000)
    want = []
    for rank in range(world):
        lo, hi = rdist.shard_by_residues(lengths, world)[rank]
        rng = np.random.default_rng(100 + rank)
        n = 37 + 100 * rank
        want.append(np.stack([rng.integers(0, 256, n), rng.integers(lo, hi, n), rng.integers(0, 5000, n)], axis=1).astype(np.int32))
    assert np.array_equal(got, np.concatenate(want))
    assert np.array_equal(got2, want[0])


def _text_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    text = "".join("q%d\tt%d\t%d\n" % (rank, k, k * k) for k in range(5 + 3 * rank)) if rank != 1 else ""
    got = rdist.gather_text(text, dst=0)
    dist.barrier()
    if rank == 0:
        q.put(got)
    dist.destroy_process_group()


def test_gather_text_world3_gloo():
    """the end-of-search exchange of search_sharded: per-rank hit tables (one may be empty) -> rank 0, rank order"""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_text_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = q.get(timeout=120)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = "".join("q0\tt%d\t%d\n" % (k, k * k) for k in range(5)) + "".join("q2\tt%d\t%d\n" % (k, k * k) for k in range(11))
    assert got == want


def _records_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7 + rank)
    n = 0 if rank == 1 else 11 + 50 * rank                 # one rank without hits
    rec = torch.from_numpy(rng.integers(0, 1 << 20, (n, 3)).astype(np.int32))
    got = rdist.gather_records_device(rec)                 # tensors in, tensor out: the same call runs on GPU buffers over RCCL
    dist.barrier()
    q.put((rank, got.numpy().copy()))
    dist.destroy_process_group()


def test_gather_records_world3_gloo():
    """the N > 1 leg of bench.py: per-rank hit-record buffers (tensors; on the GPU the buffers the kernel appended to) ->
    every rank, rank order, no host round trip in the call itself"""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_records_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.concatenate([np.random.default_rng(7 + r).integers(0, 1 << 20, (0 if r == 1 else 11 + 50 * r, 3)).astype(np.int32) for r in range(world)])
    for r in range(world):
        assert np.array_equal(res[r], want)


def test_merge_rskdb_containers():
    """the prepared containers of consecutive chain slices (one per rank of a multi-GPU self search from a .bca file) -> one"""
    import struct
    from reseek_amd import dist as rdist
    mk = lambda n, body: b"RSKDB1\0\0" + struct.pack("<II", n, 8) + body
    assert rdist.merge_rskdb([mk(2, b"ab"), mk(0, b""), mk(3, b"cde")]) == mk(5, b"abcde")
    assert rdist.merge_rskdb([np.frombuffer(mk(1, b"z"), np.uint8)]) == mk(1, b"z")
    with pytest.raises(ValueError):
        rdist.merge_rskdb([b"NOTRSKDB" + b"\0" * 8])
    with pytest.raises(ValueError):
        rdist.merge_rskdb([mk(1, b"a"), b"RSKDB1\0\0" + struct.pack("<II", 1, 7) + b"b"])
