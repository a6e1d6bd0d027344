"""CPU: the shard arithmetic behind the multi-GPU searches, through the C-ABI (rsk_shard_range; no device).
kind 0 = self search: shard r scores the pairs (i <= j) whose target j lies in [lo_r, hi_r) -- the rectangle
chains[0, lo) x chains[lo, hi) plus the triangle of chains[lo, hi) -- with equal DP cells per shard;
kind 1 = -db search: contiguous chain ranges with equal residues.  The same function serves the one-process form
(DBSearcher::m_Devices / RSK_DEVICES: one context per device) and the one-process-per-GPU form (shard_index / shard_count)."""
import os

import numpy as np
import pytest

import fixtures as fx
from reseek_amd import capi


def lengths_sets():
    rng = np.random.default_rng(3)
    yield "scop40", fx.scop40_lengths().astype(np.uint32)
    yield "lognormal_tail", np.clip(rng.lognormal(np.log(250), 0.75, 20000), 20, 5000).astype(np.uint32)
    yield "sorted", np.sort(fx.scop40_lengths()[:3000]).astype(np.uint32)
    yield "tiny", np.array([7, 300, 12], np.uint32)


@pytest.mark.parametrize("count", [1, 2, 3, 8])
@pytest.mark.parametrize("kind", [0, 1])
def test_shards_tile_the_set_in_order(kind, count):
    for name, L in lengths_sets():
        prev = 0
        for r in range(count):
            lo, hi = capi.shard_range(kind, L, r, count)
            assert lo == prev and lo <= hi <= len(L), (name, r)
            prev = hi
        assert prev == len(L), name


@pytest.mark.parametrize("count", [2, 3, 8])
def test_self_shards_balance_cells_and_cover_every_pair_once(count):
    for name, L in lengths_sets():
        if len(L) < 100:
            continue
        Lf = L.astype(np.float64)
        pre = np.cumsum(Lf)                                  # residues of chains 0..j
        cells_of_target = pre * Lf                           # cells of the pairs (i <= j) of target j
        total = cells_of_target.sum()
        shares, pairs = [], 0
        for r in range(count):
            lo, hi = capi.shard_range(0, L, r, count)
            shares.append(cells_of_target[lo:hi].sum())
            # rectangle [0, lo) x [lo, hi) + triangle of [lo, hi)
            pairs += lo * (hi - lo) + (hi - lo) * (hi - lo + 1) // 2
        assert pairs == len(L) * (len(L) + 1) // 2, name
        assert abs(sum(shares) - total) <= 1e-6 * total
        # each share within one target's worth of the ideal
        biggest = cells_of_target.max()
        for s in shares:
            assert abs(s - total / count) <= biggest + 1e-6 * total, (name, shares)


@pytest.mark.parametrize("count", [2, 3, 8])
def test_residue_shards_balance_residues(count):
    for name, L in lengths_sets():
        if len(L) < 100:
            continue
        total = int(L.astype(np.int64).sum())
        for r in range(count):
            lo, hi = capi.shard_range(1, L, r, count)
            assert abs(int(L[lo:hi].astype(np.int64).sum()) - total / count) <= int(L.max()), (name, r)


def test_more_shards_than_chains_and_bad_arguments():
    L = np.array([50, 60], np.uint32)
    got = [capi.shard_range(1, L, r, 8) for r in range(8)]
    assert sum(hi - lo for lo, hi in got) == 2 and all(lo <= hi for lo, hi in got)
    with pytest.raises(RuntimeError):
        capi.shard_range(0, L, 3, 3)
    with pytest.raises(RuntimeError):
        capi.shard_range(3, L, 0, 1)


@pytest.mark.parametrize("count", [1, 2, 3, 8])
def test_self_windows_of_the_length_order(count):
    """kind 2 (r06): windows of POSITIONS of the set's length order (stable sort by length) -- the shards of a self search with
    a Mu filter.  They tile [0, n) in order and hold equal DP cells of the triangle: position p closes the pairs of the chain
    there with every chain at a position <= p."""
    for name, L in lengths_sets():
        if len(L) < 100:
            continue
        Ls = np.sort(L, kind="stable").astype(np.float64)
        cells = Ls * np.cumsum(Ls)                         # cells position p closes
        got = [capi.shard_range(2, L, r, count) for r in range(count)]
        assert got[0][0] == 0 and got[-1][1] == len(L) and all(got[r][1] == got[r + 1][0] for r in range(count - 1)), (name, got)
        share = [cells[lo:hi].sum() / cells.sum() for lo, hi in got]
        assert max(abs(x - 1.0 / count) for x in share) <= cells.max() / cells.sum() + 1e-12, (name, share)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("scheme", ["targets", "fold"])
def test_shardplan_launch_lists_tile_the_triangle(world, scheme):
    """tools/exp/shardplan.py (the rectangle + triangle cuts tools/exp/shard_times.py measures against the window scheme
    bench.py runs): every pair i <= j in exactly one launch of one rank, cell shares within one target's worth of equal."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "exp"))
    import shardplan as sp
    for name, L in lengths_sets():
        if len(L) < 100:
            continue
        Ls = np.sort(L)
        p = sp.plan(Ls, world, scheme)
        assert len(p) == world and sp.check_plan(len(Ls), p), (name, world, scheme)
        shares = sp.cell_shares(Ls, world, scheme)
        assert abs(sum(shares) - 1.0) < 1e-9
        biggest = (np.cumsum(Ls.astype(np.float64)) * Ls).max() / sp.cell_prefix(Ls)[-1]
        assert max(shares) - 1.0 / world <= 2 * biggest + 1e-9, (name, shares)
    # a plan that loses or doubles a box is caught
    L0 = np.sort(list(lengths_sets())[0][1])
    bad = sp.plan(L0, 2, "targets")
    bad[1] = bad[1][:-1]
    assert not sp.check_plan(len(L0), bad)
