"""Cuts of the self-search pair space {(i, j): i <= j} of ONE length-sorted chain set into per-rank launch lists
(pairs are independent, the reference deals them to threads through one locked counter, runself.cpp:72-99;
here a rank's share is fixed up front, balanced by DP cells, and made of a few LARGE launches -- the gapless ring kernel
wants >= ~2,000 (ring, target block) work items per launch).

A plan is, per rank, a list of launches (q_lo, q_hi, t_lo, t_hi, triangle):
  triangle = True : the pairs i <= j of chains[q_lo:q_hi) (t range == q range)
  triangle = False: the rectangle chains[q_lo:q_hi) x chains[t_lo:t_hi), q_hi <= t_lo
Every pair (i <= j) is in exactly one launch of exactly one rank (checked by `check_plan`, tests/test_shard_ranges.py).

Schemes (`plan(lens, world, scheme)`):
  "targets"  rank r = one target range [b_r, b_r+1) of `world` cell-balanced ranges: rectangle chains[0:b_r) x range +
             the range's triangle.  The last rank's range is thin (the ~2 % longest chains against everything).
  "fold"     2 * world cell-balanced target ranges, rank r takes ranges r and 2 * world - 1 - r: every rank has a wide
             range of short targets AND a thin range of long ones, so whatever a thin range costs beyond its cells (one
             LDS profile build per <= 64 targets instead of per 1,024) is spread over all ranks instead of landing on the
             last one.  Four launches per rank.
Pure host arithmetic (numpy); no GPU, no torch."""
import numpy as np


def cell_prefix(lens):
    """cum[j] = DP cells of the pairs (i <= t) with target t < j"""
    lens = np.asarray(lens, np.float64)
    return np.concatenate([[0.0], np.cumsum(lens * np.cumsum(lens))])


def target_bounds(lens, parts):
    cum = cell_prefix(lens)
    n = len(lens)
    b = [int(np.searchsorted(cum, cum[-1] * r / parts, side="left")) for r in range(parts)] + [n]
    for k in range(1, len(b)):
        b[k] = max(b[k], b[k - 1])
    return b


def _range_launches(lo, hi):
    out = []
    if hi > lo:
        if lo > 0:
            out.append((0, lo, lo, hi, False))
        out.append((lo, hi, lo, hi, True))
    return out


def plan(lens, world, scheme="fold"):
    """-> list (per rank) of launch lists"""
    if world <= 1:
        return [[(0, len(lens), 0, len(lens), True)]]
    if scheme == "targets":
        b = target_bounds(lens, world)
        return [_range_launches(b[r], b[r + 1]) for r in range(world)]
    if scheme == "fold":
        b = target_bounds(lens, 2 * world)
        return [_range_launches(b[r], b[r + 1]) + _range_launches(b[2 * world - 1 - r], b[2 * world - r]) for r in range(world)]
    raise ValueError("unknown shard scheme " + scheme)


def launch_cells(lens, launch):
    lens = np.asarray(lens, np.float64)
    q_lo, q_hi, t_lo, t_hi, tri = launch
    if tri:
        blk = lens[q_lo:q_hi]
        return float((blk * np.cumsum(blk)).sum())
    return float(lens[q_lo:q_hi].sum() * lens[t_lo:t_hi].sum())


def launch_pairs(launch):
    q_lo, q_hi, t_lo, t_hi, tri = launch
    n = q_hi - q_lo
    return n * (n + 1) // 2 if tri else n * (t_hi - t_lo)


def cell_shares(lens, world, scheme="fold"):
    p = plan(lens, world, scheme)
    c = [sum(launch_cells(lens, la) for la in rank) for rank in p]
    tot = sum(c)
    return [x / tot for x in c]


def check_plan(n, p):
    """every pair i <= j exactly once: compares pair counts and checks that the launches are disjoint boxes"""
    boxes = [la for rank in p for la in rank]
    pairs = sum(launch_pairs(la) for la in boxes)
    if pairs != n * (n + 1) // 2:
        return False
    # target ranges of the boxes that share a target range must agree on it and tile [0, j] along the queries
    by_t = {}
    for q_lo, q_hi, t_lo, t_hi, tri in boxes:
        by_t.setdefault((t_lo, t_hi) if not tri else (q_lo, q_hi), []).append((q_lo, q_hi, tri))
    ranges = sorted(by_t)
    if ranges and (ranges[0][0] != 0 or ranges[-1][1] != n):
        return False
    for (a, b), (c, d) in zip(ranges, ranges[1:]):
        if b != c:
            return False
    for (t_lo, t_hi), parts in by_t.items():
        parts.sort()
        cover = 0
        for q_lo, q_hi, tri in parts:
            if q_lo != cover:
                return False
            cover = q_hi
            if tri and (q_lo, q_hi) != (t_lo, t_hi):
                return False
        if cover != t_hi:
            return False
    return True
