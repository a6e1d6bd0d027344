#!/usr/bin/env python3
"""bench.py -- reseek -search hot-path benchmark on MI355X (driver contract: one JSON line on rank 0).

Workload (BASELINE.json configs[1]): SCOP40-shaped all-vs-all, "swgaplessint kernel only":
11,211 synthetic Mu-letter chains (lengths = the empirical SCOP40 length list, letters iid from the
SCOP40 Mu frequency table, planted homologs), every pair i <= j scored with the gapless integer
kernel (SWFastGapless_Int, swgaplessint.cpp:7) -> uint16 score matrix in HBM.
One "step" = one full all-vs-all pass with the chain set already resident in HBM.

metric = aligned DP cells/s (sum LA*LB over scored pairs / wall), chain-pairs/s reported beside it.
N > 1 (one process per GPU, torchrun): strong scaling -- ONE SCOP40-shaped set, rank r scores the pairs whose
target lies in its range of the triangle (ranges balanced by DP cells), value = all cells / max-over-ranks time;
the only exchange is an RCCL all_gather of the per-rank hit buffers.  The whole `-search` call is sharded the same
way (`search`).  --weak gives every rank an independent set instead.
Beside the contract's fields the line carries `roofline_live` (the kernels the live -search path runs) and
`search_bca` (whole call from a .bca file, with the reference binary timed on this box's host cores).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SCOP40 Mu letter frequencies A..Z a..j (measured from test_data/scop40.mu.fa)
MU_FREQ = np.array([
    0.0220, 0.0025, 0.0102, 0.0084, 0.0472, 0.0239, 0.0174, 0.0196, 0.0436, 0.0761, 0.0082, 0.0448, 0.0266,
    0.0510, 0.0474, 0.0645, 0.0299, 0.1021, 0.0248, 0.0031, 0.0212, 0.0082, 0.0110, 0.0162, 0.0205, 0.0086,
    0.0385, 0.0368, 0.0039, 0.0297, 0.0100, 0.0120, 0.0206, 0.0263, 0.0104, 0.0530])
MU_CHARS = "ABCDEFGHIJLKMNOPQRSTUVWXYZabcdefghij"   # sic: letter 10 = L, 11 = K (alpha.cpp g_LetterToCharMu)

# Packed VALU peak: 256 CUs x 4 SIMDs x 64 lanes / 4 cycles x 2.4 GHz.  VOP3P (v_pk_*: packed int16 and packed f16 alike)
# issues at one wave64 instruction per 4 cycles per SIMD on gfx950 (tools/ubench_valu.hip measures 35-37 T lane-ops/s for
# v_pk_add_i16, v_pk_max_i16, v_pk_add_f16, v_pk_maximum3_f16; 32-bit VOP2 ops run at twice that).  A lane-op = one lane of
# one packed instruction = two DP cells' worth of one operation.
# Gapless kernel: per 4 cells (a ring dword, two target letters) two clamped adds and one three-operand maximum.
GAPLESS_LANEOPS_PER_CELL = 0.75
PEAK_VALU_LANEOPS = 256 * 4 * 16 * 2.4e9
PEAK_HBM_GBS = 8000.0


def scop40_lengths():
    with open(os.path.join(ROOT, "tests", "golden", "scop40_lengths.txt")) as f:
        return np.array([int(x) for x in f.read().split()], dtype=np.int64)


def synth_mu_chains(seed, nchains=None):
    """SCOP40-shaped synthetic Mu chains, sorted by length; ~10% are mutated copies (sub .3/ins .1/del .1,
    cf. test_para.cpp:150-174) of another chain so that a realistic fraction of pairs scores high."""
    rng = np.random.default_rng(seed)
    lens = scop40_lengths()
    if nchains is not None:
        lens = lens[rng.choice(len(lens), nchains, replace=nchains > len(lens))]
    p = MU_FREQ / MU_FREQ.sum()
    seqs = [rng.choice(36, int(L), p=p).astype(np.uint8) for L in lens]
    nhom = len(seqs) // 10
    for k in rng.choice(len(seqs), nhom, replace=False):
        src = seqs[int(rng.integers(0, len(seqs)))]
        r = rng.random(len(src))
        out = []
        for c, u in zip(src, r):
            if u < 0.1:
                continue
            if u < 0.2:
                out.append(int(rng.integers(0, 36)))
            out.append(int(rng.integers(0, 36)) if u < 0.5 else int(c))
        L = len(seqs[k])
        out = (out * (L // max(1, len(out)) + 1))[:L] if len(out) < L else out[:L]
        seqs[k] = np.array(out, np.uint8)
    order = np.argsort([len(s) for s in seqs], kind="stable")
    return [seqs[i] for i in order]


def usable_cpus():
    """CPUs this process may actually use: min(affinity mask, cgroup CPU quota).  (The GPU boxes expose 256 hardware
    threads but cap the container at a 16-CPU quota; threads beyond the quota only get throttled.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(seqs, seconds_target=15.0):
    """Reference CPU kernels timed on this box's host cores on a bounded sample of the same pairs.
    kind "reference": oracle/_ref/ref_harness (the unmodified reference objects, built from
    /root/reference by oracle/Makefile.ref) runs SWFastPinopGapless on a strided sample of the
    triangle; falls back to the C restatement ("port", 1 core) if that binary did not travel."""
    cores = usable_cpus()
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if os.path.exists(harness):
        with tempfile.TemporaryDirectory() as td:
            fa = os.path.join(td, "bench.mu.fa")
            with open(fa, "w") as f:
                for i, s in enumerate(seqs):
                    f.write(">c%d\n%s\n" % (i, "".join(MU_CHARS[c] for c in s)))
            # ~0.12 Gcells/s/thread scalar: size the sample for ~seconds_target
            mean_cells = float(np.mean([len(s) for s in seqs])) ** 2
            npairs = int(max(2000, seconds_target * 0.12e9 * cores / mean_cells))
            try:
                out = subprocess.run([harness, "benchmu", fa, str(npairs), str(cores)], capture_output=True, text=True,
                                     timeout=600, check=True).stdout.strip().splitlines()[-1]
                r = json.loads(out)
                return {"value": r["cells"] / r["gapless_secs"], "unit": "cells/s", "cores": cores, "kind": "reference",
                        "sample": "%d pairs strided over the all-vs-all triangle (%.3g cells), SWFastPinopGapless "
                                  "(swfastpinopgapless.cpp:6) via oracle/_ref/ref_harness, %d std::threads = the container's CPU quota (%d hardware threads visible); same binary's "
                                  "AVX2 parasail fwd filter: %.3g cells/s" % (r["pairs"], r["cells"], cores, os.cpu_count() or 1,
                                                                              r["cells"] / r["parasail_fwd_secs"])}
            except Exception as e:  # noqa: BLE001
                sys.stderr.write("cpu_baseline: reference harness failed (%s); using the C port\n" % e)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    rng = np.random.default_rng(1)
    n = len(seqs)
    npairs = 20000
    ia = rng.integers(0, n, npairs)
    ib = rng.integers(0, n, npairs)
    t0 = time.perf_counter()
    ol.mu_gapless_pairs(seqs, ia, ib)
    dt = time.perf_counter() - t0
    cells = float(sum(len(seqs[a]) * len(seqs[b]) for a, b in zip(ia, ib)))
    return {"value": cells / dt, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": "%d random pairs (%.3g cells), oracle/rsk_oracle.c rsko_mu_gapless, 1 thread" % (npairs, cells)}


class GpuClockPoller:
    """Shader clock and board power while a leg runs, from the amdgpu hwmon files (freq1_input, power1_input, power1_cap) and
    gpu_busy_percent of every card that has them, sampled every 10 ms on a thread: mean / min clock over the samples taken while the
    card was busy, mean power, the power cap.  A slow BOX (lower clock at the same cap, or a lower cap) shows here; a slow BUILD
    does not (the driver saw +21 % on the configs[4] share between rounds and the line could not tell which)."""

    def __init__(self, period=0.01):
        import glob
        import threading
        self.cards = []
        for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")):
            hw = os.path.dirname(f)
            dev = os.path.dirname(os.path.dirname(hw))
            if os.path.exists(os.path.join(dev, "gpu_busy_percent")):
                self.cards.append((dev, hw))
        self.period = period
        self.samples = [[] for _ in self.cards]
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True) if self.cards else None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().split()[0])
        except (OSError, ValueError, IndexError):
            return None

    def _run(self):
        while not self._stop.is_set():
            for k, (dev, hw) in enumerate(self.cards):
                self.samples[k].append((self._read(os.path.join(dev, "gpu_busy_percent")), self._read(os.path.join(hw, "freq1_input")),
                                        self._read(os.path.join(hw, "power1_input"))))
            self._stop.wait(self.period)

    def __enter__(self):
        if self._th:
            self._th.start()
        return self

    def __exit__(self, *exc):
        if self._th:
            self._stop.set()
            self._th.join()
        return False

    def summary(self):
        """the busiest card's numbers (a 1-GPU leg uses one card; its index under /sys/class/drm need not be the HIP ordinal)"""
        best = None
        for k, (dev, hw) in enumerate(self.cards):
            busy = [s for s in self.samples[k] if s[0] is not None and s[0] >= 50.0 and s[1]]
            if not busy:
                continue
            e = {"card": os.path.basename(os.path.dirname(dev)), "samples_busy": len(busy), "samples": len(self.samples[k]),
                 "sclk_busy_mean_ghz": round(sum(s[1] for s in busy) / len(busy) / 1e9, 3), "sclk_busy_min_ghz": round(min(s[1] for s in busy) / 1e9, 3),
                 "power_busy_mean_w": round(sum(s[2] for s in busy if s[2]) / max(1, sum(1 for s in busy if s[2])) / 1e6, 1),
                 "power_busy_max_w": round(max([s[2] for s in busy if s[2]] or [0]) / 1e6, 1),
                 "power_cap_w": (self._read(os.path.join(hw, "power1_cap")) or 0) / 1e6}
            if best is None or e["samples_busy"] > best["samples_busy"]:
                best = e
        return best


def box_info():
    """what tells one GPU box from another in a bench line: power cap, the top shader-clock state, visible cards"""
    p = GpuClockPoller()
    if not p.cards:
        return None
    dev, hw = p.cards[0]
    top = None
    try:
        with open(os.path.join(dev, "pp_dpm_sclk")) as f:
            top = max(int(x.split(":")[1].strip().lower().replace("mhz", "").replace("*", "").strip()) for x in f.read().splitlines() if ":" in x)
    except (OSError, ValueError):
        pass
    return {"cards": len(p.cards), "power_cap_w": (p._read(os.path.join(hw, "power1_cap")) or 0) / 1e6, "sclk_top_mhz": top}


def search_end_to_end(seqs):
    """SURVEY 8d metric (ii): chain-pairs/s of the whole `-search -sensitive` call (container load + upload + Mu filter +
    SW/traceback/LDDT + long-chain path + hit replay + TSV) on the same SCOP40-shaped set, with synthetic profile bytes
    and CA traces added (tools/bench_search.py); second of two runs.  Reported beside the kernel metric, not as `value`."""
    import torch
    import reseek_amd
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_search
    ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    with tempfile.TemporaryDirectory() as td:
        db, out = os.path.join(td, "syn.rskdb"), os.path.join(td, "hits.tsv")
        bench_search.write_rskdb(db, seqs, np.random.default_rng(5))
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            nhits, st = ctx.search_rskdb(db, out, "sensitive")
            dt = time.perf_counter() - t0
            best = {"mode": "-search -sensitive, all-vs-all, whole call", "seconds": dt, "chain_pairs": int(st[0]),
                    "chain_pairs_per_sec": st[0] / dt, "mu_filter_survivors": int(st[5]), "long_chain_pairs": int(st[4]),
                    "hits": int(nhits)}
    ctx.close()
    return best


def _load_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def _latest_profile(suffix):
    """newest profiles/rNN_<suffix> (the records are named per round)"""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return os.path.basename(c[-1]) if c else "r00_" + suffix


def _verified_pmc(name=None):
    """Counter records of tools/prof_live.sh, per kernel -- only those whose kernel SOURCE file still has the sha256 it had
    when the counters were collected (a kernel edit must not leave stale counters in the bench line); the others read
    {"stale": reason}."""
    import hashlib
    name = name or _latest_profile("live_pmc.json")
    rec = _load_json(name)
    if not rec:
        return {}
    shas = rec.get("kernel_source_sha256", {})
    out = {}
    for k, e in rec.items():
        if k == "kernel_source_sha256" or not isinstance(e, dict):
            continue
        src = e.get("kernel_source")
        try:
            with open(os.path.join(ROOT, src), "rb") as f:
                cur = hashlib.sha256(f.read()).hexdigest()
        except (OSError, TypeError):
            cur = None
        if src and cur and shas.get(src) == cur:
            out[k] = dict(e, source_verified="sha256 of %s unchanged since the counters were collected (profiles/%s)" % (src, name))
        else:
            out[k] = {"stale": "profiles/%s: %s changed since the counters were collected (or no hash on record): re-run tools/prof_live.sh" % (name, src)}
    return out


def _with_clock(e, pmc_entry):
    """the nominal peak assumes 2.4 GHz; the counters give the clock the kernel actually held (a kernel at the chip's power limit
    runs below it) and the share of its cycles in which a SIMD issued a VALU instruction"""
    ck = (pmc_entry or {}).get("sustained_clock_ghz")
    if ck and "frac" in e:
        e["sustained_clock_ghz"] = ck
        e["frac_at_sustained_clock"] = e["frac"] * 2.4 / ck
        e["valu_issue_frac_of_cycles"] = (pmc_entry or {}).get("valu_issue_frac")
    return e


def prefilter_entry(ctx, name, what, seqs, pmc, reps=2):
    """One `roofline_live` entry for k_prefilter (prefiltermu.cpp:382, twohitdiag.cpp:368-398): the scan of every chain of
    `seqs` against the neighbourhood index of the same set (idxt, the `-fast -db` configuration).  Two byte models:
      survey  SURVEY 8d's model of the REFERENCE's data flow: TL + 8 x seed items x (6 B posting + 4 B bag write + 4 B read
              back)... read here as 14 B per seed item + 2 B per diagonal cell (the judge's recomputation formula)
      ours    what this kernel has to move: 4 B per seed item (the posting; bit addresses stay in LDS), 2 B per diagonal
              cell (a query and a target letter), 12 B per result triple, TL per target.
    The seed walk is the HBM / latency side, the diagonal scans the VALU side (5.5 instructions per cell); both fractions
    are reported, `bound` names the larger share of the kernel time (RSK_PF_DEBUG=1 splits them: profiles/)."""
    import torch
    import reseek_amd
    n = len(seqs)
    q = reseek_amd.Db.from_mu_seqs(ctx, seqs)
    cap = int(min(n * n + 16, 200_000_000))
    dq, dt, ds = (torch.zeros(cap, dtype=torch.int32, device="cuda") for _ in range(3))
    dn = torch.zeros(1, dtype=torch.int32, device="cuda")
    ms = []
    for _ in range(reps + 1):                       # the first call builds the index
        ctx.mu_prefilter_dev(q, q, dq.data_ptr(), dt.data_ptr(), ds.data_ptr(), cap, dn.data_ptr(), neighbourhood=2)
        torch.cuda.synchronize()
        ms.append(ctx.last_kernel_ms())
    ms_ = float(np.median(ms[1:]))
    items, postings, twohit, cells = ctx.mu_prefilter_last_work()
    triples = int(dn.item())
    nres = float(sum(len(s) for s in seqs))
    q.close()
    ours = 4.0 * items + 2.0 * cells + 12.0 * triples + nres
    survey = 14.0 * items + 2.0 * cells + nres
    return _with_clock({"kernel": "k_prefilter", "workload": name, "what": what, "kernel_ms": ms_, "chains": n, "index_postings": int(postings), "seed_items": int(items),
            "seed_items_per_s": items / ms_ * 1e3, "twohit_diagonals": int(twohit), "diagonal_cells": int(cells), "result_triples": triples,
            # the diagonal scans are the larger share of the kernel time (RSK_PF_DEBUG=1 stops after the seed walk: 0.10 of 0.64 s on
            # the config-2 letters, 30 of 93 ms on SCOP40), so the entry's bound is the VALU one; the byte models follow under `hbm`
            "bound": "valu", "unit": "T lane-instr/s", "instructions_per_cell": 5.5, "achieved": 5.5 * cells / ms_ * 1e3 / 1e12,
            "peak": PEAK_VALU_LANEOPS / 1e12, "frac": 5.5 * cells / ms_ * 1e3 / PEAK_VALU_LANEOPS,
            "note": "diagonal scans (FindHSP): perm + extract + LDS gather + add + 2 max per cell = 5.5 wave64 instructions per cell "
                    "at one instruction per 4 cycles per SIMD; the seed walk (one posting read, one LDS read, one or two LDS atomics "
                    "per item) is latency / HBM bound and not in this fraction",
            "hbm": {"algorithmic_bytes": ours, "achieved_GBs": ours / ms_ * 1e3 / 1e9, "frac": ours / ms_ * 1e3 / 1e9 / PEAK_HBM_GBS,
                    "formula": "4 B x seed items (the posting) + 2 B x diagonal cells (letters; L2 / LDS resident in practice) + 12 B x "
                               "result triples + target letters",
                    "survey_model": {"bytes": survey, "achieved_GBs": survey / ms_ * 1e3 / 1e9, "frac": survey / ms_ * 1e3 / 1e9 / PEAK_HBM_GBS,
                                     "formula": "14 B x seed items + 2 B x diagonal cells + target letters (SURVEY 8d)"},
                    "posting_reads_GBs": 4.0 * items / ms_ * 1e3 / 1e9},
            "pmc": pmc.get("k_prefilter") if name.startswith("config2") else None}, pmc.get("k_prefilter") if name.startswith("config2") else None)


def predicted_scaling(ctx, seqs, schemes=("window",), reps=2, worlds=(2, 4, 8)):
    """What each rank of an N-GPU run of this bench would take, measured by running every rank's launches one after the
    other on THIS GPU (kernel ms from the library's HIP events; no collective, no host overlap): per N the per-rank ms, the
    slowest rank, and the strong-scaling efficiency t(1) / (N x slowest rank) it predicts.  (VERDICT r04 #3a: equal cells are
    not equal time -- the thin range of the longest targets builds one LDS profile per <= 64 targets.)"""
    import torch
    import reseek_amd
    n = len(seqs)
    lens = np.array([len(s) for s in seqs], np.float64)
    HIT_MIN, HIT_CAP = 120, 1 << 22
    rec = torch.zeros((HIT_CAP, 3), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    dbs = {}

    def db_of(lo, hi):
        if (lo, hi) not in dbs:
            dbs[(lo, hi)] = reseek_amd.Db.from_mu_seqs(ctx, seqs[lo:hi])
        return dbs[(lo, hi)]

    def launch_ms(la):
        q_lo, q_hi, t_lo, t_hi, tri = la
        q, t = db_of(q_lo, q_hi), db_of(t_lo, t_hi)
        out = torch.zeros((q_hi - q_lo, t_hi - t_lo), dtype=torch.int16, device="cuda")
        v = []
        for _ in range(reps + 1):
            ctx.mu_gapless_hits_dev(q, t, tri, HIT_MIN, rec.data_ptr(), HIT_CAP, cnt.data_ptr(), d_scores_ptr=out.data_ptr(), ldo=t_hi - t_lo,
                                    q_base=q_lo, t_base=t_lo)
            torch.cuda.synchronize()
            v.append(ctx.last_kernel_ms())
        del out
        return float(np.median(v[1:]))

    t1 = launch_ms((0, n, 0, n, True))
    res = {"one_gpu_kernel_ms": t1, "method": "every rank's launches run one after the other on one GPU, kernel ms from HIP events; "
           "efficiency = t(1) / (N x slowest rank)"}
    if "window" in schemes:
        # the scheme bench.py --gpus N runs: every rank keeps the whole set and takes one window of target positions
        full = db_of(0, n)
        out = torch.zeros((n, n), dtype=torch.int16, device="cuda")
        r = {}
        for N in worlds:
            ms, cells = [], []
            for k in range(N):
                lo, hi = ctx.mu_gapless_shard_window(full, k, N)
                v = []
                for _ in range(reps + 1):
                    ctx.mu_gapless_hits_window_dev(full, lo, hi, HIT_MIN, rec.data_ptr(), HIT_CAP, cnt.data_ptr(), d_scores_ptr=out.data_ptr(), ldo=n)
                    torch.cuda.synchronize()
                    v.append(ctx.last_kernel_ms())
                ms.append(float(np.median(v[1:])))
                cells.append(ctx.mu_gapless_last_work()[1])
            r["n%d" % N] = {"rank_ms": [round(x, 3) for x in ms], "max_rank_ms": round(max(ms), 3), "max_over_mean_ms": round(max(ms) / (sum(ms) / N), 4),
                            "cells_max_over_mean": round(max(cells) * N / sum(cells), 4), "efficiency": round(t1 / (N * max(ms)), 4),
                            "launches_per_rank": [1] * N, "launch_Tcells_per_s": [[round(c / (m * 1e9), 2)] for c, m in zip(cells, ms)]}
        res["window"] = r
        del out
    for scheme in schemes:
        if scheme == "window":
            continue
        sys.path.insert(0, os.path.join(ROOT, "tools", "exp"))
        import shardplan                        # the r04 rectangle + triangle cuts, kept for tools/exp/shard_times.py
        r = {}
        for N in worlds:
            p = shardplan.plan(lens, N, scheme)
            per = [[launch_ms(la) for la in rank] for rank in p]
            ms = [sum(x) for x in per]
            cells = shardplan.cell_shares(lens, N, scheme)
            r["n%d" % N] = {"rank_ms": [round(x, 3) for x in ms], "max_rank_ms": round(max(ms), 3), "max_over_mean_ms": round(max(ms) / (sum(ms) / N), 4),
                            "cells_max_over_mean": round(max(cells) * N, 4), "efficiency": round(t1 / (N * max(ms)), 4),
                            "launches_per_rank": [len(rank) for rank in p],
                            "launch_Tcells_per_s": [[round(shardplan.launch_cells(lens, la) / (m * 1e9), 2) for la, m in zip(rank, x)] for rank, x in zip(p, per)]}
        res[scheme] = r
        for d in list(dbs.values()):
            d.close()
        dbs.clear()
    return res


def predicted_search_scaling(seqs, worlds=(2, 4, 8), reps=2, bca_worlds=(8,)):
    """`predicted_scaling.search`: what each rank of an N-GPU run of the LIVE self search would take -- the whole
    `rsk_search -sensitive` call with shard_index k / shard_count N (RunSelfShard, host/dbsearcher.cpp; the reference deals the pairs
    to threads through a locked counter, runself.cpp:72-99), every shard run one after the other on THIS GPU from the prepared
    container (.rskdb: profiles and self-rev scores inside) and, for `bca_worlds`, from the .bca file: a rank featurises ONE slice of
    the chains, the prepared containers are exchanged (reseek_amd.dist.featurise_sharded), then it searches its shard.  Last of `reps` runs per shard; efficiency =
    t(1) / (N x slowest shard); `hits_sum` must equal the unsharded call's hits."""
    import torch
    import reseek_amd
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_search
    ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    res = {"method": "rsk_search shard_index k / shard_count N, shards one after the other on one GPU, wall seconds of the whole call "
                     "(last of %d runs); efficiency = t(1) / (N x slowest shard)" % reps}
    try:
        with tempfile.TemporaryDirectory() as td:
            db, out = os.path.join(td, "syn.rskdb"), os.path.join(td, "hits.tsv")
            bench_search.write_rskdb(db, seqs, np.random.default_rng(5))
            bca = os.path.join(td, "syn.bca")
            bench_search.write_bca_records(bca, bench_search.gen_bca_chains(scop40_lengths(), np.random.default_rng(7)))

            def shard_s(src, k, N, r):
                dt, nh, st = 0.0, 0, None
                for _ in range(r):
                    t0 = time.perf_counter()
                    nh, st = ctx.search(src, out, "sensitive", shard_index=k, shard_count=N)
                    dt = time.perf_counter() - t0
                return dt, int(nh), int(st[0])

            from reseek_amd import dist as rdist
            for tag, src, ws, r in (("rskdb", db, worlds, reps), ("bca", bca, bca_worlds, reps)):
                t1, h1, p1 = shard_s(src, 0, 1, reps)
                e = {"one_gpu_seconds": round(t1, 4), "hits": h1, "chain_pairs": p1}
                for N in ws:
                    feat, exch_bytes, t_merge, shard_src = [0.0] * N, 0, 0.0, src
                    if tag == "bca":
                        # what reseek_amd.dist.search_sharded does with a .bca file: rank k featurises slice k of the chains
                        # (rsk_bca_to_rskdb), the containers are all-gathered (not timed here: one GPU; `exchange_bytes` is what every
                        # rank receives) and merged, then every rank searches its shard of the prepared set
                        parts = []
                        for k in range(N):
                            part = os.path.join(td, "part%d.rskdb" % k)
                            t0 = time.perf_counter()
                            ctx.bca_to_rskdb(bca, part, "sensitive", shard_index=k, shard_count=N)
                            feat[k] = time.perf_counter() - t0
                            parts.append(np.fromfile(part, dtype=np.uint8))
                            os.remove(part)
                        t0 = time.perf_counter()
                        shard_src = os.path.join(td, "all%d.rskdb" % N)
                        with open(shard_src, "wb") as f:
                            f.write(rdist.merge_rskdb(parts))
                        t_merge = time.perf_counter() - t0
                        exch_bytes = int(sum(len(x) for x in parts))
                    v = [shard_s(shard_src, k, N, r) for k in range(N)]
                    secs = [x[0] + feat[k] + t_merge for k, x in enumerate(v)]
                    e["n%d" % N] = {"shard_seconds": [round(x, 4) for x in secs], "max_over_mean": round(max(secs) * N / sum(secs), 4),
                                    "efficiency": round(t1 / (N * max(secs)), 4), "hits_sum": sum(x[1] for x in v),
                                    "pairs_sum": sum(x[2] for x in v), "pairs_max_over_mean": round(max(x[2] for x in v) * N / max(1, sum(x[2] for x in v)), 4)}
                    if tag == "bca":
                        e["n%d" % N].update({"featurise_slice_seconds": [round(x, 4) for x in feat], "merge_seconds": round(t_merge, 4),
                                             "exchange_bytes": exch_bytes, "search_seconds": [round(x[0], 4) for x in v]})
                        os.remove(shard_src)
                    assert e["n%d" % N]["hits_sum"] == h1 and e["n%d" % N]["pairs_sum"] == p1, \
                        "%s: the %d shards' hits / pairs do not add up to the unsharded call's: %s" % (tag, N, e)
                res[tag] = e
                if tag == "rskdb":
                    for N in ws:
                        res["n%d" % N] = e["n%d" % N]
    finally:
        ctx.close()
    return res


def config2_mu_letters():
    """Mu letters of the seeded 11,211-chain synthetic .bca (BASELINE configs[2]'s input; host featurisation, no GPU)"""
    import reseek_amd
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench_search
    import fixtures as fx
    with tempfile.TemporaryDirectory() as td:
        bca, fa = os.path.join(td, "s.bca"), os.path.join(td, "s.mu.fa")
        bench_search.write_bca(bca, scop40_lengths(), np.random.default_rng(7))
        reseek_amd.capi.bca_to_mu_fasta(bca, fa)
        return fx.read_mu_fasta(fa)[1]


def live_kernels(ctx, seqs, db, reps=3):
    """`roofline_live`: the kernels reseek -search actually runs (the gapless kernel of `value` has no caller in the
    reference), timed with the library's HIP events on the launch stream, on the same SCOP40-shaped set:
      k_mu_sw     Mu SW filter forward pass over the whole triangle (parasail_mu.cpp:120), packed half floats holding integers, two queries per register (k_mu_sw2: 7.5 ops per cell pair)
      k_sw_float  float SW + trace of the -sensitive filter survivors (sw.cpp:79; per-pair kernel)
      k_sw_qp     float SW + trace, 64 queries x all chains (query-profile kernel; the -db / -verysensitive regime)
    Bounds: VALU issue (one wave64 instruction per 4 cycles per SIMD for mixed VOP2/VOP3/DPP streams, profiles/r02_ubench_valu.txt)
    and, for k_sw_float, the LDS (8 random ds_read_b32 per cell).  `pmc` = issue / LDS-busy fractions from the rocprofv3
    counter passes of `bench.py --live-only` committed as profiles/rNN_live_pmc.json (tools/prof_live.sh; the newest round's record is read)."""
    import torch
    import reseek_amd
    n = len(seqs)
    lens = np.array([len(s) for s in seqs], np.float64)
    tri_cells = float((lens * np.cumsum(lens[::-1])[::-1]).sum())
    out8 = torch.zeros((n, n), dtype=torch.uint8, device="cuda")
    pmc = _verified_pmc()
    res = []

    def med(f):
        v = []
        for _ in range(reps):
            f()
            torch.cuda.synchronize()
            v.append(ctx.last_kernel_ms())
        return float(np.median(v))

    # --- k_mu_sw
    ctx.mu_sw_matrix_dev(db, db, True, False, out8.data_ptr(), n)
    ms = med(lambda: ctx.mu_sw_matrix_dev(db, db, True, False, out8.data_ptr(), n))
    alg = float((lens * (n - np.arange(n))).sum() + np.cumsum(lens[::-1])[::-1].sum() + n * (n + 1) / 2)      # LA + LB + 1 per pair
    res.append({"kernel": "k_mu_sw", "what": "Mu SW filter, forward pass, all pairs i<=j", "kernel_ms": ms, "cells": tri_cells,
                "cells_per_s": tri_cells / ms * 1e3, "bound": "valu", "valu_ops_per_cell": 3.75, "unit": "T lane-ops/s",
                "achieved": tri_cells * 3.75 / ms * 1e3 / 1e12, "peak": PEAK_VALU_LANEOPS / 1e12,
                "frac": tri_cells * 3.75 / ms * 1e3 / PEAK_VALU_LANEOPS,
                "hbm": {"algorithmic_bytes": alg, "achieved_GBs": alg / ms * 1e3 / 1e9, "frac": alg / ms * 1e3 / 1e9 / PEAK_HBM_GBS},
                "pmc": pmc.get("k_mu_sw"),
                "pmc_dispatch": "the counters are those of the LONGEST of the pass's dispatches (k_mu_sw2 over the query pairs of its "
                                "largest class), not of the whole pass"})
    _with_clock(res[-1], pmc.get("k_mu_sw"))
    # --- survivors of the -sensitive filter -> float SW (per-pair kernel)
    cap = 4_000_000
    pq = torch.zeros(cap, dtype=torch.int32, device="cuda")
    pt = torch.zeros(cap, dtype=torch.int32, device="cuda")
    nn = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.mu_filter_dev(db, db, True, 12.0, 20.0, out8.data_ptr(), n, pq.data_ptr(), pt.data_ptr(), 0, 0, cap, nn.data_ptr())
    torch.cuda.synchronize()
    ns = int(nn.item())
    ia = pq[:ns].cpu().numpy().astype(np.uint32)
    ib = pt[:ns].cpu().numpy().astype(np.uint32)
    o = np.lexsort((ib, ia))
    ia, ib = ia[o], ib[o]
    rng = np.random.default_rng(3)
    li = np.array([len(s) for s in seqs], np.uint32)
    tot = int(li.sum())
    prof = np.concatenate([np.concatenate([rng.integers(0, 20, (1, int(L))), rng.integers(0, 16, (7, int(L)))]).astype(np.uint8).reshape(-1)
                           for L in li])
    xyz = tuple(np.cumsum(rng.normal(0, 2.2, tot)).astype(np.float32) for _ in range(3))
    dbs = reseek_amd.Db(ctx, li, mu=np.concatenate(seqs), prof=prof, xyz=xyz, selfrev=np.zeros(n, np.float32))

    def sw_entry(name, what, a, b, minfwd, valu_per_cell, lds):
        ctx.align_pairs(dbs, dbs, a, b, min_fwd_score=minfwd, collect=False)      # warm the allocator pool
        v = []
        for _ in range(reps):
            ctx.align_pairs(dbs, dbs, a, b, min_fwd_score=minfwd, collect=False)
            v.append(ctx.last_kernel_ms())
        ms_ = float(np.median(v))
        p_, cells, tb = ctx.align_last_work()
        la, lb = li[a].astype(np.float64), li[b].astype(np.float64)
        alg_ = float((8 * (la + lb)).sum() + tb + (la + lb).sum())                # 8(LA+LB) + trace + path
        e = {"kernel": name, "what": what, "kernel_ms": ms_, "pairs": int(p_), "cells": float(cells), "cells_per_s": cells / ms_ * 1e3,
             "bound": "valu", "valu_ops_per_cell": valu_per_cell, "unit": "T lane-ops/s",
             "achieved": cells * valu_per_cell / ms_ * 1e3 / 1e12, "peak": PEAK_VALU_LANEOPS / 1e12,
             "frac": cells * valu_per_cell / ms_ * 1e3 / PEAK_VALU_LANEOPS,
             "hbm": {"algorithmic_bytes": alg_, "trace_bytes": float(tb), "achieved_GBs": alg_ / ms_ * 1e3 / 1e9,
                     "frac": alg_ / ms_ * 1e3 / 1e9 / PEAK_HBM_GBS},
             "pmc": pmc.get(name)}
        _with_clock(e, pmc.get(name))
        if lds:
            # 8 ds_read_b32 gathers per cell; conflict-free LDS rate = 32 lanes/clk/CU
            peak_g = 256 * 32 * 2.4e9
            e["lds"] = {"gathers_per_cell": 8, "achieved_Tgathers_per_s": 8 * cells / ms_ * 1e3 / 1e12, "peak_conflict_free": peak_g / 1e12,
                        "frac_conflict_free": 8 * cells / ms_ * 1e3 / peak_g,
                        "note": "random 4-byte gathers: ~3 distinct addresses on the busiest bank of a 32-lane group, i.e. the "
                                "attainable rate is about a third of the conflict-free one"}
        return e

    res.append(sw_entry("k_sw_float", "float SW + trace of the %d -sensitive Mu-filter survivors (per-pair kernel)" % ns, ia, ib, 7.0, 41.0, True))
    nq = 64
    order = np.random.default_rng(4).permutation(n)[:nq].astype(np.uint32)
    qa = np.repeat(order, n)
    qb = np.tile(np.arange(n, dtype=np.uint32), nq)
    res.append(sw_entry("k_sw_qp", "float SW + trace, %d queries x %d chains (query-profile kernel)" % (nq, n), qa, qb, 0.0, 21.2, False))      # 254 VALU instructions per 12-row column in the ISA of the hot loop (R = 12 instance: 18.5 per cell for the recurrence, trace masks and best cell + ~32 per step; r04b: 278)
    # --- k_traceback and k_lddt (+ k_lddt_long): the stages behind the Smith-Waterman kernels, on STRUCTURES (the set above has
    # iid profile letters: its alignments are a few columns long).  3,000 synthetic chains of tools/bench_search.py's generator
    # (persistent random walks, SCOP40 lengths; the look-alike structures of the config legs), featurised by the library's
    # DSS (host), 24 queries against all of them, every pair -- the -verysensitive regime of BASELINE configs[4].
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_search
        from reseek_amd import capi
        rng2 = np.random.default_rng(21)
        slen = scop40_lengths()
        slen = slen[rng2.choice(len(slen), 3000)]
        recs = bench_search.gen_bca_chains(slen, rng2)
        mus, profs, xs, ys, zs = [], [], [], [], []
        for seq, ic, L in recs:
            c = (np.frombuffer(ic, np.uint16).reshape(L, 3).astype(np.float32) / 10.0 - 1000.0)      # pdbchain.h:90
            pr, mu = capi.dss_featurize(seq, c[:, 0], c[:, 1], c[:, 2])
            mus.append(mu); profs.append(pr.reshape(-1)); xs.append(c[:, 0]); ys.append(c[:, 1]); zs.append(c[:, 2])
        sdb = reseek_amd.Db(ctx, slen.astype(np.uint32), mu=np.concatenate(mus), prof=np.concatenate(profs),
                            xyz=(np.concatenate(xs), np.concatenate(ys), np.concatenate(zs)), selfrev=np.zeros(len(slen), np.float32))
        nq2, nt2 = 24, len(slen)
        qa2, qb2 = np.repeat(np.arange(nq2, dtype=np.uint32), nt2), np.tile(np.arange(nt2, dtype=np.uint32), nq2)
        ctx.align_pairs(sdb, sdb, qa2, qb2, min_fwd_score=0.0, collect=False)
        tb_ms, st_ms = [], []
        for _ in range(reps):
            alns = ctx.align_pairs(sdb, sdb, qa2, qb2, min_fwd_score=0.0)
            t_sw, t_tb, t_st = ctx.align_last_times()
            tb_ms.append(t_tb); st_ms.append(t_st)
        sdb.close()
        tb_ms, st_ms = float(np.median(tb_ms)), float(np.median(st_ms))
        plen = np.array([a.path_len for a, _ in alns], np.float64)
        ncol = np.array([p.count("M") for _, p in alns], np.float64)
        steps, longest = float(plen.sum()), float(plen.max())
        tests = float((ncol * (ncol - 1) / 2).sum())
        res.append({"kernel": "k_traceback", "what": "TraceBackBitSW of %d pairs (%d queries x %d synthetic structures, every pair): one thread per pair, one dependent 16-byte load per step" % (len(alns), nq2, nt2),
                    "kernel_ms": tb_ms, "pairs": len(alns), "steps": steps, "longest_walk_steps": longest, "steps_per_s": steps / tb_ms * 1e3,
                    "bound": "latency", "ns_per_step_of_the_longest_walk": tb_ms * 1e6 / max(longest, 1.0),
                    "floor_ns_per_step": {"l2_hit": 212.0 / 2.4, "hbm_miss": 900.0 / 2.4},
                    "frac": (longest * 212.0 / 2.4 * 1e-6) / tb_ms,
                    "note": "every walker is resident at once (one wave per 64 pairs), so the kernel lasts as long as its longest walk; frac = "
                            "(longest walk x L2-hit latency of a dependent load, ~212 cycles: MI355X_MICROARCH.md) / kernel time -- a trace block "
                            "was just written by k_sw_qp (GBs per call) and is read from HBM (~900 cycles per miss), four steps of a diagonal share a line"})
        res.append({"kernel": "k_lddt", "what": "GetLDDT_mu_fast of the same pairs (k_lddt: a wave per pair; alignments of > 256 columns: k_lddt_long, a workgroup per pair)",
                    "kernel_ms": st_ms, "pairs": len(alns), "aligned_columns_mean": float(ncol.mean()), "column_pair_tests": tests,
                    "bound": "valu", "valu_ops_per_test": 15.0, "unit": "T lane-ops/s", "achieved": tests * 15.0 / st_ms * 1e3 / 1e12,
                    "peak": PEAK_VALU_LANEOPS / 1e12, "frac": tests * 15.0 / st_ms * 1e3 / PEAK_VALU_LANEOPS,
                    "note": "work = the unordered column pairs, 15 VALU instructions per 64 tests (8 packed ops for the two squared distances, "
                            "2 comparisons, partner address); beside them per pair: path expansion, the pairs within R0 (two correctly rounded "
                            "square roots + thresholds, ~10 % of the tests), per-column fractions and the reference's sequential column sum"})
    except Exception as e:  # noqa: BLE001
        sys.stderr.write("bench: traceback / lddt live entries failed: %s\n" % e)
    dbs.close()
    del out8, pq, pt
    torch.cuda.empty_cache()
    # --- k_prefilter: the `-fast -db` prefilter scan on the two letter statistics it meets
    try:
        res.append(prefilter_entry(ctx, "config2: Mu letters of the 11,211-chain synthetic .bca (low complexity: ~430 seed items per pair)",
                                   "k-mer prefilter scan, all chains vs the idxt neighbourhood index of the same set", config2_mu_letters(), pmc))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import fixtures as fx
        res.append(prefilter_entry(ctx, "real SCOP40 Mu letters (tests/golden/scop40.mu.fa.gz: ~21 seed items per pair)",
                                   "k-mer prefilter scan, all chains vs the idxt neighbourhood index of the same set",
                                   fx.read_mu_fasta("scop40.mu.fa.gz")[1], pmc))
    except Exception as e:  # noqa: BLE001
        sys.stderr.write("bench: prefilter live entry failed: %s\n" % e)
    return res


def search_vs_reference(nsample=1500, reps=3):
    """The whole `-search -sensitive` call from a .bca file next to the REFERENCE BINARY on this box's host cores.
    Ours: the full SCOP40-shaped synthetic .bca (11,211 chains: DSS featurisation + self-rev + filter + SW + long-chain
    path + hit table), second of two runs.  Reference: oracle/_ref/reseek -search on an every-k-th-chain sample of the
    same file (all usable cores, median of `reps`), reported as chain-pairs/s; rule for the full set: seconds =
    pairs_full / that rate (the sample is unbiased in length, the cost of the search is per pair).  The sorted hit table of
    the sample is compared with ours on the same sample."""
    import torch
    import reseek_amd
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_search
    ref = os.path.join(ROOT, "oracle", "_ref", "reseek")
    lens = scop40_lengths()
    recs = bench_search.gen_bca_chains(lens, np.random.default_rng(7))
    ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    out = {}
    try:
        with tempfile.TemporaryDirectory() as td:
            full, samp = os.path.join(td, "syn.bca"), os.path.join(td, "sample.bca")
            bench_search.write_bca_records(full, recs)
            idx = np.linspace(0, len(recs) - 1, nsample).astype(np.int64)
            bench_search.write_bca_records(samp, [recs[i] for i in idx], labels=["syn%05d" % i for i in idx])
            hits = os.path.join(td, "hits.tsv")
            for _ in range(2):
                t0 = time.perf_counter()
                nh, st = ctx.search_rskdb(full, hits, "sensitive")
                dt = time.perf_counter() - t0
            out = {"mode": "-search -sensitive all-vs-all from a synthetic .bca (featurisation + self-rev inside the call)",
                   "chains": len(recs), "seconds": dt, "chain_pairs": int(st[0]), "chain_pairs_per_sec": st[0] / dt,
                   "sw_pairs": int(st[5]), "long_chain_pairs": int(st[4]), "hits": int(nh)}
            ours_s = os.path.join(td, "ours_sample.tsv")
            t0 = time.perf_counter()
            nh_s, st_s = ctx.search_rskdb(samp, ours_s, "sensitive")
            t_ours_s = time.perf_counter() - t0
            if os.path.exists(ref):
                cores = usable_cpus()
                ref_tsv = os.path.join(td, "ref.tsv")
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    subprocess.run([ref, "-search", samp, "-sensitive", "-output", ref_tsv, "-threads", str(cores)], check=True,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=td, timeout=900)
                    ts.append(time.perf_counter() - t0)
                tm = float(np.median(ts))
                a = sorted(open(ref_tsv).read().splitlines())
                b = sorted(open(ours_s).read().splitlines())
                pairs_s = float(st_s[0])
                sa, sb = set(a), set(b)
                diff = sorted((sa - sb) | (sb - sa))
                long_labels = {"syn%05d" % i for i in idx if recs[i][2] >= 600}      # m_MKFL of -sensitive
                diff_long = [r for r in diff if r.split("\t")[0] in long_labels or r.split("\t")[1] in long_labels]
                out["cpu_baseline"] = {
                    "value": pairs_s / tm, "unit": "chain-pairs/s", "cores": cores, "kind": "reference",
                    "sample": "oracle/_ref/reseek -search sample.bca -sensitive -threads %d: every %.2f-th chain of the same .bca "
                              "(%d chains, %d pairs); median of %d runs" % (cores, len(recs) / nsample, nsample, int(pairs_s), reps),
                    "seconds_runs": ts, "seconds_median": tm,
                    "rule": "full-set seconds = chain_pairs / value",
                    "extrapolated_full_set_seconds": out["chain_pairs"] / (pairs_s / tm),
                    "speedup_whole_call": (out["chain_pairs"] / (pairs_s / tm)) / out["seconds"],
                    "ours_on_the_same_sample_seconds": t_ours_s}
                out["hit_table_on_sample"] = {
                    "identical": a == b, "reference_rows": len(a), "our_rows": len(b), "rows_differing": len(diff),
                    "rows_differing_without_a_chain_of_600_or_more": len(diff) - len(diff_long),
                    "note": "with several threads the reference is not reproducible on pairs with a chain >= 600 residues (its "
                            "banded X-drop traceback reads uninitialised trace cells, DESIGN.md section 5); its 1-thread table equals ours"}
    finally:
        ctx.close()
    return out


def config_shares(which=("config2", "config3", "config4")):
    """Driver-visible numbers for BASELINE configs[2..4] beside the headline (each a whole `rsk_search` call from .bca files,
    second of two runs for the -db shares (config2: one run), with an equality bit of the sorted hit table against oracle/_ref/reseek on a
    sample of the same files):
      config2  `-search Q -db Q -fast`, Q = the 11,211-chain SCOP40-shaped synthetic .bca: Mu k-mer prefilter + two-hit
               diagonals on the GPU, then the candidates under the sensitive preset (search.cpp:76-111)
      config3  256 queries x 125,000-chain DB `-sensitive`  = one GPU's share of "256 x 1M, pairs sharded over 8 GPUs"
      config4  1000 queries x 87,500-chain DB `-verysensitive` = one GPU's share of "1k x 700k (PDB scale)"
    The databases are written by tools/bench_search.py write_bca_fast (seeded; SCOP40 length distribution)."""
    import torch
    import reseek_amd
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_search
    ref = os.path.join(ROOT, "oracle", "_ref", "reseek")
    lens = scop40_lengths()
    cores = usable_cpus()
    ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    out = {}

    def sample_check(td, q, db, mode, nq_s, ndb_s):
        """reference binary vs ours on the first nq_s queries x an every-k-th-chain sample of the DB -> dict"""
        if not os.path.exists(ref):
            return {"identical": None, "note": "oracle/_ref/reseek did not travel"}
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import make_tail_bca
        qs, dbs = os.path.join(td, "qs.bca"), os.path.join(td, "dbs.bca")
        rq, lq = make_tail_bca.read_bca(q)
        bench_search.write_bca_records(qs, rq[:nq_s], labels=lq[:nq_s])
        if db == q:
            dbs = qs
        else:
            rd, ld = make_tail_bca.read_bca(db) if os.path.getsize(db) < (200 << 20) else (None, None)
            if rd is None:
                return {"identical": None, "note": "DB too large to sample in Python"}
            idx = np.linspace(0, len(rd) - 1, ndb_s).astype(np.int64)
            bench_search.write_bca_records(dbs, [rd[i] for i in idx], labels=[ld[i] for i in idx])
        ours, theirs = os.path.join(td, "ours_s.tsv"), os.path.join(td, "ref_s.tsv")
        ctx.search(qs, ours, mode, db=dbs)
        t0 = time.perf_counter()
        subprocess.run([ref, "-search", qs, "-db", dbs, "-" + mode, "-output", theirs, "-threads", "1"], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, cwd=td, timeout=900)
        tr = time.perf_counter() - t0
        a, b = sorted(open(theirs).read().splitlines()), sorted(open(ours).read().splitlines())
        return {"identical": a == b, "rows": len(a), "sample": "%d queries x %d DB chains, reference -threads 1 (%.1f s)" % (nq_s, ndb_s if db != q else nq_s, tr)}

    def run(name, what, q, db, mode, reps):
        hits = q + ".hits.tsv"
        best = None
        for _ in range(reps):
            ctx.path_counters_reset()
            with GpuClockPoller() as poll:
                t0 = time.perf_counter()
                nh, st = ctx.search(q, hits, mode, db=db)
                dt = time.perf_counter() - t0
            pc = ctx.path_counters()
            best = {"what": what, "seconds": dt, "chain_pairs": int(st[0]) if not st[7] else None, "prefilter_candidates": int(st[0]) if st[7] else None,
                    "sw_pairs": int(st[5]), "long_chain_pairs": int(st[4]), "hits": int(nh), "tsv_bytes": os.path.getsize(hits),
                    # the clock the box held during the call (hwmon, while busy) and the one k_sw_qp itself held (s_memtime / s_memrealtime
                    # over its workgroups): a slow box is told from a slow build by these two
                    "clock": poll.summary(), "swqp_clock_ghz": pc.get("swqp_clock_ghz"),
                    "sw_pairs_scored_frac": pc.get("scored_frac"), "sw_pairs_rescored": pc.get("sw_pairs_rescored"),
                    "upload_copies": pc.get("upload_copies"), "upload_MB": round(pc.get("upload_bytes", 0) / 1e6, 1) if pc else None,
                    "db_batches": pc.get("db_batches"), "loader_seconds": pc.get("loader_seconds"), "featurise_seconds": pc.get("featurise_seconds"),
                    "upload_seconds": pc.get("upload_seconds")}
        os.remove(hits)
        return best

    try:
        with tempfile.TemporaryDirectory() as td:
            if "config2" in which:
                rng = np.random.default_rng(7)
                q = os.path.join(td, "syn11211.bca")
                bench_search.write_bca(q, lens, rng)                   # the set of search_bca / the full-size goldens
                e = run("config2", "-search Q -db Q -fast, Q = 11,211 synthetic chains (prefilter + two-hit diagonals on the GPU, candidates "
                        "under the sensitive preset)", q, q, "fast", 1)
                e["chain_pairs"] = 11211 * 11211
                e["chain_pairs_per_sec"] = e["chain_pairs"] / e["seconds"]
                e["vs_reference_on_sample"] = sample_check(td, q, q, "fast", 400, 400)
                out["config2_fast_db_11211x11211"] = e
            for key, nq, ndb, mode, tag in (("config3", 256, 125000, "sensitive", "config3_share_256x125000_sensitive"),
                                            ("config4", 1000, 87500, "verysensitive", "config4_share_1000x87500_verysensitive")):
                if key not in which:
                    continue
                rng = np.random.default_rng(11)
                q, db = os.path.join(td, key + "_q.bca"), os.path.join(td, key + "_db.bca")
                bench_search.write_bca_fast(q, lens[rng.choice(len(lens), nq)], rng, "q")
                t0 = time.perf_counter()
                bench_search.write_bca_fast(db, lens[rng.choice(len(lens), ndb)], rng, "d")
                tgen = time.perf_counter() - t0
                e = run(key, "-search Q -db DB -%s, %d queries x %d-chain .bca DB (one GPU's share; DSS featurisation + self-rev of the DB "
                        "inside the call)" % (mode, nq, ndb), q, db, mode, 2)
                e["chain_pairs_per_sec"] = e["chain_pairs"] / e["seconds"]
                e["db_generation_seconds"] = tgen
                e["vs_reference_on_sample"] = sample_check(td, q, db, mode, 32 if key == "config3" else 8, 1500 if key == "config3" else 600)
                out[tag] = e
                os.remove(db)
    finally:
        ctx.close()
    out["kernel_time_split"] = "rocprofv3 kernel traces of these three calls: profiles/r06_search_*_rocprofv3.txt (tools/prof_search.sh)"
    out["reference_cores_on_this_box"] = cores
    return out


def search_sharded_leg(ctx, seqs, rank, world, dist, coll_dev):
    """N > 1: the whole `-search -sensitive` all-vs-all call with the triangle cut into target ranges, one per rank
    (rsk_search shard_index / shard_count), hit tables gathered on rank 0 over the process group; max-over-ranks wall
    time of the second of two runs."""
    import torch
    from reseek_amd import dist as rdist
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_search
    with tempfile.TemporaryDirectory() as td:
        db, out = os.path.join(td, "syn.rskdb"), os.path.join(td, "hits.tsv")
        bench_search.write_rskdb(db, seqs, np.random.default_rng(5))
        for _ in range(2):
            dist.barrier()
            t0 = time.perf_counter()
            nhits, st = rdist.search_sharded(ctx, db, out, "sensitive", device=coll_dev)
            dist.barrier()
            dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        p = torch.tensor([float(st[0])], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(p, op=dist.ReduceOp.SUM)
        if rank != 0:
            return None
        return {"mode": "-search -sensitive, all-vs-all, whole call, triangle sharded by target range over %d ranks" % world,
                "seconds": float(t.item()), "chain_pairs": int(p.item()), "chain_pairs_per_sec": float(p.item()) / float(t.item()),
                "hits_gathered": int(nhits)}


REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline")
LINE_LIMIT = 4096


def _r(x, nd=4):
    return round(float(x), nd) if isinstance(x, (int, float)) and not isinstance(x, bool) else x


def compact_line(res):
    """The ONE line the driver parses: the contract's keys, `roofline` and `cpu_baseline` as small objects of
    scalars, and scalar summaries of the other legs -- strict JSON, one line, < 4096 bytes whatever the legs produced.  Everything
    long (`roofline_live`, `predicted_scaling`, `configs`, the notes) lives in the detail record (`emit_detail`)."""
    cfg = res.get("config", {})
    hr = cfg.get("hit_records") or {}
    rf = res.get("roofline", {})
    hbm = rf.get("hbm", {})
    out = {k: res.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                   "vs_baseline", "dtype", "data")}
    out["ms_per_step"] = _r(out["ms_per_step"], 4)
    out["chain_pairs_per_sec"] = _r(res.get("chain_pairs_per_sec"), 1)
    out["config"] = {"workload": str(cfg.get("workload", ""))[:200], "pairs_total": cfg.get("pairs_total"), "cells_total": cfg.get("cells_total"),
                     "collective_backend": cfg.get("collective_backend"), "collective_world": cfg.get("collective_world"),
                     "windows": cfg.get("windows") if len(cfg.get("windows") or []) <= 16 else None,
                     "hit_records": {"min_score": hr.get("min_score"), "rank0_per_step": hr.get("rank0_per_step"),
                                     "gathered_all_ranks": hr.get("gathered_all_ranks"), "gather_check": hr.get("gather_check")}}
    out["roofline"] = {"bound": rf.get("bound"), "kernel": rf.get("kernel"), "achieved": _r(rf.get("achieved")), "peak": _r(rf.get("peak")),
                       "unit": rf.get("unit"), "frac": _r(rf.get("frac")), "kernel_ms": _r(rf.get("kernel_ms")),
                       "lane_ops_per_cell": rf.get("lane_ops_per_cell"), "frac_vs_guide_vop2_rate": _r(rf.get("frac_vs_guide_vop2_rate")),
                       "traffic": rf.get("traffic"), "algorithmic_bytes": _r(hbm.get("algorithmic_bytes"), 0),
                       "hbm_achieved_GBs": _r(hbm.get("achieved"), 1), "hbm_peak_GBs": hbm.get("peak"), "hbm_frac": _r(hbm.get("frac"))}
    cb = res.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": _r(cb.get("value"), 1), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "sample": str(cb.get("sample", ""))[:160]}
    out["dtype_note"] = "integer scores n/2048 in packed f16 (exact on [0, 2048]); int32 rescoring at the ceiling"
    se, sb, cf, ps = res.get("search"), res.get("search_bca"), res.get("configs") or {}, res.get("predicted_scaling") or {}
    if se:
        out["search_s"] = _r(se.get("seconds"))
        out["search_chain_pairs_per_sec"] = _r(se.get("chain_pairs_per_sec"), 1)
        if "hits_gathered" in se:
            out["search_hits_gathered"] = se["hits_gathered"]
    if sb:
        out["search_bca_s"] = _r(sb.get("seconds"))
        cbb = sb.get("cpu_baseline") or {}
        out["speedup"] = _r(cbb.get("speedup_whole_call"), 1)
        out["speedup_ref_cores"] = cbb.get("cores")
        out["search_bca_identical_on_sample"] = (sb.get("hit_table_on_sample") or {}).get("identical")
    for key, tag in (("config2_s", "config2_fast_db_11211x11211"), ("config3_s", "config3_share_256x125000_sensitive"),
                     ("config4_s", "config4_share_1000x87500_verysensitive")):
        e = cf.get(tag)
        if e:
            out[key] = _r(e.get("seconds"))
            out[key[:-2] + "_identical_on_sample"] = (e.get("vs_reference_on_sample") or {}).get("identical")
            if e.get("clock"):
                out[key[:-2] + "_sclk_ghz"] = e["clock"].get("sclk_busy_mean_ghz")
            if e.get("swqp_clock_ghz"):
                out[key[:-2] + "_swqp_clock_ghz"] = _r(e["swqp_clock_ghz"], 3)
    w = ps.get("window") or {}
    for N in (2, 4, 8):
        if "n%d" % N in w:
            out["predicted_eff_n%d" % N] = w["n%d" % N].get("efficiency")
    sp = ps.get("search") or {}
    for N in (2, 4, 8):
        if "n%d" % N in sp:
            out["predicted_search_eff_n%d" % N] = sp["n%d" % N].get("efficiency")
    if "n8" in (sp.get("bca") or {}):
        out["predicted_search_bca_eff_n8"] = sp["bca"]["n8"].get("efficiency")
    if res.get("box"):
        out["box"] = res["box"]
    out["detail"] = res.get("detail_file")
    line = json.dumps(out, allow_nan=False, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:                   # never let an over-long string field take the line over the limit
        for k in ("dtype_note", "box"):
            out.pop(k, None)
        out["config"]["workload"] = out["config"]["workload"][:80]
        out["config"]["windows"] = None
        if "cpu_baseline" in out:
            out["cpu_baseline"].pop("sample", None)
        line = json.dumps(out, allow_nan=False, separators=(",", ":"))
    assert len(line) < LINE_LIMIT and "\n" not in line, "bench line too long (%d bytes)" % len(line)
    return line


def _nan_to_none(o):
    if isinstance(o, float) and (o != o or o in (float("inf"), float("-inf"))):
        return None
    if isinstance(o, dict):
        return {k: _nan_to_none(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_nan_to_none(v) for v in o]
    return o


def emit_detail(res, path):
    """Everything the compact line leaves out: the full record to `path` (under gpurun_out/: it travels back from the GPU box) and,
    section by section, to EARLIER stdout lines prefixed `bench-detail` (not JSON lines: nothing but the last line starts with `{`)."""
    res = _nan_to_none(res)
    try:
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        with open(path, "w") as f:
            json.dump(res, f, allow_nan=False)
        res["detail_file"] = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError as e:
        sys.stderr.write("bench: could not write %s: %s\n" % (path, e))
        res["detail_file"] = None
    for k in ("roofline", "cpu_baseline", "predicted_scaling", "roofline_live", "search", "search_bca", "configs"):
        if k in res:
            print("bench-detail %s: %s" % (k, json.dumps(res[k], allow_nan=False)))
    return res


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torchrun (the reference fans out by itself: `-threads`,
    runself.cpp:101-145): one process per GPU under torch.distributed.run on this node, rendezvous on 127.0.0.1, same arguments.
    Fails loudly when the box has fewer than N devices (RSK_BENCH_ONE_DEVICE=1: all ranks on cuda:0 over gloo, a plumbing check
    whose numbers mean nothing)."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (librsk has no CPU fallback)")
    ndev = torch.cuda.device_count()
    if n > ndev and os.environ.get("RSK_BENCH_ONE_DEVICE", "") != "1":
        raise SystemExit("bench.py --gpus %d: this box has %d GPU(s) (set RSK_BENCH_ONE_DEVICE=1 to run all ranks on cuda:0 over gloo; "
                         "such a run only checks the plumbing)" % (n, ndev))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cpus() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chains", type=int, default=0, help="0 = the full SCOP40-shaped set (11,211)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--weak", action="store_true", help="N > 1: an independent SCOP40-shaped set per rank instead of shards of one")
    ap.add_argument("--no-search", action="store_true", help="skip the end-to-end -search legs (rank 0, 1 GPU only)")
    ap.add_argument("--no-predict", action="store_true", help="skip the predicted-scaling leg (the N = 2 / 4 / 8 windows run one after the other on this GPU)")
    ap.add_argument("--no-live", action="store_true", help="skip the live-path kernel rooflines (rank 0, 1 GPU only)")
    ap.add_argument("--live-only", action="store_true", help="only the live-path kernels (the command tools/prof_live.sh profiles)")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE configs[2..4] legs (rank 0, 1 GPU only)")
    ap.add_argument("--configs-only", default="", help="only these legs, e.g. config3,config4 (prints their JSON and exits)")
    ap.add_argument("--search-scaling-only", action="store_true", help="only predicted_scaling.search (prints its JSON and exits)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"), help="where the full record goes (the last "
                    "stdout line is the compact < 4 KB contract line)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)                     # does not return

    import torch
    import reseek_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch `python bench.py --gpus N`, or torchrun with --nproc-per-node N "
                         "AND --gpus N)" % (args.gpus, world))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (librsk has no CPU fallback)")
    # RSK_BENCH_ONE_DEVICE=1 (debug only): all ranks share cuda:0 and talk over gloo, to exercise the N > 1
    # code path on a single-GPU box; the numbers of such a run mean nothing.
    one_device = os.environ.get("RSK_BENCH_ONE_DEVICE", "") == "1"
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    # RSK_DIST_FORCE=1: the collective path with a process group of ONE rank (RCCL on a single-GPU box; tests/test_gpu_rccl.py)
    if world > 1 or os.environ.get("RSK_DIST_FORCE", "") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    coll_dev = torch.device("cpu") if one_device else torch.device("cuda", local)

    # N > 1: ONE SCOP40-shaped set on every rank (strong scaling, BASELINE metric "SCOP40 all-vs-all, 1/2/4/8 GPUs").  Every rank
    # keeps the whole set (41 MB) and scores the pairs whose later member, in the kernel's processing order, stands in the
    # rank's WINDOW of positions (rsk_mu_gapless_shard_window: windows of equal modelled cost; rsk_mu_gapless_hits_window_dev:
    # one launch of the same shape as the whole triangle -- the same rings against fewer targets).  r01-r04 gave a rank a target
    # range of the length-sorted set as a rectangle + a small triangle of its own: measured r05 (predicted_scaling), the small
    # triangles ran at 4-37 T cells/s against 41-44 and the rank of the longest chains took 1.29 x its share.  --weak: an
    # independent set per rank (seed + rank), as in round 1.
    seqs = synth_mu_chains(0x5EED5EEC + (rank if args.weak else 0), args.chains or None)
    n = len(seqs)
    lens = np.array([len(s) for s in seqs], np.float64)
    windowed = world > 1 and not args.weak
    stream = torch.cuda.current_stream()
    ctx = reseek_amd.Ctx(local, stream=stream.cuda_stream)
    db = reseek_amd.Db.from_mu_seqs(ctx, seqs)                                  # inputs resident in HBM before the timed region
    win = ctx.mu_gapless_shard_window(db, rank, world) if windowed else (0, n)
    windows = [list(ctx.mu_gapless_shard_window(db, r, world)) for r in range(world)] if windowed else [[0, n]]
    if args.live_only:
        print(json.dumps({"roofline_live": live_kernels(ctx, seqs, db, reps=1)}))
        return
    if args.configs_only:
        print(json.dumps({"configs": config_shares(tuple(args.configs_only.split(",")))}))
        return
    if args.search_scaling_only:
        print(json.dumps({"predicted_scaling": {"search": predicted_search_scaling(seqs, bca_worlds=(2, 4, 8))}}))
        return
    out = torch.zeros((n, n), dtype=torch.int16, device="cuda")               # the dense matrix of the whole set; a rank writes its pairs' cells
    summary = torch.zeros(2, dtype=torch.int64, device="cuda")
    # what a search keeps of the pair space: the kernel itself appends {query, target, score} (indices of the whole set)
    # for the pairs scoring >= HIT_MIN; the dense uint16 matrix is written as well (it is the contract's output)
    HIT_MIN, HIT_CAP = 120, 1 << 22
    rec = torch.zeros((HIT_CAP, 3), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int32, device="cuda")

    def step():
        if windowed:
            ctx.mu_gapless_hits_window_dev(db, win[0], win[1], HIT_MIN, rec.data_ptr(), HIT_CAP, cnt.data_ptr(), d_scores_ptr=out.data_ptr(), ldo=n)
        else:
            ctx.mu_gapless_hits_dev(db, db, True, HIT_MIN, rec.data_ptr(), HIT_CAP, cnt.data_ptr(), d_scores_ptr=out.data_ptr(), ldo=n)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if dist is not None:
        # the path's only collective: gather of the per-rank hit records (query, target, score) the kernels appended --
        # device buffers, all_gather over RCCL / xGMI, nothing goes through the host
        from reseek_amd import dist as rdist
        torch.cuda.synchronize()
        c = cnt.cpu()
        rows = rec[:min(int(c[0]), HIT_CAP)]
        if one_device:
            gathered = rdist.gather_rows(rows.cpu().numpy(), dst=0, device=coll_dev, all_ranks=True)
        else:
            gathered = rdist.gather_records_device(rows)
        summary[0] = gathered.shape[0]
    barrier()
    dt = time.perf_counter() - t0
    # per-launch kernel time (the library's HIP events on the launch stream) and work of this rank's launches, untimed pass
    step()
    torch.cuda.synchronize()
    kernel_ms = ctx.last_kernel_ms()
    pairs, cells, slots = ctx.mu_gapless_last_work()
    hits_rank = int(cnt[0].item())

    # the same pass without the dense matrix (hit records only: what a search needs), untimed, for the record
    kernel_ms_hits_only = None
    if not windowed:
        ctx.mu_gapless_hits_dev(db, db, True, HIT_MIN, rec.data_ptr(), HIT_CAP, cnt.data_ptr())
        torch.cuda.synchronize()
        kernel_ms_hits_only = ctx.last_kernel_ms()

    tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
    tot = torch.tensor([float(cells), float(pairs)], dtype=torch.float64, device=coll_dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    total_cells, total_pairs = float(tot[0].item()), float(tot[1].item())
    # N > 1: the gathered hit records must be the records of the whole triangle -- every pair in exactly one rank's window
    # (VERDICT r04 #3c).  Rank 0 scores the whole triangle once more, untimed, and compares the counts; the per-rank counts
    # must add up to the gathered total as well.
    gather_check = None
    if dist is not None and windowed:
        hsum = torch.tensor([float(hits_rank)], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(hsum, op=dist.ReduceOp.SUM)
        if rank == 0:
            ctx.mu_gapless_hits_dev(db, db, True, HIT_MIN, rec.data_ptr(), HIT_CAP, cnt.data_ptr())
            torch.cuda.synchronize()
            one_gpu_hits = int(cnt[0].item())
            gather_check = {"gathered_all_ranks": int(summary[0].item()), "sum_of_rank_counts": int(hsum.item()), "one_gpu_hit_records": one_gpu_hits}
            assert gather_check["gathered_all_ranks"] == one_gpu_hits == gather_check["sum_of_rank_counts"], \
                "sharded hit records differ from the one-GPU triangle: %s" % gather_check
    search_n = None
    if dist is not None and not args.no_search and not args.weak:
        search_n = search_sharded_leg(ctx, seqs, rank, world, dist, coll_dev)

    if rank == 0:
        cells_per_s = total_cells * args.steps / dt
        k_cells_per_s = cells / (kernel_ms * 1e-3)
        nres = float(sum(len(s) for s in seqs))
        # algorithmic HBM bytes per launch: (LA + LB + 8) per pair, score-only
        npairs_all = n * (n + 1) / 2.0
        alg_bytes = float(lens.sum() * (n + 1) + 8.0 * npairs_all) * (pairs / npairs_all)      # sum over the pairs i <= j of LA + LB + 8; a window: its share of the pairs
        # HBM traffic per launch: rocprofv3 PMC counters of this same command (tools/prof_bench.sh -> tools/prof_traffic_json.py),
        # committed under profiles/ together with the sha256 of the kernel's source file: reported only while that file is
        # unchanged and the workload is the one that was profiled -- otherwise null, with the reason
        traffic, traffic_src = None, None
        try:
            import hashlib
            tname = _latest_profile("traffic.json")
            with open(os.path.join(ROOT, "profiles", tname)) as f:
                tj = json.load(f)
            with open(os.path.join(ROOT, tj["kernel_source"]), "rb") as f:
                sha = hashlib.sha256(f.read()).hexdigest()
            if sha != tj["kernel_source_sha256"]:
                traffic_src = "profiles/%s is STALE (%s changed since the counters were collected): re-run tools/prof_bench.sh" % (tname, tj["kernel_source"])
            elif n != 11211 or args.chains or world != 1:
                traffic_src = "profiles/%s holds the 1-GPU full-set workload only" % tname
            else:
                traffic = float(tj["traffic_bytes_per_launch"])
                traffic_src = "profiles/%s (rocprofv3 PMC FETCH_SIZE x 2 + WRITE_SIZE per dispatch, same command; kernel source sha256 verified)" % tname
        except (OSError, ValueError, KeyError) as e:
            traffic_src = "no traffic record (%s)" % e
        res = {
            "metric": "aligned cells/sec (SCOP40-shaped all-vs-all, gapless int Mu kernel)",
            "value": cells_per_s, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": "int16-exact", "data": "synthetic",
            "chain_pairs_per_sec": total_pairs * args.steps / dt,
            "config": {"workload": "BASELINE configs[1]: SCOP40-shaped (%d chains, %d residues%s) all-vs-all "
                                   "i<=j, swgaplessint kernel only" % (n, int(nres), " per GPU" if args.weak else ""),
                       "pairs_total": int(total_pairs), "cells_total": total_cells, "pairs_rank0": pairs, "cells_rank0": cells,
                       "hit_records": {"min_score": HIT_MIN, "rank0_per_step": hits_rank,
                                       "gathered_all_ranks": int(summary[0].item()) if dist is not None else None,
                                       "gather_check": gather_check,
                                       "note": "appended by the kernel (rsk_mu_gapless_hits_dev / _window_dev) next to the dense uint16 matrix"},
                       "windows": windows,
                       "collective_backend": (dist.get_backend() if dist is not None else None),
                       "collective_world": (dist.get_world_size() if dist is not None else None),
                       "sharding": "one independent set per GPU (--weak)" if args.weak else
                                   "one set on every rank; rank r scores the pairs whose later member (kernel processing order) stands in its "
                                   "window of positions, windows of equal modelled cost (rsk_mu_gapless_shard_window), one launch per "
                                   "rank and step (rsk_mu_gapless_hits_window_dev); no data-path collective, hit buffers gathered over RCCL"},
            "roofline": {
                "bound": "valu", "kernel": "k_gapless_ring<16,16> (+<8,16>)",
                "achieved": k_cells_per_s * GAPLESS_LANEOPS_PER_CELL / 1e12, "peak": PEAK_VALU_LANEOPS / 1e12, "unit": "T lane-ops/s",
                "frac": k_cells_per_s * GAPLESS_LANEOPS_PER_CELL / PEAK_VALU_LANEOPS,
                "lane_ops_per_cell": GAPLESS_LANEOPS_PER_CELL,
                "frac_vs_guide_vop2_rate": k_cells_per_s * GAPLESS_LANEOPS_PER_CELL / (2.0 * PEAK_VALU_LANEOPS),
                "peak_guide_vop2_rate": 2.0 * PEAK_VALU_LANEOPS / 1e12,
                "note": "work = 0.75 packed VALU lane-op per DP cell: scores as n/2048 in packed half floats (exact), per ring "
                        "dword and pair of target letters two v_pk_add_f16 clamp (add + floor at 0) and one v_pk_maximum3_f16 "
                        "(r02: packed int16, add-saturate + max per letter = 1 lane-op per cell, 31 T cells/s); pairs that reach "
                        "the clamp's ceiling (2048) are rescored in integers by the wave that found them; peak = VOP3P issue "
                        "rate 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz = 39.3 T lane-ops/s (ubench: 35-37).  The guide's plain-VOP2 rate "
                        "(157.3 TFLOP/s / 2 = 78.6 T lane-ops/s: one wave64 instruction per 2 cycles, which only runs of one identical "
                        "VOP2 opcode reach on this part) is twice that: against it the same work is `frac_vs_guide_vop2_rate`.  LDS "
                        "2 B/cell; kernel time from HIP events on the launch stream",
                "kernel_ms": kernel_ms, "cell_slots_issued": slots, "slot_efficiency": cells / max(1, slots),
                "kernel_ms_hit_records_only": kernel_ms_hits_only,
                "hbm": {"bound": "hbm", "achieved": alg_bytes / (kernel_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": alg_bytes / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                        "algorithmic_bytes": alg_bytes, "traffic": traffic},
                "traffic": traffic, "traffic_source": traffic_src},
        }
        res["box"] = box_info()
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(seqs)
        if world == 1 and not args.chains and not args.no_predict:
            try:
                res["predicted_scaling"] = predicted_scaling(ctx, seqs)
            except Exception as e:  # noqa: BLE001 -- the kernel metric above stands on its own
                sys.stderr.write("bench: predicted-scaling leg failed: %s\n" % e)
        if not args.no_live and world == 1 and not args.chains:
            try:
                del out
                torch.cuda.empty_cache()
                res["roofline_live"] = live_kernels(ctx, seqs, db)
            except Exception as e:  # noqa: BLE001 -- the kernel metric above stands on its own
                sys.stderr.write("bench: live-kernel leg failed: %s\n" % e)
        if search_n is not None:
            res["search"] = search_n
        if world == 1:
            # the legs below are whole searches on contexts of their own: give the kernel legs' device memory back first (the
            # pools of this context and of the parked helper contexts hold tens of GB of trace scratch)
            try:
                reseek_amd.capi.lib().rsk_ctx_trim(ctx.h)
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001
                sys.stderr.write("bench: trim failed: %s\n" % e)
        if not args.no_search and world == 1 and not args.chains:
            try:
                res["search"] = search_end_to_end(seqs)
                res["search_bca"] = search_vs_reference()
            except Exception as e:  # noqa: BLE001 -- the kernel metric above stands on its own
                sys.stderr.write("bench: end-to-end search leg failed: %s\n" % e)
            if not args.no_predict:
                try:
                    res.setdefault("predicted_scaling", {})["search"] = predicted_search_scaling(seqs)
                except Exception as e:  # noqa: BLE001
                    sys.stderr.write("bench: predicted search-scaling leg failed: %s\n" % e)
        if not args.no_configs and not args.no_search and world == 1 and not args.chains:
            try:
                res["configs"] = config_shares()
            except Exception as e:  # noqa: BLE001
                sys.stderr.write("bench: configs[2..4] leg failed: %s\n" % e)
        res = emit_detail(res, args.detail)
        final_line = compact_line(res)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(final_line, flush=True)              # the LAST line of stdout: the one the driver parses


if __name__ == "__main__":
    main()
