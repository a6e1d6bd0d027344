"""The N > 1 path end to end on ONE GPU: two processes (torch.distributed, gloo; RSK_BENCH_ONE_DEVICE=1 makes both use
cuda:0) run reseek_amd.dist.search_sharded on the q100 fixture -- self search, -db mode and the two-stage -fast -db
path with its all_gather of the prefilter triples (also with bags that overflow) -- and rank 0 compares the gathered hit tables with the reference's
goldens (tools/search_dist_demo.py); bench.py --gpus 2 runs its strong-scaling shards the same way."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(script, nproc, extra=()):
    env = dict(os.environ, RSK_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), script] + list(extra)
    return subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)


def test_search_sharded_world2_gloo_one_device():
    r = _torchrun(os.path.join(ROOT, "tools", "search_dist_demo.py"), 2)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("identical to the reference") == 5, r.stdout


def test_bench_strong_scaling_world2_gloo_one_device():
    r = _torchrun(os.path.join(ROOT, "bench.py"), 2, ["--gpus", "2", "--steps", "2", "--warmup", "1", "--chains", "1500"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["scaling"] == "strong"
    n = 1500
    assert res["config"]["pairs_total"] == n * (n + 1) // 2      # the windows partition the triangle
    assert res["value"] > 0
    w = res["config"]["windows"]
    assert res["config"]["collective_world"] == 2 and len(w) == 2 and w[0][0] == 0 and w[0][1] == w[1][0] and w[1][1] == n
    # bench.py itself compares the gathered records with a one-GPU pass of the whole triangle (and asserts)
    gc = res["config"]["hit_records"]["gather_check"]
    assert gc["gathered_all_ranks"] == gc["one_gpu_hit_records"] == gc["sum_of_rank_counts"] > 0
    # the hit records the kernels appended on both ranks (gathered) == the records of the unsharded run on the same set
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--chains", "1500", "--no-cpu-baseline",
                          "--no-search", "--no-live"], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    r1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["config"]["hit_records"]["gathered_all_ranks"] == r1["config"]["hit_records"]["rank0_per_step"] > 0


def test_bench_gpus_2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO torchrun in the command (`--gpus` was parsed and never read): bench.py
    re-executes itself under torch.distributed.run with two ranks (RSK_BENCH_ONE_DEVICE=1: both on cuda:0 over gloo) and the line
    says so; without that variable, and with fewer than N devices, it must refuse loudly."""
    import torch
    env = dict(os.environ, RSK_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--chains", "1500", "--steps", "2"], capture_output=True,
                       text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < 4096
    res = json.loads(last)
    assert res["n_gpus"] == 2 and res["config"]["collective_world"] == 2 and res["steps"] == 2
    gc = res["config"]["hit_records"]["gather_check"]
    assert gc["gathered_all_ranks"] == gc["one_gpu_hit_records"] == gc["sum_of_rank_counts"] > 0
    assert res["config"]["pairs_total"] == 1500 * 1501 // 2
    if torch.cuda.device_count() < 2:
        env.pop("RSK_BENCH_ONE_DEVICE")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--chains", "1500", "--steps", "2"], capture_output=True,
                           text=True, env=env, cwd=ROOT, timeout=300)
        assert r.returncode != 0 and "this box has 1 GPU" in r.stderr, r.stderr[-2000:]
