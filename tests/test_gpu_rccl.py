"""RCCL itself on the hardware a build box has: ONE process, ONE GPU, torch.distributed backend "nccl" (= RCCL on ROCm),
world size 1.  The collectives of the multi-GPU paths -- the device-to-device gather of the hit records the gapless kernel
appends (reseek_amd.dist.gather_records_device), the padded uint8 all_gather of the hit tables (search_sharded), the
all_gather of the prefilter triples on DEVICE tensors, bench.py's barrier / all_reduce bracket -- all run through the
library the driver's 8-GPU launch will load, so that launch is not the first time RCCL initialises.  (The other N > 1 tests
use gloo on one device: tests/test_gpu_dist.py.)"""
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


def _torchrun1(script, extra=()):
    env = dict(os.environ, RSK_DIST_FORCE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RSK_BENCH_ONE_DEVICE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), script] + list(extra)
    return subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)


def test_search_sharded_and_record_gather_over_rccl_world1():
    r = _torchrun1(os.path.join(ROOT, "tools", "search_dist_demo.py"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("identical to the reference") == 4, r.stdout
    assert "gather_records_device over nccl: 5 records" in r.stdout, r.stdout


def test_bench_under_torchrun_world1_rccl():
    r = _torchrun1(os.path.join(ROOT, "bench.py"), ["--gpus", "1", "--steps", "2", "--warmup", "1", "--chains", "1500", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 1 and res["config"]["collective_backend"] == "nccl"
    n = 1500
    assert res["config"]["pairs_total"] == n * (n + 1) // 2
    # the records gathered over RCCL are the records the kernel appended
    assert res["config"]["hit_records"]["gathered_all_ranks"] == res["config"]["hit_records"]["rank0_per_step"] > 0
    # the sharded whole-search leg ran through the same group
    assert res["search"]["hits_gathered"] > 0
    assert res["config"]["collective_world"] == 1 and res["config"]["windows"] == [[0, n]]
