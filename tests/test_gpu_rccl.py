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
    assert r.stdout.count("identical to the reference") == 5, r.stdout
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
    assert res["search_hits_gathered"] > 0 and res["search_s"] > 0
    assert res["config"]["collective_world"] == 1 and res["config"]["windows"] == [[0, n]]


def test_rsk_gather_hits_over_rccl_world1_through_the_c_abi():
    """rsk_comm_* / rsk_gather_hits (reseek_amd/csrc/rsk_comm.hip): the hit-record gather of a multi-process C++ caller, no
    Python collective involved -- a world-1 RCCL communicator on the box's GPU gathers the records the gapless kernel appended
    (device to device), twice (the result buffer is reused), and an empty contribution."""
    import ctypes as C
    import numpy as np
    import torch
    import reseek_amd
    ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    uid = ctx.comm_unique_id()
    assert len(uid) == 128
    comm = ctx.comm_create(uid, 0, 1)
    assert reseek_amd.capi.lib().rsk_comm_world(comm) == 1 and reseek_amd.capi.lib().rsk_comm_rank(comm) == 0
    rng = np.random.default_rng(5)
    seqs = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in rng.integers(30, 300, 300)]
    for k in range(0, 100, 5):
        seqs[k + 1] = seqs[k][:len(seqs[k + 1])].copy() if len(seqs[k]) >= len(seqs[k + 1]) else seqs[k + 1]
    db = reseek_amd.Db.from_mu_seqs(ctx, seqs)
    cap = 1 << 14
    rec = torch.zeros((cap, 3), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    for thr in (60, 90):
        ctx.mu_gapless_hits_dev(db, db, True, thr, rec.data_ptr(), cap, cnt.data_ptr())
        torch.cuda.synchronize()
        n = int(cnt.item())
        assert n > 5
        ptr, total, counts = ctx.gather_hits(comm, rec.data_ptr(), n, 12, 1)
        torch.cuda.synchronize()
        assert (total, counts) == (n, [n])
        back = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
        hip = C.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(C.c_void_p(back.data_ptr()), C.c_void_p(ptr), C.c_size_t(n * 12), 3) == 0      # device to device
        assert torch.equal(back, rec[:n])
    ptr, total, counts = ctx.gather_hits(comm, 0, 0, 12, 1)
    assert (total, counts) == (0, [0])
    ctx.comm_destroy(comm)
    db.close()
    ctx.close()
