#!/usr/bin/env python3
"""What N = 2 / 4 / 8 ranks of bench.py would each take, run one after the other on ONE GPU: kernel ms (HIP events of the
library) of every rank's launches under the shard schemes of tools/exp/shardplan.py.  usage: shard_times.py [scheme ...]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import reseek_amd  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import shardplan  # noqa: E402

schemes = sys.argv[1:] or ["targets", "fold"]
seqs = bench.synth_mu_chains(0x5EED5EEC, None)
lens = np.array([len(s) for s in seqs], np.float64)
ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
res = bench.predicted_scaling(ctx, seqs, schemes=schemes, reps=3)
print(json.dumps(res, indent=1))
