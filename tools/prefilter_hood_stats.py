import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, reseek_amd, fixtures as fx
labels, seqs = fx.read_mu_fasta("scop40.mu.fa.gz", limit=int(sys.argv[1]))
ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
q = reseek_amd.Db.from_mu_seqs(ctx, seqs)
cap = 80_000_000
dq = torch.zeros(cap, dtype=torch.int32, device="cuda"); dt = torch.zeros(cap, dtype=torch.int32, device="cuda"); ds = torch.zeros(cap, dtype=torch.int32, device="cuda"); dn = torch.zeros(1, dtype=torch.int32, device="cuda")
for mode in (0, 2):
    for rep in range(2):
        t0 = time.perf_counter()
        ctx.mu_prefilter_dev(q, q, dq.data_ptr(), dt.data_ptr(), ds.data_ptr(), cap, dn.data_ptr(), neighbourhood=mode)
        torch.cuda.synchronize()
        print("mode", mode, "rep", rep, "seconds %.3f" % (time.perf_counter() - t0), "kernel_ms %.1f" % ctx.last_kernel_ms(), "triples", int(dn.item()), flush=True)
