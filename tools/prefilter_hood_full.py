import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, reseek_amd, fixtures as fx, hashlib
labels, seqs = fx.read_mu_fasta("scop40.mu.fa.gz")
ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
q = reseek_amd.Db.from_mu_seqs(ctx, seqs)
cap = 80_000_000
dq = torch.zeros(cap, dtype=torch.int32, device="cuda"); dt = torch.zeros(cap, dtype=torch.int32, device="cuda"); ds = torch.zeros(cap, dtype=torch.int32, device="cuda"); dn = torch.zeros(1, dtype=torch.int32, device="cuda")
for rep in range(2):
    t0 = time.perf_counter()
    ctx.mu_prefilter_dev(q, q, dq.data_ptr(), dt.data_ptr(), ds.data_ptr(), cap, dn.data_ptr(), neighbourhood=-1)
    torch.cuda.synchronize()
    print("rep", rep, "seconds", time.perf_counter() - t0, "kernel_ms", ctx.last_kernel_ms(), "triples", int(dn.item()), flush=True)
n = int(dn.item())
qq = dq[:n].cpu().numpy().astype(np.uint32); tt = dt[:n].cpu().numpy().astype(np.uint32); ss = ds[:n].cpu().numpy().astype(np.uint32)
t0 = time.perf_counter()
with open("/tmp/hood_tmp.tsv", "w") as f: pass
rq, rt, rs = reseek_amd.capi.rsb_select(qq, tt, ss, len(seqs), 1500, tmp_tsv_path="/tmp/hood_tmp.tsv")
print("rsb seconds", time.perf_counter() - t0, "kept", len(rq))
lines = ["%s\t%s\t%d" % (labels[a], labels[b], c) for a, b, c in zip(rq.tolist(), rt.tolist(), rs.tolist())]
lines.sort()
print("lines", len(lines), "sorted_scores_md5", hashlib.md5(("\n".join(lines) + "\n").encode()).hexdigest(), "tmp_tsv_md5", hashlib.md5(open("/tmp/hood_tmp.tsv","rb").read()).hexdigest())
