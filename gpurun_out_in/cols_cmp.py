import os, subprocess, sys, torch
sys.path.insert(0, "/root/repo")
import reseek_amd
ROOT = "/root/repo"
q, db = ROOT + "/gpurun_out_in/q.bca", ROOT + "/gpurun_out_in/db.bca"
cols = "query+target+dpscore+lddt+newts+evalue+ql+tl"
ref = ROOT + "/oracle/_ref/reseek"
subprocess.run([ref, "-search", q, "-db", db, "-fast", "-columns", cols, "-output", "/tmp/ref_cols.tsv", "-threads", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.search(q, "/tmp/our_cols.tsv", "fast", db=db, columns=cols) if hasattr(ctx, "search") else None
for f in ("/tmp/ref_cols.tsv", "/tmp/our_cols.tsv"):
    print(f)
    for l in open(f):
        if "\tsyn00160\t" in l and (l.startswith("syn00000\t") or l.startswith("syn00135\t")):
            print("  ", l.strip())
