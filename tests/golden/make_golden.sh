#!/bin/bash
# Regenerates every fixture in tests/golden/ from the reference itself, run in THIS container:
#   oracle/_ref/reseek       = unmodified reference binary   (oracle/Makefile.ref)
#   oracle/_ref/ref_harness  = our main() linked against the reference objects (oracle/ref_harness.cpp)
# Fixtures are data only (inputs + the reference's outputs).  /root/reference is needed to run
# this script but never at test time.
set -euo pipefail
cd "$(dirname "$0")/../.."
make -f oracle/Makefile.ref -j8 >/dev/null
R=oracle/_ref/reseek
H=oracle/_ref/ref_harness
T=/root/reference/test_data
G=tests/golden
TMP=$(mktemp -d)
COLS=query+target+qlo+qhi+ql+tlo+thi+tl+pctid+pvalue+evalue+cigar+dpscore+lddt+newts+ids+gaps+aq

# 1. constant tables -> product header (generated data)
$H tables reseek_amd/csrc/rsk_tables_data.h

# 2. per-chain inputs (profile bytes, Mu letters, 3-mers, CA coords, self-rev) per mode
for m in sensitive verysensitive fast; do
  $H db $T/q100.bca $TMP/q100_$m.rskdb -- -$m
  gzip -9n < $TMP/q100_$m.rskdb > $G/q100_$m.rskdb.gz
done
$H dbq $T/q100.bca $TMP/q100_sensitive_dbq.rskdb -- -sensitive
gzip -9n < $TMP/q100_sensitive_dbq.rskdb > $G/q100_sensitive_dbq.rskdb.gz
$H db $T/q10.bca $TMP/q10.rskdb -- -sensitive
gzip -9n < $TMP/q10.rskdb > $G/q10_sensitive.rskdb.gz
$H db $T/palms.bca $TMP/palms.rskdb -- -sensitive
gzip -9n < $TMP/palms.rskdb > $G/palms_sensitive.rskdb.gz

# 3. per-pair intermediates (all i<=j pairs)
$H pairs $T/q100.bca $TMP/pairs_q100_sensitive.bin 100 -- -sensitive
gzip -9n < $TMP/pairs_q100_sensitive.bin > $G/pairs_q100_sensitive.bin.gz
$H pairs $T/q100.bca $TMP/pairs_q32_verysensitive.bin 32 -- -verysensitive
gzip -9n < $TMP/pairs_q32_verysensitive.bin > $G/pairs_q32_verysensitive.bin.gz

# 4. end-to-end hit tables of the reference binary (sorted; -threads 1)
for m in sensitive verysensitive fast; do
  $R -search $T/q100.bca -$m -columns $COLS -output $TMP/q100_$m.tsv -threads 1 -quiet >/dev/null 2>&1
  sort $TMP/q100_$m.tsv | gzip -9n > $G/hits_q100_$m.tsv.gz
  $R -search $T/q100.bca -$m -output $TMP/q100_${m}_std.tsv -threads 1 -quiet >/dev/null 2>&1
  sort $TMP/q100_${m}_std.tsv | gzip -9n > $G/hits_q100_${m}_std.tsv.gz
done
$R -search $T/q10.bca -sensitive -columns $COLS -output $TMP/q10.tsv -threads 1 -quiet >/dev/null 2>&1
sort $TMP/q10.tsv > $G/hits_q10_sensitive.tsv
$R -search $T/palms.bca -sensitive -columns $COLS -output $TMP/palms.tsv -threads 1 -quiet >/dev/null 2>&1
sort $TMP/palms.tsv | gzip -9n > $G/hits_palms_sensitive.tsv.gz
$R -search $T/q100.bca -db $T/q100.bca -sensitive -columns $COLS -output $TMP/q100db.tsv -threads 1 -quiet >/dev/null 2>&1
sort $TMP/q100db.tsv | gzip -9n > $G/hits_q100_db_q100_sensitive.tsv.gz

# 5. Mu-letter known answers: real SCOP40 Mu sequences (first 160 of the reference's own
#    test_data/scop40.mu.fa, all ordered pairs) and seeded random/adversarial pairs
$H mukat $T/scop40.mu.fa 0 160 $TMP/mukat_scop40_160.bin
gzip -9n < $TMP/mukat_scop40_160.bin > $G/mukat_scop40_160.bin.gz
$H randkat 0x5EED5EEC 3000 $TMP/randkat_3000.bin
gzip -9n < $TMP/randkat_3000.bin > $G/randkat_3000.bin.gz
# the reference's own -test_xdrop / testsw peptide pairs (+300 random ones) through SWFast, SWGapless, XDropFwd/Bwd, MergeFwdBwd
$H xdropkat 0x5EED 300 $TMP/xdropkat_309.bin
gzip -9n < $TMP/xdropkat_309.bin > $G/xdropkat_309.bin.gz

# 6. the empirical SCOP40 chain-length list (used by the synthetic generator)
python3 - <<EOF
import gzip
L=[]; n=0
for line in open("$T/scop40.mu.fa"):
    if line.startswith(">"):
        if n: L.append(n)
        n=0
    else: n+=len(line.strip())
L.append(n)
open("$G/scop40_lengths.txt","w").write("\n".join(map(str,L))+"\n")
print(len(L), sum(L))
EOF
# 7. k-mer prefilter (exact k-mers; -threads 1 because -threads N is not deterministic):
#    the reference's own test data file scop40.mu.fa (11,211 Mu sequences) travels as a fixture.
gzip -9n < $T/scop40.mu.fa > $G/scop40.mu.fa.gz
awk 'BEGIN{n=0} /^>/{n++} n<=1000' $T/scop40.mu.fa > $TMP/sub1000.mu.fa
$R -prefilter_mu $TMP/sub1000.mu.fa -db $TMP/sub1000.mu.fa -output $TMP/t1000.tsv -output2 $TMP/s1000.tsv -threads 1 -quiet >/dev/null 2>&1
$R -prefilter_mu $TMP/sub1000.mu.fa -db $TMP/sub1000.mu.fa -output $TMP/t1000_b50.tsv -output2 $TMP/s1000_b50.tsv -rsb_size 50 -threads 1 -quiet >/dev/null 2>&1
gzip -9n < $TMP/t1000.tsv > $G/prefilter_sub1000_tmp.tsv.gz
LC_ALL=C sort $TMP/s1000.tsv | gzip -9n > $G/prefilter_sub1000_scores.tsv.gz
gzip -9n < $TMP/t1000_b50.tsv > $G/prefilter_sub1000_b50_tmp.tsv.gz
LC_ALL=C sort $TMP/s1000_b50.tsv | gzip -9n > $G/prefilter_sub1000_b50_scores.tsv.gz
$R -prefilter_mu $T/scop40.mu.fa -db $T/scop40.mu.fa -output $TMP/tfull.tsv -output2 $TMP/sfull.tsv -threads 1 -quiet >/dev/null 2>&1
echo "lines $(wc -l < $TMP/sfull.tsv) sorted_scores_md5 $(LC_ALL=C sort $TMP/sfull.tsv | md5sum | cut -d' ' -f1) tmp_tsv_md5 $(md5sum < $TMP/tfull.tsv | cut -d' ' -f1)" > $G/prefilter_scop40_full.md5.txt
cat $G/prefilter_scop40_full.md5.txt
# 8. `-search -fast -db`: k-mer neighbourhood prefilter (MuPreFilter) + PostMuFilter under the sensitive preset
mkdir -p $TMP/kt
TMPDIR=$TMP/kt $R -search $T/q100.bca -db $T/q100.bca -fast -columns $COLS -output $TMP/q100fast.tsv -threads 1 -keeptmp -quiet >/dev/null 2>&1
sort $TMP/q100fast.tsv | gzip -9n > $G/hits_q100_db_q100_fast.tsv.gz
gzip -9n < $TMP/kt/rce.*.tmp > $G/prefilter_q100_db_q100_fast_tmp.tsv.gz
$R -search $T/q100.bca -db $T/q100.bca -fast -output $TMP/q100fast_std.tsv -threads 1 -quiet >/dev/null 2>&1
sort $TMP/q100fast_std.tsv | gzip -9n > $G/hits_q100_db_q100_fast_std.tsv.gz
#    a bag of 5 overflows for nearly every query: the kept candidates depend on RankedScoresBag's truncation sequence and
#    quicksort tie order (rankedscoresbag.cpp:34-51) -- the fixture of the multi-GPU exchange tests
mkdir -p $TMP/kt4
TMPDIR=$TMP/kt4 $R -search $T/q100.bca -db $T/q100.bca -fast -rsb_size 5 -columns $COLS -output $TMP/q100fast_b5.tsv -threads 1 -keeptmp -quiet >/dev/null 2>&1
sort $TMP/q100fast_b5.tsv | gzip -9n > $G/hits_q100_db_q100_fast_rsb5.tsv.gz
gzip -9n < $TMP/kt4/rce.*.tmp > $G/prefilter_q100_db_q100_fast_rsb5_tmp.tsv.gz
#    neighbourhood prefilter alone on Mu FASTA inputs (ref_harness prefhood = MuPreFilter + RankedScoresBag::ToScoreTsv)
awk 'BEGIN{n=0} /^>/{n++} n<=80' $T/scop40.mu.fa > $TMP/sub80.mu.fa
$H prefhood $TMP/sub80.mu.fa $TMP/sub1000.mu.fa $TMP/h80_scores.tsv $TMP/h80_tmp.tsv -- -fast -threads 1
$H prefhood $TMP/sub80.mu.fa $TMP/sub1000.mu.fa $TMP/h80t_scores.tsv $TMP/h80t_tmp.tsv -- -fast -threads 1 -idxt
$H prefhood $TMP/sub1000.mu.fa $TMP/sub1000.mu.fa $TMP/h1000_scores.tsv $TMP/h1000_tmp.tsv -- -fast -threads 1
for n in h80 h80t; do LC_ALL=C sort $TMP/${n}_scores.tsv | gzip -9n > $G/prefilter_hood_${n}_scores.tsv.gz; gzip -9n < $TMP/${n}_tmp.tsv > $G/prefilter_hood_${n}_tmp.tsv.gz; done
LC_ALL=C sort $TMP/h1000_scores.tsv | gzip -9n > $G/prefilter_hood_h1000_scores.tsv.gz
# 9. -convert (Mu FASTA, .bca round trip) and -dbmu
$R -convert $T/q100.bca -feature_fasta $TMP/q100.mu.fa -threads 1 -quiet >/dev/null 2>&1
gzip -9n < $TMP/q100.mu.fa > $G/q100_convert.mu.fa.gz
$R -convert $T/q10.bca -bca $TMP/q10_rt.bca -threads 1 -quiet >/dev/null 2>&1
cmp $TMP/q10_rt.bca $T/q10.bca          # the reference's own writer reproduces its test file
mkdir -p $TMP/kt2
TMPDIR=$TMP/kt2 $R -search $T/q100.bca -db $T/q100.bca -fast -dbmu $TMP/q100.mu.fa -columns $COLS -output $TMP/q100fast_dbmu.tsv -threads 1 -keeptmp -quiet >/dev/null 2>&1
sort $TMP/q100fast_dbmu.tsv | gzip -9n > $G/hits_q100_db_q100_fast_dbmu.tsv.gz
gzip -9n < $TMP/kt2/rce.*.tmp > $G/prefilter_q100_db_q100_fast_dbmu_tmp.tsv.gz
for f in q10 q100 palms; do gzip -9n < $T/$f.bca > $G/$f.bca.gz; done
# 10. D1 leftovers on real chains
$H d1pairs $T/q100.bca $TMP/d1pairs_q40.bin 40 -- -sensitive
gzip -9n < $TMP/d1pairs_q40.bin > $G/d1pairs_q40_sensitive.bin.gz
# 11. edge cases: chains of 1..65 residues, a 2099-residue chain, a duplicate chain and a duplicated label
python3 $G/make_edge_bca.py $TMP/edge.bca
gzip -9n < $TMP/edge.bca > $G/edge.bca.gz
for m in sensitive fast verysensitive; do
  $R -search $TMP/edge.bca -$m -columns $COLS -output $TMP/edge_$m.tsv -threads 1 -quiet >/dev/null 2>&1
  sort $TMP/edge_$m.tsv | gzip -9n > $G/hits_edge_$m.tsv.gz
done
$R -search $TMP/edge.bca -db $TMP/edge.bca -sensitive -columns $COLS -output $TMP/edge_db.tsv -threads 1 -quiet >/dev/null 2>&1
sort $TMP/edge_db.tsv | gzip -9n > $G/hits_edge_db.tsv.gz
$R -search $TMP/edge.bca -db $TMP/edge.bca -sensitive -noself -columns $COLS -output $TMP/edge_db_noself.tsv -threads 1 -quiet >/dev/null 2>&1
sort $TMP/edge_db_noself.tsv | gzip -9n > $G/hits_edge_db_noself.tsv.gz
mkdir -p $TMP/kt3
TMPDIR=$TMP/kt3 $R -search $TMP/edge.bca -db $TMP/edge.bca -fast -columns $COLS -output $TMP/edge_fastdb.tsv -threads 1 -keeptmp -quiet >/dev/null 2>&1
sort $TMP/edge_fastdb.tsv | gzip -9n > $G/hits_edge_fastdb.tsv.gz
gzip -9n < $TMP/kt3/rce.*.tmp > $G/prefilter_edge_fastdb_tmp.tsv.gz
rm -rf $TMP
ls -la $G
# 12. BASELINE configs[3] / configs[4] shapes at fixture size: -db searches in -sensitive and -verysensitive, with a
#     PDB-like length tail up to 5,000 residues (Mu fallback > 2048, row groups > 1024, long-chain path >= 600)
python3 $G/make_tail_bca.py subset $T/q100.bca $TMP/q32.bca 32
$R -search $TMP/q32.bca -db $T/q100.bca -verysensitive -columns $COLS -output $TMP/q32v.tsv -threads 1 -quiet >/dev/null 2>&1
sort $TMP/q32v.tsv | gzip -9n > $G/hits_q32_db_q100_verysensitive.tsv.gz
# BASELINE configs[0], literally: `reseek -search <32-chain .bca subset> -sensitive`, all-vs-all, on the CPU (all columns + default columns)
$R -search $TMP/q32.bca -sensitive -columns $COLS -output $TMP/q32s.tsv -threads 1 -quiet >/dev/null 2>&1
sort $TMP/q32s.tsv | gzip -9n > $G/hits_q32_sensitive.tsv.gz
$R -search $TMP/q32.bca -sensitive -output $TMP/q32s_std.tsv -threads 1 -quiet >/dev/null 2>&1
sort $TMP/q32s_std.tsv | gzip -9n > $G/hits_q32_sensitive_std.tsv.gz
python3 $G/make_tail_bca.py tail $TMP/tailq.bca $TMP/taildb.bca
gzip -9n < $TMP/tailq.bca > $G/tailq.bca.gz
gzip -9n < $TMP/taildb.bca > $G/taildb.bca.gz
for m in sensitive verysensitive; do
  $R -search $TMP/tailq.bca -db $TMP/taildb.bca -$m -columns $COLS -output $TMP/tail_$m.tsv -threads 1 -quiet >/dev/null 2>&1
  sort $TMP/tail_$m.tsv | gzip -9n > $G/hits_tail_db_$m.tsv.gz
done
$R -search $TMP/taildb.bca -verysensitive -columns $COLS -output $TMP/tail_self_v.tsv -threads 1 -quiet >/dev/null 2>&1
sort $TMP/tail_self_v.tsv | gzip -9n > $G/hits_taildb_self_verysensitive.tsv.gz
# 13. the long-chain path stage by stage (chained HSPs, mega scores, XDropHSP start, XDropFwd / XDropBwd, MergeFwdBwd) for every
#     long-chain pair of palms.bca, from the reference's own functions (checked against DSSAligner::AlignMKF inside the harness)
$H xdrophsp $T/palms.bca $TMP/xdrophsp_palms.bin 39 -- -sensitive
gzip -9n < $TMP/xdrophsp_palms.bin > $G/xdrophsp_palms_sensitive.bin.gz
# 13b. the same for the 48-chain set with the length tail (17 .. 5,000 residues; 1,215 long-chain pairs): bands that outgrow
#      the device kernel's LDS ring, extensions next to chain ends, short partners
$H db $TMP/taildb.bca $TMP/taildb.rskdb -- -sensitive
gzip -9n < $TMP/taildb.rskdb > $G/taildb_sensitive.rskdb.gz
$H xdrophsp $TMP/taildb.bca $TMP/xdrophsp_taildb.bin 48 -- -sensitive
gzip -9n < $TMP/xdrophsp_taildb.bin > $G/xdrophsp_taildb_sensitive.bin.gz
