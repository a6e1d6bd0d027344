#!/bin/bash
# fresh reference-binary comparisons on the final r04 kernels (new seeds; PDB-like length tails): tests/compare_with_reference.py
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
echo "# tests/compare_with_reference.py on the final r04 tree (seed 404; reference = oracle/_ref/reseek on this box)"
echo "## 64 queries x 4000 DB chains, PDB-like tail (DB chains up to 5,000 residues), -verysensitive, reference -threads 16"
RSK_COMPARE_SEED=404 RSK_COMPARE_TAIL=1 timeout 1500 python tests/compare_with_reference.py 64 verysensitive 4000 16 2>/dev/null
echo "## 48 queries x 1500 DB chains, PDB-like tail, -sensitive, reference -threads 1"
RSK_COMPARE_SEED=405 RSK_COMPARE_TAIL=1 timeout 1500 python tests/compare_with_reference.py 48 sensitive 1500 1 2>/dev/null
echo "## 400 x 400 -fast -db (k-mer prefilter path), reference -threads 1"
RSK_COMPARE_SEED=406 timeout 1500 python tests/compare_with_reference.py 400 fast 400 1 2>/dev/null
echo "## 1500 chains all-vs-all -sensitive with 6 long chains (MKF / X-drop path), reference -threads 1"
RSK_COMPARE_SEED=407 RSK_COMPARE_LONG=6 timeout 1500 python tests/compare_with_reference.py 1500 sensitive 0 1 2>/dev/null
} > gpurun_out/r04_compare_final.txt 2>&1
grep -c '"identical": true' gpurun_out/r04_compare_final.txt
grep '"identical"\|_seconds\|rows\|##' gpurun_out/r04_compare_final.txt
