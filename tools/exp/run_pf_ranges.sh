#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3 4; do
  echo "== RSK_PF_RANGES=$r"
  RSK_PF_RANGES=$r RSK_TRACE=1 python bench.py --configs-only config2 2>&1 | grep "target ranges\|MuPreFilter\] index\|top-B\|\"seconds\"" | cut -c1-200 | head -4
  RSK_PF_RANGES=$r python bench.py --configs-only config2 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['configs']['config2_fast_db_11211x11211']['seconds'])"
done
