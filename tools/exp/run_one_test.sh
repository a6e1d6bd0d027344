#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
ulimit -c 0
timeout 600 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -40
