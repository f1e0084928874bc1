#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
for i in 1 2 3; do
for v in 0 1; do
DOSX_LATE_SELF_FLUSH=$v timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('late_self_flush=$v', d['ms_per_step'])"
done
done
