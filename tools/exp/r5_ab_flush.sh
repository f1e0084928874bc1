#!/bin/bash
# A/B: where the GNN layers' weight-gradient groups are flushed (DOSX_SPLIT_LATE_FLUSH 0 / 1 / 2), interleaved
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
for i in 1 2 3; do
for v in 1 2; do
DOSX_SPLIT_LATE_FLUSH=$v timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('late_flush=$v ', d['ms_per_step'], d['roofline']['avg_us'], d['roofline']['launches_per_step'])"
done
done
