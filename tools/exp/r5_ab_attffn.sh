#!/bin/bash
# attention (<= 16 keys) inside the feed-forward launch also for the 6528-row source encoder
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
for i in 1 2 3; do
timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('default            ', d['ms_per_step'])"
DOSX_ATT_FFN_MAX_ROWS=8192 timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('att in 32-row ffn  ', d['ms_per_step'])"
DOSX_ATT_FFN_MAX_ROWS=8192 DOSX_FFN_HALF_MAX=204 timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('att in 16-row ffn  ', d['ms_per_step'])"
done
