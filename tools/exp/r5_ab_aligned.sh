#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
timeout 900 python3 -m pytest tests/test_gpu_round4.py -x -q -k "attention_inside" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
for i in 1 2 3; do
DOSX_ATT_ALIGNED=0 timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('two launches      ', d['ms_per_step'])"
timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --kernels-out gpurun_out/sites_aligned.json 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('aligned mfma tiles', d['ms_per_step'])"
done
