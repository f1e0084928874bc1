#!/bin/bash
# A/B: 8 (kernel-argument table of 4 KB) vs 16 weight-gradient jobs per dosx_grad_flush launch
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
DOSX_LIB=dostransformer_amd/csrc/build/libdosx_j16.so timeout 600 python3 -m pytest tests/test_gpu_round3.py -x -q -k "wgrad or flush or sink" 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2 3; do
timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('jobs=8  ', d['ms_per_step'], d['roofline']['avg_us'], d['roofline']['frac'])"
DOSX_LIB=dostransformer_amd/csrc/build/libdosx_j16.so timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('jobs=16 ', d['ms_per_step'], d['roofline']['avg_us'], d['roofline']['frac'])"
done
