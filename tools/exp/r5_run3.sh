#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_run3
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python3 -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -25 > "$OUT/t5.log"
cat "$OUT/t5.log"
timeout 900 python3 -m pytest tests/test_gpu_models.py -x -q -k "oracle_live or golden or ghost" 2>&1 | tail -15 > "$OUT/tm.log"
cat "$OUT/tm.log"
for i in 1 2; do
DOSX_FACTOR_MIN_GF=4 timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --kernels-out "$OUT/sites_old.json" 2> "$OUT/bench_old.err" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('old ', d['ms_per_step'])"
DOSX_EDGE_ONE_LAUNCH_BWD=0 timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --kernels-out "$OUT/sites_fwd.json" 2> "$OUT/bench_fwd.err" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd1 ', d['ms_per_step'])"
timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --kernels-out "$OUT/sites_new.json" 2> "$OUT/bench_new.err" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('both ', d['ms_per_step'])"
done
tail -3 "$OUT"/bench_new.err
