#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_run5
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python3 -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -25 > "$OUT/t5.log"
cat "$OUT/t5.log"
DOSX_LIB=dostransformer_amd/csrc/build/libdosx_stamps.so python3 tools/bench_edge.py phonon 64 2>&1 | tail -8
for i in 1 2; do
DOSX_NODE_GRAD_ONE_LAUNCH=0 timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --kernels-out "$OUT/sites_nong.json" 2> "$OUT/bench_nong.err" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('nong ', d['ms_per_step'])"
timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --kernels-out "$OUT/sites_new.json" 2> "$OUT/bench_new.err" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('all ', d['ms_per_step'])"
done
DOSX_NODE_GRAD_ONE_LAUNCH=0 timeout 300 python3 bench.py --config edos_h256_b64 --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --kernels-out "$OUT/sites_edos_old.json" 2> "$OUT/bench_eold.err" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('edos nong ', d['ms_per_step'])"
timeout 300 python3 bench.py --config edos_h256_b64 --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --kernels-out "$OUT/sites_edos_new.json" 2> "$OUT/bench_enew.err" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('edos all ', d['ms_per_step'])"
