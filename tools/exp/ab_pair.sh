#!/bin/bash
# A/B: the two output heads as one launch (DOSX_GEMM_PAIR)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for rep in 1 2 3 4; do
  for v in 0 1; do
    echo -n "cfg2 pair=$v: "; DOSX_GEMM_PAIR=$v python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms
  done
  for v in 0 1; do
    echo -n "edos pair=$v: "; DOSX_GEMM_PAIR=$v python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  done
done
python3 tools/predict_latency.py 2>/dev/null | grep "^predict"
