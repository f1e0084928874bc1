#!/bin/bash
# A/B: the output heads with their per-crystal K-segments multiplied once per crystal (DOSX_FACTOR_HEADS)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for rep in 1 2 3 4; do
  for v in 0 1; do
    echo -n "cfg2 heads=$v: "; DOSX_FACTOR_HEADS=$v python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms
    echo -n "edos heads=$v: "; DOSX_FACTOR_HEADS=$v python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
    echo -n "edos_t4_b32 heads=$v: "; DOSX_FACTOR_HEADS=$v python3 bench.py --config edos_h256_t4_b32 --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | ms
  done
done
