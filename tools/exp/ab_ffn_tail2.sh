#!/bin/bash
# A/B: largest tail (rows beyond the last full round) that still goes to the concurrent chain (DOSX_FFN_TAIL_MAX)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for rep in 1 2 3; do
  for v in 2048 6000; do
    echo -n "edos tail_max=$v: "; DOSX_FFN_TAIL_MAX=$v python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
    echo -n "edos_t4_b32 tail_max=$v: "; DOSX_FFN_TAIL_MAX=$v python3 bench.py --config edos_h256_t4_b32 --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | ms
  done
done
