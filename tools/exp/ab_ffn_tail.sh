#!/bin/bash
# A/B: the tail rows of the unfused feed-forward layers as a concurrent chain on a side stream (DOSX_FFN_TAIL: 1 forward, 2 + backward)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for rep in 1 2 3 4; do
  for v in 0 1 2; do
    echo -n "edos ffn_tail=$v: "; DOSX_FFN_TAIL=$v python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  done
done
