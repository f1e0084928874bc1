#!/bin/bash
# A/B: the unfused feed-forward forward as a tail chain (1) or as two equal row chains on two streams (3)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for rep in 1 2 3; do
  for v in 0 1 3; do
    echo -n "edos ffn_tail=$v: "; DOSX_FFN_TAIL=$v python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  done
done
