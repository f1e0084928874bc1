#!/bin/bash
# A/B: the last message-passing layer with the aggregation in front of its second Linear (DOSX_FACTOR_LAST) - Electron-DOS shapes
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for rep in 1 2 3 4; do
  echo -n "edos last=0: "; DOSX_FACTOR_LAST=0 python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  echo -n "edos last=1: "; python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  echo -n "edos_t4_b32 last=0: "; DOSX_FACTOR_LAST=0 python3 bench.py --config edos_h256_t4_b32 --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | ms
  echo -n "edos_t4_b32 last=1: "; python3 bench.py --config edos_h256_t4_b32 --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | ms
done
