#!/bin/bash
# A/B: factored EdgeModel Linear: off | forward + weight gradient | + input gradient  (Electron-DOS shapes; cfg2 is below the size limit)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for rep in 1 2 3 4; do
  echo -n "edos off: "; DOSX_FACTOR_EDGE_WGRAD=0 python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  echo -n "edos fwd+wgrad: "; DOSX_FACTOR_DGRAD=0 python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  echo -n "edos fwd+wgrad+dgrad: "; python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  echo -n "edos_t4_b32 fwd+wgrad: "; DOSX_FACTOR_DGRAD=0 python3 bench.py --config edos_h256_t4_b32 --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | ms
  echo -n "edos_t4_b32 fwd+wgrad+dgrad: "; python3 bench.py --config edos_h256_t4_b32 --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | ms
done
