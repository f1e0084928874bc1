#!/bin/bash
# A/B: small plain dgrad GEMMs on the vector-ALU sliver kernel (DOSX_SLIVER_MAX_GF: flop limit in GF, 0 = never)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for rep in 1 2 3 4; do
  for G in 0 2.0 4.0 8.0; do
    echo -n "cfg2 sliver<=$G: "; DOSX_SLIVER_MAX_GF=$G python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms
  done
  for G in 0 2.0 4.0 8.0; do
    echo -n "edos sliver<=$G: "; DOSX_SLIVER_MAX_GF=$G python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  done
done
