#!/bin/bash
# longer A/B of DOSX_WGRAD_ROUND (see ab_round.sh): 5 interleaved pairs per configuration
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for rep in 1 2 3 4 5; do
  for R in 0 512; do
    echo -n "cfg2 round=$R: "; DOSX_WGRAD_ROUND=$R python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms
  done
done
for rep in 1 2 3 4 5; do
  for R in 0 1024; do
    echo -n "edos round=$R: "; DOSX_WGRAD_ROUND=$R python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  done
done
