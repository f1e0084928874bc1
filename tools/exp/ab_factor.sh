#!/bin/bash
# A/B: EdgeModel weight gradient factored into node-sum jobs (DOSX_FACTOR_EDGE_WGRAD) on / off
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_us'])"; }
for rep in 1 2 3 4; do
  for F in 0 1; do
    echo -n "cfg2 factor=$F: "; DOSX_FACTOR_EDGE_WGRAD=$F python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms
    echo -n "edos factor=$F: "; DOSX_FACTOR_EDGE_WGRAD=$F python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  done
done
