#!/bin/bash
# A/B: the last layer's aggregate-first form at the Phonon-DOS benchmark shape (0.59 GF: below the default limit of 1.3 GF)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for rep in 1 2 3 4; do
  for v in 1.3 0; do
    echo -n "cfg2 last_min_gf=$v: "; DOSX_FACTOR_LAST_MIN_GF=$v python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms
  done
done
