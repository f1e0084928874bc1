#!/bin/bash
# A/B: --shuffle line with / without bucket promotion (Trainer(promote=...): DOSX_BENCH_PROMOTE)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['slots'])"; }
for rep in 1 2 3; do
  for v in 0 0.08; do
    echo -n "shuffle promote=$v: "; DOSX_BENCH_PROMOTE=$v python3 bench.py --shuffle --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms
  done
done
