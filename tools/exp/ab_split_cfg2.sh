#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "cfg2 split=$v: "; DOSX_GEMM_SPLIT=$v python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms
  done
done
