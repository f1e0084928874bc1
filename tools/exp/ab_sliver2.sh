#!/bin/bash
# cfg2 only, sliver off / on (2 GF), 6 interleaved rounds, alternating which comes first
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -4
for rep in 1 2 3 4 5 6; do
  if [ $((rep % 2)) = 1 ]; then order="0 2.0"; else order="2.0 0"; fi
  for G in $order; do
    echo -n "cfg2 sliver<=$G: "; DOSX_SLIVER_MAX_GF=$G python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | ms
  done
done
