#!/bin/bash
# A/B: node-row half of the factored EdgeModel input gradient on the side stream next to the edge-row GEMM (DOSX_FACTOR_SIDE)
# (knob removed after the measurement - profiles/r04_ab_factor_side.log: +0.3 % / neutral; the branch wrapped node_sums and the two N-row GEMMs of gnn_bwd in sink.on_side + sink.join)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for rep in 1 2 3 4; do
  for v in 0 1; do
    echo -n "edos side=$v: "; DOSX_FACTOR_SIDE=$v python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
    echo -n "edos_t4_b32 side=$v: "; DOSX_FACTOR_SIDE=$v python3 bench.py --config edos_h256_t4_b32 --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | ms
  done
done
