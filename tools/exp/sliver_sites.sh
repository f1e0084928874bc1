#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for G in 0 2.0; do
  DOSX_SLIVER_MAX_GF=$G python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline --kernels-out gpurun_out/r4_sites_edos_$G.json > gpurun_out/r4_sites_edos_$G.line 2>/dev/null
  DOSX_SLIVER_MAX_GF=$G python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --kernels-out gpurun_out/r4_sites_cfg2_$G.json > gpurun_out/r4_sites_cfg2_$G.line 2>/dev/null
done
