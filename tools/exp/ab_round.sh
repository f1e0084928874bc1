#!/bin/bash
# A/B: weight-gradient groups issued as successive launches of at most R workgroups (DOSX_WGRAD_ROUND, csrc/gemm.hip
# dosx_grad_flush) against one launch per group (0).  Interleaved.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for rep in 1 2; do
  for R in 0 256 512 768 1024; do
    echo -n "edos round=$R: "; DOSX_WGRAD_ROUND=$R python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  done
done
for rep in 1 2; do
  for R in 0 256 512; do
    echo -n "cfg2 round=$R: "; DOSX_WGRAD_ROUND=$R python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms
  done
done
