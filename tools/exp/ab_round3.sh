#!/bin/bash
# the shipped round policy (auto: 512 / 1024 workgroups per launch by workgroup lifetime) against one launch per group
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for rep in 1 2 3; do
  for R in 0 auto; do
    export DOSX_WGRAD_ROUND=$R
    echo -n "cfg2 round=$R: "; python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms
    echo -n "edos round=$R: "; python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  done
  for R in 0 512 1024 auto; do
    export DOSX_WGRAD_ROUND=$R
    echo -n "edos_t4_b32 round=$R: "; python3 bench.py --config edos_h256_t4_b32 --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | ms
    echo -n "phonon_h64_b8 round=$R: "; python3 bench.py --config phonon_h64_b8 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | ms
  done
done
