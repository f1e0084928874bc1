#!/bin/bash
# A/B: 48-entry row phases (NJ = 3) for 33-48 keys (the 41-key Electron-DOS cross attention) against the 64-entry ones.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for rep in 1 2 3; do
  for V in 1 0; do
    echo -n "edos NJ3=$V: "; DOSX_ATTN_NJ3=$V python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  done
done
echo "--- microbench (NJ3 on)"; python3 tools/bench_kernels.py --what attn 2>/dev/null | grep eDOS
echo "--- microbench (NJ3 off)"; DOSX_ATTN_NJ3=0 python3 tools/bench_kernels.py --what attn 2>/dev/null | grep eDOS
