#!/bin/bash
# A/B: LayerNorm->PReLU backward row kernel in its lean form (64 registers, 4 KB of LDS: fits next to a weight-gradient group) - DOSX_LN_LEAN
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for v in 0 1; do echo "== DOSX_LN_LEAN=$v"; DOSX_LN_LEAN=$v python3 tools/exp/last_layer_kernels.py 2>/dev/null | grep ln_prelu; done
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "edos lean=$v: "; DOSX_LN_LEAN=$v python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
    echo -n "edos_t4_b32 lean=$v: "; DOSX_LN_LEAN=$v python3 bench.py --config edos_h256_t4_b32 --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | ms
  done
done
DOSX_LN_LEAN=1 python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline --kernels-out gpurun_out/r4_sites_lean.json > /dev/null 2>&1
python3 -c "
import json
d=json.load(open('gpurun_out/r4_sites_lean.json'))
for s in d['sites']:
    if 'ln_prelu' in s['site']: print(s['site'], s['avg_us'], s['launches_per_step'])
"
