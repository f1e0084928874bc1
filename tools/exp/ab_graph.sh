#!/bin/bash
# replayed launch list vs HIP graph of the same step (three streams captured), both configs
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r.get('host_ms_per_step'))"; }
for rep in 1 2 3; do
  for m in replay graph; do
    echo -n "cfg2 $m: "; python3 bench.py --launch $m --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms
  done
  for m in replay graph; do
    echo -n "edos $m: "; python3 bench.py --launch $m --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  done
done
