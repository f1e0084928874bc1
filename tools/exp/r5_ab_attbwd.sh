#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
timeout 900 python3 -m pytest tests/test_gpu_models.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -6
for i in 1 2 3; do
DOSX_FUSED_ATT_BWD=0 timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('attention bwd separate', d['ms_per_step'])"
timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --kernels-out gpurun_out/sites_attbwd.json 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('attention bwd in ffn  ', d['ms_per_step'])"
done
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/sites_attbwd.json'))
for s in sorted(d['sites'], key=lambda s:-s['us_per_step'])[:12]:
    print(f"{s['us_per_step']:8.1f} {s['launches_per_step']:4} {s['avg_us']:7.1f} {s['site']}")
PY
