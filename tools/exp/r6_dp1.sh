#!/bin/bash
# what the data-parallel step costs one rank (1-rank RCCL group) with / without the mid gradient bucket and the late early-bucket hook
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O="$R/gpurun_out/r6_dp1"
mkdir -p "$O"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_predict.py -x -q 2>&1 | tail -3
run() {
  name=$1; shift
  env "$@" timeout 300 python bench.py --allow-env --no-secondary --no-cpu-baseline --steps 200 ${EXTRA:-} 2> /dev/null | \
    python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', r['ms_per_step'], r.get('dp',{}).get('plan'))" | tee -a "$O/ab.log"
}
for i in 1 2 3; do
  EXTRA="" run "single" DOSX_X=0
  EXTRA="--force-dist" run "dp1 mid+late" DOSX_X=0
  EXTRA="--force-dist" run "dp1 nomid+late" DOSX_DP_MID_BUCKET=0
  EXTRA="--force-dist" run "dp1 mid, hook early" DOSX_MID_HOOK_LATE=0
  EXTRA="--force-dist" run "dp1 nomid, hook early" DOSX_DP_MID_BUCKET=0 DOSX_MID_HOOK_LATE=0
done
