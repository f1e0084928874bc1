#!/bin/bash
# hardware-queue count of the HIP runtime against the data-parallel step's stream waits
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O="$R/gpurun_out/r6_dp2"
mkdir -p "$O"
cd "$R"
run() {
  name=$1; shift
  env "$@" timeout 300 python bench.py --allow-env --no-secondary --no-cpu-baseline --steps 200 ${EXTRA:-} 2> /dev/null | \
    python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', r['ms_per_step'])" | tee -a "$O/ab.log"
}
for i in 1 2 3; do
  EXTRA="" run "single q4" DOSX_X=0
  EXTRA="" run "single q8" GPU_MAX_HW_QUEUES=8
  EXTRA="" run "single q6" GPU_MAX_HW_QUEUES=6
  EXTRA="--force-dist" run "dp1 nomid q4" DOSX_DP_MID_BUCKET=0
  EXTRA="--force-dist" run "dp1 nomid q8" DOSX_DP_MID_BUCKET=0 GPU_MAX_HW_QUEUES=8
  EXTRA="--force-dist" run "dp1 mid q8" GPU_MAX_HW_QUEUES=8
  EXTRA="--force-dist" run "dp1 mid q6" GPU_MAX_HW_QUEUES=6
done
