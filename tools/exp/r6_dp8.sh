#!/bin/bash
# how long the 8-ranks-on-one-GPU full-size run takes, by hardware queues per process
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
for q in 2 1; do
  O="$R/gpurun_out/r6_dp8/q$q"; mkdir -p "$O"
  t0=$(date +%s)
  for r in 0 1 2 3 4 5 6 7; do
    GPU_MAX_HW_QUEUES=$q HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python tests/dp_worker.py $r 8 $((29600 + q)) "$O" full > "$O/rank$r.log" 2>&1 &
  done
  wait
  echo "queues=$q: $(( $(date +%s) - t0 )) s"; grep dp_worker "$O/rank0.log"
  rm -f "$O"/*.npz
done
