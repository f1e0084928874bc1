#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O="$R/gpurun_out/r6_run4"
mkdir -p "$O"
cd "$R"
run() {   # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --allow-env --no-secondary --no-cpu-baseline --steps 200 2> /dev/null | \
    python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', r['ms_per_step'], r['check']['loss_last'], r['check']['replay_eq_eager'])" | tee -a "$O/ab.log"
}
for i in 1 2 3; do
  run "F default (cs fwd, dense chain, cs bwd only there)" DOSX_X=0
  run "G + mid hook late"    DOSX_MID_HOOK_LATE=1
  run "D cs bwd everywhere"  DOSX_MLP_LN_CS_BWD=1
  run "H late self flush 0"  DOSX_LATE_SELF_FLUSH=0
  run "I split late flush 2" DOSX_SPLIT_LATE_FLUSH=2
  run "J split late flush 0" DOSX_SPLIT_LATE_FLUSH=0
  run "K flush behind node"  DOSX_GNN_FLUSH_BEFORE_NODE=0
done
