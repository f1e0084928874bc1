#!/bin/bash
# Round 6, third job: which of the column-split / chained forms of the node-side BACKWARD pays inside the step
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O="$R/gpurun_out/r6_run3"
mkdir -p "$O"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_graph.py -x -q -k "residual_path" 2>&1 | tail -2
run() {   # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --allow-env --no-secondary --no-cpu-baseline --steps 200 2> /dev/null | \
    python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', r['ms_per_step'], r['check']['loss_last'], r['check']['replay_eq_eager'])" | tee -a "$O/ab.log"
}
for i in 1 2 3; do
  run "A cs_fwd+cs_bwd"            DOSX_NODE_CHAIN=0 DOSX_DENSE_CHAIN=0
  run "B cs_fwd only"              DOSX_NODE_CHAIN=0 DOSX_DENSE_CHAIN=0 DOSX_MLP_LN_CS_BWD=0
  run "C chain, flush after"       DOSX_NODE_CHAIN=1 DOSX_DENSE_CHAIN=1 DOSX_FLUSH_AFTER_CHAIN=1
  run "D dense chain only"         DOSX_NODE_CHAIN=0 DOSX_DENSE_CHAIN=1
  run "E no cs at all"             DOSX_MLP_LN_CS=0
done
