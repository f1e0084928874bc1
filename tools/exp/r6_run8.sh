#!/bin/bash
# phonon edge encoder in one launch: tests + interleaved A/B
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O="$R/gpurun_out/r6_run8"
mkdir -p "$O"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_graph.py -x -q -k "edge_encoder_in_one" 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 900 python -m pytest tests/test_gpu_models.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
run() {
  name=$1; cfg=$2; steps=$3; shift 3
  env "$@" timeout 300 python bench.py --allow-env --no-secondary --no-cpu-baseline --config $cfg --steps $steps 2> /dev/null | \
    python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', '$cfg', r['ms_per_step'], r['check']['loss_last'], r['check']['replay_eq_eager'])" | tee -a "$O/ab.log"
}
for i in 1 2 3; do
  run "edge enc 2 launches" phonon_h128_b64 200 DOSX_EDGE_ENC_ONE_LAUNCH=0
  run "edge enc 1 launch  " phonon_h128_b64 200 DOSX_EDGE_ENC_ONE_LAUNCH=1
done
