#!/bin/bash
# an encoder stack's layers in one forward launch: tests + interleaved A/B
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O="$R/gpurun_out/r6_run7"
mkdir -p "$O"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_ffn.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_step.py tests/test_gpu_predict.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
run() {
  name=$1; cfg=$2; steps=$3; shift 3
  env "$@" timeout 300 python bench.py --allow-env --no-secondary --no-cpu-baseline --config $cfg --steps $steps 2> /dev/null | \
    python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', '$cfg', r['ms_per_step'], r['check']['loss_last'], r['check']['replay_eq_eager'])" | tee -a "$O/ab.log"
}
for i in 1 2 3; do
  run "layer by layer" phonon_h128_b64 200 DOSX_FFN_MULTI=0
  run "stack per launch" phonon_h128_b64 200 DOSX_FFN_MULTI=1
done
