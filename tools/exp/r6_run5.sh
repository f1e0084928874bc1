#!/bin/bash
# heads backward in one launch: test + interleaved A/B (cfg2 and Electron-DOS)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O="$R/gpurun_out/r6_run5"
mkdir -p "$O"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -k "heads_backward" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "oracle_live or g5 or g6" 2>&1 | tail -3
run() {
  name=$1; cfg=$2; steps=$3; shift 3
  env "$@" timeout 300 python bench.py --allow-env --no-secondary --no-cpu-baseline --config $cfg --steps $steps 2> /dev/null | \
    python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', '$cfg', r['ms_per_step'], r['check']['loss_last'], r['check']['replay_eq_eager'])" | tee -a "$O/ab.log"
}
for i in 1 2 3; do
  run "heads 3 launches" phonon_h128_b64 200 DOSX_HEADS_BWD_ONE_LAUNCH=0
  run "heads 1 launch  " phonon_h128_b64 200 DOSX_HEADS_BWD_ONE_LAUNCH=1
  run "heads 3 launches" edos_h256_b64 40 DOSX_HEADS_BWD_ONE_LAUNCH=0
  run "heads 1 launch  " edos_h256_b64 40 DOSX_HEADS_BWD_ONE_LAUNCH=1
done
