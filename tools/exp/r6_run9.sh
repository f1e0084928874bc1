#!/bin/bash
# Electron-DOS (cfg3) and its T4 / 32-crystal shard against this round's schedule changes
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O="$R/gpurun_out/r6_run9"
mkdir -p "$O"
cd "$R"
run() {
  name=$1; cfg=$2; steps=$3; shift 3
  env "$@" timeout 300 python bench.py --allow-env --no-secondary --no-cpu-baseline --config $cfg --steps $steps 2> /dev/null | \
    python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', '$cfg', r['ms_per_step'], r['check']['loss_last'], r['check']['replay_eq_eager'])" | tee -a "$O/ab.log"
}
for i in 1 2 3; do
  run "default          " edos_h256_b64 40 DOSX_X=0
  run "early hook early " edos_h256_b64 40 DOSX_MID_HOOK_LATE=0
  run "dense chain off  " edos_h256_b64 40 DOSX_DENSE_CHAIN=0
  run "default          " edos_h256_t4_b32 40 DOSX_X=0
  run "early hook early " edos_h256_t4_b32 40 DOSX_MID_HOOK_LATE=0
done
