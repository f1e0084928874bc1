#!/bin/bash
# Electron-DOS: the self stack's weight-gradient group flushed BEHIND the heads' small backward kernels (now possible with the
# factored heads: the B-row jobs that read the side stream's row sum go with the next flush)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O="$R/gpurun_out/r6_run10"
mkdir -p "$O"
cd "$R"
run() {
  name=$1; cfg=$2; steps=$3; shift 3
  env "$@" timeout 300 python bench.py --allow-env --no-secondary --no-cpu-baseline --config $cfg --steps $steps 2> /dev/null | \
    python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', '$cfg', r['ms_per_step'], r['check']['loss_last'], r['check']['replay_eq_eager'])" | tee -a "$O/ab.log"
}
for i in 1 2 3; do
  run "flush in front (auto)" edos_h256_b64 40 DOSX_X=0
  run "flush behind heads   " edos_h256_b64 40 DOSX_LATE_SELF_FLUSH=1
  run "flush behind stack 1 " edos_h256_b64 40 DOSX_LATE_SELF_FLUSH=2
  run "flush in front (auto)" edos_h256_t4_b32 40 DOSX_X=0
  run "flush behind heads   " edos_h256_t4_b32 40 DOSX_LATE_SELF_FLUSH=1
  run "flush behind stack 1 " edos_h256_t4_b32 40 DOSX_LATE_SELF_FLUSH=2
done
