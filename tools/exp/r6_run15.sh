#!/bin/bash
# --shuffle mode (fresh batch collated on the device every step): greedy tiles over the whole batch against the crystal-aligned table
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O="$R/gpurun_out/r6_run15"
mkdir -p "$O"
cd "$R"
run() {
  name=$1; shift
  env "$@" timeout 300 python bench.py --allow-env --no-secondary --no-cpu-baseline --shuffle --steps 200 2> /dev/null | \
    python -c "import json,sys; r=json.loads(sys.stdin.read()); s={x['site']: x['us_per_step'] for x in r['top_sites']}; print('$name', r['ms_per_step'], r['value'], r['check']['loss_last'], s.get('edge_mlp_fwd[H128]'), s.get('edge_mlp_bwd[H128]'))" | tee -a "$O/ab.log"
}
for i in 1 2 3; do
  run "crystal-aligned tiles" DOSX_COLLATE_GREEDY_TILES=0
  run "greedy over the batch" DOSX_COLLATE_GREEDY_TILES=1
done
