#!/bin/bash
# per-site tables of the cfg2 step with / without a switch: bash tools/exp/r6_sites.sh VAR  (values 0 and 1)
set -u
V=${1:-DOSX_MLP_LN_CS}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O="$R/gpurun_out/r6_sites"
mkdir -p "$O"
cd "$R"
for v in 0 1; do
  env $V=$v timeout 300 python bench.py --allow-env --no-secondary --no-cpu-baseline --steps 100 --kernels-out "$O/sites_$v.json" 2> /dev/null > "$O/bench_$v.json"
done
python - <<PY
import json
a=json.load(open("$O/sites_0.json")); b=json.load(open("$O/sites_1.json"))
print("ms", a["ms_per_step"], b["ms_per_step"])
def tab(d): return {s["site"]:(s["launches_per_step"], s["avg_us"], s["us_per_step"]) for s in d["sites"]}
ta, tb = tab(a), tab(b)
for k in sorted(set(ta)|set(tb), key=lambda k: -(ta.get(k,(0,0,0))[2]+tb.get(k,(0,0,0))[2])):
    print(f"{k[:52]:52s} {str(ta.get(k)):28s} {str(tb.get(k)):28s}")
PY
