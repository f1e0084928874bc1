#!/bin/bash
# Round 6, second job: tests of the chained node-side backward + A/B of the chain switches + ffn phase stamps
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O="$R/gpurun_out/r6_run2"
mkdir -p "$O"
cd "$R"
timeout 1200 python -m pytest tests/test_gpu_graph.py -k "mlp_ln or column_split or node_side or dense_key" tests/test_gpu_step.py tests/test_gpu_models.py  -x -q > "$O/pytest.log" 2>&1
echo "pytest rc=$?" >> "$O/pytest.log"
tail -6 "$O/pytest.log"
for i in 1 2 3; do
  for v in "0 0" "1 0" "1 1"; do
    set -- $v
    DOSX_NODE_CHAIN=$1 DOSX_DENSE_CHAIN=$2 timeout 300 python bench.py --allow-env --no-secondary --no-cpu-baseline --steps 200 2> /dev/null | \
      python -c "import json,sys; r=json.loads(sys.stdin.read()); print('node_chain=$1 dense_chain=$2', r['ms_per_step'], r['check']['loss_last'], r['check']['replay_eq_eager'])" | tee -a "$O/ab.log"
  done
done
for t in stamp_ffn.py stamp_ffn_att.py stamp_ffn_att_bwd.py; do
  echo "== $t" >> "$O/stamps.log"
  DOSX_LIB=$R/dostransformer_amd/csrc/build/libdosx_stamps.so timeout 200 python tools/$t >> "$O/stamps.log" 2>&1
done
grep -v "amdgpu.ids" "$O/stamps.log" | tail -30
