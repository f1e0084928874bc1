#!/bin/bash
# Round 6: the column-split NodeModel kernels - tests, kernel times under rocprofv3, interleaved A/B of the cfg2 step
# (DOSX_MLP_LN_CS=0 / 1).  gpurun -- 'bash tools/exp/r6_cs.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O="$R/gpurun_out/r6_cs"
mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_graph.py -k "mlp_ln or column_split or node_side or dense_key" -x -q > "$O/pytest.log" 2>&1
echo "pytest rc=$?" >> "$O/pytest.log"
tail -5 "$O/pytest.log"
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o st -- python3 "$R/tools/bench_kernels.py" --what nmlp > "$O/nmlp.log" 2>&1 < /dev/null
f=$(find "$O/trace" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$O/nmlp_kernel_stats.csv" && grep -E "mlp_ln|gemm" "$f" | cut -c1-160
grep "^nmlp" "$O/nmlp.log"
rm -rf "$O/trace"
cd "$R"
for i in 1 2 3; do
  for v in 0 1; do
    DOSX_MLP_LN_CS=$v timeout 300 python bench.py --allow-env --no-secondary --no-cpu-baseline --steps 200 2> /dev/null | \
      python -c "import json,sys; r=json.loads(sys.stdin.read()); print('cs=$v', r['ms_per_step'], r['check'])" | tee -a "$O/ab.log"
  done
done
