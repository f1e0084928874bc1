#!/bin/bash
# per-queue timeline of one replayed step (rocprofv3 kernel trace) -> gpurun_out/r5_trace/streams.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_trace
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o st -- \
  python3 "$ROOT/bench.py" --steps 60 --warmup 20 --no-cpu-baseline --no-secondary --kernels-out "$OUT/trace.kernels.json" ${EXTRA:-} > "$OUT/trace.log" 2>&1 < /dev/null
cd "$ROOT"
python3 tools/step_streams.py "$OUT/trace" 0.0 > "$OUT/streams.txt" 2>&1
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
rm -rf "$OUT/trace"
tail -2 "$OUT/trace.log" | cut -c1-300
