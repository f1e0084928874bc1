#!/bin/bash
# round-5 baseline: bench line + per-queue timeline of one replayed step (cfg2), factor on/off
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_base
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --kernels-out "$OUT/sites.json" > "$OUT/bench.json" 2> "$OUT/bench.err" < /dev/null
DOSX_FACTOR_MIN_GF=0.5 timeout 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --kernels-out "$OUT/sites_fac.json" > "$OUT/bench_fac.json" 2> "$OUT/bench_fac.err" < /dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o st -- \
  python3 "$ROOT/bench.py" --steps 60 --warmup 20 --no-cpu-baseline --no-secondary --kernels-out "$OUT/trace.kernels.json" > "$OUT/trace.log" 2>&1 < /dev/null
cd "$ROOT"
python3 tools/step_streams.py "$OUT/trace" 0.0 > "$OUT/streams.txt" 2>&1
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
rm -rf "$OUT/trace"
head -c 600 "$OUT/bench.json"; echo; head -c 600 "$OUT/bench_fac.json"; echo
