#!/bin/bash
# The measurement job behind profiles/<round>_*: run on the GPU box
#     gpurun -- 'bash tools/profile_round.sh r02 <git head>'
# then copy gpurun_out/prof_<round>/summary/* into profiles/.  rocprofv3 passes are separate (kernel trace | one PMC
# counter each), the program comes directly after `--`, outputs are CSV.
set -u
R=${1:-r06}
HEAD=${2:-unknown}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$R
S=$OUT/summary
mkdir -p "$S"
export TMPDIR=/tmp
cd "$ROOT"
# PMC passes FIRST: the traffic file must exist (with this tree's source hash) when the bench lines below are produced
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -o pmc -- \
    python3 "$ROOT/bench.py" --steps 20 --warmup 10 --no-cpu-baseline --no-secondary --kernels-out "$OUT/pmc_$c.kernels.json" > "$OUT/pmc_$c.log" 2>&1 < /dev/null
done
ff=$(find "$OUT/pmc_FETCH_SIZE" -name "*counter_collection.csv" | head -1)
fw=$(find "$OUT/pmc_WRITE_SIZE" -name "*counter_collection.csv" | head -1)
if [ -n "$ff" ] && [ -n "$fw" ]; then
  python3 "$ROOT/tools/pmc_traffic.py" "$ff" "$fw" "$S/${R}_pmc_traffic.json" "$HEAD" > "$OUT/pmc_traffic.log" 2>&1
  cp "$S/${R}_pmc_traffic.json" "$ROOT/profiles/${R}_pmc_traffic.json"       # bench.py below reads it from profiles/
  python3 "$ROOT/tools/pmc_summary.py" "$ff" FETCH_SIZE "$S/${R}_pmc_fetch_size_summary.csv"
  python3 "$ROOT/tools/pmc_summary.py" "$fw" WRITE_SIZE "$S/${R}_pmc_write_size_summary.csv"
fi
# the same two passes over the Electron-DOS configuration: its sites carry `traffic` too
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_edos_$c" -o pmc -- \
    python3 "$ROOT/bench.py" --config edos_h256_b64 --steps 10 --warmup 5 --no-cpu-baseline --no-secondary --kernels-out "$OUT/pmc_edos_$c.kernels.json" > "$OUT/pmc_edos_$c.log" 2>&1 < /dev/null
done
ef=$(find "$OUT/pmc_edos_FETCH_SIZE" -name "*counter_collection.csv" | head -1)
ew=$(find "$OUT/pmc_edos_WRITE_SIZE" -name "*counter_collection.csv" | head -1)
if [ -n "$ef" ] && [ -n "$ew" ]; then
  python3 "$ROOT/tools/pmc_traffic.py" "$ef" "$ew" "$S/${R}_pmc_traffic_edos.json" "$HEAD" edos_h256_b64 > "$OUT/pmc_traffic_edos.log" 2>&1
  cp "$S/${R}_pmc_traffic_edos.json" "$ROOT/profiles/${R}_pmc_traffic_edos.json"
fi
# north_star IN THE STEP (round 6): MFMA-busy of every kernel of the replayed cfg2 and cfg3 steps (one pass each; program directly
# after `--`), summarised per symbol + the launches that contain the attention / the scatter-add (tools/pmc_step.py)
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma_step" -o pmc -- \
  python3 "$ROOT/bench.py" --steps 20 --warmup 10 --no-cpu-baseline --no-secondary --kernels-out "$OUT/pmc_mfma_step.kernels.json" > "$OUT/pmc_mfma_step.log" 2>&1 < /dev/null
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma_step_edos" -o pmc -- \
  python3 "$ROOT/bench.py" --config edos_h256_b64 --steps 10 --warmup 5 --no-cpu-baseline --no-secondary --kernels-out "$OUT/pmc_mfma_step_edos.kernels.json" > "$OUT/pmc_mfma_step_edos.log" 2>&1 < /dev/null
ms=$(find "$OUT/pmc_mfma_step" -name "*counter_collection.csv" | head -1)
me=$(find "$OUT/pmc_mfma_step_edos" -name "*counter_collection.csv" | head -1)
if [ -n "$ms" ] && [ -f "$S/${R}_pmc_traffic.json" ]; then
  if [ -n "$me" ] && [ -f "$S/${R}_pmc_traffic_edos.json" ]; then
    python3 "$ROOT/tools/pmc_step.py" "$S/${R}_pmc_mfma_step.json" "$HEAD" "$ms" "$S/${R}_pmc_traffic.json" "$me" "$S/${R}_pmc_traffic_edos.json" > "$OUT/pmc_step.log" 2>&1
  else
    python3 "$ROOT/tools/pmc_step.py" "$S/${R}_pmc_mfma_step.json" "$HEAD" "$ms" "$S/${R}_pmc_traffic.json" > "$OUT/pmc_step.log" 2>&1
  fi
  cp "$S/${R}_pmc_mfma_step.json" "$ROOT/profiles/${R}_pmc_mfma_step.json"       # bench.py reads it from profiles/ (north_star.in_step)
fi
# MFMA utilisation of the attention kernels from the counters (north_star: "MFMA utilisation for attention against CDNA4 peak"):
# the attention microbenchmark (BASELINE shapes + roofline scale) under ONE pmc pass
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma_attn" -o pmc -- \
  python3 "$ROOT/tools/bench_kernels.py" --what attn > "$OUT/pmc_mfma_attn.log" 2>&1 < /dev/null
fm=$(find "$OUT/pmc_mfma_attn" -name "*counter_collection.csv" | head -1)
[ -n "$fm" ] && python3 "$ROOT/tools/pmc_mfma.py" "$fm" "$S/${R}_pmc_mfma_attention.csv" attn > "$OUT/pmc_mfma_attn.summary" 2>&1
# ... and the achieved HBM rate of the scatter-add kernel from the counters (north_star: "rocprof-reported achieved HBM GB/s for
# the scatter-add"): the scatter microbenchmark (cfg2 size + roofline scale) under the two traffic passes
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_scatter_$c" -o pmc -- \
    python3 "$ROOT/tools/bench_kernels.py" --what scatter > "$OUT/pmc_scatter_$c.log" 2>&1 < /dev/null
done
sf=$(find "$OUT/pmc_scatter_FETCH_SIZE" -name "*counter_collection.csv" | head -1)
sw=$(find "$OUT/pmc_scatter_WRITE_SIZE" -name "*counter_collection.csv" | head -1)
[ -n "$sf" ] && [ -n "$sw" ] && python3 "$ROOT/tools/pmc_scatter.py" "$sf" "$sw" "$S/${R}_pmc_scatter_add.csv" > "$OUT/pmc_scatter.summary" 2>&1
# the two north_star figures in one hash-tagged file (bench.py puts it on its line as `north_star`)
if [ -n "$fm" ] && [ -f "$S/${R}_pmc_scatter_add.csv" ]; then
  python3 "$ROOT/tools/pmc_north_star.py" "$fm" "$S/${R}_pmc_scatter_add.csv" "$S/${R}_north_star.json" "$HEAD" > "$OUT/north_star.log" 2>&1
  cp "$S/${R}_north_star.json" "$ROOT/profiles/${R}_north_star.json"
fi
# kernel traces of the two replayed steps FIRST (their per-site kernel-only fractions go on the bench lines below:
# tools/kernel_only.py -> profiles/<round>_kernel_only*.json, hash-tagged)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o st -- \
  python3 "$ROOT/bench.py" --steps 100 --warmup 20 --no-cpu-baseline --no-secondary --kernels-out "$OUT/trace.kernels.json" > "$OUT/trace.log" 2>&1 < /dev/null
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then
  cp "$f" "$S/${R}_bench_phonon_h128_b64_kernel_stats.csv"
  python3 "$ROOT/tools/kernel_only.py" "$f" "$OUT/trace.kernels.json" "$S/${R}_kernel_only.json" "$HEAD" > "$OUT/kernel_only.log" 2>&1
  cp "$S/${R}_kernel_only.json" "$ROOT/profiles/${R}_kernel_only.json"
fi
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_edos" -o st -- \
  python3 "$ROOT/bench.py" --config edos_h256_b64 --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --kernels-out "$OUT/trace_edos.kernels.json" > "$OUT/trace_edos.log" 2>&1 < /dev/null
f=$(find "$OUT/trace_edos" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then
  cp "$f" "$S/${R}_bench_edos_h256_b64_kernel_stats.csv"
  python3 "$ROOT/tools/kernel_only.py" "$f" "$OUT/trace_edos.kernels.json" "$S/${R}_kernel_only_edos.json" "$HEAD" > "$OUT/kernel_only_edos.log" 2>&1
  cp "$S/${R}_kernel_only_edos.json" "$ROOT/profiles/${R}_kernel_only_edos.json"
fi
cd "$ROOT"
# THE bench line (the driver's invocation: default flags; secondaries = eDOS H256 + shuffle inside the same record) + its per-site table
timeout 600 python3 bench.py --kernels-out "$S/${R}_bench_phonon_h128_b64_sites.json" > "$S/${R}_bench_phonon_h128_b64.json" 2> "$OUT/bench_phonon.err" < /dev/null
timeout 400 python3 bench.py --shuffle --no-cpu-baseline --no-secondary --kernels-out "$OUT/shuffle.kernels.json" > "$S/${R}_bench_phonon_h128_b64_shuffle.json" 2> "$OUT/bench_shuffle.err" < /dev/null
timeout 500 python3 bench.py --config edos_h256_b64 --kernels-out "$S/${R}_bench_edos_h256_b64_sites.json" > "$S/${R}_bench_edos_h256_b64.json" 2> "$OUT/bench_edos.err" < /dev/null
timeout 300 python3 tools/bench_wgroup.py > "$S/${R}_wgrad_groups.log" 2> /dev/null < /dev/null
# the three launches of a factored message-passing layer alone + their phase stamps (stamps build, if it travelled)
timeout 200 python3 tools/bench_edge.py phonon 64 2> /dev/null < /dev/null | grep -v amdgpu.ids > "$S/${R}_edge_kernels.log"
[ -f dostransformer_amd/csrc/build/libdosx_stamps.so ] && DOSX_LIB=dostransformer_amd/csrc/build/libdosx_stamps.so timeout 200 python3 tools/bench_edge.py phonon 64 2> /dev/null < /dev/null | grep -E "wave 0" >> "$S/${R}_edge_kernels.log"
# the N > 1 bench path on this one GPU (two ranks share cuda:0 over gloo: plumbing check of the split replay plan, no scaling meaning)
timeout 400 python3 bench.py --gpus 2 --dist-backend gloo --share-gpu --steps 20 --warmup 5 --no-cpu-baseline > "$S/${R}_bench_2ranks_shared_gpu.json" 2> "$OUT/bench_2ranks.err" < /dev/null
timeout 600 python3 tools/bench_kernels.py > "$S/${R}_kernel_microbench.log" 2> "$OUT/microbench.err" < /dev/null
timeout 200 python3 tools/predict_latency.py 2> /dev/null | grep "^predict" >> "$S/${R}_kernel_microbench.log"
ls -la "$S"
