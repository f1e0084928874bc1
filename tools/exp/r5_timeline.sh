#!/bin/bash
# one replayed step per HIP queue at the current head (kernel trace only); usage: r5_timeline.sh [config]
C=${1:-phonon_h128_b64}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl/tr -- python3 $R/bench.py --config $C --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-dp1 > $R/gpurun_out/tl/bench_$C.log 2>&1
python3 $R/tools/step_streams.py $R/gpurun_out/tl/tr 0.0 > $R/gpurun_out/tl/streams_$C.txt 2>&1
rm -rf $R/gpurun_out/tl/tr
grep ms_per_step $R/gpurun_out/tl/bench_$C.log | cut -c1-300
