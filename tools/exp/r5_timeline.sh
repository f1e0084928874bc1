#!/bin/bash
# one replayed cfg2 step per HIP queue at the current head (kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl/tr -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-dp1 > $R/gpurun_out/tl/bench.log 2>&1
python3 $R/tools/step_streams.py $R/gpurun_out/tl/tr 0.0 > $R/gpurun_out/tl/streams.txt 2>&1
rm -rf $R/gpurun_out/tl/tr
tail -1 $R/gpurun_out/tl/bench.log | cut -c1-300
