#!/bin/bash
# A/B: attention inside the feed-forward launch with crystal-aligned tiles (DosxFfn.att_aligned) on / off, and which form
# the <= 16-key / <= 4096-row layers take.  Needs tools/exp/patches/ffn_att_aligned.diff applied (git apply) and a rebuild:
# the form was measured slower (profiles/r04_ab_att_aligned.log) and is not in the tree.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for rep in 1 2 3 4; do
  echo -n "aligned=0: "; DOSX_ATT_ALIGNED=0 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms
  echo -n "aligned=1 rows_first=1: "; DOSX_ATT_ALIGNED=1 DOSX_ATT_ROWS_FIRST=1 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms
  echo -n "aligned=1 rows_first=0: "; DOSX_ATT_ALIGNED=1 DOSX_ATT_ROWS_FIRST=0 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms
done
python3 tools/bench_kernels.py --what layer 2>/dev/null | grep "^layer"
python3 tools/predict_latency.py 2>/dev/null | grep "^predict phonon"
