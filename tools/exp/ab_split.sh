#!/bin/bash
# A/B: mixed-height single-launch split of dosx_gemm (DOSX_GEMM_SPLIT) against / with the concurrent feed-forward tail (DOSX_FFN_TAIL)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for v in 0 1; do echo "== DOSX_GEMM_SPLIT=$v"; DOSX_GEMM_SPLIT=$v python3 tools/bench_kernels.py --what edosffn 2>/dev/null | grep gemm | head -4; done
for rep in 1 2 3; do
  for cfg in "0 1" "1 0" "1 1"; do
    set -- $cfg
    echo -n "edos split=$1 ffn_tail=$2: "; DOSX_GEMM_SPLIT=$1 DOSX_FFN_TAIL=$2 python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  done
  for v in 0 1; do
    echo -n "edos_t4_b32 split=$v: "; DOSX_GEMM_SPLIT=$v python3 bench.py --config edos_h256_t4_b32 --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | ms
  done
done
