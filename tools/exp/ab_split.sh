#!/bin/bash
# A/B: tail-split launches of dosx_gemm (DOSX_GEMM_SPLIT): microbenchmark of the four Electron-DOS feed-forward GEMMs, then the steps
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for v in 0 1; do echo "== DOSX_GEMM_SPLIT=$v"; DOSX_GEMM_SPLIT=$v python3 tools/bench_kernels.py --what edosffn 2>/dev/null | grep gemm; done
for rep in 1 2 3 4; do
  for v in 0 1; do
    echo -n "edos split=$v: "; DOSX_GEMM_SPLIT=$v python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  done
  for v in 0 1; do
    echo -n "edos_t4_b32 split=$v: "; DOSX_GEMM_SPLIT=$v python3 bench.py --config edos_h256_t4_b32 --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | ms
  done
  for v in 0 1; do
    echo -n "cfg2 split=$v: "; DOSX_GEMM_SPLIT=$v python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms
  done
done
