#!/bin/bash
# one workgroup per CU for the large 64-row-tile GEMMs (DOSX_GEMM_SOLO_WG = smallest grid that gets the LDS pad)
# (the knob is not in the tree: git apply tools/exp/patches/gemm_solo_wg.diff, rebuild, run; result: profiles/r04_ab_gemm_solo.log)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for v in 0 300; do echo "== DOSX_GEMM_SOLO_WG=$v"; DOSX_GEMM_SOLO_WG=$v python3 tools/bench_kernels.py --what edosffn 2>/dev/null | grep gemm; done
for rep in 1 2 3; do
  for v in 0 300; do
    echo -n "edos solo=$v: "; DOSX_GEMM_SOLO_WG=$v python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
  done
done
