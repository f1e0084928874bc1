#!/bin/bash
# mixed-height grid: the small tail tiles at the end of the grid (0) or at its start (1)
# (knob removed after the measurement - profiles/r04_ab_gemm_tail_first.log: no difference; gemm_mixed_kernel would take a flag that maps blockIdx.x < n2 to the tail tiles)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for v in 0 1; do echo "== DOSX_GEMM_TAIL_FIRST=$v"; DOSX_GEMM_TAIL_FIRST=$v python3 tools/bench_kernels.py --what edosffn 2>/dev/null | grep gemm; done
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "edos tail_first=$v (ffn_tail off): "; DOSX_FFN_TAIL=0 DOSX_GEMM_TAIL_FIRST=$v python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
    echo -n "edos_t4_b32 tail_first=$v: "; DOSX_GEMM_TAIL_FIRST=$v python3 bench.py --config edos_h256_t4_b32 --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | ms
  done
done
