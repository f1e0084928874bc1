#!/bin/bash
# Tile policy of the Electron-DOS feed-forward GEMMs (M = 25728: 402 row tiles of 64 -> 3.14 rounds of 256 workgroups at N = 256)
cd "$(dirname "$0")/../.."
out=gpurun_out/r4_ab_tiles.log
: > $out
for cfg in "default" "DOSX_GEMM_RT=1" "DOSX_GEMM_RT=2" "DOSX_GEMM_RT=3" "DOSX_GEMM_BN=256" "DOSX_GEMM_BN=256 DOSX_GEMM_RT=1" "DOSX_GEMM_BN=256 DOSX_GEMM_RT=3"; do
  echo "== $cfg" >> $out
  if [ "$cfg" = default ]; then python3 tools/bench_kernels.py --what edosffn 2>/dev/null | grep gemm >> $out
  else env $cfg python3 tools/bench_kernels.py --what edosffn 2>/dev/null | grep gemm >> $out; fi
done
