"""Round quantisation of the 512-column row-epilogue GEMMs of the hidden-256 message passing (32-row tiles, one workgroup per CU):
E = 17880 edges = 559 tiles = 2.18 rounds; 16384 = exactly 2."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_kernels as bk
from dostransformer_amd import ops
for M in (16384, 17880, 8192, 8940):
    bk.gemm_case("edge da (PRELU_LN_BWD epi)", M, 512, 256, wl=1, epi=ops.EPI_PRELU_LN_BWD)
    bk.gemm_case("edge gemm1 (EPI_LN)", M, 512, 256, epi=ops.EPI_LN)
    bk.gemm_case("edge gemm2-like plain N256 K512", M, 256, 512)
