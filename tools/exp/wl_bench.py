import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_kernels as bk
from dostransformer_amd import ops
for (M,N,K) in [(25728,1024,256),(25728,256,1024),(16128,768,512),(16128,512,768),(9344,384,256),(9344,256,384),(6528,512,128),(6528,128,512)]:
    for wl in (0,1):
        bk.gemm_case(f"plain wl={wl}", M,N,K,wl=wl)
bk.gemm_case("relu-mask epi wl=1", 25728,1024,256,wl=1,epi=ops.EPI_PRELU_BWD)
