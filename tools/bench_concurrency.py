"""Can two libdosx kernels from different HIP streams share the GPU?  (each one alone is a single,
partially filled wave of workgroups at BASELINE sizes)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import ops
DEV = "cuda"
H = 128
M = 9000
a = torch.randn(M, 3 * H, device=DEV); w = torch.randn(2 * H, 3 * H, device=DEV); out = torch.empty(M, 2 * H, device=DEV); rstd = torch.empty(M, device=DEV)
dy = torch.randn(M, 2 * H, device=DEV); ns = ops.wgrad_splits(M, 2 * H, 3 * H); slab = torch.empty(ns, 2 * H, 3 * H, device=DEV)
a2 = torch.randn(6528, H, device=DEV); w2 = torch.randn(4 * H, H, device=DEV); out2 = torch.empty(6528, 4 * H, device=DEV)
def k_gemm(): ops.gemm(M, 2 * H, [ops.seg(a)], w, out, epi=ops.EPI_LN, aux_out=rstd)
def k_wgrad(): ops.wgrad(M, 2 * H, ops.seg(dy), [ops.seg(a)], slab, None, ns)
def k_gemm2(): ops.gemm(6528, 4 * H, [ops.seg(a2)], w2, out2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(fa, fb, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        if fa:
            with torch.cuda.stream(s1): fa()
        if fb:
            with torch.cuda.stream(s2): fb()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, fa, fb in [("gemm(edge1) + wgrad(edge W1)", k_gemm, k_wgrad), ("gemm(edge1) + gemm(fc1)", k_gemm, k_gemm2), ("wgrad + gemm(fc1)", k_wgrad, k_gemm2)]:
    run(fa, fb, 5)
    ta, tb, tab = run(fa, None), run(None, fb), run(fa, fb)
    print(f"{name}: A alone {ta:.1f} us, B alone {tb:.1f} us, both streams {tab:.1f} us per pair (sum {ta+tb:.1f})")
