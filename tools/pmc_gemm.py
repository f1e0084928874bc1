import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import ops
DEV = "cuda"
H = 128
cases = [(9000, 2 * H, 3 * H, 0, ops.EPI_LN), (6528, 4 * H, H, 0, 0), (6528, H, 4 * H, 1, 0), (450, 2 * H, 2 * H, 0, ops.EPI_LN)]
for (M, N, K, wl, epi) in cases:
    a = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) if wl == 0 else torch.randn(K, N, device=DEV)
    out = torch.empty(M, N, device=DEV)
    kw = dict(epi=epi, aux_out=torch.empty(M, device=DEV)) if epi == ops.EPI_LN else {}
    for _ in range(3):
        ops.gemm(M, N, [ops.seg(a)], w, out, w_layout=wl, **kw)
torch.cuda.synchronize()
