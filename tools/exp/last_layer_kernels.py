#!/usr/bin/env python3
"""The two row kernels of the aggregate-first last layer at the Electron-DOS size, launched alone (E 17880, N 1554, W 512)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dostransformer_amd import ops
from tools.bench_kernels import timeit
DEV = "cuda"
N, E, W, H = 1554, 17880, 512, 256
dst = torch.sort(torch.randint(0, N, (E,), device=DEV))[0].to(torch.int32)
deg = torch.bincount(dst.long(), minlength=N)
rowptr = torch.zeros(N + 1, device=DEV, dtype=torch.int32); rowptr[1:] = torch.cumsum(deg, 0).to(torch.int32)
xhat, rstd = torch.randn(E, W, device=DEV), torch.rand(E, device=DEV) + 0.5
gam, bet, alpha, bias = torch.randn(W, device=DEV), torch.randn(W, device=DEV), torch.tensor([0.25], device=DEV), torch.randn(H, device=DEV)
S, R = torch.empty(N, W, device=DEV), torch.empty(N, H, device=DEV)
us = timeit(lambda: ops.act_segment_sum(xhat, rowptr, None, gam, bet, alpha, bias, S, R, N, E, W, H))
print(f"act_segment_sum      {us:7.1f} us  {4.0 * (E * W + N * (W + H)) / us / 1e3:7.1f} GB/s")
dnode = torch.randn(N, W, device=DEV)
rows = ops.ln_prelu_bwd_partial_rows(E)
dz, part = torch.empty(E, W, device=DEV), torch.empty(rows, 2 * W + 4, device=DEV)
us = timeit(lambda: ops.ln_prelu_bwd_gather(dnode, dst, None, xhat, rstd, gam, bet, alpha, dz, part, E, W))
print(f"ln_prelu_bwd_gather  {us:7.1f} us  {8.0 * E * W / us / 1e3:7.1f} GB/s")
dact = torch.randn(E, W, device=DEV)
us = timeit(lambda: ops.ln_prelu_bwd(dact, xhat, rstd, gam, bet, alpha, dz, part, E, W))
print(f"ln_prelu_bwd (plain) {us:7.1f} us  {12.0 * E * W / us / 1e3:7.1f} GB/s")
