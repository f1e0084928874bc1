#!/usr/bin/env python3
"""Diagnostic: ONE plain weight-gradient job large enough to fill the chip, timed alone, at the split count given by
DOSX_WGRAD_MAXSPLIT (static in the library: one process per setting, tools/exp/ab_sat.sh).  Tells what two co-resident
workgroups per CU give over one (same tiles, half / double the splits) and what the tile height gives (DOSX_WGRAD_NT)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import ops  # noqa: E402
from bench_wgroup import timeit  # noqa: E402

DEV = "cuda"
for (M, N, K) in [(25728, 1024, 256), (25728, 256, 1024), (17880, 512, 768), (262144, 512, 128)]:
    dy, a = torch.randn(M, N, device=DEV), torch.randn(M, K, device=DEV)
    ns = ops.wgrad_splits(M, N, K)
    slab = torch.empty(max(ops.wgrad_scratch_floats(N, K, ns), 1), device=DEV)
    dw = torch.empty(N, K, device=DEV)
    g = ops.wgrad_desc(M, N, ops.seg(dy), [ops.seg(a)], slab, None, ns, dst=dw)
    us = timeit(lambda: ops.grad_flush([g], ()), iters=20)
    fl = 2.0 * M * N * K
    print(f"NT={os.environ.get('DOSX_WGRAD_NT', 'auto')} maxsplit={os.environ.get('DOSX_WGRAD_MAXSPLIT', 'def')} "
          f"M={M} N={N} K={K} splits={ns} tiles64={ops.wgrad_tiles(N, K)}: {us:8.1f} us {fl / us / 1e6:6.1f} TF/s ({100 * fl / us / 1e6 / 157.3:4.1f} %)")
