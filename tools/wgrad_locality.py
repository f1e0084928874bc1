"""Does the wgrad kernel's speed depend on WHICH XCD last wrote its operands?  (cross-kernel L2 locality)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import ops
DEV = "cuda"
def timeit(fn, pre, iters=50):
    for _ in range(5): pre(); fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(iters):
        pre()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / iters * 1e3
for name, M, N, K in [("fc1 wgrad", 6528, 512, 128), ("fc2 wgrad", 6528, 128, 512), ("edge W1", 9000, 256, 384)]:
    dy = torch.randn(M, N, device=DEV); a = torch.randn(M, K, device=DEV)
    ns = ops.wgrad_splits(M, N, K)
    slab = torch.empty(ns, N, K, device=DEV)
    f = lambda: ops.wgrad(M, N, ops.seg(dy), [ops.seg(a)], slab, None, ns)
    big = torch.empty(64 << 20, device=DEV)
    t_warm = timeit(f, lambda: None)
    t_dy = timeit(f, lambda: dy.mul_(1.0))
    t_both = timeit(f, lambda: (dy.mul_(1.0), a.mul_(1.0)))
    t_cold = timeit(f, lambda: big.zero_())          # 256 MiB write: flushes L2 and most of the MALL
    print(f"{name:10s} M={M} N={N} K={K}: operands as left by the previous wgrad {t_warm:5.1f} us | dY rewritten by an "
          f"elementwise kernel {t_dy:5.1f} | dY and A rewritten {t_both:5.1f} | after a 256 MiB memset {t_cold:5.1f}")
