"""How much does a torch.cuda.Event bracket add to one kernel launch?  (calibration for bench.py's per-site timings)"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import ops
DEV = "cuda"
M, N, K = 9000, 256, 384
a = torch.randn(M, K, device=DEV); w = torch.randn(N, K, device=DEV); out = torch.empty(M, N, device=DEV)
kw = dict(epi=ops.EPI_LN, aux_out=torch.empty(M, device=DEV))
f = lambda: ops.gemm(M, N, [ops.seg(a)], w, out, w_layout=0, **kw)
for _ in range(5): f()
torch.cuda.synchronize()
def bracket(fn, n=100):
    ev = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); ev.append((s, e))
    torch.cuda.synchronize()
    return statistics.median(s.elapsed_time(e) * 1e3 for s, e in ev)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(200): f()
e.record(); torch.cuda.synchronize()
print("back-to-back per launch %.2f us" % (s.elapsed_time(e) * 1e3 / 200))
print("empty bracket %.2f us" % bracket(lambda: None))
print("bracketed gemm %.2f us" % bracket(f))
x = torch.empty(1 << 10, device=DEV)
print("bracketed tiny fill %.2f us" % bracket(lambda: ops.fill(x, 0.0)))
