"""Diagnostic: phase stamps of ffn_fwd_kernel<false> (needs the -DDOSX_STAMPS build, see tools/stamp_gemm.py)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import ops, _lib
lib = _lib.load()
lib.dosx_debug_read_ffn_stamps.argtypes = [C.c_void_p]
M, H = 6528, 128
x = torch.randn(M, H, device="cuda"); stats = torch.rand(M, 2, device="cuda")
g, b = torch.randn(H, device="cuda"), torch.randn(H, device="cuda")
w1, b1 = torch.randn(4 * H, H, device="cuda"), torch.randn(4 * H, device="cuda")
w2, b2 = torch.randn(H, 4 * H, device="cuda"), torch.randn(H, device="cuda")
h = torch.empty(M, 4 * H, device="cuda"); out = torch.empty(M, H, device="cuda")
for _ in range(5):
    ops.ffn_fwd(M, H, x, stats, g, b, w1, b1, w2, b2, h, out)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 64)()
lib.dosx_debug_read_ffn_stamps(buf)
s = [buf[i] for i in range(8)]
names = ["start", "first barrier (LN1 tile + chunk 0)", "fc1 loop end (16 chunks + 4 T writes)", "T-complete barrier", "fc2 loop end (16 chunks)", "C tile barrier", "rows done"]
for i in range(1, 7):
    print(f"{names[i]:45s} +{s[i] - s[i - 1]:6d} clk   (at {s[i] - s[0]})")
