"""Diagnostic: s_memtime stamps of attn_bwd_dkv_kernel<2> at the Electron-DOS self-attention size (needs the -DDOSX_STAMPS build):
   make -C dostransformer_amd/csrc stamps
   DOSX_LIB=$PWD/dostransformer_amd/csrc/build/libdosx_stamps.so python tools/stamp_attn_dkv.py
Matrix wave 0 stamps every chunk start (slot 2 + c), staging thread 256 its store / issue phases (32 + ...)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import ops, _lib
from dostransformer_amd._lib import Attn
DEV = "cuda"
lib = _lib.load()
lib.dosx_debug_read_attn_stamps.argtypes = [C.c_void_p]
Sq, Bq, Nk, Bk, H = 201, 128, 201, 128, 256
x = torch.randn(Sq * Bq, H, device=DEV); kv = torch.randn(Nk * Bk, H, device=DEV)
g, b = torch.randn(H, device=DEV), torch.randn(H, device=DEV)
out = torch.empty(Sq * Bq, H, device=DEV); probs = torch.empty(Bq, Sq, Nk, device=DEV)
qs, os_ = torch.empty(Sq * Bq, 2, device=DEV), torch.empty(Sq * Bq, 2, device=DEV)
a = Attn()
a.Sq, a.Bq, a.Nk, a.Bk, a.H, a.q_stride_s, a.q_stride_b = Sq, Bq, Nk, Bk, H, Bq, 1
a.x, a.kvhat, a.gamma0, a.beta0 = x.data_ptr(), kv.data_ptr(), g.data_ptr(), b.data_ptr()
a.out, a.probs, a.qstats, a.out_stats = out.data_ptr(), probs.data_ptr(), qs.data_ptr(), os_.data_ptr()
ops.attention_fwd(a)
dout = torch.randn(Sq * Bq, H, device=DEV); dx = torch.empty(Sq * Bq, H, device=DEV)
dsc = torch.empty(Bq, Sq, Nk, device=DEV); dkv = torch.zeros(Nk * Bk, H, device=DEV)
nqt, nkt = (Sq + 31) // 32, (Nk + 31) // 32
part = torch.empty(Bq * nqt + Bk * max(nkt, (Nk + 15) // 16), 2 * H, device=DEV)
a.dout, a.dx, a.dscores, a.dkvhat, a.dkv_accumulate = dout.data_ptr(), dx.data_ptr(), dsc.data_ptr(), dkv.data_ptr(), 1
a.partials_q, a.partials_kv = part.data_ptr(), part.data_ptr() + 4 * Bq * nqt * 2 * H
for _ in range(3):
    ops.attention_bwd(a)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 64)()
lib.dosx_debug_read_attn_stamps(buf)
t0 = buf[0]
print("matrix wave 0: start 0 | zeroed+barrier", buf[1] - t0, "| chunk starts", [buf[2 + c] - t0 for c in range(13)], "| epilogue", buf[30] - t0)
print("staging t256:  ", [(i, buf[32 + i] - t0) for i in range(0, 16)])
