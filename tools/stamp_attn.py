"""Diagnostic: per-phase s_memtime stamps of attn_fwd_kernel (needs the -DDOSX_STAMPS build):
   make -C dostransformer_amd/csrc stamps
   DOSX_LIB=$PWD/dostransformer_amd/csrc/build/libdosx_stamps.so python tools/stamp_attn.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import ops, _lib
from dostransformer_amd._lib import Attn
DEV = "cuda"
lib = _lib.load()
lib.dosx_debug_read_attn_stamps.argtypes = [C.c_void_p]
names = ["start", "x+K loads issued", "LN done", "barrier", "QK+store", "barrier", "softmax", "barrier", "PV+store", "barrier", "epilogue"]
for name, Sq, Bq, Nk, Bk, H in [("phonon cross", 51, 128, 12, 64, 128), ("phonon self", 51, 128, 51, 128, 128), ("eDOS self", 201, 128, 201, 128, 256)]:
    x = torch.randn(Sq * Bq, H, device=DEV); kv = torch.randn(Nk * Bk, H, device=DEV)
    g, b = torch.randn(H, device=DEV), torch.randn(H, device=DEV)
    out = torch.empty(Sq * Bq, H, device=DEV); probs = torch.empty(Bq, Sq, Nk, device=DEV)
    qs, os_ = torch.empty(Sq * Bq, 2, device=DEV), torch.empty(Sq * Bq, 2, device=DEV)
    a = Attn()
    a.Sq, a.Bq, a.Nk, a.Bk, a.H, a.q_stride_s, a.q_stride_b = Sq, Bq, Nk, Bk, H, Bq, 1
    a.x, a.kvhat, a.gamma0, a.beta0 = x.data_ptr(), kv.data_ptr(), g.data_ptr(), b.data_ptr()
    a.out, a.probs, a.qstats, a.out_stats = out.data_ptr(), probs.data_ptr(), qs.data_ptr(), os_.data_ptr()
    for _ in range(5):
        ops.attention_fwd(a)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    lib.dosx_debug_read_attn_stamps(buf)
    t0 = buf[0]
    print(f"== {name} Sq={Sq} Bq={Bq} Nk={Nk} H={H}: " + " | ".join(f"{n} {buf[i] - t0}" for i, n in enumerate(names)))
