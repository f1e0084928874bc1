"""Diagnostic: s_memtime stamps of attn_bwd_dkv_kernel (streamed dK/dV, Nk > 64), workgroup (0,0): matrix wave 0 and staging
wave 0.  make -C dostransformer_amd/csrc stamps; DOSX_LIB=.../build/libdosx_stamps.so python tools/stamp_dkv.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import ops, _lib
from dostransformer_amd._lib import Attn
DEV = "cuda"
lib = _lib.load()
lib.dosx_debug_read_attn_stamps.argtypes = [C.c_void_p]
for name, Sq, Bq, Nk, Bk, H in [("eDOS self", 201, 128, 201, 128, 256), ("phonon-like 70", 70, 64, 70, 64, 128)]:
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
    part = torch.empty(Bq * nqt + Bk * nkt, 2 * H, device=DEV)
    a.dout, a.dx, a.dscores, a.dkvhat, a.dkv_accumulate = dout.data_ptr(), dx.data_ptr(), dsc.data_ptr(), dkv.data_ptr(), 1
    a.partials_q, a.partials_kv = part.data_ptr(), part.data_ptr() + 4 * Bq * nqt * 2 * H
    for _ in range(3):
        ops.attention_bwd(a)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    lib.dosx_debug_read_attn_stamps(buf)
    t0 = buf[0]
    nit = (Bq // Bk) * ((Sq + 15) // 16)
    print(f"== {name}: chunks {nit}; matrix wave: first barrier {buf[1]-t0} | chunk starts " + " ".join(str(buf[2+c]-t0) for c in range(min(nit, 14))) + f" | loop end {buf[30]-t0}")
    print("   staging wave: start", buf[32]-t0, "first stored", buf[33]-t0, "| [top, done] " + " ".join(f"[{buf[34+c]-t0} {buf[35+c]-t0}]" for c in range(0, min(nit, 14), 2)))
