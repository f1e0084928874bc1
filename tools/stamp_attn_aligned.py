"""Diagnostic: phase stamps (s_memtime, core clocks) of workgroup 0 / wave 0 of the crystal-aligned attention kernels
(csrc/attention_aligned.hip) at the Electron-DOS cross-attention shape; needs DOSX_LIB=.../build/libdosx_stamps.so.
Forward slots (first tile; +16 second tile): 1 rows normalised, 2 barrier, 3 scores, 4 barrier, 5 softmax, 6 barrier, 7 P.K,
8 barrier, 9 output written; 63 end.  Backward: 1 phase a, 2 barrier, 3 dP, 4, 5 dS + row operands requested, 6, 7 dq, 8,
9 LayerNorm-0 backward, 10, 11 share product, 12 barrier; 60 tiles done, 61 share + partial rows published, 63 end."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import ops, _lib
from dostransformer_amd._lib import Attn
lib = _lib.load()
lib.dosx_debug_read_attn_aligned_stamps.argtypes = [C.c_void_p]
DEV = "cuda"
for Sq, Bq, Nk, Bk, H in ((201, 128, 41, 64, 256), (201, 64, 41, 64, 256), (51, 128, 51, 128, 128)):
    x, kv = torch.randn(Sq * Bq, H, device=DEV), torch.randn(Nk * Bk, H, device=DEV)
    g, b = torch.randn(H, device=DEV), torch.randn(H, device=DEV)
    out, probs = torch.empty(Sq * Bq, H, device=DEV), torch.empty(Bq, Sq, Nk, device=DEV)
    qs, os_ = torch.empty(Sq * Bq, 2, device=DEV), torch.empty(Sq * Bq, 2, device=DEV)
    a = Attn()
    a.Sq, a.Bq, a.Nk, a.Bk, a.H, a.q_stride_s, a.q_stride_b = Sq, Bq, Nk, Bk, H, Bq, 1
    a.x, a.kvhat, a.gamma0, a.beta0 = x.data_ptr(), kv.data_ptr(), g.data_ptr(), b.data_ptr()
    a.out, a.probs, a.qstats, a.out_stats = out.data_ptr(), probs.data_ptr(), qs.data_ptr(), os_.data_ptr()
    dout, dx = torch.randn(Sq * Bq, H, device=DEV), torch.empty(Sq * Bq, H, device=DEV)
    dkv = torch.zeros(Nk * Bk, H, device=DEV)
    nqt = (Sq + 31) // 32
    part = torch.empty(Bq * nqt + Bk * ((Nk + 15) // 16), 2 * H, device=DEV)
    kvp = torch.empty(Bq * nqt * Nk, H, device=DEV)
    a.dout, a.dx, a.dscores, a.dkvhat, a.dkv_accumulate = dout.data_ptr(), dx.data_ptr(), None, dkv.data_ptr(), 1
    a.partials_q, a.partials_kv = part.data_ptr(), part.data_ptr() + 4 * Bq * nqt * 2 * H
    a.dkv_part, a.dkv_cnt = kvp.data_ptr(), ops.COUNTERS.take(DEV, Bk)
    for which, fn in (("fwd", ops.attention_fwd), ("bwd", ops.attention_bwd)):
        for _ in range(3):
            fn(a)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 64)()
        lib.dosx_debug_read_attn_aligned_stamps(buf)
        s = [buf[i] for i in range(64)]
        t0 = s[0]
        print(f"{which} Sq={Sq} Bq={Bq} Nk={Nk} H={H}:", {i: int(s[i] - t0) for i in range(64) if s[i] and s[i] >= t0})
