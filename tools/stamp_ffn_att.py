"""Diagnostic: phase stamps of ffn_fwd_kernel<false, 64, 2> - the crystal-aligned attention prologue (ffn_att_tile) + the
feed-forward half (needs the -DDOSX_STAMPS build: DOSX_LIB=dostransformer_amd/csrc/build/libdosx_stamps.so)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import ops, _lib
lib = _lib.load()
lib.dosx_debug_read_ffn_stamps.argtypes = [C.c_void_p]
H, Sq, Bq = 128, 51, 128
for Nk, Bk in ((12, 64), (51, 128)):
    M = Sq * Bq
    x = torch.randn(M, H, device="cuda")
    kv = torch.randn(Nk * Bk, H, device="cuda")
    g, b = torch.randn(H, device="cuda"), torch.randn(H, device="cuda")
    flat = torch.randn(4 * H * H + 4 * H + H * 4 * H + H, device="cuda") * 0.05
    w1, b1 = flat[:4 * H * H].view(4 * H, H), flat[4 * H * H:4 * H * H + 4 * H]
    o = 4 * H * H + 4 * H
    w2, b2 = flat[o:o + 4 * H * H].view(H, 4 * H), flat[o + 4 * H * H:]
    h, out = torch.empty(M, 4 * H, device="cuda"), torch.empty(M, H, device="cuda")
    att = dict(kvhat=kv, gamma0=g, beta0=b, Nk=Nk, Bk=Bk, Bq=Bq, Sq=Sq, qs=Bq, qb=1, probs=torch.empty(Bq, Sq, Nk, device="cuda"),
               qstats=torch.empty(M, 2, device="cuda"), x1=torch.empty(M, H, device="cuda"), st1=torch.empty(M, 2, device="cuda"),
               mask=None, aligned=True)
    for _ in range(5):
        ops.ffn_fwd(M, H, x, None, g, b, w1, b1, w2, b2, h, out, att=att)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    lib.dosx_debug_read_ffn_stamps(buf)
    s = [buf[i] for i in range(32)]
    t0 = s[0]
    print(f"Nk={Nk}:", {i: int(s[i] - t0) for i in range(32) if s[i]})
