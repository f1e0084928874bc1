"""Diagnostic: phase stamps of ffn_bwd_kernel<false, 64, 2> - the feed-forward half's backward + the crystal-aligned attention
backward behind it (ffn_att_bwd_tile) - through a real encoder layer (needs DOSX_LIB=.../build/libdosx_stamps.so).
Slots: 0 kernel start, 1 dy tile + first chunk staged, 2 fc2 input-gradient loop done (4 column blocks), 3 dh tile barrier, 4 fc1
input-gradient loop done, 5 C tile barrier, 16 attention epilogue start (= row epilogue done), 17 operands requested / keys stored, 18-19 phase a + barrier, 20 dP, 21 dS,
22-23 dq + barrier, 24 LayerNorm-0 backward / query partial rows, 25 dK + dV product, 26 drained, 27 ticket known, 28 reduction
(last arriver only), 29 end."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import functional as Fn, ops, _lib
lib = _lib.load()
lib.dosx_debug_read_ffn_stamps.argtypes = [C.c_void_p]
H, Sq, Bq, T = 128, 51, 128, 1
for Nk, Bk in ((12, 64), (51, 128)):
    gen = torch.Generator().manual_seed(1)
    P = {}
    lp = "e.layers.0"
    shapes = {lp + ".layer_norms.0.weight": (H,), lp + ".layer_norms.0.bias": (H,), lp + ".layer_norms.1.weight": (H,),
              lp + ".layer_norms.1.bias": (H,), lp + ".fc1.weight": (4 * H, H), lp + ".fc1.bias": (4 * H,),
              lp + ".fc2.weight": (H, 4 * H), lp + ".fc2.bias": (H,), "e.layer_norm.weight": (H,), "e.layer_norm.bias": (H,)}
    flat = torch.randn(sum(int(torch.tensor(v).prod()) for v in shapes.values()), generator=gen).cuda() * 0.05
    o = 0
    for k, shp in shapes.items():
        n = int(torch.tensor(shp).prod())
        P[k] = flat[o:o + n].view(*shp)
        o += n
    x = torch.randn(Sq * Bq, H, generator=gen).cuda()
    kvhat = torch.randn(Nk * Bk, H, generator=gen).cuda()
    for _ in range(3):
        y, ctx = Fn.encoder_fwd(P, "e", x, Sq, Bq, Bq, 1, kvhat, Nk, Bk, H, T)
        G = {k: torch.zeros_like(v) for k, v in P.items()}
        dkv = torch.zeros(Nk * Bk, H, device="cuda")
        sink = ops.GradSink("cuda")
        dx = Fn.encoder_bwd(P, G, "e", ctx, torch.randn(Sq * Bq, H, device="cuda"), dkv, sink)
        sink.flush()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    lib.dosx_debug_read_ffn_stamps(buf)
    s = [buf[i] for i in range(32)]
    t0 = s[0]
    print(f"Nk={Nk}:", {i: int(s[i] - t0) for i in range(32) if s[i] and s[i] >= t0})
