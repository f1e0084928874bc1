"""Diagnostic: per-phase s_memtime stamps of gemm_kernel (needs the -DDOSX_STAMPS build):
   make -C dostransformer_amd/csrc stamps
   DOSX_LIB=$PWD/dostransformer_amd/csrc/build/libdosx_stamps.so python tools/stamp_gemm.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import ops, _lib
DEV = "cuda"
lib = _lib.load()
lib.dosx_debug_read_stamps.argtypes = [C.c_void_p]
cases = [("node gemm1", 450, 256, 256, 0, ops.EPI_LN), ("fc2", 6528, 128, 512, 0, 0), ("fc1", 6528, 512, 128, 0, 0), ("edge gemm1", 9000, 256, 384, 0, ops.EPI_LN)]
for name, M, N, K, wl, epi in cases:
    a = torch.randn(M, K, device=DEV); w = torch.randn(N, K, device=DEV); out = torch.empty(M, N, device=DEV)
    kw = dict(epi=epi, aux_out=torch.empty(M, device=DEV)) if epi == ops.EPI_LN else {}
    for _ in range(5):
        ops.gemm(M, N, [ops.seg(a)], w, out, w_layout=wl, **kw)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (64 * 64))()
    lib.dosx_debug_read_stamps(buf)
    nk = (K + 31) // 32
    nwg = ((M + 31) // 32 + 36) // 37
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(20):
        ops.gemm(M, N, [ops.seg(a)], w, out, w_layout=wl, **kw)
    en.record(); torch.cuda.synchronize()
    print(f"== {name} M={M} N={N} K={K}: chunks={nk}  kernel {st.elapsed_time(en) * 50:.1f} us   (stamps: 10 ns ticks)")
    for wg in [0]:
        s = [buf[wg * 64 + i] for i in range(64)]
        t0 = s[0]
        rel = lambda i: (s[i] - t0)
        per = [f"[start {rel(3+3*kt)} issued {rel(4+3*kt)}]" for kt in range(min(nk, 17))]
        print(f" wg#0 matrix wave 0: first barrier passed {rel(1)} | " + " ".join(per[:4]) + " ... " + " ".join(per[-2:]))
        ss = [buf[32 * 64 + i] for i in range(64)]
        rs = lambda i: (ss[i] - t0)
        pers = [f"[top {rs(2+3*kt)} stored {rs(3+3*kt)} issued {rs(4+3*kt)}]" for kt in range(min(nk, 8))]
        print(f"      staging wave 0: start {rs(0)} chunk0 stored {rs(1)} | " + " ".join(pers))
        print(f"      loop end {rel(55)} | (last row block) Cs written {rel(56)} | barrier {rel(57)} | rows done {rel(58)}   (shader-clock ticks)")
