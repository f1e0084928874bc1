"""Diagnostic: per-phase s_memtime stamps of the weight-gradient kernel body (needs the -DDOSX_STAMPS build:
make -C dostransformer_amd/csrc stamps; DOSX_LIB=dostransformer_amd/csrc/build/libdosx_stamps.so python tools/stamp_wgrad.py).
Workgroup 0 only: matrix wave 0 (row 0 of the stamp buffer) and staging wave 0 (row 32).  Raw s_memtime ticks."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import ops, _lib
DEV = "cuda"
lib = _lib.load()
lib.dosx_debug_read_stamps.argtypes = [C.c_void_p]
for name, M, N, K, pro in [("fc1 wgrad (ROWLN)", 6528, 512, 128, ops.PRO_ROWLN), ("fc2 wgrad", 6528, 128, 512, 0), ("edge W1", 9344, 256, 384, 0),
                            ("saturated (512 workgroups of 100 chunks)", 25728, 1024, 256, 0)]:
    dy = torch.randn(M, N, device=DEV); a = torch.randn(M, K, device=DEV)
    ns = ops.wgrad_splits(M, N, K)
    slab = torch.empty(max(ops.wgrad_scratch_floats(N, K, ns), 1), device=DEV)
    dw = torch.empty(N, K, device=DEV)
    kw = {}
    if pro == ops.PRO_ROWLN:
        kw = dict(pro=pro, pro_gamma=torch.randn(K, device=DEV), pro_beta=torch.randn(K, device=DEV), pro_stats=torch.rand(M, 2, device=DEV))
    g = ops.wgrad_desc(M, N, ops.seg(dy), [ops.seg(a)], slab, None, ns, dst=dw, **kw)
    for _ in range(5):
        ops.wgrad_grouped([g])
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (64 * 64))()
    lib.dosx_debug_read_stamps(buf)
    t0 = buf[0]
    nch = ((M + ns - 1) // ns + 31) // 32
    m = [f"[{buf[2+2*c]-t0} {buf[3+2*c]-t0}]" for c in range(min(nch, 12))]
    st = [f"[{buf[32*64+2+3*c]-t0} {buf[32*64+3+3*c]-t0} {buf[32*64+4+3*c]-t0}]" for c in range(min(nch, 12))]
    print(f"== {name} M={M} N={N} K={K} splits={ns} chunks/WG={nch}")
    print(f"   matrix wave: first barrier {buf[1]-t0} | [start, mma issued] " + " ".join(m) + f" | loop end {buf[60]-t0} | finish: tile in LDS {buf[61]-t0} stores drained {buf[62]-t0} ticket {buf[63]-t0}")
    print(f"   staging wave: start {buf[32*64]-t0} chunk0 stored {buf[32*64+1]-t0} | [top stored issued] " + " ".join(st))
