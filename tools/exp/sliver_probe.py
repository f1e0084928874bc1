#!/usr/bin/env python3
"""Does a tiny-footprint workgroup get a CU slot under a weight-gradient group?  (tools/exp/sliver/sliver.hip)
For each probe kernel: time alone, and time when launched on a second stream right behind an Electron-DOS-sized
weight-gradient group (4 jobs, ~1000 workgroups of ~70 us) on the first stream."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dostransformer_amd import ops  # noqa: E402

DEV = "cuda"
lib = C.CDLL(os.path.join(ROOT, "tools", "exp", "sliver", "libsliver.so"))
for f in (lib.sliver_launch, lib.fat_launch, lib.sliver_prio_launch, lib.sliver_valu_launch, lib.sliver_valu_prio_launch):
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    f.restype = C.c_int
keep = []


def job(M, N, K):
    dy, a = torch.randn(M, N, device=DEV), torch.randn(M, K, device=DEV)
    ns = ops.wgrad_splits(M, N, K)
    slab = torch.empty(max(ops.wgrad_scratch_floats(N, K, ns), 1), device=DEV)
    sb = torch.empty(ns * ((N + 63) // 64) * 64, device=DEV)
    dw, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    keep.extend([dy, a, slab, sb, dw, db])
    return ops.wgrad_desc(M, N, ops.seg(dy), [ops.seg(a)], slab, sb, ns, dst=dw, dst_bias=db)


def main():
    M = 25728
    group = [job(M, 1024, 256), job(M, 256, 1024), job(M, 1024, 256), job(M, 256, 1024)]
    buf = torch.empty(1 << 22, device=DEV)
    A, B = torch.cuda.Stream(), torch.cuda.Stream()
    # a real light chain kernel for comparison: the node-MLP dgrad of the Electron-DOS step
    xa, w, out = torch.randn(1554, 512, device=DEV), torch.randn(512, 512, device=DEV), torch.empty(1554, 512, device=DEV)
    probes = {
        "sliver  MFMA, s_setprio 3,           256 WGs, short": lambda s: lib.sliver_prio_launch(buf.data_ptr(), 256, 30, 4096, s),
        "sliver  vector ALU only,             256 WGs, short": lambda s: lib.sliver_valu_launch(buf.data_ptr(), 256, 120, 4096, s),
        "sliver  vector ALU only, s_setprio 3, 256 WGs, short": lambda s: lib.sliver_valu_prio_launch(buf.data_ptr(), 256, 120, 4096, s),
        "sliver  4 waves  52 VGPR   4 KB LDS, 256 WGs, short": lambda s: lib.sliver_launch(buf.data_ptr(), 256, 30, 4096, s),
        "sliver  4 waves  52 VGPR   4 KB LDS,  64 WGs, short": lambda s: lib.sliver_launch(buf.data_ptr(), 64, 30, 4096, s),
        "sliver  4 waves  52 VGPR  24 KB LDS, 256 WGs, short": lambda s: lib.sliver_launch(buf.data_ptr(), 256, 30, 24576, s),
        "fat     8 waves 150 VGPR  96 KB LDS, 256 WGs, short": lambda s: lib.fat_launch(buf.data_ptr(), 256, 5, 96 * 1024, s),
        "fat     8 waves 150 VGPR   8 KB LDS,  64 WGs, short": lambda s: lib.fat_launch(buf.data_ptr(), 64, 5, 8192, s),
        "sliver  4 waves  52 VGPR   4 KB LDS, 256 WGs": lambda s: lib.sliver_launch(buf.data_ptr(), 256, 400, 4096, s),
        "sliver  4 waves  52 VGPR   8 KB LDS, 512 WGs": lambda s: lib.sliver_launch(buf.data_ptr(), 512, 200, 8192, s),
        "sliver  4 waves  52 VGPR  24 KB LDS, 256 WGs": lambda s: lib.sliver_launch(buf.data_ptr(), 256, 400, 24576, s),
        "fat     8 waves 150 VGPR  96 KB LDS, 256 WGs": lambda s: lib.fat_launch(buf.data_ptr(), 256, 70, 96 * 1024, s),
        "fat     8 waves 150 VGPR   8 KB LDS, 256 WGs": lambda s: lib.fat_launch(buf.data_ptr(), 256, 70, 8192, s),
    }

    lib.sliver_gemm_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.sliver_gemm_launch.restype = C.c_int
    shapes = {"head dgrad M 12864 N 256 K 256": (12864, 256, 256), "node-MLP dgrad M 1554 N 512 K 512": (1554, 512, 512),
              "cfg2 dgrad M 6528 N 128 K 128": (6528, 128, 128)}
    for label, (Mm, Nn, Kk) in shapes.items():
        a_, w_ = torch.randn(Mm, Kk, device=DEV), torch.randn(Kk, Nn, device=DEV)
        c_, c2_ = torch.empty(Mm, Nn, device=DEV), torch.empty(Mm, Nn, device=DEV)
        keep.extend([a_, w_, c_, c2_])
        rc = lib.sliver_gemm_launch(a_.data_ptr(), Kk, w_.data_ptr(), Nn, None, 0, c_.data_ptr(), Nn, Mm, Nn, Kk, 1,
                                    torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ref = a_.double() @ w_.double()
        print(f"sliver GEMM {label}: rc {rc}, max rel err {float((c_.double() - ref).abs().max() / ref.abs().max()):.2e}")
        fl = 2.0 * Mm * Nn * Kk
        for prio in (0, 1):
            probes[f"VALU sliver GEMM {label} prio {3 * prio} [{fl / 1e9:.2f} GF]"] = \
                (lambda s, a_=a_, w_=w_, c_=c_, Mm=Mm, Nn=Nn, Kk=Kk, prio=prio:
                 lib.sliver_gemm_launch(a_.data_ptr(), Kk, w_.data_ptr(), Nn, None, 0, c_.data_ptr(), Nn, Mm, Nn, Kk, prio, s))

        def mfma(s, a_=a_, w_=w_, c2_=c2_, Mm=Mm, Nn=Nn):
            with torch.cuda.stream(torch.cuda.ExternalStream(s)):
                ops.gemm(Mm, Nn, [ops.seg(a_)], w_, c2_, w_layout=1)
            return 0
        probes[f"MFMA  dosx_gemm  {label}"] = mfma

    def gemm_probe(s):
        with torch.cuda.stream(torch.cuda.ExternalStream(s)):
            ops.gemm(1554, 512, [ops.seg(xa)], w, out, w_layout=1)
        return 0
    probes["real    node-MLP dgrad GEMM M 1554 N 512 K 512"] = gemm_probe

    def group_us():
        with torch.cuda.stream(A):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.wgrad_grouped(group)
            e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3
    for _ in range(3):
        g_alone = group_us()
    print(f"weight-gradient group alone: {g_alone:.1f} us")
    big = group
    for label, grp, strm in (("4 jobs (>= 1024 workgroups: a backlog)", big, B), ("2 jobs (512 workgroups: no backlog)", big[:2], B)):
        group = grp
        print(f"--- group: {label}: alone {group_us():.1f} us")
        run_probes(probes, group, A, strm)


def run_probes(probes, group, A, B):
    for name, fn in probes.items():
        if not ("GEMM" in name or "gemm" in name):
            continue
        def timed(under):
            res = []
            for _ in range(8):
                torch.cuda.synchronize()
                if under:
                    with torch.cuda.stream(A):
                        ops.wgrad_grouped(group)
                        evg = torch.cuda.Event(enable_timing=True)
                    with torch.cuda.stream(B):
                        torch.cuda._sleep(60000)          # ~25 us: the group's first round is resident when the probe arrives
                with torch.cuda.stream(B):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    rc = fn(B.cuda_stream)
                    assert rc == 0, rc
                    e.record()
                torch.cuda.synchronize()
                res.append(s.elapsed_time(e) * 1e3)
            res.sort()
            return res[len(res) // 2]
        t0, t1 = timed(False), timed(True)
        print(f"{name}: alone {t0:7.1f} us | behind the group {t1:7.1f} us  (x{t1 / t0:.1f})")


if __name__ == "__main__":
    main()
