#!/usr/bin/env python3
"""How much do a dgrad GEMM (main stream) and a wgrad kernel (side stream) really overlap on one MI355X?
Times n iterations of: serial on one stream | two streams, independent | two streams with the per-pair
fork (event record on main, wait on side) the training step uses."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import ops  # noqa: E402

DEV = "cuda"


def run(M, N, K, Kw, n=200):
    a = torch.randn(M, K, device=DEV); w = torch.randn(K, N, device=DEV); out = torch.empty(M, N, device=DEV)
    dy = torch.randn(M, N, device=DEV); act = torch.randn(M, Kw, device=DEV)
    ns = ops.wgrad_splits(M, N, Kw)
    slab = torch.empty(ns, N, Kw, device=DEV); sb = torch.empty(ns, N, device=DEV)
    g = lambda: ops.gemm(M, N, [ops.seg(a)], w, out, w_layout=1)
    wg = lambda: ops.wgrad(M, N, ops.seg(dy), [ops.seg(act)], slab, sb, ns)
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()

    def timed(body):
        for _ in range(10):
            body()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            body()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e6

    def only_g():
        g()

    def only_w():
        wg()

    def serial():
        g(); wg()

    def indep():
        g()
        with torch.cuda.stream(side):
            wg()

    def forked():
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            wg()
        g()

    res = {k: timed(f) for k, f in [("gemm", only_g), ("wgrad", only_w), ("serial", serial), ("indep", indep), ("forked", forked)]}
    # the same through recorded programs (host cost ~5 us per launch instead of ~15)
    def rec(body):
        ops.RECORDER.begin(); body(); p = ops.RECORDER.end(); return p
    p_serial, p_indep = rec(serial), None
    sink_side = side
    def indep_rec():
        ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
        ops.RECORDER.prog.append((ev.record, (main,))); ops.RECORDER.prog.append((side.wait_event, (ev,)))
        with torch.cuda.stream(side):
            wg()
        g()
    p_fork = rec(indep_rec)
    res["serial(replay)"] = timed(p_serial.run)
    res["forked(replay)"] = timed(p_fork.run)
    print(f"M={M} N={N} K={K} Kw={Kw}: " + "  ".join(f"{k} {v:6.1f}" for k, v in res.items()), flush=True)


run(6528, 128, 512, 512)      # ffn dgrad fc1 + wgrad fc2
run(9000, 384, 256, 384)      # edge dgrad1 + wgrad W1
run(9000, 256, 128, 256)
