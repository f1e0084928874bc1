#!/usr/bin/env python3
"""Feasibility probe (not part of the product): do TWO independent half-batch training chains on two stream sets finish
sooner than ONE full-batch chain?  Two separate models/trainers with B/2 crystals each, steps issued alternately from one
host thread on two streams, against one trainer with B crystals.  Aggregate crystals/s printed for both."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as Bn                                             # noqa: E402
from dostransformer_amd import ops                             # noqa: E402
from dostransformer_amd.batch import bucket_sizes, collate, pad_batch   # noqa: E402
from dostransformer_amd.train import Trainer                   # noqa: E402

dev = torch.device("cuda:0")
kind, L, T, H, B = Bn.CONFIGS["phonon_h128_b64"]
NCH = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = 300


def mk(bsz, seed0):
    model = Bn.build_model(kind, L, T, H, dev).to(dev)
    tr = Trainer(model, lr=1e-4, beta=1.0, replay=True)
    bs = []
    for k in range(8):
        g = collate(Bn.make_crystals(kind, bsz, seed=seed0 + k, dtype=torch.float32))
        g = pad_batch(g, *bucket_sizes(g.meta.num_nodes, g.meta.num_edges, 8, 128))
        bs.append(g.to(dev))
    return tr, bs


def run(chains):
    streams = [torch.cuda.Stream(device=dev) for _ in chains]
    dicts = [dict() for _ in chains]

    def one(i, it):
        ops.GradSink._side_streams = dicts[i]
        with torch.cuda.stream(streams[i]):
            tr, bs = chains[i]
            tr.step(bs[it % 8], bs[it % 8].meta.num_graphs)
    for it in range(12):
        for i in range(len(chains)):
            one(i, it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        for i in range(len(chains)):
            one(i, it)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


one = run([mk(B, 0)])
print(f"1 chain  x {B:3d} crystals: {one * 1e3:.4f} ms per round -> {B / one:9.0f} crystals/s")
per = B // NCH
many = run([mk(per, 100 * (i + 1)) for i in range(NCH)])
print(f"{NCH} chains x {per:3d} crystals: {many * 1e3:.4f} ms per round -> {per * NCH / many:9.0f} crystals/s")
