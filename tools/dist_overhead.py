"""Attribute the cost of the data-parallel exchanges at world size 1 (run under torch.distributed.run)."""
import os, sys, time, torch
import torch.distributed as td
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import synth
from dostransformer_amd.batch import bucket_sizes, pad_batch
from dostransformer_amd.dist import DataParallel
from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
from dostransformer_amd.train import Trainer
dev = "cuda"
td.init_process_group("nccl", device_id=torch.device("cuda:0")) if os.environ.get("DEVID") else td.init_process_group("nccl")
torch.manual_seed(0)
model = DOSTransformer_phonon(3, 2, 118, 4, 128, dev, 0.0).to(dev)
gs = []
for k in range(8):
    g = synth.phonon_batch(64, seed=k, dtype=torch.float32)
    gs.append(pad_batch(g, *bucket_sizes(g.meta.num_nodes, g.meta.num_edges)).to(dev))

class NoGrad(DataParallel):
    def all_reduce_grads(self, flat_grad): pass
class NoSse(DataParallel):
    def all_reduce_sse(self, sse): pass
class Neither(DataParallel):
    def all_reduce_grads(self, flat_grad): pass
    def all_reduce_sse(self, sse): pass

def run(name, dp, bucketed=True):
    tr = Trainer(model, replay=True, dist=dp)
    tr.bucketed = tr.bucketed and bucketed
    for i in range(16): tr.step(gs[i % 8], 64)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 200
    for i in range(n): tr.step(gs[i % 8], 64)
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step", flush=True)
run("no dist", None)
run("dist: both all-reduces", DataParallel())
run("dist: no grad all-reduce", NoGrad())
run("dist: no SSE all-reduce", NoSse())
run("dist: split programs only", Neither())
run("dist: one bucket (no overlap)", DataParallel(), bucketed=False)
run("dist: two buckets", DataParallel())
run("no dist", None)
