import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import synth, ops
from dostransformer_amd.batch import bucket_sizes, pad_batch
from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
from dostransformer_amd.train import Trainer
dev = "cuda"
torch.manual_seed(0)
model = DOSTransformer_phonon(3, 2, 118, 4, 128, dev, 0.0).to(dev)
g = synth.phonon_batch(64, seed=0, dtype=torch.float32)
g = pad_batch(g, *bucket_sizes(g.meta.num_nodes, g.meta.num_edges)).to(dev)
for side in (True, False):
    ops.GradSink.use_side_stream = side
    tr = Trainer(model, replay=True)
    import dostransformer_amd.train as T
    # _record forces side stream on; patch for the experiment
    orig = ops.GradSink.use_side_stream
    tr.step(g); tr.step(g)
    slot = list(tr._slots.values())[0]
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        slot.prog_a.run()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print(f"entries {len(slot.prog_a)}: host enqueue {t_host*1e3:.3f} ms/step, with sync {t_all*1e3:.3f} ms/step")
    break
# same program, all on one stream: rebuild with side stream disabled inside _record
import types
def rec_noside(self, slot, fp, ng):
    import dostransformer_amd.ops as o
    o.RECORDER.begin()
    with torch.no_grad():
        st = self._part_a(fp, slot.g, slot.g.meta); loss = self._part_b(fp, slot.g.meta, st, ng)
    slot.prog_a = o.RECORDER.end(); slot.keep = (st, loss); slot.loss, slot.out, slot.sse = loss, st["out"], st.get("sse")
ops.GradSink.use_side_stream = False
tr2 = Trainer(model, replay=True)
tr2._record = types.MethodType(rec_noside, tr2)
tr2.step(g); tr2.step(g)
slot = list(tr2._slots.values())[0]
torch.cuda.synchronize(); n = 50; t0 = time.perf_counter()
for _ in range(n): slot.prog_a.run()
t_host = (time.perf_counter() - t0) / n
torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / n
print(f"single stream: entries {len(slot.prog_a)}: host enqueue {t_host*1e3:.3f} ms/step, with sync {t_all*1e3:.3f} ms/step")
