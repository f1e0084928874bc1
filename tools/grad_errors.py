#!/usr/bin/env python3
"""Per-tensor gradient error of the HIP path against the fp64 oracle (diagnostic; prints the tensors sorted by error).
usage: grad_errors.py [H T B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dos_oracle as O                     # noqa: E402  (diagnostic tool, not product code)
from dostransformer_amd import synth                   # noqa: E402
from dostransformer_amd.train import Trainer           # noqa: E402
from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon  # noqa: E402

H, T, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (128, 2, 16)
torch.manual_seed(0)
model = DOSTransformer_phonon(3, T, 118, 4, H, "cuda", 0.0)
p64 = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
model = model.to("cuda")
g64 = synth.phonon_batch(B, seed=11, dtype=torch.float64)
g = synth.phonon_batch(B, seed=11, dtype=torch.float32).to("cuda")
torch.set_num_threads(8)
_, gr = O.train_step("phonon", p64, {}, g64, 3, T, lr=1e-4, beta=1.0)
tr = Trainer(model, lr=1e-4)
tr.forward_backward(g)
torch.cuda.synchronize()
fp = model.flat_params()
rows = []
for k, a in gr.items():
    if a is None:
        continue
    d = (fp.G[k].cpu().double() - a).abs()
    rows.append((float(d.max() / (a.abs().max() + 1e-12)), k, float(a.abs().max()), int((d > 1e-4 * a.abs().max()).sum()), a.numel()))
for e, k, mx, nbad, n in sorted(rows, reverse=True)[:25]:
    print(f"{e:10.3e}  {k:60s} max|g| {mx:9.3e}   elements off by > 1e-4*max: {nbad}/{n}")
k = sorted(rows, reverse=True)[0][1]
a = gr[k]
d = (fp.G[k].cpu().double() - a)
idx = torch.nonzero(d.abs() > 1e-4 * a.abs().max())
print("worst tensor", k, tuple(a.shape), "bad element indices (first 20):", idx[:20].tolist())
if a.dim() == 2:
    print("bad rows:", sorted(set(idx[:, 0].tolist()))[:40], "bad cols:", sorted(set(idx[:, 1].tolist()))[:40])
