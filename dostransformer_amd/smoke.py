"""Tiny end-to-end check used by ``__graft_entry__.smoke()``: one phonon training step on the GPU
(forward, loss, backward, AdamW — all libdosx) compared with the oracle on identical inputs."""
from __future__ import annotations

import os
import sys

import torch


def run(device: str = "cuda:0") -> None:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import dos_oracle as O           # the checker (allowed in smoke(), never in the product path)
    from dostransformer_amd import synth
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    from dostransformer_amd.train import Trainer

    torch.cuda.set_device(device)
    torch.manual_seed(0)
    model = DOSTransformer_phonon(3, 1, 118, 4, 64, device, 0.0).to(device)
    ref_params = {k: v.detach().cpu().double().clone() for k, v in model.state_dict().items()}
    g64 = synth.phonon_batch(8, seed=1, dtype=torch.float64)
    g = synth.phonon_batch(8, seed=1, dtype=torch.float32).to(device)
    tr = Trainer(model, lr=1e-4, beta=1.0)
    loss = float(tr.step(g))
    dg, _, ds = tr.last_outputs
    state = {}
    with torch.no_grad():
        rg, _, rs = O.dostransformer_phonon_forward(ref_params, g64, 3, 1)
    ref_loss, _ = O.train_step("phonon", ref_params, state, g64, 3, 1, lr=1e-4, beta=1.0)
    rmse = float(torch.sqrt(torch.mean((dg.cpu().double() - rg) ** 2)))
    rmse_s = float(torch.sqrt(torch.mean((ds.cpu().double() - rs) ** 2)))
    dpar = max(float((model.state_dict()[k].cpu().double() - v).abs().max()) for k, v in ref_params.items()
               if v.is_floating_point())
    print(f"smoke: loss {loss:.6f} (oracle {float(ref_loss):.6f}); DOS rmse vs oracle {rmse:.2e}/{rmse_s:.2e}; "
          f"max |param - oracle| after AdamW {dpar:.2e}")
    assert rmse < 1e-4 and rmse_s < 1e-4, "DOS vector deviates from the oracle by more than 1e-4 RMSE"
    assert abs(loss - float(ref_loss)) < 1e-4
    assert dpar < 1e-5
