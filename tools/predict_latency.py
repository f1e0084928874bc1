"""Batch-1 / small-batch inference latency: eager forward (Python-issued launches) vs predict.Predictor
(recorded forward program, dosx_replay).  SURVEY.md §8f-2.   usage: python tools/predict_latency.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dostransformer_amd import synth
from dostransformer_amd.batch import collate
from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
from dostransformer_amd.embedder_eDOS.DOSTransformer import DOSTransformer
from dostransformer_amd.predict import Predictor

dev = torch.device("cuda:0")


def lat(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


for kind in ("phonon", "edos"):
    torch.manual_seed(0)
    if kind == "phonon":
        model = DOSTransformer_phonon(3, 2, 118, 4, 128, dev, 0.0).to(dev).eval()
        cs = synth.phonon_crystals(64, seed=1, dtype=torch.float32)
    else:
        model = DOSTransformer(3, 2, 200, 41, 2, 256, dev, 0.0).to(dev).eval()
        cs = synth.edos_crystals(64, seed=1, dtype=torch.float32)
    pred = Predictor(model)
    for B in (1, 8, 64):
        g = collate(cs[:B]).to(dev)

        def eager():
            with torch.no_grad():
                model(g)
        e = lat(eager)
        r = lat(lambda: pred(g))
        slot = next(iter(pred._slots.values())) if len(pred._slots) == 1 else list(pred._slots.values())[-1]
        gp = slot.g
        ld = lat(lambda: slot.load(gp))
        rn = lat(lambda: slot.prog_a.run())
        print(f"predict {kind:6s} B={B:3d}: eager {e:8.1f} us   replay {r:8.1f} us   ({B / r * 1e6:9.0f} crystals/s)"
              f"   [load only {ld:6.1f} us, program only {rn:6.1f} us, {slot.prog_a._n} launches]", flush=True)
