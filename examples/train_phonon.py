#!/usr/bin/env python3
"""End-to-end Phonon-DOS run on the MI355X path: structures -> graphs (GPU neighbour list) -> device-resident
dataset -> fused training steps -> evaluation -> checkpoint.  The build's own counterpart of the reference driver
`main_phDOS.py` (whose model constructor call and `test_phonon` unpacking do not match its own modules, SURVEY.md
§3.1 — this driver uses the signatures the modules actually have).  No dataset ships with the reference, so by
default it trains on synthetic structures; pass --pickle with a list of
{symbols, positions, cell, phdos, crystal_system, mp_id} dicts to use real ones.

    python examples/train_phonon.py --epochs 5 --crystals 512
"""
import argparse
import os
import pickle
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import checkpoint, evaluate, featurize, synth  # noqa: E402
from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon  # noqa: E402
from dostransformer_amd.loader import DeviceDataset  # noqa: E402
from dostransformer_amd.predict import Predictor  # noqa: E402
from dostransformer_amd.train import Trainer  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=3)          # utils.py:29-42 defaults
    ap.add_argument("--transformer", type=int, default=2)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--beta", type=float, default=1.0)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--r-max", type=float, default=4.0)       # main_phDOS.py:21
    ap.add_argument("--crystals", type=int, default=512)
    ap.add_argument("--pickle", default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="phonon_best.pt")
    args = ap.parse_args(argv)
    dev = torch.device("cuda:0")
    torch.manual_seed(args.seed)

    entries = pickle.load(open(args.pickle, "rb")) if args.pickle else synth.phonon_structures(args.crystals, args.seed)
    t0 = time.perf_counter()
    crystals = featurize.build_data_all(entries, r_max=args.r_max, device=dev, dtype=torch.float32)
    print(f"{len(crystals)} crystals -> graphs in {time.perf_counter() - t0:.2f} s "
          f"({sum(c['edge_index'].shape[1] for c in crystals)} edges)")
    perm = np.random.default_rng(args.seed).permutation(len(crystals))
    n_val = max(1, len(perm) // 10)
    split = {"valid": perm[:n_val], "test": perm[n_val:2 * n_val], "train": perm[2 * n_val:]}
    ds = {k: DeviceDataset([crystals[i] for i in idx], dev) for k, idx in split.items()}

    model = DOSTransformer_phonon(args.layers, args.transformer, 118, 4, args.hidden, dev, 0.0).to(dev)
    # coarse shape buckets: reshuffled batches then fall into a few dozen (N, E, n_max) buckets that are all recorded
    # within the first epoch (ghost padding is exact; it costs a few per cent of extra rows)
    trainer = Trainer(model, lr=args.lr, beta=args.beta, replay=True, bucket=(32, 1024), promote=0.08)
    predictor = Predictor(model, bucket=(32, 1024))
    best, history = float("inf"), []
    for epoch in range(args.epochs):
        model.train()
        t0, losses, seen = time.perf_counter(), [], 0
        for batch in ds["train"].batches(args.batch_size, shuffle=True, seed=args.seed + epoch):
            losses.append(trainer.step(batch))
            seen += batch.num_graphs
        loss = float(torch.stack(losses).mean())                  # one host read per epoch
        dt = time.perf_counter() - t0
        history.append(loss)
        rmse, mse, mae, r2v = evaluate.test_phonon(predictor, ds["valid"].batches(args.batch_size))
        print(f"[epoch {epoch + 1}/{args.epochs}] loss {loss:.4f} | {seen / dt:8.0f} crystals/s | "
              f"valid rmse {rmse:.4f} mse {mse:.4f} mae {mae:.4f} r2 {r2v:.4f}")
        if rmse < best:
            best = rmse
            checkpoint.save(args.out, model, trainer)
            t = evaluate.test_phonon(predictor, ds["test"].batches(args.batch_size))
            print(f"            test rmse {t[0]:.4f} mse {t[1]:.4f} mae {t[2]:.4f} r2 {t[3]:.4f}   (saved {args.out})")
    return {"best_valid_rmse": best, "train_loss": history}


if __name__ == "__main__":
    main()
