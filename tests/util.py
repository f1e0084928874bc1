"""Shared helpers for the tests: golden-fixture loading."""
import os

import numpy as np
import torch

from dostransformer_amd.batch import CrystalBatch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def sub(z, prefix):
    """All entries of an npz under ``prefix`` as torch tensors (prefix stripped)."""
    out = {}
    for k in z.files:
        if k.startswith(prefix):
            v = z[k]
            if v.dtype.kind in "fiub":
                out[k[len(prefix):]] = torch.from_numpy(np.array(v))
    return out


def batch_from(z, prefix="b/"):
    f = sub(z, prefix)
    nb = int(f.pop("num_graphs"))
    g = CrystalBatch(f, nb)
    if prefix + "mp_id" in z.files:
        g.mp_id = [str(s) for s in z[prefix + "mp_id"]]
    return g


def rmse(a, b):
    a = torch.as_tensor(a).detach().to(torch.float64)
    b = torch.as_tensor(b).detach().to(torch.float64)
    return float(torch.sqrt(torch.mean((a - b) ** 2)))


def maxabs(a, b):
    a = torch.as_tensor(a).detach().to(torch.float64)
    b = torch.as_tensor(b).detach().to(torch.float64)
    return float((a - b).abs().max()) if a.numel() else 0.0
