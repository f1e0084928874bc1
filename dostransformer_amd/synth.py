"""Synthetic crystal graphs with the reference's input schema.

No dataset ships with the reference (its data needs Materials-Project downloads:
`data/mat2graph.py:238-259`, `utils.py:152-176`), so throughput and parity runs use
these seeded distributions.  Shapes/dtypes follow `utils.py:249-303` (phonon) and
`data/mat2graph.py:81-143,155-158,207-232` (eDOS).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from .batch import CrystalBatch, collate

PH_BINS = 51      # DOSTransformer_phonon.py:19
E_BINS = 201      # DOSTransformer.py:17
PH_ATOM_FEATS = 118   # main_phDOS.py:60
PH_BOND_FEATS = 4     # main_phDOS.py:61
E_ATOM_FEATS = 200    # main_eDOS.py:60 (matscholar embedding width, mat2graph.py:155)
E_BOND_FEATS = 41     # mat2graph.py:167-179,215 : arange(0, 8.2, 0.2)
E_GLOB_FEATS = 2      # mat2graph.py:86


def phonon_crystal(gen: torch.Generator, n_atoms: Optional[int] = None, n_out: int = 20,
                   dtype=torch.float64) -> Dict[str, object]:
    """One synthetic phonon crystal: one-hot(Z)*mass nodes (`utils.py:259-260,293`),
    ``n_out`` out-edges per atom whose first is the zero-length self edge that
    ``neighbor_list(..., self_interaction=True)`` produces (`utils.py:267`)."""
    if n_atoms is None:
        n_atoms = int(torch.randint(2, 13, (1,), generator=gen))
    z = torch.randint(0, PH_ATOM_FEATS, (n_atoms,), generator=gen)
    mass = torch.rand(n_atoms, generator=gen, dtype=torch.float64) * 199.0 + 1.0
    x = torch.zeros(n_atoms, PH_ATOM_FEATS, dtype=torch.float64)
    x[torch.arange(n_atoms), z] = mass
    src = torch.arange(n_atoms).repeat_interleave(n_out)
    dst = torch.randint(0, n_atoms, (n_atoms * n_out,), generator=gen)
    vec = (torch.rand(n_atoms * n_out, 3, generator=gen, dtype=torch.float64) * 2 - 1) * (4.0 / math.sqrt(3.0))
    self_pos = torch.arange(n_atoms) * n_out
    dst[self_pos] = torch.arange(n_atoms)
    vec[self_pos] = 0.0
    return {
        "x": x.to(dtype),
        "edge_index": torch.stack([src, dst], 0),
        "edge_vec": vec.to(dtype),
        "system": torch.randint(0, 7, (1,), generator=gen)[0],
        "phdos": torch.rand(1, PH_BINS, generator=gen, dtype=torch.float64).to(dtype),
    }


def edos_crystal(gen: torch.Generator, n_atoms: Optional[int] = None, n_out: int = 12,
                 dtype=torch.float32, idx: int = 0) -> Dict[str, object]:
    """One synthetic eDOS crystal: ``n_atoms`` real atoms + the all-zero phantom
    node appended by `mat2graph.py:155-158` (isolated: `get_bond_info` only emits
    edges between real atoms, `:207-232`); Gaussian-expanded distances as edge
    features (`:167-179`)."""
    if n_atoms is None:
        n_atoms = int(torch.randint(2, 41, (1,), generator=gen))
    x = torch.randn(n_atoms + 1, E_ATOM_FEATS, generator=gen, dtype=torch.float64)
    x[n_atoms] = 0.0
    src = torch.arange(n_atoms).repeat_interleave(n_out)
    dst = torch.randint(0, n_atoms, (n_atoms * n_out,), generator=gen)
    d = torch.rand(n_atoms * n_out, generator=gen, dtype=torch.float64) * 7.0 + 1.0
    mu = torch.arange(E_BOND_FEATS, dtype=torch.float64) * 0.2
    edge_attr = torch.exp(-((d[:, None] - mu[None, :]) ** 2) / 0.2 ** 2)
    y_ft = torch.rand(E_BINS, generator=gen, dtype=torch.float64)
    # a few negative targets so the harness' clamp-at-zero (main_eDOS.py:111-112) is exercised
    y_ft[torch.randint(0, E_BINS, (3,), generator=gen)] *= -1.0
    return {
        "x": x.to(dtype),
        "edge_index": torch.stack([src, dst], 0),
        "edge_attr": edge_attr.to(dtype),
        "glob": torch.randn(E_GLOB_FEATS, generator=gen, dtype=torch.float64).to(dtype),
        "system": torch.randint(0, 7, (1,), generator=gen)[0],
        "y_ft": y_ft.to(dtype),
        "mp_id": f"synth-{idx}",
    }


def phonon_batch(batch_size: int, seed: int, dtype=torch.float64, sort_edges: bool = True,
                 n_atoms: Optional[List[int]] = None) -> CrystalBatch:
    gen = torch.Generator().manual_seed(seed)
    cs = [phonon_crystal(gen, None if n_atoms is None else n_atoms[i], dtype=dtype) for i in range(batch_size)]
    return collate(cs, sort_edges=sort_edges)


def edos_batch(batch_size: int, seed: int, dtype=torch.float32, sort_edges: bool = True,
               n_atoms: Optional[List[int]] = None) -> CrystalBatch:
    gen = torch.Generator().manual_seed(seed)
    cs = [edos_crystal(gen, None if n_atoms is None else n_atoms[i], dtype=dtype, idx=i) for i in range(batch_size)]
    return collate(cs, sort_edges=sort_edges)


def phonon_crystals(batch_size: int, seed: int, dtype=torch.float64):
    gen = torch.Generator().manual_seed(seed)
    return [phonon_crystal(gen, dtype=dtype) for _ in range(batch_size)]


def edos_crystals(batch_size: int, seed: int, dtype=torch.float32):
    gen = torch.Generator().manual_seed(seed)
    return [edos_crystal(gen, dtype=dtype, idx=i) for i in range(batch_size)]


def phonon_structures(count: int, seed: int):
    """Synthetic *structures* (what `utils.py:178-196` load_data yields per row, before build_data): random triclinic
    cells of 2-12 atoms with a smooth random target DOS.  Input of ``featurize.build_data_all``."""
    import numpy as np
    from .featurize import CRYSTAL_SYSTEMS, SYMBOLS
    rng = np.random.default_rng(seed)
    systems = list(CRYSTAL_SYSTEMS) + ["Triclinic"]
    out = []
    for k in range(count):
        n = int(rng.integers(2, 13))
        cell = np.diag(rng.uniform(3.0, 6.5, 3)) + rng.uniform(-0.8, 0.8, (3, 3))
        grid = np.linspace(0.0, 1.0, PH_BINS)
        dos = sum(a * np.exp(-((grid - c) / w) ** 2) for a, c, w in
                  zip(rng.uniform(0.2, 1.0, 3), rng.uniform(0.1, 0.9, 3), rng.uniform(0.05, 0.2, 3)))
        out.append({"symbols": [SYMBOLS[z] for z in rng.integers(0, 83, n)], "positions": rng.uniform(0, 1, (n, 3)) @ cell,
                    "cell": cell, "phdos": dos / dos.max(), "crystal_system": systems[int(rng.integers(0, 7))],
                    "mp_id": f"synth-{k}"})
    return out
