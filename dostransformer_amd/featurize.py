"""Phonon input featurisation on the GPU (counterpart of `utils.py:249-303` build_data).

The reference builds one PyG ``Data`` per crystal on the host: ASE's periodic ``neighbor_list("ijS", cutoff=r_max,
self_interaction=True)`` (`utils.py:267`), ``edge_vec = pos[dst] - pos[src] + shift @ lattice`` (`:271-273`), node
features ``x = diag(atomic mass)[Z-1]`` (`:259-260,293`), ``z = one_hot(Z-1)``, the crystal-system code
(`:277-290`) and the target ``phdos``.  Here the neighbour search of the WHOLE dataset is one libdosx call pair
(``ops.neighbor_list``: one GPU thread per ordered atom pair, exact minimal image boxes, no ASE); the rest is index
arithmetic.  ``build_data_all`` returns host crystal dicts in the schema ``batch.collate`` / ``loader.DeviceDataset``
consume, so a dataset goes  structures -> build_data_all -> DeviceDataset -> Trainer  without ASE or PyG.

Entries may be the reference's pandas rows (``entry.structure`` an ASE ``Atoms``: ``.symbols``, ``.positions``,
``.cell.array``; ``entry.phdos``, ``entry.crystal_system``, ``entry.mp_id``) or plain dicts with the keys
``symbols, positions, cell, phdos, crystal_system, mp_id``.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import ops

SYMBOLS = ("H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y Zr "
           "Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir Pt "
           "Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr Rf Db Sg Bh Hs Mt Ds Rg Cn Nh Fl Mc "
           "Lv Ts Og").split()
# Standard atomic weights, IUPAC 2016 (the table ASE ships as ase.data.atomic_masses and `Atom(Z).mass` returns,
# `utils.py:254-257`); ASE is not available here to pin them — pass ``masses=`` to override.
ATOMIC_MASSES = (
    1.008, 4.002602, 6.94, 9.0121831, 10.81, 12.011, 14.007, 15.999, 18.998403163, 20.1797, 22.98976928, 24.305,
    26.9815385, 28.085, 30.973761998, 32.06, 35.45, 39.948, 39.0983, 40.078, 44.955908, 47.867, 50.9415, 51.9961,
    54.938044, 55.845, 58.933194, 58.6934, 63.546, 65.38, 69.723, 72.630, 74.921595, 78.971, 79.904, 83.798, 85.4678,
    87.62, 88.90584, 91.224, 92.90637, 95.95, 97.90721, 101.07, 102.90550, 106.42, 107.8682, 112.414, 114.818, 118.710,
    121.760, 127.60, 126.90447, 131.293, 132.90545196, 137.327, 138.90547, 140.116, 140.90766, 144.242, 144.91276,
    150.36, 151.964, 157.25, 158.92535, 162.500, 164.93033, 167.259, 168.93422, 173.054, 174.9668, 178.49, 180.94788,
    183.84, 186.207, 190.23, 192.217, 195.084, 196.966569, 200.592, 204.38, 207.2, 208.98040, 208.98243, 209.98715,
    222.01758, 223.01974, 226.02541, 227.02775, 232.0377, 231.03588, 238.02891, 237.04817, 244.06421, 243.06138,
    247.07035, 247.07031, 251.07959, 252.0830, 257.09511, 258.09843, 259.1010, 262.110, 267.122, 268.126, 271.134,
    270.133, 269.1338, 278.156, 281.165, 281.166, 285.177, 286.182, 289.190, 289.194, 293.204, 293.208, 294.214)
# `utils.py:277-290`
CRYSTAL_SYSTEMS = {"Cubic": 0, "Hexagonal": 1, "Tetragonal": 2, "Trigonal": 3, "Orthorhombic": 4, "Monoclinic": 5}
_Z_OF = {s: i for i, s in enumerate(SYMBOLS)}


def _get(entry, key):
    if isinstance(entry, dict):
        return entry[key]
    st = getattr(entry, "structure", None)
    if key == "symbols":
        return list(st.symbols)
    if key == "positions":
        return np.asarray(st.positions)
    if key == "cell":
        return np.asarray(getattr(st.cell, "array", st.cell))
    return getattr(entry, key)


def build_data_all(entries: Sequence, r_max: float = 5.0, device="cuda:0", masses: Optional[Sequence[float]] = None,
                   dtype: torch.dtype = torch.float64) -> List[Dict[str, object]]:
    """`utils.py:249-303` for a whole dataset at once.  Returns one dict per entry with the reference's ``Data``
    fields: pos, lattice, symbol, x, z, edge_index (row 0 = central atom, row 1 = neighbour), edge_shift, edge_vec,
    edge_len (rounded to 2 decimals, `:276`), phdos [1,51], system, mp_id — host tensors, ``dtype`` floating point."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("build_data_all runs the neighbour search on an MI355X through libdosx (no CPU fallback)")
    entries = list(entries)
    if not entries:
        return []
    mass = torch.tensor(ATOMIC_MASSES if masses is None else list(masses), dtype=torch.float64)
    if mass.numel() != len(SYMBOLS):
        raise ValueError(f"masses must have {len(SYMBOLS)} entries")
    sym = [list(_get(e, "symbols")) for e in entries]
    pos = [np.asarray(_get(e, "positions"), np.float64).reshape(-1, 3) for e in entries]
    cell = np.stack([np.asarray(_get(e, "cell"), np.float64).reshape(3, 3) for e in entries])
    n = np.array([p.shape[0] for p in pos], np.int64)
    for s, p in zip(sym, pos):
        if len(s) != p.shape[0] or p.shape[0] == 0:
            raise ValueError("every entry needs >= 1 atom and one symbol per position")
    atom_ptr = np.concatenate([[0], np.cumsum(n)])
    nl = ops.neighbor_list(torch.from_numpy(np.concatenate(pos)).to(dev), torch.from_numpy(cell).to(dev),
                           torch.from_numpy(atom_ptr.astype(np.int32)).to(dev), float(r_max), self_interaction=True)
    src, dst = nl["src"].cpu().long(), nl["dst"].cpu().long()
    shift, vec, eptr = nl["shift"].cpu(), nl["edge_vec"].cpu(), nl["edge_ptr"].cpu().tolist()
    eye = torch.eye(len(SYMBOLS), dtype=dtype)
    out = []
    for c, e in enumerate(entries):
        z = torch.tensor([_Z_OF[s] for s in sym[c]], dtype=torch.int64)
        a, b = eptr[c], eptr[c + 1]
        ev = vec[a:b]
        x = torch.zeros(len(z), len(SYMBOLS), dtype=dtype)
        x[torch.arange(len(z)), z] = mass[z].to(dtype)
        system = CRYSTAL_SYSTEMS.get(_get(e, "crystal_system"), 6)
        out.append({
            "pos": torch.from_numpy(pos[c]).to(dtype), "lattice": torch.from_numpy(cell[c]).to(dtype).unsqueeze(0),
            "symbol": sym[c], "x": x, "z": eye[z],
            "edge_index": torch.stack([src[a:b], dst[a:b]], 0),
            "edge_shift": shift[a:b].to(dtype), "edge_vec": ev.to(dtype),
            "edge_len": torch.from_numpy(np.around(ev.norm(dim=1).numpy(), decimals=2)),
            "phdos": torch.as_tensor(np.asarray(_get(e, "phdos"), np.float64)).reshape(1, -1).to(dtype),
            "system": torch.tensor(system), "mp_id": _get(e, "mp_id"),
        })
    return out
