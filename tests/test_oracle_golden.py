"""Pin the oracle (oracle/dos_oracle.py) against vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import dos_oracle as O
from tests.util import batch_from, load, maxabs, sub

F32 = 2e-6
F64 = 1e-12


@pytest.mark.parametrize("name,tol", [("f32", F32), ("f64", 5e-7)])
def test_g1_mha(name, tol):
    # f64 tolerance is 5e-7 because the reference forces the softmax to fp32
    # (multihead_attention.py:69); the oracle does the same, so they agree to fp32 rounding.
    z = load("g1_mha.npz")
    q, kv = torch.from_numpy(z[f"{name}/q"]), torch.from_numpy(z[f"{name}/kv"])
    out = O.multihead_attention(q, kv, kv)
    assert maxabs(out, z[f"{name}/out"]) < tol
    if name == "f64":       # and it is NOT a pure fp64 softmax
        assert out.dtype == torch.float64


@pytest.mark.parametrize("mode", ["cross", "self"])
def test_g2_encoder(mode):
    z = load("g2_encoder.npz")
    p = {k: v.requires_grad_(v.is_floating_point()) for k, v in sub(z, "p/").items()}
    x = torch.from_numpy(z[f"{mode}/x"]).requires_grad_(True)
    if mode == "cross":
        kv = torch.from_numpy(z["cross/kv"]).requires_grad_(True)
    else:
        kv = x
    y = O.transformer_encoder({"enc." + k: v for k, v in p.items()}, "enc", x, kv, kv, 2)
    assert maxabs(y, z[f"{mode}/y"]) < F32
    (y * torch.from_numpy(z[f"{mode}/w"])).sum().backward()
    assert maxabs(x.grad, z[f"{mode}/dx"]) < 1e-5
    if mode == "cross":
        assert maxabs(kv.grad, z["cross/dkv"]) < 1e-5
    dead = set(str(s) for s in z[f"{mode}/dead"])
    for k, v in p.items():
        if k == "version":
            continue
        if k in dead:
            assert v.grad is None, k            # in/out_proj never get a gradient
        else:
            assert maxabs(v.grad, z[f"{mode}/g/{k}"]) < 2e-5, k
    assert any("in_proj_weight" in d for d in dead) and any("out_proj" in d for d in dead)


@pytest.mark.parametrize("name", ["mean", "sum"])
def test_g3_processor(name):
    z = load("g3_processor.npz")
    p = {"proc." + k: v.requires_grad_(True) for k, v in sub(z, f"{name}/p/").items()}
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    e = torch.from_numpy(z["e"]).requires_grad_(True)
    ei = torch.from_numpy(z["edge_index"])
    ox, oe = O.processor(p, "proc", x, ei, e, mean=(name == "mean"))
    assert maxabs(ox, z[f"{name}/ox"]) < F32 and maxabs(oe, z[f"{name}/oe"]) < F32
    ((ox * torch.from_numpy(z[f"{name}/wx"])).sum() + (oe * torch.from_numpy(z[f"{name}/we"])).sum()).backward()
    assert maxabs(x.grad, z[f"{name}/dx"]) < 1e-5 and maxabs(e.grad, z[f"{name}/de"]) < 1e-5
    dead = set(str(s) for s in z[f"{name}/dead"])
    assert all("node_mlp_1" in d for d in dead) and len(dead) == 7
    for k, v in p.items():
        kk = k[len("proc."):]
        if kk in dead:
            assert v.grad is None
        else:
            assert maxabs(v.grad, z[f"{name}/g/{kk}"]) < 2e-5, kk


def test_g4_edge_features():
    z = load("g4_edge_features.npz")
    out = O.edge_features_sh1(torch.from_numpy(z["edge_vec"]))
    assert maxabs(out, z["edge_attr"]) < 1e-14
    ref = z["edge_attr"]
    assert np.allclose(ref[0], [1, 0, 0, 0])              # zero-length self edge
    assert ref[6, 0] == 0 and ref[5, 0] == 0              # beyond the cutoff
    assert 0 < ref[4, 0] < 1                              # on the cosine ramp


def _check_train(z, kind, n_layers, n_t, ftol, gtol, ptol):
    g = batch_from(z)
    p0 = sub(z, "p0/")
    params = {k: v.clone() for k, v in p0.items()}
    state = {}
    fwd = O.dostransformer_phonon_forward if kind == "phonon" else O.dostransformer_forward
    with torch.no_grad():
        dg, xn, ds = fwd(params, g, n_layers, n_t)
    assert maxabs(dg, z["dos_global"]) < ftol
    assert maxabs(ds, z["dos_system"]) < ftol
    assert maxabs(xn, z["x_nodes"]) < ftol
    dead = set(str(s) for s in z["dead_params"])
    for step in (1, 2, 3):
        loss, grads = O.train_step(kind, params, state, g, n_layers, n_t, lr=1e-4, beta=1.0)
        if step == 1:
            assert abs(float(loss) - float(z["loss"])) < ftol
            for k, gr in grads.items():
                if k in dead:
                    assert gr is None, k
                else:
                    assert gr is not None, k
                    assert maxabs(gr, z["g/" + k]) < gtol, k
        if step in (1, 3):
            ref = sub(z, f"p{step}/")
            for k, v in params.items():
                assert maxabs(v, ref[k]) < ptol, (step, k)
            for k in dead:                                  # dead params: no update, no decay
                assert torch.equal(params[k], p0[k])


def test_g5_phonon_full():
    z = load("g5_phonon.npz")
    _check_train(z, "phonon", 3, 1, 1e-6, 1e-6, 1e-9)      # fp64 except the fp32 softmax


def test_g6_edos_full():
    z = load("g6_edos.npz")
    _check_train(z, "edos", 3, 2, 2e-5, 2e-4, 2e-6)


def test_g7_batch_composition():
    z = load("g7_batch_composition.npz")
    p = sub(z, "p0/")
    a = batch_from(z, "alone/b/")
    b = batch_from(z, "both/b/")
    with torch.no_grad():
        oa = O.dostransformer_phonon_forward(p, a, 3, 1)
        ob = O.dostransformer_phonon_forward(p, b, 3, 1)
    assert maxabs(oa[0], z["alone/dos_global"]) < 1e-6 and maxabs(oa[2], z["alone/dos_system"]) < 1e-6
    assert maxabs(ob[0], z["both/dos_global"]) < 1e-6 and maxabs(ob[2], z["both/dos_system"]) < 1e-6
    # the unmasked zero padding makes a crystal's output depend on its batch mates
    assert maxabs(oa[0][0], ob[0][0]) > 1e-3
    # ... and forcing the alone batch to the same Nmax reproduces the batched row exactly
    from dostransformer_amd.batch import graph_meta
    graph_meta(a, n_max=11)
    with torch.no_grad():
        oa2 = O.dostransformer_phonon_forward(p, a, 3, 1)
    assert maxabs(oa2[0][0], ob[0][0]) < 1e-6


def test_g8_graphnetworks():
    z = load("g8_graphnetwork_phonon.npz")
    p = {k: v.requires_grad_(True) for k, v in sub(z, "p0/").items()}
    dos = O.graphnetwork_phonon_forward(p, batch_from(z), 3)
    assert maxabs(dos, z["dos"]) < 1e-10
    (dos * torch.from_numpy(z["w"])).sum().backward()
    dead = set(str(s) for s in z["dead_params"])
    for k, v in p.items():
        if k in dead:
            assert v.grad is None
        else:
            assert maxabs(v.grad, z["g/" + k]) < 1e-9, k
    z = load("g8_graphnetwork_edos.npz")
    p = {k: v.requires_grad_(True) for k, v in sub(z, "p0/").items()}
    dos, xn = O.graphnetwork_forward(p, batch_from(z), 3)
    assert maxabs(dos, z["dos"]) < 2e-5 and maxabs(xn, z["x_nodes"]) < 2e-5
    (dos * torch.from_numpy(z["w"])).sum().backward()
    dead = set(str(s) for s in z["dead_params"])
    for k, v in p.items():
        if k in dead:
            assert v.grad is None
        else:
            assert maxabs(v.grad, z["g/" + k]) < 5e-4, k


def test_g9_eval_loops():
    """utils.py:61-143 (`test`, `test_phonon`, `r2`) run by the reference itself over two batches each."""
    z = load("g9_eval.npz")
    loader = [batch_from(z, "ph/b0/"), batch_from(z, "ph/b1/")]
    m = O.eval_phonon(sub(z, "ph/p0/"), loader, 3, 1)
    assert np.allclose(m, z["ph/metrics"], rtol=1e-9, atol=1e-11)
    loader = [batch_from(z, "e/b0/"), batch_from(z, "e/b1/")]
    m, (ids, preds, y, emb) = O.eval_edos(sub(z, "e/p0/"), loader, 3, 2)
    assert np.allclose(m, z["e/metrics"], rtol=2e-5, atol=1e-6)
    assert ids == [str(s) for s in z["e/mp_id"]]
    assert maxabs(preds, z["e/preds"]) < 1e-5 and maxabs(y, z["e/y"]) == 0.0 and maxabs(emb, z["e/embeddings"]) < 1e-4


# ---- §8f-3: the brute-force periodic neighbour list is pinned by crystallographic known answers --------------------
def _nl(pos, cell, rc, si=False):
    from oracle.dos_oracle import neighbor_list_bruteforce
    return neighbor_list_bruteforce(np.asarray(pos, float), np.asarray(cell, float), rc, si)


@pytest.mark.parametrize("name,pos,shells", [
    ("sc", [[0, 0, 0]], [(1.01, 6), (1.42, 18), (1.74, 26), (2.01, 32)]),
    ("fcc", [[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0]], [(0.71, 12), (1.01, 18), (1.23, 42)]),
    ("bcc", [[0, 0, 0], [.5, .5, .5]], [(0.87, 8), (1.01, 14), (1.42, 26)]),
])
def test_neighbor_oracle_coordination_shells(name, pos, shells):
    for rc, expect in shells:
        i, j, S, D = _nl(pos, np.eye(3), rc)
        assert (np.bincount(i, minlength=len(pos)) == expect).all(), (name, rc)
        assert (np.linalg.norm(D, axis=1) < rc).all() and (np.linalg.norm(D, axis=1) > 0).all()


def test_neighbor_oracle_hcp_and_symmetries():
    a, c = 1.0, np.sqrt(8.0 / 3.0)                       # ideal hcp: 12 nearest neighbours at distance a
    cell = np.array([[a, 0, 0], [-a / 2, a * np.sqrt(3) / 2, 0], [0, 0, c]])
    frac = np.array([[1 / 3, 2 / 3, 0.25], [2 / 3, 1 / 3, 0.75]])
    pos = frac @ cell
    i, j, S, D = _nl(pos, cell, 1.01)
    assert (np.bincount(i) == 12).all()
    # self_interaction adds exactly the n zero-length (i, i, 0) pairs
    i2, j2, S2, D2 = _nl(pos, cell, 1.01, si=True)
    assert len(i2) == len(i) + 2 and ((np.abs(D2).sum(1) == 0).sum() == 2)
    # (i, j, S) is an edge  <=>  (j, i, -S) is one
    fwd = set(zip(i.tolist(), j.tolist(), map(tuple, S.tolist())))
    assert fwd == {(b, a_, tuple(-np.array(s))) for a_, b, s in fwd}
    # translating an atom by a lattice vector changes shifts, not the set of difference vectors
    pos_u = pos.copy(); pos_u[1] += 2 * cell[0] - 3 * cell[2]
    i3, j3, S3, D3 = _nl(pos_u, cell, 1.01)
    key = lambda ii, jj, DD: sorted((int(p), int(q), *np.round(d, 9)) for p, q, d in zip(ii, jj, DD))
    assert key(i, j, D) == key(i3, j3, D3)
