"""Evaluation loops of the reference drivers (`utils.py:61-143`: ``test``, ``test_phonon``, ``r2``) on the
libdosx-backed modules.  Same signatures and return values as upstream, so
``main_eDOS.py:135-139`` / ``main_phDOS.py:124-128`` can import them from here unchanged; the metrics stay on
the device until the end of a batch (upstream round-trips every batch through sklearn on the host).
"""
from __future__ import annotations

from typing import Callable, Iterable, Optional

import torch


def r2(x1: torch.Tensor, x2: torch.Tensor) -> float:
    """`utils.py:20-23`: ``r2_score(x1.flatten(), x2.flatten(), multioutput='variance_weighted')`` — on the
    flattened arrays this is 1 - SS_res / SS_tot with ``x1`` the target."""
    t, p = x1.double().flatten(), x2.double().flatten()
    return float(1.0 - ((t - p) ** 2).sum() / ((t - t.mean()) ** 2).sum())


def _pool_sum(x: torch.Tensor, g, num_graphs: int) -> torch.Tensor:
    """scatter_sum(x, batch) of `utils.py:96-99` (the per-crystal embedding the reference stores): dosx_graph_pool on the batch's
    graph_ptr - the kernel the decoder's own sum-pooling uses (`DOSTransformer.py:151-161`)."""
    from . import ops
    from .batch import graph_meta
    m = graph_meta(g, x.device)
    xf = x.detach().to(torch.float32).contiguous()
    out = torch.empty(num_graphs, xf.shape[1], dtype=torch.float32, device=x.device)
    ops.graph_pool(xf, m.graph_ptr, out.data_ptr(), xf.shape[1], num_graphs, xf.shape[1])
    return out.to(x.dtype)


def test_phonon(model, data_loader: Iterable, criterion: Optional[Callable] = None, r2: Callable = r2, device=None):
    """`utils.py:117-143`.  Returns (rmse, mse, mae, r2), each the mean over batches of the per-batch value."""
    criterion = criterion if criterion is not None else torch.nn.L1Loss()
    model.eval()
    n = 0
    rmse = mse = mae = r2s = 0.0
    with torch.no_grad():
        for batch in data_loader:
            if device is not None:
                batch.to(device)
            preds_global, _, preds_system = model(batch)
            y = batch.phdos.reshape(preds_global.shape[0], -1).to(preds_system.dtype)
            mse_sys = ((y - preds_system) ** 2).mean(dim=1)
            rmse = rmse + torch.sqrt(mse_sys).mean()
            mse = mse + mse_sys.mean()
            mae = mae + criterion(preds_system, y)
            r2s += r2(y, preds_system)
            n += 1
    return float(rmse) / n, float(mse) / n, float(mae) / n, r2s / n


def test(model, data_loader: Iterable, criterion: Optional[Callable] = None, r2: Callable = r2, device=None):
    """`utils.py:61-112` (eDOS).  Target and prediction are clamped at 0 (`:76-78`).  Returns
    (rmse, mse, mae, r2, [[mp_id, preds, y, embeddings]]) with numpy arrays in the last item like upstream."""
    criterion = criterion if criterion is not None else torch.nn.L1Loss()
    model.eval()
    n = 0
    rmse = mse = mae = r2s = 0.0
    ids, preds, ys, embs = [], [], [], []
    with torch.no_grad():
        for batch in data_loader:
            if device is not None:
                batch.to(device)
            _, embeddings, preds_system = model(batch)
            nb = len(batch.mp_id)
            y = torch.clamp(batch.y_ft, min=0.0).reshape(nb, -1).to(preds_system.dtype)
            preds_system = torch.clamp(preds_system, min=0.0)
            mse_sys = ((y - preds_system) ** 2).mean(dim=1)
            rmse = rmse + torch.sqrt(mse_sys).mean()
            mse = mse + mse_sys.mean()
            mae = mae + criterion(preds_system, y)
            r2s += r2(y, preds_system)
            ids += list(batch.mp_id)
            preds.append(preds_system)
            ys.append(y)
            embs.append(_pool_sum(embeddings, batch, nb))
            n += 1
    preds_y = [[ids, torch.cat(preds).cpu().numpy(), torch.cat(ys).cpu().numpy(), torch.cat(embs).cpu().numpy()]]
    return float(rmse) / n, float(mse) / n, float(mae) / n, r2s / n, preds_y
