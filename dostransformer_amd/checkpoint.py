"""Checkpoint I/O (the reference has none).  The model part is a plain ``state_dict`` in the
reference's key layout, so weights move both ways between this build and upstream-trained
models; the optimizer part is ``Trainer.state_dict()`` (AdamW moments keyed by parameter name)."""
from __future__ import annotations

from typing import Optional

import torch


def _plain(v, where: str):
    """`extra` as something ``weights_only=True`` can read back: python scalars / str / None, tensors, and lists / tuples /
    dicts of those; numpy scalars and arrays are converted, anything else is refused HERE rather than at load time."""
    import numpy as np
    if v is None or isinstance(v, (bool, int, float, str, torch.Tensor)):
        return v
    if isinstance(v, np.generic):
        return v.item()
    if isinstance(v, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(v))
    if isinstance(v, (list, tuple)):
        return type(v)(_plain(x, where) for x in v)
    if isinstance(v, dict):
        return {str(k): _plain(x, f"{where}.{k}") for k, x in v.items()}
    raise TypeError(f"checkpoint extra {where}: {type(v).__name__} cannot be stored (use numbers, str, tensors, lists, dicts)")


def save(path: str, model: torch.nn.Module, trainer=None, extra: Optional[dict] = None) -> None:
    extra = _plain(extra or {}, "extra")
    ckpt = {"format": "dostransformer_amd/1",
            "model": {k: v.detach().cpu() for k, v in model.state_dict().items()},
            "optimizer": trainer.state_dict() if trainer is not None else None,
            "extra": extra or {}}
    torch.save(ckpt, path)


def load(path: str, model: torch.nn.Module, trainer=None, strict: bool = True, trust_pickle: bool = False) -> dict:
    """Loads a checkpoint written by :func:`save`, or a bare reference ``state_dict`` file.

    The format holds only tensors, str, int, float, tuple and dict, so files are read with ``weights_only=True``
    (no arbitrary unpickling: upstream-trained ``state_dict`` files are third-party input).  ``trust_pickle=True`` is
    the explicit opt-in for a legacy file that really needs the full unpickler."""
    import pickle
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=not trust_pickle)
    except pickle.UnpicklingError as e:
        raise pickle.UnpicklingError(f"{path}: not readable with weights_only=True ({e}); if the file is trusted (e.g. a "
                                     f"checkpoint with arbitrary objects in `extra`), load it with trust_pickle=True") from e
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    model.load_state_dict(sd, strict=strict)
    if trainer is not None and isinstance(ckpt, dict) and ckpt.get("optimizer") is not None:
        trainer.load_state_dict(ckpt["optimizer"])
    return ckpt.get("extra", {}) if isinstance(ckpt, dict) else {}
