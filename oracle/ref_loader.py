"""TEST INFRASTRUCTURE ONLY -- loads the *real* reference model in the build container.

This file reads ``/root/reference`` and therefore only works in the build
container; nothing on the GPU box (``-m gpu`` tests, ``smoke()``, ``bench.py``)
imports it.  It is used by ``oracle/gen_golden.py`` (to emit the committed
fixtures under ``tests/golden/``) and by ``tests/test_oracle_vs_reference.py``
(skipped when ``/root/reference`` is absent) to pin ``oracle/restatement.py``
to the reference's own arithmetic.

How the reference is loaded:
``/root/reference/mmidas/nn_model.py`` needs Python >= 3.12 only for two lines
(``assert_never`` import at :3, PEP-695 ``def avg[T]`` at :85).  Those two lines
are neutralised *in memory*; no reference source is copied into this repo.
A second, unmodified loader for ``build/lib/mmidas/nn_model.py`` (older snapshot,
identical arithmetic) is offered as a cross-check.
"""
from __future__ import annotations

import contextlib
import importlib.util
import os
import sys
import types
from typing import Any, Dict, List

REFERENCE_ROOT = "/root/reference"
_PRIMARY = os.path.join(REFERENCE_ROOT, "mmidas", "nn_model.py")
_OLD = os.path.join(REFERENCE_ROOT, "build", "lib", "mmidas", "nn_model.py")


def reference_available() -> bool:
    return os.path.isfile(_PRIMARY)


def load_reference_nn_model() -> types.ModuleType:
    """Return the reference ``mmidas.nn_model`` module (current semantics)."""
    sys.dont_write_bytecode = True
    with open(_PRIMARY, "r") as fh:
        src = fh.read()
    a = "from typing import Optional, List, Iterable, Sequence, assert_never"
    b = "def avg[T](x: Sequence[T]) -> T:"
    if a not in src or b not in src:
        raise RuntimeError("reference nn_model.py changed; loader recipe no longer applies")
    src = src.replace(a, "from typing import Optional, List, Iterable, Sequence")
    src = src.replace(b, "def avg(x):")
    mod = types.ModuleType("ref_nn_model")
    mod.__file__ = _PRIMARY
    sys.modules["ref_nn_model"] = mod  # dataclasses resolves cls.__module__ through sys.modules
    exec(compile(src, "ref_nn_model", "exec"), mod.__dict__)
    return mod


def load_reference_nn_model_old() -> types.ModuleType:
    """Cross-check oracle: the packaged older snapshot, loads unmodified on py3.10."""
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("ref_nn_model_old", _OLD)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _RecordingDropout:
    """Stands in for ``nn.Dropout`` on the reference model (``x_dp`` / ``s_dp``).

    Semantics are torch's: keep-mask ~ Bernoulli(1-p) drawn from the global
    generator, output = x * mask / (1-p); identity when not training or p == 0.
    Either records the mask it draws or replays masks handed to it.
    """

    def __init__(self, p: float, owner, replay: List[Any] | None = None):
        self.p = float(p)
        self.owner = owner
        self.replay = list(replay) if replay is not None else None
        self.record: List[Any] = []

    def __call__(self, x):
        import torch

        if (not self.owner.training) or self.p == 0.0:
            return x
        if self.replay is not None:
            mask = self.replay.pop(0).to(x.dtype)
        else:
            mask = torch.bernoulli(torch.full_like(x, 1.0 - self.p))
        self.record.append(mask.to(torch.uint8).clone())
        return x * mask / (1.0 - self.p)


@contextlib.contextmanager
def explicit_noise(model, noise: Dict[str, Any] | None = None):
    """Run reference ``forward`` with recorded or replayed noise.

    ``noise`` (replay) holds per-arm lists: ``x_mask`` [A][B,D] uint8,
    ``u_gumbel`` [A][B,C] float, ``u_state`` [A][B,S] float, ``s_mask`` [A][B,S].
    Yields a dict that, after the block, holds what was drawn in consumption
    order (per arm bernoulli[B,D] -> rand[B,1,C] ->
    rand_like[B,S] -> bernoulli[B,S] iff s_drop > 0).
    """
    import torch

    rec: Dict[str, list] = {"x_mask": [], "u_gumbel": [], "u_state": [], "s_mask": []}
    rp = noise
    x_dp_old, s_dp_old = model.x_dp, model.s_dp
    xd = _RecordingDropout(x_dp_old.p, model, None if rp is None else rp["x_mask"])
    sd = _RecordingDropout(s_dp_old.p, model, None if rp is None else rp.get("s_mask", []))
    # nn.Module.__setattr__ refuses non-Module values for registered children.
    object.__setattr__(model, "_oracle_xd", xd)
    del model._modules["x_dp"], model._modules["s_dp"]
    model.__dict__["x_dp"] = xd
    model.__dict__["s_dp"] = sd

    rand_old, rand_like_old = torch.rand, torch.rand_like
    ug = None if rp is None else list(rp["u_gumbel"])
    us = None if rp is None else list(rp["u_state"])

    def rand(*size, **kw):
        shape = size[0] if len(size) == 1 and not isinstance(size[0], int) else size
        if ug is not None:
            t = ug.pop(0).reshape(tuple(shape)).clone()
        else:
            t = rand_old(*size, **kw)
        rec["u_gumbel"].append(t.reshape(t.shape[0], -1).clone())
        return t

    def rand_like(t0, **kw):
        if us is not None:
            t = us.pop(0).reshape(t0.shape).clone()
        else:
            t = rand_like_old(t0, **kw)
        rec["u_state"].append(t.clone())
        return t

    torch.rand, torch.rand_like = rand, rand_like
    try:
        yield rec
    finally:
        torch.rand, torch.rand_like = rand_old, rand_like_old
        del model.__dict__["x_dp"], model.__dict__["s_dp"]
        model._modules["x_dp"], model._modules["s_dp"] = x_dp_old, s_dp_old
        rec["x_mask"] = xd.record
        rec["s_mask"] = sd.record


def reference_step(model, xs, temp, noise=None, eval_flag=False):
    """forward + loss exactly as ``cpl_mixvae.py:435-460`` drives them."""
    with explicit_noise(model, noise) as rec:
        out = model(xs, temp, 0.0, eval=eval_flag)
    x_recs, _, _, x_lows, cs, s_smps, c_smps, s_means, s_logvars, c_probs = out
    loss_out = model.loss(x_recs, [], [], xs, s_means, s_logvars, cs, c_smps, 0.0)
    return out, loss_out, rec


_UTILS = os.path.join(REFERENCE_ROOT, "mmidas", "_utils.py")
_CONSENSUS_FUNCS = ("classify", "compute_confmat", "confmat_normalize", "compute_confmat_naive",
                    "confmat_normalize_naive", "confmat_mean")


def load_reference_consensus_utils() -> types.SimpleNamespace:
    """The reference's own ``classify`` / ``compute_confmat`` / ``confmat_normalize`` / ``confmat_mean``
    (mmidas/_utils.py:79-129), compiled in memory from the reference file where it lies.  Only these function
    definitions are taken (the module's top level imports packages this image lacks); nothing is copied into the repo."""
    import ast
    import numpy as np
    with open(_UTILS, "r") as fh:
        tree = ast.parse(fh.read(), filename=_UTILS)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in _CONSENSUS_FUNCS]
    mod = ast.Module(body=keep, type_ignores=[])
    ns: Dict[str, Any] = {"np": np}
    exec(compile(mod, _UTILS, "exec"), ns)
    return types.SimpleNamespace(**{k: ns[k] for k in _CONSENSUS_FUNCS})


_UDAGAN = os.path.join(REFERENCE_ROOT, "mmidas", "augmentation", "udagan.py")
_AUG_UTILS = os.path.join(REFERENCE_ROOT, "mmidas", "augmentation", "aug_utils.py")


def load_reference_augmenter():
    """The reference's own ``Augmenter_smartseq`` class (mmidas/augmentation/udagan.py:217-329) and ``reparam_trick``
    (aug_utils.py:51-65), compiled in memory from the reference files where they lie (only these two definitions: the
    modules' top levels import the whole package).  Nothing is copied into the repo."""
    import ast
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    ns: Dict[str, Any] = {"torch": torch, "nn": nn, "F": F}
    for path, names, kind in ((_AUG_UTILS, ("reparam_trick",), ast.FunctionDef), (_UDAGAN, ("Augmenter_smartseq",), ast.ClassDef)):
        with open(path, "r") as fh:
            tree = ast.parse(fh.read(), filename=path)
        keep = [n for n in tree.body if isinstance(n, kind) and n.name in names]
        assert len(keep) == len(names), (path, names)
        exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns["Augmenter_smartseq"]


_TRAINER = os.path.join(REFERENCE_ROOT, "mmidas", "cpl_mixvae.py")


def load_reference_trainer(ref_nn=None) -> types.SimpleNamespace:
    """The reference's own trainer class ``cpl_mixVAE`` (mmidas/cpl_mixvae.py:150-1650: ``__init__``, ``init_model``,
    ``train``, ``eval_model``) compiled in memory from the reference file where it lies, bound to the reference model
    (``load_reference_nn_model``) and the reference's consensus helpers.  The module does not import on this image
    (py3.12-only ``def unwrap[T]`` at :104, ``torchvision`` / ``wandb`` absent): that one line is neutralised in
    memory, only the class and the module's small helper functions are compiled, and the names its top level would
    have imported are supplied here (``wandb`` is not needed: ``run=None`` or a stand-in logger).  Nothing is copied
    into the repo."""
    import ast
    import pickle
    import time as _time
    from functools import reduce

    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    import numpy as np
    import torch
    import torch.distributed as dist
    import torch.nn.functional as F
    import torch.nn.utils.prune as prune
    import torch.optim as optim
    from torch import nn
    from torch.optim.optimizer import Optimizer
    from torch.utils.data import DataLoader, TensorDataset
    from tqdm import tqdm, trange
    from typing import Any, Iterable, Literal, Mapping, Optional, Sequence

    if ref_nn is None:
        ref_nn = load_reference_nn_model()
    with open(_TRAINER, "r") as fh:
        src = fh.read()
    a = "def unwrap[T](x: Optional[T]) -> T:"
    if a not in src:
        raise RuntimeError("reference cpl_mixvae.py changed; loader recipe no longer applies")
    tree = ast.parse(src.replace(a, "def unwrap(x):"), filename=_TRAINER)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) or (isinstance(n, ast.ClassDef) and n.name == "cpl_mixVAE")]
    # helpers of mmidas/_utils.py the trainer imports (:35-41 of the trainer)
    with open(_UTILS, "r") as fh:
        utree = ast.parse(fh.read(), filename=_UTILS)
    ukeep = [n for n in utree.body if isinstance(n, ast.FunctionDef)
             and n.name in ("to_np", "classify", "compute_confmat", "confmat_mean", "confmat_normalize")]
    ns: Dict[str, Any] = dict(os=os, pickle=pickle, time=_time, reduce=reduce, plt=plt, np=np, torch=torch, th=torch,
                              dist=dist, nn=nn, F=F, prune=prune, optim=optim, Optimizer=Optimizer, DataLoader=DataLoader,
                              TensorDataset=TensorDataset, tqdm=tqdm, trange=trange, Optional=Optional, Literal=Literal,
                              Sequence=Sequence, Iterable=Iterable, Any=Any, Mapping=Mapping, wandb=None,
                              mixVAE_model=ref_nn.mixVAE_model, VAEConfig=ref_nn.VAEConfig)
    exec(compile(ast.Module(body=ukeep, type_ignores=[]), _UTILS, "exec"), ns)
    exec(compile(ast.Module(body=keep, type_ignores=[]), _TRAINER, "exec"), ns)
    return types.SimpleNamespace(cpl_mixVAE=ns["cpl_mixVAE"], nn_model=ref_nn, ns=ns)
