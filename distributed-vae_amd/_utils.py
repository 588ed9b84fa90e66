"""Consensus utilities of mmidas/_utils.py:79-129 on the device.

Same names and meaning as the reference; inputs and outputs are CUDA tensors instead of numpy arrays, the arithmetic
runs in the HIP library (csrc/consensus.hip) and the values are bit-identical to the reference's numpy code
(integer counts; fp64 division and numpy's summation order for the mean).  No CPU fallback.
"""
from typing import Optional

import torch

from . import _native as N


def classify(probs: torch.Tensor) -> torch.Tensor:
    """_utils.py:79-80 ``np.argmax(probs, axis=-1)``: int32 labels, first maximum on ties."""
    return N.classify(probs)


def confmat_counts(n_arm: int, K: int, device) -> torch.Tensor:
    """Zeroed int64 [n_arm (n_arm - 1) / 2, K, K] accumulator for ``mixVAE_model.eval_labels(..., counts=)``."""
    return torch.zeros(max(n_arm * (n_arm - 1) // 2, 1), K, K, dtype=torch.int64, device=device)


def compute_confmat(labels1: torch.Tensor, labels2: torch.Tensor, K: Optional[int] = None) -> torch.Tensor:
    """_utils.py:84-95: K x K float64 matrix with ``m[labels1[i], labels2[i]] += 1``.  K None: number of distinct
    labels of the fuller side, as the reference."""
    assert len(labels1) == len(labels2)
    assert labels1.dim() == labels2.dim() == 1
    assert labels1.dtype == labels2.dtype and labels1.dtype in (torch.int64, torch.int32)
    if K is None:
        K = max(int(torch.unique(labels1).numel()), int(torch.unique(labels2).numel()))
    lab = torch.stack([labels1, labels2]).to(torch.int32)
    return N.confmat_accumulate(lab, K)[0].to(torch.float64)


def confmat_normalize(cm: torch.Tensor) -> torch.Tensor:
    """_utils.py:98-100: divide column j by max(column sum j, row sum j), 0 where that is 0."""
    counts = cm.to(torch.int64).unsqueeze(0)
    if not torch.equal(counts[0].to(cm.dtype), cm):
        raise ValueError("confmat_normalize on the device takes a matrix of counts (integers)")
    return N.consensus(counts, want_norm=True)[1][0]


def confmat_mean(cm: torch.Tensor) -> torch.Tensor:
    """_utils.py:128-129: mean of the diagonal (fp64, numpy's pairwise summation order), a 0-d device tensor."""
    d = torch.diagonal(cm).to(torch.float64)
    return _np_pairwise_sum(d) / d.numel()


def _np_pairwise_sum(v: torch.Tensor) -> torch.Tensor:
    """numpy's pairwise_sum on a 1-D fp64 device tensor, same association order (bit-identical result)."""
    n = v.numel()
    if n < 8:
        r = torch.zeros((), dtype=torch.float64, device=v.device)
        for i in range(n):
            r = r + v[i]
        return r
    if n > 128:
        n2 = n // 2
        n2 -= n2 % 8
        return _np_pairwise_sum(v[:n2]) + _np_pairwise_sum(v[n2:])
    m = n - n % 8
    acc = v[0:8].clone()
    for i in range(8, m, 8):
        acc = acc + v[i:i + 8]
    res = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]))
    for i in range(m, n):
        res = res + v[i]
    return res


def consensus_from_counts(counts: torch.Tensor) -> torch.Tensor:
    """``confmat_mean(confmat_normalize(cm))`` for every arm pair in one launch: float64 [pairs]."""
    return N.consensus(counts)
