"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's label / consensus arithmetic.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(distributed-vae_amd/_utils.py -> csrc/consensus.hip) never does.

Follows mmidas/_utils.py of the reference:
  classify            :79-80     np.argmax(probs, axis=-1)
  compute_confmat     :84-95     K x K float64 zeros, np.add.at(matrix, (labels1, labels2), 1)
  confmat_normalize   :98-100    maxes = max(column sums, row sums); cm / maxes (broadcast over the last axis), 0 where 0
  confmat_mean        :128-129   np.mean(np.diag(cm))
and the per-epoch loop of mmidas/cpl_mixvae.py:563-657 (`epoch_consensus`): labels of every arm over all batches,
one confusion matrix per arm pair a < b, mean over the pairs.

Pinned by the reference's own known-answer tests (tests/test_utils.py:18-105 of the reference, restated as data in
tests/golden/consensus_kat.json) and, where /root/reference is present, against the live reference functions
(tests/test_oracle_vs_reference.py).
"""
import numpy as np


def classify(probs):
    return np.argmax(probs, axis=-1)


def compute_confmat(labels1, labels2, K=None):
    labels1 = np.asarray(labels1)
    labels2 = np.asarray(labels2)
    assert len(labels1) == len(labels2)
    assert labels1.ndim == labels2.ndim == 1
    if K is None:
        K = max(len(np.unique(labels1)), len(np.unique(labels2)))
    m = np.zeros((K, K))
    np.add.at(m, (labels1, labels2), 1)
    return m


def confmat_normalize(cm):
    cm = np.asarray(cm, dtype=np.float64)
    maxes = np.maximum(np.sum(cm, axis=0), np.sum(cm, axis=1))
    return np.divide(cm, maxes, out=np.zeros_like(cm), where=maxes != 0)


def confmat_mean(cm):
    return np.mean(np.diag(cm))


def epoch_consensus(labels, K):
    """labels: int array [A, n] (classify of c per arm over the whole set) -> (per-pair consensus list, mean)."""
    A = len(labels)
    vals = []
    for a in range(A):
        for b in range(a + 1, A):
            vals.append(confmat_mean(confmat_normalize(compute_confmat(labels[a], labels[b], K))))
    return vals, float(np.mean(np.array(vals)))
