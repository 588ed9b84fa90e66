"""CPU: the consensus oracle (oracle/consensus.py) against the reference's own known-answer vectors
(tests/golden/consensus_kat.json, restated from the reference's tests/test_utils.py:18-105) and, in the build
container, against the reference's functions themselves on random cases."""
import json
import os

import numpy as np
import pytest

from oracle import consensus as OC
from oracle import ref_loader as RL

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "consensus_kat.json")))


def test_compute_confmat_kat():
    for k in KAT["compute_confmat"]:
        got = OC.compute_confmat(np.array(k["labels1"]), np.array(k["labels2"]))
        assert got.dtype == np.float64
        assert np.array_equal(got, np.array(k["expected"], dtype=np.float64))


def test_confmat_normalize_kat():
    for k in KAT["confmat_normalize"]:
        assert np.array_equal(OC.confmat_normalize(np.array(k["cm"], dtype=float)), np.array(k["expected"], dtype=float))


def test_confmat_mean_kat():
    for k in KAT["confmat_mean"]:
        assert OC.confmat_mean(np.array(k["cm"])) == k["expected"]


def test_classify_kat():
    for k in KAT["classify"]:
        assert OC.classify(np.array(k["probs"])).tolist() == k["expected"]


def test_edge_cases():
    # an empty class (row and column of zeros) normalises to zeros, not NaN
    cm = OC.compute_confmat(np.array([0, 0, 2]), np.array([0, 2, 2]), 4)
    n = OC.confmat_normalize(cm)
    assert np.isfinite(n).all() and n[1].sum() == 0 and n[:, 3].sum() == 0
    assert OC.confmat_mean(n) == (0.5 + 0.0 + 0.5 + 0.0) / 4
    # ties: first maximum
    assert OC.classify(np.array([[0.4, 0.4, 0.2]])).tolist() == [0]
    # three arms: pairs in (0,1), (0,2), (1,2) order
    lab = np.array([[0, 1, 2, 2], [0, 1, 2, 1], [1, 1, 2, 2]])
    vals, mean = OC.epoch_consensus(lab, 3)
    assert len(vals) == 3 and mean == pytest.approx(np.mean(vals))
    assert vals[0] == OC.confmat_mean(OC.confmat_normalize(OC.compute_confmat(lab[0], lab[1], 3)))


@pytest.mark.skipif(not RL.reference_available(), reason="/root/reference not present")
def test_oracle_equals_reference_functions():
    ref = RL.load_reference_consensus_utils()
    rng = np.random.default_rng(3)
    for K, n in [(4, 50), (92, 5000), (92, 300), (128, 20000), (7, 1)]:
        l1 = rng.integers(0, K, n).astype(np.int64)
        l2 = np.where(rng.random(n) < 0.7, l1, rng.integers(0, K, n)).astype(np.int64)
        cm_ref = ref.compute_confmat(l1, l2, K)
        cm = OC.compute_confmat(l1, l2, K)
        assert np.array_equal(cm, cm_ref) and np.array_equal(cm_ref, ref.compute_confmat_naive(l1, l2, K))
        assert np.array_equal(OC.confmat_normalize(cm), ref.confmat_normalize(cm_ref))
        assert OC.confmat_mean(OC.confmat_normalize(cm)) == ref.confmat_mean(ref.confmat_normalize(cm_ref))
    p = rng.random((40, 9))
    assert np.array_equal(OC.classify(p), ref.classify(p))
