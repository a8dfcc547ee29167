"""GPU k-NN (ugn_knn_predict through ugaitnet_amd.knn) against the oracle."""
import numpy as np
import pytest
import torch

from oracle import knn_oracle

pytestmark = pytest.mark.gpu


def _data(rng, ncls, ng, nq, d, spread):
    centers = rng.normal(size=(ncls, d)).astype(np.float32) * spread
    yg = rng.integers(0, ncls, ng)
    g = (centers[yg] + rng.normal(size=(ng, d))).astype(np.float32)
    yq = rng.integers(0, ncls, nq)
    q = (centers[yq] + rng.normal(size=(nq, d))).astype(np.float32)
    return g, yg, q, yq


@pytest.mark.parametrize("k,ng,nq,d", [(1, 70, 33, 64), (3, 500, 129, 256), (5, 200, 64, 100), (16, 100, 10, 36), (3, 90, 40, 50),
                                       (2, 65, 65, 33)])
def test_knn_matches_oracle(k, ng, nq, d):
    from ugaitnet_amd.knn import KNeighborsClassifier
    rng = np.random.default_rng(k * 1000 + ng)
    g, yg, q, _ = _data(rng, 10, ng, nq, d, 2.0)
    clf = KNeighborsClassifier(n_neighbors=k).fit(g, yg.astype(np.float64))   # the reference passes float labels
    nbr, pred = clf.kneighbors_and_predict(q)
    nbr_ref, pred_ref = knn_oracle.knn_predict(g, yg.astype(np.float64), q, k)
    assert np.array_equal(nbr, nbr_ref)          # continuous random data: no distance ties
    assert np.array_equal(pred, pred_ref)


def test_knn_signature_sized_codes():
    """The evaluation shape: 15,872-dimensional signatures (62 bins x 256), k = 3, ragged tile counts."""
    from ugaitnet_amd.knn import KNeighborsClassifier
    rng = np.random.default_rng(9)
    g, yg, q, yq = _data(rng, 20, 333, 157, 15872, 0.05)
    clf = KNeighborsClassifier(n_neighbors=3).fit(g, yg)
    pred = clf.predict(q)
    _, pred_ref = knn_oracle.knn_predict(g, yg, q, 3)
    assert np.mean(pred == pred_ref) >= 0.99      # fp32 Gram form may reorder neighbours that agree to 6 digits
    assert abs(clf.score(q, yq) - np.mean(pred_ref == yq)) <= 0.02


def test_knn_exact_duplicates_and_tied_votes():
    from ugaitnet_amd.knn import KNeighborsClassifier
    g = np.array([[0.0, 0.0], [1.0, 0.0], [2.0, 0.0], [3.0, 0.0], [3.0, 0.0]], np.float32)
    y = np.array([7, 3, 9, 1, 5])
    q = np.array([[0.4, 0.0], [2.9, 0.0]], np.float32)
    for k in (2, 4):
        clf = KNeighborsClassifier(n_neighbors=k).fit(g, y)
        nbr, pred = clf.kneighbors_and_predict(q)
        nbr_ref, pred_ref = knn_oracle.knn_predict(g, y, q, k)
        assert np.array_equal(pred, pred_ref)
        assert np.array_equal(nbr, nbr_ref)       # equal distances: lower gallery index first


def test_knn_rejects_bad_arguments():
    from ugaitnet_amd.knn import KNeighborsClassifier
    with pytest.raises(ValueError):
        KNeighborsClassifier(n_neighbors=17)
    clf = KNeighborsClassifier(n_neighbors=3)
    with pytest.raises(ValueError):
        clf.fit(np.zeros((2, 4), np.float32), np.zeros(2))
    clf = KNeighborsClassifier(n_neighbors=1).fit(np.zeros((2, 4), np.float32), np.zeros(2))
    with pytest.raises(ValueError):
        clf.predict(np.zeros((1, 5), np.float32))
