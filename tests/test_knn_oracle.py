"""Pins oracle/knn_oracle.py against scikit-learn (the library the reference calls, mains/mj_testUWYHGaitNet_open_tum.py:331-341)."""
import numpy as np
import pytest

from oracle import knn_oracle


@pytest.mark.parametrize("k", [1, 3, 5])
def test_oracle_matches_sklearn(k):
    sk = pytest.importorskip("sklearn.neighbors")
    rng = np.random.default_rng(100 + k)
    ncls, d = 12, 40
    centers = rng.normal(size=(ncls, d)) * 2.0
    yg = rng.integers(0, ncls, 300)
    g = centers[yg] + rng.normal(size=(300, d))
    yq = rng.integers(0, ncls, 90)
    q = centers[yq] + rng.normal(size=(90, d))
    clf = sk.KNeighborsClassifier(n_neighbors=k).fit(g, yg.astype(np.float64))
    ref = clf.predict(q)
    nbr_ref = clf.kneighbors(q, return_distance=False)
    nbr, pred = knn_oracle.knn_predict(g, yg.astype(np.float64), q, k)
    assert np.array_equal(pred, ref)
    assert np.array_equal(np.sort(nbr, 1), np.sort(nbr_ref, 1))


def test_oracle_tied_vote_takes_smallest_label():
    sk = pytest.importorskip("sklearn.neighbors")
    g = np.array([[0.0], [1.0], [2.0], [3.0]])
    y = np.array([7, 3, 9, 1])
    q = np.array([[0.4], [2.6]])
    for k in (2, 4):
        ref = sk.KNeighborsClassifier(n_neighbors=k).fit(g, y).predict(q)
        assert np.array_equal(knn_oracle.knn_predict(g, y, q, k)[1], ref)
