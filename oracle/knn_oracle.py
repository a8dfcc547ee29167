"""TEST INFRASTRUCTURE ONLY (see oracle/ugaitnet_oracle.py).  CPU restatement of the classifier the reference's evaluation
main delegates to: sklearn.neighbors.KNeighborsClassifier(n_neighbors=k) with its defaults (mains/mj_testUWYHGaitNet_open_tum.py:
328-341) -- scikit-learn is a third-party dependency of the reference (not vendored; the version it pins, 0.22-0.24, and
the 1.x series installed here document the same behaviour): brute-force Euclidean neighbours, uniform weights, the class with
the most votes, the SMALLEST class on a tie (scipy.stats.mode semantics).  Pinned against scikit-learn itself in
tests/test_knn_oracle.py."""
import numpy as np


def knn_predict(gallery, labels, probes, k):
    """gallery [G,D], labels [G], probes [Q,D] -> (neighbour indices [Q,k] nearest first, predicted labels [Q])."""
    g = np.asarray(gallery, np.float64)
    q = np.asarray(probes, np.float64)
    labels = np.asarray(labels).reshape(-1)
    d2 = (q * q).sum(1)[:, None] + (g * g).sum(1)[None, :] - 2.0 * q @ g.T
    nbr = np.argsort(d2, axis=1, kind="stable")[:, :k]
    pred = np.empty(q.shape[0], labels.dtype)
    for i in range(q.shape[0]):
        vals, cnt = np.unique(labels[nbr[i]], return_counts=True)   # sorted ascending: argmax takes the smallest on ties
        pred[i] = vals[np.argmax(cnt)]
    return nbr, pred
