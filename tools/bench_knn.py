"""Evaluation-time k-NN: GPU ugn_knn_predict vs scikit-learn on the host, signature-sized codes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ugaitnet_amd.knn import KNeighborsClassifier

ng, nq, d, k = 3000, 2000, 15872, 3
rng = np.random.default_rng(0)
centers = rng.normal(size=(150, d)).astype(np.float32) * 0.05
yg = rng.integers(0, 150, ng)
g = centers[yg] + rng.normal(size=(ng, d)).astype(np.float32)
yq = rng.integers(0, 150, nq)
q = centers[yq] + rng.normal(size=(nq, d)).astype(np.float32)
gd, qd = torch.from_numpy(g).cuda(), torch.from_numpy(q).cuda()
clf = KNeighborsClassifier(k).fit(gd, yg)
clf.predict(qd)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(3):
    pred = clf.predict(qd)
torch.cuda.synchronize()
tg = (time.time() - t0) / 3
print("GPU  : %.2f ms per predict (%d probes x %d gallery x %d dims, k=%d) = %.1f TFLOP/s on the distance GEMM" % (
    tg * 1e3, nq, ng, d, k, 2.0 * nq * ng * d / tg / 1e12))
try:
    from sklearn.neighbors import KNeighborsClassifier as SK
    sk = SK(n_neighbors=k).fit(g, yg)
    t0 = time.time()
    ref = sk.predict(q)
    tc = time.time() - t0
    print("sklearn (host): %.1f ms; agreement %.4f" % (tc * 1e3, float(np.mean(ref == pred))))
except ImportError:
    pass
