"""Diagnostic: GPU kNN covariances (regularisation NONE) of a scan against a numpy brute force; where do wrong rows sit?"""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
synth = importlib.import_module("pointcloud-slam_amd.synth")
import pointcloud_slam_amd as pcm
p = synth.make_pair(3, 8000, 80000)
g = pcm.GicpRegistration(0, optimizer="LM", regularization="NONE")
g.set_input_target(p.submap); g.set_input_source(p.scan)
c1 = g.get_covariances(False).reshape(-1, 3, 3) if g.get_covariances(False).ndim == 2 and g.get_covariances(False).shape[1] == 9 else g.get_covariances(False)
X = p.scan[:, :3].astype(np.float32)
n = len(X)
bad = []
kth = np.zeros(n)
ref = np.zeros((n, 3, 3))
for i in range(n):
    d2 = ((X - X[i]) ** 2).astype(np.float32).sum(1)
    nb = np.argsort(d2, kind="stable")[:20]
    kth[i] = np.sqrt(d2[nb[-1]])
    Y = X[nb].astype(np.float64)
    m = Y.mean(0)
    ref[i] = (Y - m).T @ (Y - m) / 20
c1 = np.asarray(c1, np.float64).reshape(n, -1)
if c1.shape[1] == 6:
    full = np.zeros((n, 3, 3)); iu = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    for j, (a, b) in enumerate(iu): full[:, a, b] = c1[:, j]; full[:, b, a] = c1[:, j]
    c1 = full
else:
    c1 = c1.reshape(n, 3, 3)
err = np.abs(c1 - ref).reshape(n, -1).max(1)
badm = err > 1e-7 * max(1.0, np.abs(ref).max())
print("bad rows", int(badm.sum()), "of", n)
res = 0.5
vox = np.round(X / res).astype(np.int64)
key = (vox[:, 0] * 1000003 + vox[:, 1]) * 1000003 + vox[:, 2]
_, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
print("k-th distance: all median %.3f  bad median %.3f max %.3f" % (np.median(kth), np.median(kth[badm]) if badm.any() else 0, kth[badm].max() if badm.any() else 0))
print("points in own voxel: all median %d  bad median %d" % (np.median(cnt[inv]), np.median(cnt[inv][badm]) if badm.any() else 0))
tr = np.trace(c1, axis1=1, axis2=2); trr = np.trace(ref, axis1=1, axis2=2)
print("trace(gpu)/trace(ref) of bad rows: ", np.round(np.percentile((tr / trr)[badm], [0, 25, 50, 75, 100]), 3) if badm.any() else "-")
idx = np.nonzero(badm)[0][:12]
for i in idx:
    print(i, "kth %.3f voxel pts %d  tr ratio %.3f  err %.2e" % (kth[i], cnt[inv][i], tr[i] / trr[i], err[i]))
