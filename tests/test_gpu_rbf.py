"""RBF-kernel covariances of FastVGICPCuda (NearestNeighborMethod::GPU_RBF_KERNEL, cuda/covariance_estimation_rbf.cu:59-151;
pcm_config.covariance_method = PCM_COV_RBF_KERNEL): the device kernel (block-culled, input-order float sums) against the oracle's
plain O(N^2) restatement and against an independent float64 statement of the same weighted moments.  Not bit-comparable with
the real reference (CUDA expf / fused multiply-adds): "parity unpinned".  ``-m gpu`` except the float64 check of the oracle."""
import numpy as np
import pytest

from helpers import pose_error, rel_err


def _moments64(pts, kw, max_dist):
    """sum w, sum w p, sum w p p^T with w = exp(-kw |x - p|^2) over |x - p|^2 <= max_dist^2, padding points at the origin
    included (the reference pads its last block of 512 with zeros, covariance_estimation_rbf.cu:126-129), in float64."""
    x = pts[:, :3].astype(np.float64)
    n = len(x)
    pad = (-n) % 512
    p = np.vstack([x, np.zeros((pad, 3))])
    d2 = ((x[:, None, :] - p[None, :, :]) ** 2).sum(-1)
    w = np.where(d2 <= np.float64(np.float32(max_dist)) ** 2, np.exp(-np.float64(np.float32(kw)) * d2), 0.0)
    sw = w.sum(1)
    sm = w @ p
    sc = np.einsum("np,pa,pb->nab", w, p, p)
    mean = sm / sw[:, None]
    return (sc - mean[:, :, None] * sm[:, None, :]) / sw[:, None, None]


def _cloud(synth, n, seed):
    p = synth.make_pair(seed, n, 2 * n, density=20.0)
    return p


def test_oracle_rbf_covariances_match_a_float64_statement(synth):
    from oracle import Oracle
    p = _cloud(synth, 1500, 3)
    for kw, md in ((0.25, 3.0), (2.0, 1.0)):
        o = Oracle("VGICP_CUDA", "LM", voxel_resolution=1.0, num_neighbors=1, regularization="NONE", rbf_kernel_width=kw, rbf_max_dist=md)
        o.set_input_target(p.submap); o.set_input_source(p.scan)
        c = o.covariances(False)
        ref = _moments64(p.scan, kw, md)
        # float sums of ~1e3 terms of magnitude |p|^2 w: relative to the second moments, not to the (much smaller) covariance
        scale = (np.abs(p.scan[:, :3]).max() ** 2)
        assert np.abs(c - ref).max() < 2e-4 * scale
        assert np.median(np.abs(c - ref)) < 1e-5 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("reg", ["NONE", "PLANE", "MIN_EIG", "FROBENIUS"])
def test_gpu_rbf_covariances_match_oracle(pcm, synth, reg):
    from oracle import Oracle
    p = _cloud(synth, 3000, 4)
    kw, md = 0.25, 3.0
    o = Oracle("VGICP_CUDA", "LM", voxel_resolution=1.0, num_neighbors=1, regularization=reg, rbf_kernel_width=kw, rbf_max_dist=md)
    g = pcm.VgicpCudaRegistration(0, optimizer="LM", regularization=reg)
    g.set_nearest_neighbor_search_method("GPU_RBF_KERNEL"); g.set_kernel_width(kw, md)
    for r in (o, g):
        r.set_input_target(p.submap); r.set_input_source(p.scan)
    g.evaluate_cost(p.guess.astype(np.float64))   # covariances are computed lazily
    for target in (False, True):
        c0, c1 = o.covariances(target), g.get_covariances(target)
        # device expf vs libm expf (a few ulp) through float sums whose terms are |p|^2 w while the covariance is their small
        # difference (sum w p p^T - mean sum w p^T: the cancellation is the reference's own); the eigen-decomposition of the
        # regularisation amplifies it further on nearly isotropic neighbourhoods: a handful of points may differ more
        scale = float(np.abs((p.submap if target else p.scan)[:, :3]).max()) ** 2
        tol = 2e-6 * scale if reg == "NONE" else max(2e-6 * scale, 2e-3 * max(1.0, np.abs(c0).max()))
        bad = np.abs(c1 - c0).reshape(len(c0), -1).max(axis=1) > tol
        assert bad.sum() <= max(2, len(c0) // 500), (target, int(bad.sum()))


@pytest.mark.gpu
def test_gpu_rbf_differs_from_knn_and_kernel_width_matters(pcm, synth):
    p = _cloud(synth, 3000, 5)
    T = p.guess.astype(np.float64)
    out = []
    for method, kw in (("GPU_BRUTEFORCE", 0.25), ("GPU_RBF_KERNEL", 0.25), ("GPU_RBF_KERNEL", 4.0)):
        g = pcm.VgicpCudaRegistration(0, optimizer="LM")
        g.set_nearest_neighbor_search_method(method); g.set_kernel_width(kw)
        g.set_input_target(p.submap); g.set_input_source(p.scan)
        g.evaluate_cost(T)
        out.append(g.get_covariances(False))
    assert np.abs(out[0] - out[1]).max() > 1e-3 and np.abs(out[1] - out[2]).max() > 1e-3


@pytest.mark.gpu
def test_gpu_rbf_align_matches_oracle(pcm, synth):
    from oracle import Oracle
    from oracle.loader import result_T
    p = synth.make_pair(6, 4000, 8000, density=60.0)
    for optimizer, iters in (("GN", 1), ("LM", 3)):
        o = Oracle("VGICP_CUDA", optimizer, voxel_resolution=1.0, num_neighbors=7, max_iterations=iters, rbf_kernel_width=0.25, rbf_max_dist=3.0)
        g = pcm.VgicpCudaRegistration(0, optimizer=optimizer, num_neighbors=7, max_iterations=iters)
        g.set_nearest_neighbor_search_method("GPU_RBF_KERNEL"); g.set_kernel_width(0.25, 3.0)
        for r in (o, g):
            r.set_input_target(p.submap); r.set_input_source(p.scan)
        c0, H0, b0 = o.linearize(p.guess.astype(np.float64))
        c1, H1, b1, inl = g.evaluate_cost(p.guess.astype(np.float64))
        assert inl == o.num_inliers and rel_err(H1, H0) < 1e-3 and rel_err(b1, b0) < 1e-3
        ro, rg = o.align(p.guess), g.align(p.guess)
        dt, dr = pose_error(result_T(ro), rg.T64)
        assert dt < 1e-4 and dr < 1e-4, (optimizer, dt, dr)
