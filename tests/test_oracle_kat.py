"""CPU tests of the oracle (no GPU): building blocks against closed-form answers /
numpy, the P2PLANE model against analytic known-answer scenes, finite-difference
Jacobian agreement, LM behaviour, and the committed golden vectors.

The reference's own fixture for this path (fast_gicp/data/relative.txt) points at
two .pcd files that are not in the tree, so these self-made pins are what holds
the oracle in place ("parity unpinned" by reference fixtures -- DESIGN.md)."""
import ctypes as C
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from helpers import pose_error

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def olib():
    from oracle import lib
    return lib()


def test_so3_exp_matches_rodrigues(olib):
    rng = np.random.default_rng(0)
    for w in list(rng.normal(size=(20, 3)) * 0.3) + [np.zeros(3), np.array([1e-7, 0, 0]), np.array([0, 3.0, 0])]:
        w = np.ascontiguousarray(w, np.float64)
        R = np.zeros((3, 3))
        olib.orc_test_so3_exp(w.ctypes.data, R.ctypes.data)
        assert np.allclose(R, Rotation.from_rotvec(w).as_matrix(), atol=1e-12)


def test_ldlt_solve(olib):
    rng = np.random.default_rng(1)
    for k in range(20):
        A = rng.normal(size=(6, 6))
        A = A @ A.T + (1e-9 if k % 2 else 1.0) * np.eye(6)
        b = rng.normal(size=6)
        x = np.zeros(6)
        olib.orc_test_ldlt6_solve(np.ascontiguousarray(A).ctypes.data, b.ctypes.data, x.ctypes.data)
        assert np.allclose(A @ x, b, rtol=1e-7, atol=1e-7 * np.abs(b).max())


def test_esti_plane_against_lstsq(olib):
    rng = np.random.default_rng(2)
    for m in (3, 4, 5):
        for _ in range(50):
            n = rng.normal(size=3); n /= np.linalg.norm(n)
            P = rng.uniform(-0.4, 0.4, size=(m, 3)); P -= np.outer(P @ n, n); P += n * rng.uniform(2, 20) + rng.normal(size=3)
            pts = np.ascontiguousarray(P, np.float32)
            pl = np.zeros(4, np.float32)
            ok = olib.orc_test_esti_plane(pts.ctypes.data, m, C.c_float(0.1), pl.ctypes.data)
            x = np.linalg.lstsq(pts.astype(np.float64), -np.ones(m), rcond=None)[0]
            nn = np.linalg.norm(x)
            assert ok == 1
            assert np.allclose(pl[:3], x / nn, atol=2e-4) and abs(pl[3] - 1 / nn) < 2e-3 * max(1.0, 1 / nn)
    # off-plane point > threshold -> rejected  (common_lib.h:235-241)
    pts = np.array([[0, 0, 5], [1, 0, 5], [0, 1, 5], [1, 1, 5], [0.5, 0.5, 5.5]], np.float32)
    pl = np.zeros(4, np.float32)
    assert olib.orc_test_esti_plane(pts.ctypes.data, 5, C.c_float(0.1), pl.ctypes.data) == 0
    # fewer than MIN_NUM_MATCH_POINTS  (common_lib.h:188-190)
    assert olib.orc_test_esti_plane(pts.ctypes.data, 2, C.c_float(0.1), pl.ctypes.data) == 0


def test_knn_equals_brute_force(olib, synth):
    """5-NN in the voxel hash == brute force restricted to the 27 cells (ivox3d.h:132-204)."""
    from oracle import Oracle
    p = synth.make_pair(3, 300, 20000)
    o = Oracle("P2PLANE", "GN", voxel_resolution=0.5, num_neighbors=27)
    o.set_input_target(p.submap)
    o.set_input_source(p.scan)
    tgt = p.submap[:, :3]
    inv = np.float32(1.0 / np.float32(0.5))
    keys = np.round(tgt * inv)          # numpy rounds half to even; exact .5 products do not occur in noisy data
    q_all = (p.scan[:, :3].astype(np.float64) @ p.T_gt[:3, :3].T + p.T_gt[:3, 3]).astype(np.float32)
    for q in q_all[:120]:
        q = np.ascontiguousarray(q)
        idx = np.zeros(8, np.int32); d2 = np.zeros(8, np.float32)
        m = olib.orc_test_knn(o._h, q.ctypes.data, idx.ctypes.data, d2.ctypes.data)
        kq = np.round(q * inv)
        near = np.all(np.abs(keys - kq) <= 1, axis=1)
        cand = np.nonzero(near)[0]
        d = ((tgt[cand] - q) ** 2).sum(1)
        order = np.argsort(d, kind="stable")[:5]
        assert m == min(5, len(cand))
        assert set(idx[:m]) == set(cand[order])
        assert np.allclose(np.sort(d2[:m]), np.sort(d[order]), rtol=1e-6)


def test_voxel_key_convention(olib):
    """iVox Pos2Grid rounds half away from zero (ivox3d.h:283-286), unlike fast_gicp's floor(x/res-0.5)."""
    from oracle import Oracle
    o = Oracle("P2PLANE", "GN", voxel_resolution=0.5)
    for p, want in (([0.24, -0.24, 0.26], [0, 0, 1]), ([0.25, -0.25, 0.75], [1, -1, 2]), ([-1.3, 2.49, 100.1], [-3, 5, 200])):
        a = np.array(p, np.float32); k = np.zeros(3, np.int32)
        olib.orc_test_voxel_key(o._h, a.ctypes.data, k.ctypes.data)
        assert list(k) == want


@pytest.mark.parametrize("optimizer", ["GN", "LM"])
def test_known_answer_corner(optimizer):
    """Three orthogonal planes, noise free: the exact pose must be recovered."""
    from oracle import Oracle
    from oracle.loader import result_T
    g = np.load(os.path.join(GOLD, "corner_kat.npz"))
    o = Oracle("P2PLANE", optimizer, voxel_resolution=0.5, num_neighbors=27)
    o.set_input_target(g["submap"]); o.set_input_source(g["scan"])
    assert o.linearize(g["T"])[0] < 1e-6          # zero residual at the truth
    r = o.align(np.eye(4, dtype=np.float32))
    dt, dr = pose_error(g["T"], result_T(r))
    assert r.converged and dt < 1e-5 and dr < 1e-5
    assert r.num_inliers == len(g["scan"])


def test_gradient_matches_finite_differences(synth):
    """b = J^T e must be half the gradient of the cost w.r.t. the LEFT perturbation used by
    the update delta * x0 (lsq_registration_impl.hpp:139-143), correspondences held fixed."""
    from oracle import Oracle
    sc, sm, T = synth.corner_scene(3000, 40000, seed=2, noise=0.01)
    o = Oracle("P2PLANE", "GN", voxel_resolution=0.5, num_neighbors=27)
    o.set_input_target(sm); o.set_input_source(sc)
    T0 = T.copy(); T0[:3, 3] += [0.01, 0.02, -0.01]
    c0, H, b = o.linearize(T0)
    assert np.allclose(H, H.T) and np.all(np.linalg.eigvalsh(H) > 0)
    h = 1e-4
    for k in range(6):
        d = np.zeros(6); d[k] = h
        Dp = np.eye(4); Dp[:3, :3] = Rotation.from_rotvec(d[:3]).as_matrix(); Dp[:3, 3] = d[3:]
        Dm = np.eye(4); Dm[:3, :3] = Rotation.from_rotvec(-d[:3]).as_matrix(); Dm[:3, 3] = -d[3:]
        g = (o.compute_error(Dp @ T0) - o.compute_error(Dm @ T0)) / (2 * h)
        assert abs(g - 2 * b[k]) < 2e-3 * max(1.0, abs(2 * b[k]))


def test_lm_cost_is_monotone(synth):
    """Accepted LM steps never increase the cost evaluated on the linearisation's correspondences."""
    from oracle import Oracle
    p = synth.make_pair(24, 5000, 60000)   # (seed 4 ends in a two-cycle of correspondence sets under GN and LM alike: never "converged")
    o = Oracle("P2PLANE", "LM", voxel_resolution=0.5, num_neighbors=27)
    o.set_input_target(p.submap); o.set_input_source(p.scan)
    o.enable_trace(256)
    r = o.align(p.guess)
    assert r.converged
    assert r.num_compute_error >= r.num_linearize - 1
    tr = o.trace()
    lin = ~np.isnan(tr[:, 1])            # linearize records; the others are the trial costs in between
    costs = tr[lin, 0]
    assert costs[-1] < costs[0]
    # every accepted trial has a cost below the cost of the linearisation it started from (rho >= 0)
    i = 0
    while i < len(tr):
        assert lin[i]
        y0 = tr[i, 0]; j = i + 1
        while j < len(tr) and not lin[j]: j += 1
        if j > i + 1 and j < len(tr): assert tr[j - 1, 0] <= y0 * (1 + 1e-12)
        i = j


def test_golden_vectors_reproduce(synth):
    """The committed golden vectors (tests/make_golden.py) are reproduced bit for bit by the
    oracle on regenerated inputs: pins both the synthetic generator and the restatement."""
    from oracle import Oracle
    from oracle.loader import result_T
    g = np.load(os.path.join(GOLD, "p2plane_config1.npz"))
    names = sorted({k.split("/")[0] for k in g.files})
    assert len(names) == 6
    for name in names:
        c = {k.split("/")[1]: g[k] for k in g.files if k.startswith(name + "/")}
        p = synth.make_pair(int(c["seed"]), int(c["n_scan"]), int(c["m_map"]))
        assert np.array_equal(p.scan[:8], c["scan_head"]) and np.array_equal(p.submap[:8], c["submap_head"])
        o = Oracle("P2PLANE", str(c["optimizer"]), voxel_resolution=0.5, num_neighbors=27, num_threads=1 + (int(c["seed"]) % 3))
        o.set_input_target(p.submap); o.set_input_source(p.scan)
        o.enable_trace(128)
        r = o.align(p.guess)
        assert r.iterations == int(c["iterations"]) and r.converged == int(c["converged"])
        assert r.num_inliers == int(c["num_inliers"]) and r.num_linearize == int(c["num_linearize"])
        # thread count only changes the summation order of the per-thread partials
        assert np.allclose(result_T(r), c["T"], rtol=0, atol=1e-9)
        assert np.allclose(o.trace(), c["trace"], rtol=1e-9, equal_nan=True)     # trial records carry NaN in the H / b slots
        # converges to a pose near the ground truth of the synthetic pair (noise-limited)
        dt, dr = pose_error(c["T_gt"], c["T"])
        assert dt < 0.10 and dr < np.deg2rad(1.0)      # noise-limited (2 cm range noise, 10k points, ground-dominated scan)


def test_golden_vectors_of_the_other_models(synth):
    """tests/golden/models_small.npz (tests/make_golden.py): one small registration per residual model, pinned against
    regressions of the restatements (GICP, VGICP additive / multiplicative, VGICP of the CUDA core, NDT P2D / D2D, pclomp NDT)."""
    import importlib.util
    from oracle import Oracle
    from oracle.loader import result_T
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    g = np.load(os.path.join(GOLD, "models_small.npz"))
    for name, model, opt, kw, dense in mg.MODEL_CASES:
        c = mg.model_case(name, model, opt, dict(kw, num_threads=2), dense)
        assert int(c["iterations"]) == int(g[name + "/iterations"]) and int(c["converged"]) == int(g[name + "/converged"]), name
        assert int(c["num_linearize"]) == int(g[name + "/num_linearize"]) and int(c["num_compute_error"]) == int(g[name + "/num_compute_error"]), name
        assert np.allclose(c["T"], g[name + "/T"], rtol=0, atol=1e-8), name
        assert np.allclose(c["H"], g[name + "/H"], rtol=1e-7, atol=1e-7 * np.abs(g[name + "/H"]).max()), name


def test_edge_cases():
    from oracle import Oracle
    o = Oracle("P2PLANE", "GN")
    with pytest.raises(RuntimeError):
        o.align()                                   # no inputs
    far = np.full((16, 3), 500.0, np.float32)
    o.set_input_target(np.zeros((4, 3), np.float32) + np.arange(4, dtype=np.float32)[:, None])
    o.set_input_source(far)
    c, H, b = o.linearize(np.eye(4))
    assert c == 0.0 and not H.any() and o.num_inliers == 0


@pytest.mark.parametrize("model", ["NDT_P2D", "NDT_D2D"])
def test_ndt_oracle_properties(synth, model):
    """NDT restatement (orc_models_gauss.c): voxel statistics vs numpy, gradient vs finite
    differences on fixed correspondences, convergence near the ground truth."""
    import ctypes as C
    from oracle import Oracle, lib
    from oracle.loader import result_T
    p = synth.make_pair(0, 4000, 60000, density=60.0)
    o = Oracle(model, "LM", voxel_resolution=1.0, num_neighbors=7)
    o.set_input_target(p.submap); o.set_input_source(p.scan)
    # voxel statistics of the voxel holding the first submap point (floor(x/res-0.5) convention)
    q = np.ascontiguousarray(p.submap[0, :3], np.float32)
    mean = np.zeros(3, np.float32); cov = np.zeros(9, np.float32); n = C.c_int()
    assert lib().orc_test_gauss_voxel(o._h, q.ctypes.data, mean.ctypes.data, cov.ctypes.data, C.byref(n)) == 1
    key = np.floor(p.submap[:, :3] / np.float32(1.0) - np.float32(0.5))
    sel = np.all(key == np.floor(q / np.float32(1.0) - np.float32(0.5)), axis=1)
    pts = p.submap[sel, :3].astype(np.float64)
    assert n.value == len(pts)
    assert np.allclose(mean, pts.mean(0), atol=1e-4)
    w, V = np.linalg.eigh(np.cov(pts.T, bias=True)) if len(pts) > 1 else (np.zeros(3), np.eye(3))
    want = V @ np.diag(np.maximum(w, 1e-3)) @ V.T
    assert np.allclose(cov.reshape(3, 3), want, atol=2e-3)       # float-product sums far from the origin (see orc_models_gauss.c)
    # gradient: b = J^T M e is half the gradient of the cost with the Cauchy weights frozen ... the reference
    # differentiates with w held constant, so only check descent: a small step along -H^-1 b lowers the cost
    T0 = p.T_gt.copy(); T0[:3, 3] += [0.05, -0.03, 0.02]
    c0, H, b = o.linearize(T0)
    d = np.linalg.solve(H, -b) * 0.5
    D = np.eye(4); D[:3, :3] = Rotation.from_rotvec(d[:3]).as_matrix(); D[:3, 3] = d[3:]
    assert o.compute_error(D @ T0) < c0
    # DIRECT1 converges next to the ground truth (DIRECT7 P2D is biased by its far-cell correspondences)
    o1 = Oracle(model, "LM", voxel_resolution=1.0, num_neighbors=1)
    o1.set_input_target(p.submap); o1.set_input_source(p.scan)
    r = o1.align(p.guess)
    # 1 m voxels are coarse and LM may stop on a rejected small step (lsq_registration_impl.hpp:155-158):
    # no accuracy claim, only that the optimiser lowered its own objective
    assert r.converged and o1.linearize(result_T(r))[0] / max(1, o1.num_inliers) < 1.05 * o1.linearize(p.guess.astype(np.float64))[0] / max(1, o1.num_inliers)


def test_exact_knn_and_gicp_covariances(synth):
    """orc_gicp.c: the grid kNN equals brute force (dense and sparse clouds), and the PLANE-regularised
    covariance has eigenvalues (1e-3, 1, 1) with the small axis along the local surface normal."""
    from oracle import Oracle
    p = synth.make_pair(5, 2000, 20000)
    o = Oracle("GICP", "LM", voxel_resolution=0.5)
    o.set_input_target(p.submap); o.set_input_source(p.scan)
    rng = np.random.default_rng(0)
    sub = p.submap[:, :3]
    for i in range(40):
        q = (sub[rng.integers(len(sub))] + rng.normal(0, 0.3 if i < 30 else 40.0, 3)).astype(np.float32)
        idx, d2 = o.knn_exact(q, 20)
        d = (sub - q) ** 2
        dd = (d[:, 0] + d[:, 1]) + d[:, 2]
        order = np.lexsort((np.arange(len(dd)), dd))[:20]
        assert (order == idx).all() and np.array_equal(dd[order], d2)
    cov = o.covariances(target=True)
    w = np.linalg.eigvalsh(cov[:200])
    assert np.allclose(w, [1e-3, 1.0, 1.0], rtol=1e-9)
    ground = np.abs(sub[:, 2]) < 1e-6                     # synthetic ground plane z = 0 -> normal = z
    if ground.sum() > 50:
        gi = np.nonzero(ground)[0][:50]
        interior = [i for i in gi if np.all(np.abs(sub[o.knn_exact(sub[i], 20)[0], 2]) < 1e-6)]
        for i in interior[:10]:
            assert abs(cov[i][2, 2] - 1e-3) < 1e-9


@pytest.mark.parametrize("model,kw", [("GICP", {}), ("VGICP", {"voxel_resolution": 1.0, "num_neighbors": 1})])
def test_gicp_oracle_converges(synth, model, kw):
    from oracle import Oracle
    from oracle.loader import result_T
    p = synth.make_pair(3, 4000, 40000)
    o = Oracle(model, "LM", **kw)
    o.set_input_target(p.submap); o.set_input_source(p.scan)
    c_guess = o.linearize(p.guess.astype(np.float64))[0]
    r = o.align(p.guess)
    assert r.converged
    assert o.linearize(result_T(r))[0] < c_guess
    dt, dr = pose_error(result_T(r), p.T_gt)
    if model == "GICP":
        assert dt < 0.15 and dr < 0.05      # sparse 4k-point scan: a sanity bound, parity is tested on the GPU side


def _blob_scene(seed=1):
    """Gaussian blobs at voxel centres (1 m grid) + source points well inside the voxels: the NDT score is
    smooth there (no point changes voxel under a 1e-4 perturbation), so finite differences are meaningful."""
    rng = np.random.default_rng(seed)
    centers = np.unique(rng.integers(-5, 6, (60, 3)).astype(np.float64) + 0.5, axis=0)
    tgt = np.concatenate([c + rng.normal(0, [0.12, 0.08, 0.03], (40, 3)) @ np.linalg.qr(rng.normal(size=(3, 3)))[0].T for c in centers]).astype(np.float32)
    src = (centers[rng.integers(len(centers), size=400)] + rng.uniform(-0.15, 0.15, (400, 3))).astype(np.float32)
    return tgt, src


@pytest.mark.parametrize("nn", [0, 1, 7, 27])
def test_pclndt_oracle_derivatives_are_derivatives(nn):
    """orc_pclndt.c: gradient = d(score)/dp and Hessian = d(gradient)/dp (eq. 6.12 / 6.13) by central
    differences; the float-path Hessian equals the double-path one up to float rounding."""
    from oracle import Oracle
    tgt, src = _blob_scene()
    o = Oracle("NDT_OMP", "LM", voxel_resolution=1.0, num_neighbors=nn, translation_eps=0.1, max_iterations=35, num_threads=4)
    o.set_input_target(tgt); o.set_input_source(src)
    p = np.array([0.02, -0.03, 0.01, 0.004, -0.006, 0.005])
    s0, g, H = o.ndt_derivatives(p)
    Hd = o.ndt_hessian(p)
    num = np.zeros(6); Hn = np.zeros((6, 6))
    for i in range(6):
        d = np.zeros(6); d[i] = 1e-4
        sp, gp, _ = o.ndt_derivatives(p + d); sm, gm, _ = o.ndt_derivatives(p - d)
        num[i] = (sp - sm) / 2e-4; Hn[:, i] = (gp - gm) / 2e-4
    assert s0 > 0
    assert np.abs(g - num).max() < 1e-3 * np.abs(g).max()
    assert np.abs(Hd - Hn).max() < 1e-3 * np.abs(Hd).max()
    assert np.abs(H - Hd).max() < 1e-4 * np.abs(Hd).max()


def test_pclndt_oracle_pieces():
    """Leaf statistics vs numpy (unbiased covariance, eigenvalue inflation to 1 % of the largest, inverse),
    the Euler / pose round trip, and the SVD solve vs numpy."""
    from oracle import Oracle
    from oracle.loader import lib
    tgt, src = _blob_scene(3)
    o = Oracle("NDT_OMP", "LM", voxel_resolution=1.0, num_neighbors=7)
    o.set_input_target(tgt); o.set_input_source(src)
    key = np.floor(tgt[:, :3] * np.float32(1.0)).astype(int)
    for c in np.unique(key, axis=0)[:20]:
        pts = tgt[(key == c).all(axis=1), :3].astype(np.float64)
        leaf = o.ndt_leaf((c + 0.5).astype(np.float32))
        assert leaf is not None and leaf[2] == len(pts)
        if len(pts) < 6:
            continue
        assert np.allclose(leaf[0], pts.mean(axis=0), atol=1e-12)
        # Leaf() starts cov_ as the identity (voxel_grid_covariance_omp.h:107), so the single-pass sum carries + I:
        # :323-324: (biased estimate + I/n) times (n-1)/n
        cov = (np.cov(pts.T, bias=True) + np.eye(3) / len(pts)) * (len(pts) - 1.0) / len(pts)
        w, V = np.linalg.eigh(cov)
        w = np.maximum(w, 0.01 * w[2])
        assert np.allclose(leaf[1], np.linalg.inv(V @ np.diag(w) @ V.T), rtol=1e-6)
    rng = np.random.default_rng(0)
    for _ in range(20):
        p = np.concatenate([rng.uniform(-5, 5, 3), rng.uniform(-1.2, 1.2, 3)])
        T = np.zeros(16, np.float32); lib().orc_pclndt_pose(p.ctypes.data, T.ctypes.data)
        T = T.reshape(4, 4)
        assert np.allclose(T[:3, :3], Rotation.from_euler("XYZ", p[3:]).as_matrix(), atol=1e-6)
        R = np.ascontiguousarray(T[:3, :3]); e = np.zeros(3, np.float32)
        lib().orc_pclndt_euler(R.ctypes.data, e.ctypes.data)
        # Eigen returns the representation with the first angle in [0, pi]: the same rotation, maybe other angles
        assert 0.0 <= e[0] <= np.pi + 1e-6
        assert np.allclose(Rotation.from_euler("XYZ", e.astype(np.float64)).as_matrix(), R, atol=2e-6)
        A = rng.normal(size=(6, 6)); A = A @ A.T + 0.1 * np.eye(6); b = rng.normal(size=6); x = np.zeros(6)
        lib().orc_pclndt_svd_solve(A.ctypes.data, b.ctypes.data, x.ctypes.data)
        assert np.allclose(x, np.linalg.solve(A, b), rtol=1e-9)


def test_pclndt_oracle_align(synth):
    from oracle import Oracle
    from oracle.loader import result_T
    p = synth.make_pair(0, 10000, 100000, density=60.0)
    o = Oracle("NDT_OMP", "LM", voxel_resolution=1.0, num_neighbors=7, translation_eps=0.1, max_iterations=35)
    o.set_input_target(p.submap); o.set_input_source(p.scan)
    r = o.align(p.guess)
    assert r.converged and r.iterations >= 1
    e0, _ = pose_error(p.guess, p.T_gt)
    e1, _ = pose_error(result_T(r), p.T_gt)
    assert e1 < e0          # epsilon 0.1 m: the reference stops as soon as a step is shorter than 10 cm


def test_pclomp_covariances(synth):
    """ORC_REG_PCLOMP (pclomp computeCovariances, gicp_omp_impl.hpp:48-122): eigenvalues (0.001, 1, 1); the same 20 neighbours as
    the fast_gicp path, so the small axis agrees with PLANE up to the cancellation of the raw float second moments."""
    from oracle import Oracle
    p = synth.make_pair(6, 1500, 15000)
    covs = {}
    for reg in ("PLANE", "PCLOMP"):
        o = Oracle("GICP", "LM", voxel_resolution=0.5, regularization=reg)
        o.set_input_target(p.submap); o.set_input_source(p.scan)
        covs[reg] = o.covariances(target=True)
    w, V = np.linalg.eigh(covs["PCLOMP"])
    assert np.allclose(w, [1e-3, 1.0, 1.0], rtol=1e-9)
    _, Vp = np.linalg.eigh(covs["PLANE"])
    cosang = np.abs(np.einsum("ia,ia->i", V[:, :, 0], Vp[:, :, 0]))
    assert np.median(cosang) > 0.9999 and (cosang > 0.99).mean() > 0.95
    assert np.allclose(covs["PCLOMP"], covs["PCLOMP"].transpose(0, 2, 1), atol=1e-15)
