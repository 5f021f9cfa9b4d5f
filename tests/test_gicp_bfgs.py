"""pclomp GICP-BFGS functor: objective / gradient that jueying_slam's GICP_OMP option minimises
(ndt_omp/include/pclomp/gicp_omp_impl.hpp:246-365).  CPU: the oracle against closed forms and finite differences
("parity unpinned": the reference holds no fixture for it).  GPU: pcm_gicp_bfgs_* against the oracle."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot


def _problem(seed, n=5000, m=4000, stride=4, base_translation=False):
    rng = np.random.default_rng(seed)
    src = np.ones((n, stride), np.float32)
    src[:, :3] = rng.uniform(-20, 20, (n, 3))
    Rt = Rot.from_euler("xyz", [0.02, -0.03, 0.05]).as_matrix()
    tgt = np.ones((n + 100, stride), np.float32)
    tgt[:n, :3] = src[:, :3] @ Rt.T + [0.1, -0.2, 0.05] + rng.normal(0, 0.02, (n, 3))
    tgt[n:, :3] = rng.uniform(-20, 20, (100, 3))
    A = rng.normal(size=(n, 3, 3))
    maha = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    maha[:, :3, :3] = A @ A.transpose(0, 2, 1) + np.eye(3)
    maha_cm = np.ascontiguousarray(maha.transpose(0, 2, 1)).reshape(n, 16)      # Eigen::Matrix4f is column-major
    idx_src = rng.permutation(n)[:m].astype(np.int32)
    idx_tgt = idx_src.copy()
    idx_tgt[::50] = rng.integers(n, n + 100, len(idx_tgt[::50]))                 # some wrong matches
    base = np.eye(4, dtype=np.float32)
    base[:3, :3] = Rot.from_euler("xyz", [0.01, 0.0, -0.01]).as_matrix()
    if base_translation:
        base[:3, 3] = [0.3, -0.1, 0.2]
    x = np.array([0.05, -0.1, 0.02, 0.01, -0.02, 0.03])
    return src, tgt, idx_src, idx_tgt, maha_cm, base, x


def test_oracle_apply_state_is_zyx_euler_on_top_of_base():
    from oracle import loader as L
    _, _, _, _, _, base, x = _problem(0, base_translation=True)
    T = L.gicp_bfgs_apply_state(base, x)
    R = Rot.from_euler("ZYX", [x[5], x[4], x[3]]).as_matrix() @ base[:3, :3].astype(np.float64)
    assert np.abs(T[:3, :3] - R).max() < 3e-7                         # float rotation
    assert np.allclose(T[:3, 3], base[:3, 3] + x[:3].astype(np.float32), atol=1e-7)
    assert np.array_equal(T[3], [0, 0, 0, 1])


def test_oracle_objective_matches_closed_form_and_gradient_matches_finite_differences():
    from oracle import loader as L
    src, tgt, isrc, itgt, maha, base, x = _problem(1)
    f, g = L.gicp_bfgs_fdf(src, tgt, isrc, itgt, maha, base, x, 2)
    f0, _ = L.gicp_bfgs_fdf(src, tgt, isrc, itgt, maha, base, x, 0)
    _, g1 = L.gicp_bfgs_fdf(src, tgt, isrc, itgt, maha, base, x, 1)
    T = L.gicp_bfgs_apply_state(base, x).astype(np.float64)
    res = src[isrc, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3] - tgt[itgt, :3]
    M = maha.reshape(-1, 4, 4).transpose(0, 2, 1)[isrc, :3, :3].astype(np.float64)
    f_ref = np.einsum("ia,iab,ib->", res, M, res) / len(isrc)
    assert abs(f - f_ref) < 1e-5 * f_ref and abs(f0 - f_ref) < 1e-5 * f_ref          # float transform / float residual
    assert np.array_equal(g, g1)                                                     # df and fdf share the gradient
    num = np.zeros(6)
    for k in range(6):
        h = 1e-3
        xp, xm = x.copy(), x.copy()
        xp[k] += h; xm[k] -= h
        num[k] = (L.gicp_bfgs_fdf(src, tgt, isrc, itgt, maha, base, xp, 2)[0] - L.gicp_bfgs_fdf(src, tgt, isrc, itgt, maha, base, xm, 2)[0]) / (2 * h)
    assert np.abs(num - g).max() < 1e-4 * np.abs(g).max()


def test_oracle_rejects_an_empty_set():
    from oracle import loader as L
    src, tgt, isrc, itgt, maha, base, x = _problem(2, n=100, m=10)
    with pytest.raises(ValueError):
        L.gicp_bfgs_fdf(src, tgt, isrc[:0], itgt[:0], maha, base, x, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,m,stride,bt", [(3, 5000, 4000, 4, False), (4, 300, 7, 3, True), (5, 120000, 100000, 8, True), (6, 400000, 300000, 4, False)])
def test_gpu_functor_matches_the_oracle(pcm, seed, n, m, stride, bt):
    from oracle import loader as L
    src, tgt, isrc, itgt, maha, base, x = _problem(seed, n=n, m=m, stride=stride, base_translation=bt)
    g = pcm.GicpRegistration(0)
    g.gicp_bfgs_set_correspondences(src, tgt, isrc, itgt, maha)
    for k in range(3):                                               # several evaluations over one packed set, as the BFGS does
        xk = x * (1.0 - 0.4 * k)
        for mode in (0, 1, 2):
            fo, go = L.gicp_bfgs_fdf(src, tgt, isrc, itgt, maha, base, xk, mode)
            fg, gg = g.gicp_bfgs_fdf(base, xk, mode)
            if mode != 1:
                assert abs(fg - fo) <= 1e-12 * abs(fo)                 # same terms, another order of the double additions
            if mode != 0:
                assert np.abs(gg - go).max() <= 1e-11 * np.abs(go).max()
    f1, g1 = g.gicp_bfgs_fdf(base, x, 2)
    f2, g2 = g.gicp_bfgs_fdf(base, x, 2)
    assert f1 == f2 and np.array_equal(g1, g2)                         # fixed summation order: reproducible


@pytest.mark.gpu
def test_gpu_functor_argument_errors(pcm):
    src, tgt, isrc, itgt, maha, base, x = _problem(7, n=200, m=50)
    g = pcm.GicpRegistration(0)
    with pytest.raises(pcm.PcmError):
        g.gicp_bfgs_fdf(base, x, 2)                                     # no correspondences yet
    bad = isrc.copy(); bad[3] = 10 ** 6
    with pytest.raises(pcm.PcmError):
        g.gicp_bfgs_set_correspondences(src, tgt, bad, itgt, maha)
    g.gicp_bfgs_set_correspondences(src, tgt, isrc[:0], itgt[:0], maha)  # an empty set is accepted, evaluating it is not
    with pytest.raises(pcm.PcmError):
        g.gicp_bfgs_fdf(base, x, 2)
    g.gicp_bfgs_set_correspondences(src, tgt, isrc, itgt, maha)
    with pytest.raises(pcm.PcmError):
        g.gicp_bfgs_fdf(base, x, 5)
    f, gr = g.gicp_bfgs_fdf(base, x, 2)
    assert np.isfinite(f) and np.isfinite(gr).all()


GOLD = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden", "gicp_bfgs_functor.npz")


def test_oracle_reproduces_the_committed_functor_vectors():
    """tests/golden/gicp_bfgs_functor.npz (tests/make_golden.py: oracle outputs, not reference outputs)."""
    from oracle import loader as L
    d = np.load(GOLD)
    assert np.array_equal(L.gicp_bfgs_apply_state(d["base"], d["x"]), d["T"])
    for mode in (0, 1, 2):
        f, g = L.gicp_bfgs_fdf(d["src"], d["tgt"], d["idx_src"], d["idx_tgt"], d["maha"], d["base"], d["x"], mode)
        if mode != 1:
            assert f == float(d["f%d" % mode])
        if mode != 0:
            assert np.array_equal(g, d["g%d" % mode])


@pytest.mark.gpu
def test_gpu_functor_matches_the_committed_vectors(pcm):
    d = np.load(GOLD)
    g = pcm.GicpRegistration(0)
    g.gicp_bfgs_set_correspondences(d["src"], d["tgt"], d["idx_src"], d["idx_tgt"], d["maha"])
    for mode in (0, 2):
        f, _ = g.gicp_bfgs_fdf(d["base"], d["x"], mode)
        assert abs(f - float(d["f%d" % mode])) <= 1e-12 * abs(float(d["f%d" % mode]))
    for mode in (1, 2):
        _, gr = g.gicp_bfgs_fdf(d["base"], d["x"], mode)
        assert np.abs(gr - d["g%d" % mode]).max() <= 1e-11 * np.abs(d["g%d" % mode]).max()


# ---- correspondence step of computeTransformation (gicp_omp_impl.hpp:405-472) -----------------------------------------
def _corr_problem(synth, seed=5):
    p = synth.make_pair(40 + seed, 6000, 60000, density=30.0)
    guess = p.guess.astype(np.float32)
    # transformation_ after a few outer iterations: a small rigid correction on top of the guess
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = Rot.from_euler("xyz", [0.004, -0.003, 0.006]).as_matrix()
    T[:3, 3] = [0.03, -0.02, 0.01]
    return p, T, guess


def test_oracle_correspondence_step_against_brute_force(synth):
    """Independent check of the oracle's step in float64 numpy: brute-force nearest neighbours, (R C1 R^T + C2)^-1 by numpy.linalg."""
    from oracle import Oracle
    p, T, G = _corr_problem(synth)
    o = Oracle("GICP", "LM", regularization="PCLOMP", max_corr_dist=0.6)
    o.set_input_target(p.submap); o.set_input_source(p.scan)
    isrc, itgt, M = o.gicp_bfgs_correspondences(T, G)
    assert 1000 < len(isrc) < len(p.scan) and np.all(np.diff(isrc) > 0)          # some rejected by the threshold; source order
    TG = T.astype(np.float64) @ G.astype(np.float64)
    q = p.scan[:, :3].astype(np.float64) @ TG[:3, :3].T + TG[:3, 3]
    tgt = p.submap[:, :3].astype(np.float64)
    sel = np.zeros(len(q), bool); sel[isrc] = True
    rng = np.random.default_rng(0)
    for i in rng.choice(len(q), 300, replace=False):                              # brute force on a sample (60k targets each)
        d2 = ((tgt - q[i]) ** 2).sum(1)
        j = int(d2.argmin())
        if d2[j] < 0.36 * (1 - 1e-4):
            assert sel[i] and abs(d2[itgt[np.searchsorted(isrc, i)]] - d2[j]) <= 1e-6 * max(d2[j], 1e-12)
        elif d2[j] > 0.36 * (1 + 1e-4):
            assert not sel[i]
    C1 = o.covariances(False); C2 = o.covariances(True)
    R = TG[:3, :3]
    for k in rng.choice(len(isrc), 300, replace=False):
        ref = np.linalg.inv(R @ C1[isrc[k]] @ R.T + C2[itgt[k]])
        assert np.abs(M[k] - ref).max() <= 2e-6 * np.abs(ref).max()               # double arithmetic, cast to float


@pytest.mark.gpu
def test_gpu_correspondence_step_matches_the_oracle_and_feeds_the_functor(pcm, synth):
    from oracle import Oracle
    from oracle import loader as L
    p, T, G = _corr_problem(synth)
    o = Oracle("GICP", "LM", regularization="PCLOMP", max_corr_dist=0.6)
    o.set_input_target(p.submap); o.set_input_source(p.scan)
    g = pcm.GicpRegistration(0, regularization="PCLOMP", max_corr_dist=0.6)
    g.set_input_target(p.submap); g.set_input_source(p.scan)
    for Tk in (np.eye(4, dtype=np.float32), T):                                  # first outer iteration, and a later one
        isrc0, itgt0, M0 = o.gicp_bfgs_correspondences(Tk, G)
        m = g.gicp_bfgs_update_correspondences(Tk, G)
        isrc1, itgt1, M1 = g.gicp_bfgs_get_correspondences()
        assert m == len(isrc0) and np.array_equal(isrc1, isrc0)
        same = itgt1 == itgt0                                                    # exact distance ties may pick another of two equidistant targets
        assert same.mean() > 0.9995
        assert np.abs(M1[same] - M0[same]).max() <= 1e-6 * np.abs(M0).max()
        # the functor on the device-packed set == the oracle's functor on the oracle's set (cloud_src = guess * input, :479)
        out = np.ones((len(p.scan), 4), np.float32)
        Gd = G.astype(np.float32)
        for a in range(3):
            out[:, a] = Gd[a, 0] * p.scan[:, 0] + (Gd[a, 1] * p.scan[:, 1] + (Gd[a, 2] * p.scan[:, 2] + Gd[a, 3]))
        maha = np.tile(np.eye(4, dtype=np.float32), (len(p.scan), 1, 1))
        maha[isrc0, :3, :3] = M0
        maha_cm = np.ascontiguousarray(maha.transpose(0, 2, 1)).reshape(-1, 16)
        base = np.eye(4, dtype=np.float32)
        x = np.array([0.01, -0.02, 0.005, 0.002, -0.001, 0.003])
        tgt4 = np.ascontiguousarray(p.submap[:, :4], np.float32)
        f0, g0 = L.gicp_bfgs_fdf(out, tgt4, isrc0, itgt0, maha_cm, base, x, 2)
        f1, g1 = g.gicp_bfgs_fdf(base, x, 2)
        assert abs(f1 - f0) <= 1e-6 * abs(f0) and np.abs(g1 - g0).max() <= 1e-6 * np.abs(g0).max()
