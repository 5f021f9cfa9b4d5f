"""GPU parity of the pclomp NDT operator (PCM_MODEL_NDT_OMP) against the oracle (oracle/orc_pclndt.c)."""
import numpy as np
import pytest

from helpers import POSE_TOL_M, POSE_TOL_RAD, pose_error, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair_dense(synth):
    return synth.make_pair(0, 10000, 100000, density=60.0)    # leaves need >= 6 points


def _both(pcm, p, nn=7, res=1.0, **kw):
    from oracle import Oracle
    g = pcm.PclNdtRegistration(0, voxel_resolution=res, num_neighbors=nn, **kw)
    cfg = g.config
    o = Oracle("NDT_OMP", "LM", voxel_resolution=cfg.voxel_resolution, num_neighbors=nn, translation_eps=cfg.translation_eps, max_iterations=cfg.max_iterations,
               ndt_step_size=float(cfg.ndt_step_size), ndt_outlier_ratio=float(cfg.ndt_outlier_ratio))
    o.set_input_target(p.submap); o.set_input_source(p.scan)
    g.set_input_target(p.submap); g.set_input_source(p.scan)
    return o, g


def _pvec(T):
    from scipy.spatial.transform import Rotation
    T = np.asarray(T, np.float64)
    return np.concatenate([T[:3, 3], Rotation.from_matrix(T[:3, :3]).as_euler("XYZ")])


@pytest.mark.parametrize("nn", [0, 1, 7, 27])       # 0 = KDTREE (radius search over the leaf centroids), the default of jueying_slam's localization
@pytest.mark.parametrize("res", [1.0, 0.5])
def test_derivatives_match_oracle(pcm, pair_dense, nn, res):
    """computeDerivatives (float inner products) and computeHessian (double) at two poses."""
    p = pair_dense
    o, g = _both(pcm, p, nn, res)
    for T in (p.guess, p.T_gt):
        pv = _pvec(T)
        s0, g0, H0 = o.ndt_derivatives(pv)
        s1, g1, H1 = g.ndt_derivatives(pv, "float")
        assert s0 != 0 and abs(s1 - s0) <= 1e-6 * abs(s0)
        assert rel_err(g1, g0) < 1e-5 and rel_err(H1, H0) < 1e-5
        s2, g2, _ = g.ndt_derivatives(pv, None)
        assert abs(s2 - s0) <= 1e-6 * abs(s0) and rel_err(g2, g0) < 1e-5
        Hd0 = o.ndt_hessian(pv)
        _, _, Hd1 = g.ndt_derivatives(pv, "double")
        assert rel_err(Hd1, Hd0) < 1e-9
        sc0, sc1 = o.ndt_score(T), g.ndt_score(T)          # calculateScore
        assert sc0 != 0 and abs(sc1 - sc0) <= 1e-12 * abs(sc0)


@pytest.mark.parametrize("nn,res", [(7, 1.0), (1, 1.0), (27, 1.0), (7, 0.5), (0, 1.0), (0, 0.5)])
def test_align_matches_oracle(pcm, pair_dense, nn, res):
    from oracle.loader import result_T
    p = pair_dense
    o, g = _both(pcm, p, nn, res)
    ro, rg = o.align(p.guess), g.align(p.guess)
    dt, dr = pose_error(result_T(ro), rg.T64)
    assert dt < POSE_TOL_M and dr < POSE_TOL_RAD, (dt, dr)
    assert rg.iterations == ro.iterations and rg.converged == bool(ro.converged)
    assert rg.num_linearize == ro.num_linearize and rg.num_compute_error == ro.num_compute_error
    assert rel_err(rg.H, np.array(ro.H[:]).reshape(6, 6)) < 1e-5


def test_identity_guess_small_epsilon_and_batch(pcm, synth):
    """Identity guess (the `guess != Identity` branch, :84-89), a tighter epsilon (more line-search work), batch == singles."""
    from oracle.loader import result_T
    pairs = [synth.make_pair(30 + i, 4000 + 1000 * i, 40000 + 10000 * i, density=60.0) for i in range(3)]
    regs, orcs = [], []
    for p in pairs:
        o, g = _both(pcm, p, 7, 1.0, translation_eps=0.01)
        regs.append(g); orcs.append(o)
    ident = np.eye(4, dtype=np.float32)
    # move the scan into the map frame so that identity is a sensible start
    p = pairs[0]
    scan_w = (p.scan[:, :3] @ p.guess[:3, :3].T + p.guess[:3, 3]).astype(np.float32)
    regs[0].set_input_source(scan_w); orcs[0].set_input_source(scan_w)
    ro, rg = orcs[0].align(ident), regs[0].align(ident)
    dt, dr = pose_error(result_T(ro), rg.T64)
    assert dt < POSE_TOL_M and dr < POSE_TOL_RAD and rg.iterations == ro.iterations
    regs[0].set_input_source(p.scan); orcs[0].set_input_source(p.scan)
    singles = [g.align(q.guess) for g, q in zip(regs, pairs)]
    for s, o, q in zip(singles, orcs, pairs):
        r = o.align(q.guess)
        dt, dr = pose_error(result_T(r), s.T64)
        assert dt < POSE_TOL_M and dr < POSE_TOL_RAD and s.iterations == r.iterations
    batch = pcm.align_batch(regs, np.stack([q.guess for q in pairs]))
    for s, b in zip(singles, batch):
        assert np.array_equal(s.T64, b.T64)


@pytest.mark.parametrize("nn", [0, 1, 7, 27])
@pytest.mark.parametrize("res", [1.0, 0.5])
def test_neighbour_leaf_lists_change_nothing(pcm, pair_dense, nn, res):
    """PCM_FLAG_NEIGHBOUR_LISTS (16): the passes read the grid's neighbour-leaf lists (built with the leaves) instead of looking the
    cells up one by one (64 = never): the same leaves in the same order, hence identical sums, poses and evaluation counts -- single
    passes, single aligns, batched aligns; the default builds the lists at a target's second registration."""
    p = pair_dense
    kw = dict(voxel_resolution=res, num_neighbors=nn)
    a = pcm.PclNdtRegistration(0, flags=16, **kw)
    b = pcm.PclNdtRegistration(0, flags=64, **kw)
    c = pcm.PclNdtRegistration(0, **kw)
    for g in (a, b, c):
        g.set_input_target(p.submap); g.set_input_source(p.scan)
    for T in (p.guess, p.T_gt):
        pv = _pvec(T)
        for kind in ("float", None, "double"):
            ra, rb = a.ndt_derivatives(pv, kind), b.ndt_derivatives(pv, kind)
            assert ra[0] == rb[0] and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2])
    ra, rb = a.align(p.guess), b.align(p.guess)
    rc = [c.align(p.guess) for _ in range(3)][-1]     # the third registration against the target runs on lists built at the second
    for r in (ra, rc):
        assert np.array_equal(r.T64, rb.T64) and r.iterations == rb.iterations and r.num_linearize == rb.num_linearize
    p2 = [pcm.PclNdtRegistration(0, flags=f, **kw) for f in (16, 16, 64, 64)]
    for g in p2:
        g.set_input_target(p.submap); g.set_input_source(p.scan)
    guesses = np.stack([p.guess, p.guess])
    ba, bb = pcm.align_batch(p2[:2], guesses), pcm.align_batch(p2[2:], guesses)
    for x, y in zip(ba, bb):
        assert np.array_equal(x.T64, y.T64) and x.iterations == y.iterations
