"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on
the same seeded inputs.  Run on the MI355X box with ``-m gpu``.

Tolerances: poses within 1e-4 m / 1e-4 rad (BASELINE.json north_star); normal
equations (H, b, cost) within 1e-5 relative; inlier counts exact.
"""
import numpy as np
import pytest

from helpers import HB_RTOL, POSE_TOL_M, POSE_TOL_RAD, pose_error, rel_err

pytestmark = pytest.mark.gpu

CFG = dict(voxel_resolution=0.5, num_neighbors=27)


def _oracle(optimizer="GN", **kw):
    from oracle import Oracle
    p = dict(CFG); p.update(kw)
    return Oracle("P2PLANE", optimizer, **p)


def _gpu(pcm, optimizer="GN", **kw):
    p = dict(CFG); p.update(kw)
    return pcm.P2PlaneRegistration(0, optimizer=optimizer, **p)


@pytest.fixture(scope="module")
def pair10k(synth):
    return synth.make_pair(0, 10000, 100000)   # BASELINE config 1


def test_linearize_matches_oracle(pcm, pair10k):
    p = pair10k
    o = _oracle(); o.set_input_target(p.submap); o.set_input_source(p.scan)
    g = _gpu(pcm); g.set_input_target(p.submap); g.set_input_source(p.scan)
    for T in (p.guess.astype(np.float64), p.T_gt):
        c0, H0, b0 = o.linearize(T)
        c1, H1, b1, inl = g.evaluate_cost(T)
        assert inl == o.num_inliers
        assert rel_err(H1, H0) < HB_RTOL
        assert rel_err(b1, b0) < HB_RTOL
        assert abs(c1 - c0) <= HB_RTOL * abs(c0)
        # compute_error re-uses the planes of the linearize above
        T2 = T.copy(); T2[:3, 3] += [0.01, -0.02, 0.005]
        assert abs(g.compute_error(T2) - o.compute_error(T2)) <= HB_RTOL * abs(o.compute_error(T2))


@pytest.mark.parametrize("nn", [1, 7, 19, 27])
def test_neighbor_modes(pcm, pair10k, nn):
    p = pair10k
    o = _oracle(num_neighbors=nn); o.set_input_target(p.submap); o.set_input_source(p.scan)
    g = _gpu(pcm, num_neighbors=nn); g.set_input_target(p.submap); g.set_input_source(p.scan)
    c0, H0, b0 = o.linearize(p.T_gt)
    c1, H1, b1, inl = g.evaluate_cost(p.T_gt)
    assert inl == o.num_inliers
    assert rel_err(H1, H0) < HB_RTOL and rel_err(b1, b0) < HB_RTOL


@pytest.mark.parametrize("optimizer", ["GN", "LM"])
def test_align_matches_oracle(pcm, pair10k, optimizer):
    from oracle.loader import result_T
    p = pair10k
    o = _oracle(optimizer); o.set_input_target(p.submap); o.set_input_source(p.scan)
    g = _gpu(pcm, optimizer); g.set_input_target(p.submap); g.set_input_source(p.scan)
    ro = o.align(p.guess)
    rg = g.align(p.guess)
    dt, dr = pose_error(result_T(ro), rg.T64)
    assert dt < POSE_TOL_M and dr < POSE_TOL_RAD
    assert rg.iterations == ro.iterations and rg.converged == bool(ro.converged)
    assert rg.num_linearize == ro.num_linearize and rg.num_compute_error == ro.num_compute_error
    assert rg.num_inliers == ro.num_inliers
    assert rel_err(rg.H, np.array(ro.H[:]).reshape(6, 6)) < 1e-4
    assert np.allclose(rg.T, rg.T64.astype(np.float32))   # final_transformation_ = x0.cast<float>()


def test_known_answer_corner(pcm, synth):
    sc, sm, T = synth.corner_scene(5000, 60000, seed=1)
    g = _gpu(pcm, "LM"); g.set_input_target(sm); g.set_input_source(sc)
    r = g.align()
    dt, dr = pose_error(T, r.T64)
    assert r.converged and dt < 1e-5 and dr < 1e-5


def test_lds_path_equals_global_path(pcm, pair10k):
    """The per-tile LDS voxel grid / staged points and the per-lane global probing
    must pick exactly the same neighbours: bit-identical planes."""
    p = pair10k
    n = len(p.scan)
    a = _gpu(pcm, sort_source=0); a.set_input_target(p.submap); a.set_input_source(p.scan)
    b = _gpu(pcm, sort_source=0, flags=1); b.set_input_target(p.submap); b.set_input_source(p.scan)
    for T in (p.T_gt, p.guess.astype(np.float64)):
        ra, rb = a.evaluate_cost(T), b.evaluate_cost(T)
        pa, pb = a.get_planes(n), b.get_planes(n)
        assert np.array_equal(np.isnan(pa[:, 0]), np.isnan(pb[:, 0]))
        ok = ~np.isnan(pa[:, 0])
        assert np.array_equal(pa[ok], pb[ok])
        assert ra[3] == rb[3] and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2])
    # sorted scan: most tiles take the LDS path
    c = _gpu(pcm, sort_source=1); c.set_input_target(p.submap); c.set_input_source(p.scan)
    c.set_profiling(2); c.align(p.guess); st = c.stats()
    assert st["tiles"] > 0 and st["tiles_lds_grid"] > 0


def test_per_point_planes_match_oracle(pcm, pair10k):
    """Per-point parity: same selected set, bit-identical fitted planes."""
    p = pair10k
    n = len(p.scan)
    o = _oracle(); o.set_input_target(p.submap); o.set_input_source(p.scan)
    g = _gpu(pcm, sort_source=0); g.set_input_target(p.submap); g.set_input_source(p.scan)
    for T in (p.guess.astype(np.float64), p.T_gt):
        o.linearize(T); g.evaluate_cost(T)
        po, so = o.get_planes(n)
        pg = g.get_planes(n)
        sg = ~np.isnan(pg[:, 0])
        assert np.array_equal(so, sg)
        assert np.array_equal(po[so], pg[sg])


def test_sorted_source_same_pose(pcm, pair10k):
    """sort_source only permutes the scan: same normal equations up to summation order."""
    p = pair10k
    a = _gpu(pcm, sort_source=0); a.set_input_target(p.submap); a.set_input_source(p.scan)
    b = _gpu(pcm, sort_source=1); b.set_input_target(p.submap); b.set_input_source(p.scan)
    c0, H0, b0, i0 = a.evaluate_cost(p.T_gt)
    c1, H1, b1, i1 = b.evaluate_cost(p.T_gt)
    assert i0 == i1 and rel_err(H1, H0) < 1e-12 and rel_err(b1, b0) < 1e-10
    ra, rb = a.align(p.guess), b.align(p.guess)
    dt, dr = pose_error(ra.T64, rb.T64)
    assert dt < 1e-9 and dr < 1e-9


def test_batch_matches_single(pcm, synth):
    """pcm_align_batch over ragged, independent pairs == one pcm_align per pair."""
    pairs = [synth.make_pair(10 + i, 4000 + 1500 * i, 40000 + 9000 * i) for i in range(4)]
    regs = []
    for p in pairs:
        g = _gpu(pcm, "LM"); g.set_input_target(p.submap); g.set_input_source(p.scan); regs.append(g)
    singles = [g.align(p.guess) for g, p in zip(regs, pairs)]
    batch = pcm.align_batch(regs, np.stack([p.guess for p in pairs]))
    for s, b in zip(singles, batch):
        assert np.array_equal(s.T64, b.T64)
        assert s.iterations == b.iterations and s.num_linearize == b.num_linearize


def test_call_order_scenarios(pcm, pair10k):
    """The four call orders of the reference's AlignmentTest (gicp_test.cpp:157-200),
    here on a scene where source and target can trade places: two samplings of
    one scene.  Pins the setInput*/swap caching semantics."""
    from oracle.loader import result_T
    p = pair10k
    world_scan = (p.scan[:, :3] @ p.T_gt[:3, :3].T + p.T_gt[:3, 3]).astype(np.float32)
    A = np.ascontiguousarray(p.submap[:, :3]); B = np.ascontiguousarray(world_scan)
    T0 = np.eye(4, dtype=np.float32); T0[:3, 3] = [0.05, -0.03, 0.02]
    g = _gpu(pcm, "LM")
    g.set_input_target(A); g.set_input_source(B)
    fwd = g.align(T0)
    o = _oracle("LM"); o.set_input_target(A); o.set_input_source(B)
    dt, dr = pose_error(result_T(o.align(T0)), fwd.T64)
    assert dt < POSE_TOL_M and dr < POSE_TOL_RAD
    # swap + set source: source:=A then swap -> target=A; set source B
    g2 = _gpu(pcm, "LM")
    g2.set_input_source(A); g2.swap_source_and_target(); g2.set_input_source(B)
    assert np.array_equal(g2.align(T0).T64, fwd.T64)
    # swap + set target
    g3 = _gpu(pcm, "LM")
    g3.set_input_target(B); g3.swap_source_and_target(); g3.set_input_target(A)
    assert np.array_equal(g3.align(T0).T64, fwd.T64)
    # pointer-identity cache: same tag -> no-op even if the buffer content differs
    g4 = _gpu(pcm, "LM")
    g4.set_input_target(A, tag=1234); g4.set_input_source(B, tag=99)
    g4.set_input_target(A[: len(A)].copy() * 0 + 1.0, tag=1234)
    assert np.array_equal(g4.align(T0).T64, fwd.T64)


def test_edge_cases(pcm, synth):
    g = _gpu(pcm)
    with pytest.raises(pcm.PcmError):          # align before inputs
        g.align()
    sc, sm, T = synth.corner_scene(50, 2000, seed=3)
    g.set_input_target(sm)
    with pytest.raises(pcm.PcmError):
        g.align()
    g.set_input_source(sc)
    g.clear_source()
    with pytest.raises(pcm.PcmError):
        g.align()
    # a scan that matches nothing: zero inliers, H = 0, pose stays finite
    far = np.full((64, 3), 1.0e4, np.float32) + np.random.default_rng(0).normal(size=(64, 3)).astype(np.float32)
    g.set_input_source(far)
    c, H, b, inl = g.evaluate_cost(np.eye(4))
    assert inl == 0 and c == 0.0 and not H.any() and not b.any()
    # tiny clouds (fewer than 5 map points in reach -> the double-precision plane path)
    o = _oracle(); o.set_input_target(sm[:7]); o.set_input_source(sc)
    g.set_input_target(sm[:7]); g.set_input_source(sc)
    c0, H0, b0 = o.linearize(T)
    c1, H1, b1, inl = g.evaluate_cost(T)
    assert inl == o.num_inliers
    if inl:
        assert rel_err(H1, H0) < HB_RTOL
    with pytest.raises(pcm.PcmError):
        pcm.P2PlaneRegistration(0, num_neighbors=5)
    with pytest.raises(pcm.PcmError):
        pcm.P2PlaneRegistration(0, voxel_resolution=0.0)


def test_strided_and_device_inputs(pcm, pair10k):
    """PointXYZI-like 32-byte records and device-resident buffers give the same map."""
    import torch
    p = pair10k
    g = _gpu(pcm); g.set_input_target(p.submap); g.set_input_source(p.scan)
    ref = g.evaluate_cost(p.T_gt)
    wide_t = np.zeros((len(p.submap), 8), np.float32); wide_t[:, :3] = p.submap[:, :3]; wide_t[:, 3:] = 7.0
    wide_s = np.zeros((len(p.scan), 12), np.float32); wide_s[:, :3] = p.scan[:, :3]
    g2 = _gpu(pcm); g2.set_input_target(wide_t); g2.set_input_source(wide_s)
    got = g2.evaluate_cost(p.T_gt)
    assert got[3] == ref[3] and np.array_equal(got[1], ref[1])
    g3 = _gpu(pcm)
    g3.set_input_target(torch.from_numpy(wide_t).cuda()); g3.set_input_source(torch.from_numpy(p.scan).cuda())
    got = g3.evaluate_cost(p.T_gt)
    assert got[3] == ref[3] and np.array_equal(got[1], ref[1])


def test_golden_vectors_gpu(pcm, synth):
    """The HIP path against the committed golden vectors (tests/golden, made by tests/make_golden.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "p2plane_config1.npz"))
    for name in sorted({k.split("/")[0] for k in g.files}):
        c = {k.split("/")[1]: g[k] for k in g.files if k.startswith(name + "/")}
        p = synth.make_pair(int(c["seed"]), int(c["n_scan"]), int(c["m_map"]))
        r = _gpu(pcm, str(c["optimizer"]))
        r.set_input_target(p.submap); r.set_input_source(p.scan)
        out = r.align(p.guess)
        dt, dr = pose_error(c["T"], out.T64)
        assert dt < POSE_TOL_M and dr < POSE_TOL_RAD
        assert out.iterations == int(c["iterations"]) and out.num_inliers == int(c["num_inliers"])
        # first linearisation of the trace: H, b, cost
        cost, H, b, inl = r.evaluate_cost(p.guess.astype(np.float64))
        assert rel_err(H, c["trace"][0, 1:37].reshape(6, 6)) < HB_RTOL and rel_err(b, c["trace"][0, 37:43]) < HB_RTOL
        assert abs(cost - c["trace"][0, 0]) <= HB_RTOL * c["trace"][0, 0]


def test_full_size_properties(pcm, synth):
    """BASELINE config 2 size (100k-pt scan vs 1M-pt submap), size-independent properties:
    (1) aligning from the converged pose stays there (idempotence);
    (2) registering the scan moved by a known rigid motion D gives pose * D^-1 (equivariance);
    (3) the batch result equals the single result."""
    p = synth.make_pair(0, 100000, 1000000)
    g = _gpu(pcm, "GN"); g.set_input_target(p.submap); g.set_input_source(p.scan)
    r = g.align(p.guess)
    assert r.converged and r.num_inliers > 0.8 * len(p.scan)
    r2 = g.align(r.T)
    dt, dr = pose_error(r.T64, r2.T64)
    assert r2.iterations <= 2 and dt < 2e-3 and dr < 2e-3
    D = np.eye(4); D[:3, 3] = [0.05, -0.02, 0.01]
    moved = p.scan.copy(); moved[:, :3] = p.scan[:, :3] @ D[:3, :3].T + D[:3, 3]
    g.set_input_source(moved.astype(np.float32))
    r3 = g.align((p.guess.astype(np.float64) @ np.linalg.inv(D)).astype(np.float32))
    dt, dr = pose_error(r.T64, r3.T64 @ D)
    assert dt < 5e-3 and dr < 5e-3
    g.set_input_source(p.scan)
    rb = pcm.align_batch([g], p.guess[None])[0]
    assert np.array_equal(rb.T64, r.T64)


@pytest.mark.parametrize("optimizer", ["GN", "LM"])
def test_batch_window_equals_singles(pcm, synth, optimizer):
    """pcm_config.batch_window (speed only): queued pairs take over the slots of finished ones on the device;
    every pair's result is the one it gets alone."""
    pairs = [synth.make_pair(60 + i, 2000 + 300 * i, 20000 + 2000 * i) for i in range(7)]
    for window in (1, 3):
        regs = []
        for p in pairs:
            g = pcm.P2PlaneRegistration(0, optimizer=optimizer, batch_window=window)
            g.set_input_target(p.submap); g.set_input_source(p.scan); regs.append(g)
        singles = [g.align(p.guess) for g, p in zip(regs, pairs)]
        batch = pcm.align_batch(regs, np.stack([p.guess for p in pairs]))
        for s, b in zip(singles, batch):
            assert np.array_equal(s.T64, b.T64) and s.iterations == b.iterations and s.converged == b.converged


def test_fused_step_equals_separate_step(pcm, synth):
    """With PCM_FLAG_FUSED_STEP (= 2) late GN rounds take the step in the search kernel's last workgroup:
    same sums in the same order, so the results are bit-identical; also against the oracle."""
    from oracle import Oracle
    from oracle.loader import result_T
    pairs = [synth.make_pair(80 + i, 3000 + 700 * i, 30000 + 3000 * i) for i in range(5)]
    out = {}
    for flags in (0, 2):
        regs = []
        for p in pairs:
            g = pcm.P2PlaneRegistration(0, optimizer="GN", flags=flags)
            g.set_input_target(p.submap); g.set_input_source(p.scan); regs.append(g)
        out[flags] = pcm.align_batch(regs, np.stack([p.guess for p in pairs])), [g.align(p.guess) for g, p in zip(regs, pairs)]
    for a, b, sa, sb in zip(out[0][0], out[2][0], out[0][1], out[2][1]):
        assert np.array_equal(a.T64, b.T64) and a.iterations == b.iterations
        assert np.array_equal(sa.T64, a.T64) and np.array_equal(sb.T64, b.T64)
    for p, a in zip(pairs, out[0][0]):
        o = Oracle("P2PLANE", "GN"); o.set_input_target(p.submap); o.set_input_source(p.scan)
        r = o.align(p.guess)
        dt, dr = pose_error(result_T(r), a.T64)
        assert dt < POSE_TOL_M and dr < POSE_TOL_RAD and a.iterations == r.iterations


@pytest.mark.gpu
def test_batch_window_hand_off_with_more_pairs_than_resident_workgroups(pcm, synth):
    """700 small pairs, 5 at a time: the step launch has more workgroups than the device keeps resident at once, so a queued
    pair's workgroup can run AFTER the pair that finished has handed it the slot in the same launch.  The hand-off only marks
    the pair PENDING (its own workgroup promotes it in a later launch), so every pair still gets the result it gets alone."""
    base = [synth.make_pair(90 + i, 600 + 50 * i, 6000 + 500 * i) for i in range(7)]
    n = 700
    regs, guesses = [], []
    for k in range(n):
        p = base[k % len(base)]
        g = pcm.P2PlaneRegistration(0, optimizer="GN", batch_window=5, max_iterations=12)
        g.set_input_target(p.submap); g.set_input_source(p.scan)
        regs.append(g); guesses.append(p.guess)
    batch = pcm.align_batch(regs, np.stack(guesses))
    singles = [regs[k].align(base[k].guess) for k in range(len(base))]
    for k in range(n):
        s, b = singles[k % len(base)], batch[k]
        assert np.array_equal(s.T64, b.T64) and s.iterations == b.iterations and s.num_linearize == b.num_linearize, k


@pytest.mark.gpu
@pytest.mark.parametrize("optimizer", ["GN", "LM"])
def test_host_window_equals_singles(pcm, synth, optimizer):
    """batch_window with up to 256 pairs and a window <= 64 is kept by the host: only the launch list's pairs run and a finished
    pair's place goes to the next queued one.  Every pair's result is the one it gets alone, whatever the window."""
    base = [synth.make_pair(90 + i, 900 + 150 * i, 9000 + 1500 * i) for i in range(6)]
    n = 40
    for window in (1, 7, 16):
        regs, guesses = [], []
        for k in range(n):
            p = base[k % len(base)]
            g = pcm.P2PlaneRegistration(0, optimizer=optimizer, batch_window=window, max_iterations=20)
            g.set_input_target(p.submap); g.set_input_source(p.scan)
            regs.append(g); guesses.append(p.guess)
        batch = pcm.align_batch(regs, np.stack(guesses))
        singles = [regs[k].align(base[k].guess) for k in range(len(base))]
        for k in range(n):
            s, b = singles[k % len(base)], batch[k]
            assert np.array_equal(s.T64, b.T64) and s.iterations == b.iterations and s.num_linearize == b.num_linearize, (window, k)
            assert s.num_compute_error == b.num_compute_error and s.converged == b.converged


@pytest.mark.parametrize("window,n,max_it", [(1, 24, 3), (2, 60, 5), (8, 100, 5)])
def test_host_window_with_pairs_that_never_converge(pcm, synth, window, n, max_it):
    """Round budget of the host-kept window (round-2 advisor finding): with every pair running to max_iterations each slot
    hand-over costs one stale round on top; the old budget ran out and returned unfinished pairs as PCM_OK.  Every pair must
    come back finished (status OK, max_it iterations, the pose it gets alone)."""
    base = [synth.make_pair(120 + i, 700 + 90 * i, 7000 + 900 * i) for i in range(4)]
    regs, guesses = [], []
    for k in range(n):
        p = base[k % len(base)]
        g = pcm.P2PlaneRegistration(0, optimizer="GN", batch_window=window, max_iterations=max_it, rotation_eps=1e-12, translation_eps=1e-12)
        g.set_input_target(p.submap); g.set_input_source(p.scan)
        regs.append(g); guesses.append(p.guess)
    batch = pcm.align_batch(regs, np.stack(guesses))
    singles = [regs[k].align(base[k].guess) for k in range(len(base))]
    for k in range(n):
        s, b = singles[k % len(base)], batch[k]
        # nr_iterations_ = index of the last iteration (lsq_registration_impl.hpp:63-64)
        assert b.status == 0 and not b.converged and b.iterations == max_it - 1 and b.num_linearize == max_it, (k, b.status, b.iterations)
        assert np.array_equal(s.T64, b.T64)
