"""GPU parity of the NDT models (P2D / D2D on Gaussian voxels) against the oracle."""
import numpy as np
import pytest

from helpers import HB_RTOL, POSE_TOL_M, POSE_TOL_RAD, pose_error, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair_dense(synth):
    # NDT voxels need > 6 points (ndt_compute_derivatives.cu:61): a denser submap than the LIO-like default
    return synth.make_pair(0, 10000, 100000, density=60.0)


def _both(pcm, model, nn, optimizer, p, res=1.0):
    from oracle import Oracle
    o = Oracle(model, optimizer, voxel_resolution=res, num_neighbors=nn)
    o.set_input_target(p.submap); o.set_input_source(p.scan)
    g = pcm.NdtRegistration(0, model=model, optimizer=optimizer, voxel_resolution=res, num_neighbors=nn)
    g.set_input_target(p.submap); g.set_input_source(p.scan)
    return o, g


@pytest.mark.parametrize("model", ["NDT_P2D", "NDT_D2D"])
@pytest.mark.parametrize("nn", [1, 7, 27])
def test_ndt_linearize_matches_oracle(pcm, pair_dense, model, nn):
    p = pair_dense
    o, g = _both(pcm, model, nn, "LM", p)
    for T in (p.guess.astype(np.float64), p.T_gt):
        c0, H0, b0 = o.linearize(T)
        c1, H1, b1, inl = g.evaluate_cost(T)
        assert inl == o.num_inliers and inl > 0
        assert rel_err(H1, H0) < HB_RTOL and rel_err(b1, b0) < HB_RTOL and abs(c1 - c0) <= HB_RTOL * abs(c0)
        T2 = T.copy(); T2[:3, 3] += [0.02, -0.01, 0.01]     # trial pose on the remembered correspondences
        assert abs(g.compute_error(T2) - o.compute_error(T2)) <= HB_RTOL * abs(o.compute_error(T2))


@pytest.mark.parametrize("model,nn,optimizer", [("NDT_P2D", 1, "LM"), ("NDT_D2D", 7, "LM"), ("NDT_D2D", 1, "GN"), ("NDT_P2D", 7, "GN")])
def test_ndt_align_matches_oracle(pcm, pair_dense, model, nn, optimizer):
    from oracle.loader import result_T
    p = pair_dense
    o, g = _both(pcm, model, nn, optimizer, p)
    ro, rg = o.align(p.guess), g.align(p.guess)
    dt, dr = pose_error(result_T(ro), rg.T64)
    assert dt < POSE_TOL_M and dr < POSE_TOL_RAD
    assert rg.iterations == ro.iterations and rg.converged == bool(ro.converged)
    assert rg.num_linearize == ro.num_linearize and rg.num_compute_error == ro.num_compute_error


@pytest.mark.parametrize("model", ["NDT_P2D", "NDT_D2D"])
@pytest.mark.parametrize("radius,count", [(1.0, 7), (1.5, 19), (2.0, 33)])
def test_ndt_direct_radius_matches_oracle(pcm, pair_dense, model, radius, count):
    """NeighborSearchMethod::DIRECT_RADIUS (ndt_cuda.cu:70-83): every voxel offset with |offset| <= radius + 1e-3, radius in
    voxels -- 7, 19 and 33 offsets for radius 1, 1.5 and 2."""
    from oracle.loader import result_T
    p = pair_dense
    rng = range(-int(np.ceil(radius)), int(np.ceil(radius)) + 1)
    assert sum(1 for i in rng for j in rng for k in rng if np.sqrt(i * i + j * j + k * k) <= radius + 1e-3) == count
    o, _ = _both(pcm, model, 7, "LM", p, res=0.5)
    o.set_neighbor_radius(radius)
    g = pcm.NdtRegistration(0, model=model, optimizer="LM", voxel_resolution=0.5, neighbor_search_radius=radius)
    g.set_input_target(p.submap); g.set_input_source(p.scan)
    T = p.guess.astype(np.float64)
    c0, H0, b0 = o.linearize(T)
    c1, H1, b1, inl = g.evaluate_cost(T)
    assert inl == o.num_inliers and inl > 0
    assert rel_err(H1, H0) < HB_RTOL and rel_err(b1, b0) < HB_RTOL and abs(c1 - c0) <= HB_RTOL * abs(c0)
    T2 = T.copy(); T2[:3, 3] += [0.02, -0.01, 0.01]
    assert abs(g.compute_error(T2) - o.compute_error(T2)) <= HB_RTOL * abs(o.compute_error(T2))
    if radius == 1.0:   # the same seven offsets as DIRECT7, in another order: the same correspondences
        o7, _ = _both(pcm, model, 7, "LM", p, res=0.5)
        o7.linearize(T)
        assert o7.num_inliers == inl
    ro, rg = o.align(p.guess), g.align(p.guess)
    dt, dr = pose_error(result_T(ro), rg.T64)
    assert dt < POSE_TOL_M and dr < POSE_TOL_RAD
    assert rg.iterations == ro.iterations and rg.num_linearize == ro.num_linearize and rg.num_compute_error == ro.num_compute_error


def test_direct_radius_is_refused_where_the_reference_has_none(pcm):
    with pytest.raises(pcm.PcmError):
        pcm.P2PlaneRegistration(0, neighbor_search_radius=1.5)      # "supported on only VGICP_CUDA" (+ NDTCuda)
    with pytest.raises(pcm.PcmError):
        pcm.NdtRegistration(0, neighbor_search_radius=4.0)           # the cube walked per element is capped at radius 3


def test_ndt_batch_and_swap(pcm, synth):
    pairs = [synth.make_pair(20 + i, 3000 + 1000 * i, 30000 + 8000 * i, density=60.0) for i in range(3)]
    regs = []
    for p in pairs:
        g = pcm.NdtRegistration(0); g.set_input_target(p.submap); g.set_input_source(p.scan); regs.append(g)
    singles = [g.align(p.guess) for g, p in zip(regs, pairs)]
    batch = pcm.align_batch(regs, np.stack([p.guess for p in pairs]))
    for s, b in zip(singles, batch):
        assert np.array_equal(s.T64, b.T64)
    # swapSourceAndTarget (ndt_cuda.cu:90-93) then setting both again == fresh object
    p = pairs[0]
    g = pcm.NdtRegistration(0); g.set_input_target(p.scan); g.set_input_source(p.submap)
    g.swap_source_and_target()
    assert np.array_equal(g.align(p.guess).T64, singles[0].T64)


@pytest.mark.parametrize("model", ["NDT_P2D", "NDT_D2D", "VGICP_CUDA"])
@pytest.mark.parametrize("nn", [1, 7, 27])
def test_neighbour_voxel_rows_change_nothing(pcm, pair_dense, model, nn):
    """PCM_FLAG_NEIGHBOUR_LISTS (16): k_ndt reads, for the voxel a point falls into, the row of its neighbour voxels built with the
    target (64 = never: every cell is looked up): the same correspondences, hence identical sums, trial costs and poses; the
    default builds the rows at a target's second registration."""
    p = pair_dense
    cls = pcm.VgicpCudaRegistration if model == "VGICP_CUDA" else pcm.NdtRegistration
    kw = dict(voxel_resolution=1.0, num_neighbors=nn)
    if model != "VGICP_CUDA":
        kw["model"] = model
    regs = {f: cls(0, flags=f, **kw) for f in (16, 64, 0)}
    for g in regs.values():
        g.set_input_target(p.submap); g.set_input_source(p.scan)
    for T in (p.guess.astype(np.float64), p.T_gt):
        ra, rb = regs[16].evaluate_cost(T), regs[64].evaluate_cost(T)
        assert ra[0] == rb[0] and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2]) and ra[3] == rb[3]
        T2 = T.copy(); T2[:3, 3] += [0.02, -0.01, 0.01]
        assert regs[16].compute_error(T2) == regs[64].compute_error(T2)
    ra, rb = regs[16].align(p.guess), regs[64].align(p.guess)
    rc = [regs[0].align(p.guess) for _ in range(3)][-1]
    for r in (ra, rc):
        assert np.array_equal(r.T64, rb.T64) and r.iterations == rb.iterations and r.num_inliers == rb.num_inliers
