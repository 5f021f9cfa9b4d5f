"""GPU parity of the jueying_lio measurement model (ObsModel + IEKF reduction) against the oracle."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as R

from helpers import HB_RTOL, rel_err

pytestmark = pytest.mark.gpu


def _state(T_wl, off_rpy=(0.01, -0.02, 0.03), off_t=(0.1713, 0.0, 0.05925)):   # extrinsic_T of config/livox.yaml:22
    offR = R.from_euler("xyz", off_rpy)
    Til = np.eye(4); Til[:3, :3] = offR.as_matrix(); Til[:3, 3] = off_t
    Twi = T_wl @ np.linalg.inv(Til)
    return R.from_matrix(Twi[:3, :3]).as_quat(), Twi[:3, 3], offR.as_quat(), np.asarray(off_t)


@pytest.mark.parametrize("extrinsic", [False, True])
def test_obs_model_matches_oracle(pcm, synth, extrinsic):
    from oracle import Oracle
    p = synth.make_pair(0, 10000, 100000)
    o = Oracle("P2PLANE", "GN", voxel_resolution=0.5, num_neighbors=27)
    o.set_input_target(p.submap); o.set_input_source(p.scan)
    g = pcm.P2PlaneRegistration(0, voxel_resolution=0.5, num_neighbors=27)
    g.set_input_target(p.submap); g.set_input_source(p.scan)
    # iteration -1: re-match at the propagated state; then an update without re-matching (converge = false)
    T0 = p.guess.astype(np.float64)
    T1 = T0.copy(); T1[:3, 3] += [0.03, -0.02, 0.01]
    for T, conv in ((T0, True), (T1, False), (p.T_gt, True)):
        st = _state(T)
        H0, h0, n0, s0 = o.obs_model(*st, extrinsic, conv)
        H1, h1, n1, s1, valid = g.obs_model(*st, extrinsic, conv)
        assert n1 == n0 and valid and n1 > 1000
        assert rel_err(H1, H0) < HB_RTOL and rel_err(h1, h0) < HB_RTOL and abs(s1 - s0) <= HB_RTOL * s0
        if not extrinsic:
            assert not H1[6:, :].any() and not h1[6:].any()


def test_obs_model_consistent_with_linearize(pcm, synth):
    """Same matcher as the LsqRegistration operator: identical selected set and cost; the 6x6
    blocks are related by the change of perturbation frame (world-left vs imu-right)."""
    p = synth.make_pair(1, 8000, 80000)
    g = pcm.P2PlaneRegistration(0, voxel_resolution=0.5, num_neighbors=27)
    g.set_input_target(p.submap); g.set_input_source(p.scan)
    T = p.T_gt
    cost, H, b, inl = g.evaluate_cost(T)
    st = _state(T, off_rpy=(0, 0, 0), off_t=(0, 0, 0))
    HTH, HTh, n, s2, valid = g.obs_model(*st, False, True)
    assert n == inl and abs(s2 - cost) <= 1e-4 * cost
    # translation block: sum n n^T in both parameterisations
    assert rel_err(HTH[:3, :3], H[3:, 3:]) < 1e-5


def test_obs_model_errors(pcm, synth):
    p = synth.make_pair(2, 2000, 20000)
    g = pcm.P2PlaneRegistration(0)
    g.set_input_target(p.submap); g.set_input_source(p.scan)
    st = _state(p.T_gt)
    with pytest.raises(pcm.PcmError):
        g.obs_model(*st, False, False)          # converge = false before any matching
    far = np.full((64, 3), 1.0e4, np.float32)
    g.set_input_source(far)
    H, h, n, s2, valid = g.obs_model(*st, False, True)
    assert n == 0 and not valid                 # "No Effective Points!" laser_mapping.cc:657-661


def test_sliding_map_add_filter_matches_oracle(pcm, synth):
    """Three LIO frames: match -> MapIncremental (add-filter against the matched neighbours) ->
    rebuilt voxel hash; the map content must equal the oracle's point for point."""
    from oracle import Oracle
    scene = synth.scene_for_points(1234, 60000, 8.0)
    submap = synth.sample_submap(scene, 60000, 4321)
    o = Oracle("P2PLANE", "GN", voxel_resolution=0.5, num_neighbors=27)
    g = pcm.P2PlaneRegistration(0, voxel_resolution=0.5, num_neighbors=27, sort_source=0)
    o.set_input_target(submap); g.set_input_target(submap)
    T = synth.sensor_pose(scene, 77)
    for f in range(3):
        Tf = T.copy(); Tf[:3, 3] += Tf[:3, 0] * 0.6 * f          # the sensor advances 0.6 m per frame
        scan, _ = synth.livox_scan(scene, Tf, 4000, 555 + f)
        st = _state(Tf)
        o.set_input_source(scan); g.set_input_source(scan)
        H0, h0, n0, s0 = o.obs_model(*st, False, True)
        H1, h1, n1, s1, valid = g.obs_model(*st, False, True)
        assert n0 == n1 and rel_err(H1, H0) < HB_RTOL
        a0 = o.map_incremental(*st, 0.5, True)
        a1 = g.map_incremental(*st, 0.5, True)
        assert a0 == a1 and 0 < a1 < len(scan)                   # the filter keeps some points and rejects some
        assert np.array_equal(g.get_target(), o.get_target())
    # first frame of a run (flg_EKF_inited_ false): every point is added
    scan, _ = synth.livox_scan(scene, T, 1000, 999)
    o.set_input_source(scan); g.set_input_source(scan)
    assert o.map_incremental(*_state(T), 0.5, False) == g.map_incremental(*_state(T), 0.5, False) == 1000
    assert np.array_equal(g.get_target(), o.get_target())


def test_sliding_map_lru_eviction_matches_oracle(pcm, synth):
    """IVox LRU (ivox3d.h:256-281): inserting new territory beyond the capacity drops the
    least-recently-touched voxels; a later touch protects a voxel."""
    from oracle import Oracle
    rng = np.random.default_rng(3)
    base = rng.uniform(0, 20, (3000, 3)).astype(np.float32)          # ~2500 voxels of 0.5 m... sparse cloud
    cap = 4000
    o = Oracle("P2PLANE", "GN", voxel_resolution=0.5, num_neighbors=27, map_capacity=cap)
    g = pcm.P2PlaneRegistration(0, voxel_resolution=0.5, num_neighbors=27, map_capacity=cap)
    o.set_input_target(base); g.set_input_target(base)
    g.set_input_source(base[:10]); o.set_input_source(base[:10])
    for k in range(4):
        touch = base[100 * k:100 * k + 50] + np.float32(0.01)         # re-touch some old voxels first ...
        new = (rng.uniform(0, 20, (900, 3)) + [30 * (k + 1), 0, 0]).astype(np.float32)   # ... then new territory
        for batch in (touch, new):
            o.target_insert(batch); g.target_insert(batch)
        got, want = g.get_target(), o.get_target()
        assert o.target_voxels <= cap - 1
        assert np.array_equal(got, want)
    assert len(want) < 3000 + 4 * 950                                  # something was evicted
