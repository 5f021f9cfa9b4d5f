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


@pytest.mark.parametrize("sort_source", [0, 1])
def test_obs_model_reference_semantics_across_calls_and_frames(pcm, synth, sort_source):
    """PCM_FLAG_LIO_REFERENCE_SEMANTICS: residuals_ / point_selected_surf_ live on as LaserMapping's members do
    (laser_mapping.cc:335-339, 616-636).  Three frames of different sizes, several ObsModel calls each with poses far
    enough from the truth that many selected points fail `|p_body| > 81 pd2^2`: in the reference such a point stays
    selected and contributes the residual stored for its index earlier (0 if never, or the one a previous FRAME left);
    in the clean mode it is dropped.  The two modes must differ, and the GPU must equal the oracle in both."""
    from oracle import Oracle
    scene = synth.scene_for_points(4242, 60000, 8.0)
    submap = synth.sample_submap(scene, 60000, 99)
    T = synth.sensor_pose(scene, 5)
    kw = dict(voxel_resolution=0.5, num_neighbors=27)
    o_ref = Oracle("P2PLANE", "GN", **kw); o_ref.set_lio_reference_semantics(True)
    o_cln = Oracle("P2PLANE", "GN", **kw)
    g_ref = pcm.P2PlaneRegistration(0, sort_source=sort_source, flags=pcm.capi.PCM_FLAG_LIO_REFERENCE_SEMANTICS, **kw)
    g_cln = pcm.P2PlaneRegistration(0, sort_source=sort_source, **kw)
    for r in (o_ref, o_cln, g_ref, g_cln):
        r.set_input_target(submap)
    differ = 0
    stale_used = 0
    for f, npts in enumerate((5000, 3000, 6000)):       # shrink, then grow past the first frame: resize() semantics
        scan, _ = synth.livox_scan(scene, T, npts, 700 + f)
        for r in (o_ref, o_cln, g_ref, g_cln):
            r.set_input_source(scan)
        n = scan.shape[0]
        # the IEKF's pattern (esekfom.hpp:1529,1722-1732): the first call matches; later ones may not
        steps = ((0.45, True), (0.05, False), (0.40, False), (0.30, True), (0.02, False))
        for k, (off, conv) in enumerate(steps):
            Tk = T.copy(); Tk[:3, 3] += off * np.array([0.6, -0.5, 0.62]) * (1 if (k + f) % 2 == 0 else -1)
            st = _state(Tk)
            res_before = o_ref.get_lio_members(n)[1].copy()
            ref_o = o_ref.obs_model(*st, True, conv)
            cln_o = o_cln.obs_model(*st, True, conv)
            ref_g = g_ref.obs_model(*st, True, conv)
            cln_g = g_cln.obs_model(*st, True, conv)
            for (H0, h0, n0, s0), (H1, h1, n1, s1, valid) in ((ref_o, ref_g), (cln_o, cln_g)):
                assert n1 == n0 and valid and n0 > 500
                assert rel_err(H1, H0) < HB_RTOL and rel_err(h1, h0) < HB_RTOL and abs(s1 - s0) <= HB_RTOL * max(s0, 1e-30)
            pl_o, res_o, sel_o = o_ref.get_lio_members(n)
            res_g, sel_g = g_ref.get_lio_members(n)
            assert np.array_equal(sel_g, sel_o)
            assert np.array_equal(res_g.view(np.uint32), res_o.view(np.uint32))      # residuals_: bit for bit
            differ += ref_o[2] != cln_o[2]
            # a selected point whose residual did not move in this call contributed a stale one
            stale = sel_o & (res_o == res_before) & (res_o != 0)
            stale_used += int(stale.sum())
            assert ref_o[2] >= cln_o[2]
    assert differ >= 6          # the modes really are different operators on this sequence
    assert stale_used > 100     # and non-zero residuals of earlier calls / frames were re-used


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


@pytest.mark.parametrize("sort_source", [0, 1])      # the device's re-ordering of the scan must not show in the map
def test_sliding_map_add_filter_matches_oracle(pcm, synth, sort_source):
    """Three LIO frames: match -> MapIncremental (add-filter against the matched neighbours) ->
    rebuilt voxel hash; the map content must equal the oracle's point for point."""
    from oracle import Oracle
    scene = synth.scene_for_points(1234, 60000, 8.0)
    submap = synth.sample_submap(scene, 60000, 4321)
    o = Oracle("P2PLANE", "GN", voxel_resolution=0.5, num_neighbors=27)
    g = pcm.P2PlaneRegistration(0, voxel_resolution=0.5, num_neighbors=27, sort_source=sort_source)
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
