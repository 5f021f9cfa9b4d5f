"""PCM_FLAG_REFERENCE_KNN_ORDER: the neighbours reach the plane fit in the row order of the reference's IVox::GetClosestPoint
(libstdc++ std::nth_element, ivox3d.h:173-178 / ivox3d_node.hpp:176-181).  GPU (nth_select.h restatement on the device) against the
oracle in ORC_KNN_ORDER_LIBSTDCXX mode (the container's real std::nth_element): same planes bit for bit.  ``-m gpu``."""
import numpy as np
import pytest

from helpers import HB_RTOL, POSE_TOL_M, POSE_TOL_RAD, pose_error, rel_err

pytestmark = pytest.mark.gpu

REF_ORDER = 32


def _pair(pcm, p, optimizer="GN", nn=27, flags=REF_ORDER, order="libstdcxx"):
    from oracle import Oracle
    o = Oracle("P2PLANE", optimizer, voxel_resolution=0.5, num_neighbors=nn); o.set_knn_order(order)
    o.set_input_target(p.submap); o.set_input_source(p.scan)
    g = pcm.P2PlaneRegistration(0, optimizer=optimizer, voxel_resolution=0.5, num_neighbors=nn, flags=flags, sort_source=0)
    g.set_input_target(p.submap); g.set_input_source(p.scan)
    return o, g


@pytest.mark.parametrize("nn", [7, 27])
def test_planes_bit_identical_to_the_oracle_in_reference_order(pcm, synth, nn):
    p = synth.make_pair(0, 10000, 100000)
    n = len(p.scan)
    o, g = _pair(pcm, p, nn=nn)
    oa, ga = _pair(pcm, p, nn=nn, flags=0, order="ascending")
    differs = 0
    for T in (p.guess.astype(np.float64), p.T_gt):
        c0, H0, b0 = o.linearize(T)
        c1, H1, b1, inl = g.evaluate_cost(T)
        po, so = o.get_planes(n)
        pg = g.get_planes(n)
        sg = ~np.isnan(pg[:, 0])
        assert np.array_equal(so, sg) and np.array_equal(po[so], pg[sg])
        assert inl == o.num_inliers and rel_err(H1, H0) < 1e-9 and rel_err(b1, b0) < 1e-9 and abs(c1 - c0) <= 1e-9 * abs(c0)
        ga.evaluate_cost(T)
        pa = ga.get_planes(n)
        both = sg & ~np.isnan(pa[:, 0])
        differs += int(np.sum(np.any(pa[both] != pg[both], axis=1)))
    assert differs > 0   # the mode is live: the ascending order gives other last bits on some planes


@pytest.mark.parametrize("optimizer", ["GN", "LM"])
def test_align_equals_the_oracle_in_reference_order(pcm, synth, optimizer):
    from oracle.loader import result_T
    for seed in (1, 2):
        p = synth.make_pair(seed, 10000, 100000)
        o, g = _pair(pcm, p, optimizer)
        ro, rg = o.align(p.guess), g.align(p.guess)
        dt, dr = pose_error(result_T(ro), rg.T64)
        assert dt < 1e-9 and dr < 1e-9
        assert rg.iterations == ro.iterations and rg.num_inliers == ro.num_inliers and rg.num_linearize == ro.num_linearize
        assert rg.num_compute_error == ro.num_compute_error and rg.converged == bool(ro.converged)


def test_batch_in_reference_order_equals_singles(pcm, synth):
    pairs = [synth.make_pair(60 + i, 3000 + 500 * i, 30000 + 4000 * i) for i in range(3)]
    regs = []
    for p in pairs:
        g = pcm.P2PlaneRegistration(0, optimizer="GN", flags=REF_ORDER)
        g.set_input_target(p.submap); g.set_input_source(p.scan); regs.append(g)
    batch = pcm.align_batch(regs, np.stack([p.guess for p in pairs]))
    for g, p, b in zip(regs, pairs, batch):
        s = g.align(p.guess)
        assert np.array_equal(s.T64, b.T64) and s.iterations == b.iterations


def test_dense_voxels_are_refused_not_truncated(pcm, synth):
    sc, sm, T = synth.corner_scene(1000, 340000, seed=5, noise=0.01)   # > 121 points in a 0.5 m voxel
    g = pcm.P2PlaneRegistration(0, optimizer="GN", flags=REF_ORDER)
    g.set_input_target(sm); g.set_input_source(sc)
    with pytest.raises(pcm.PcmError) as e:
        g.align(T.astype(np.float32))
    assert e.value.code == -4 and "121" in str(e.value)
