"""k_linearize_counted (counted cells: every candidate's address known up front, four per cell and trip) against the round-2
default search kernel k_linearize and the oracle: the same neighbours in the same order as the per-cell walk
(jueying_lio ivox3d.h:132-204), hence bit-identical planes, sums and poses.  Run on the MI355X box with ``-m gpu``.
"""
import numpy as np
import pytest

from helpers import HB_RTOL, POSE_TOL_M, POSE_TOL_RAD, pose_error, rel_err

pytestmark = pytest.mark.gpu

COUNTED, NO_LDS = 8, 1


def _reg(pcm, p_map, p_scan, optimizer="GN", **kw):
    g = pcm.P2PlaneRegistration(0, optimizer=optimizer, **kw)
    g.set_input_target(p_map); g.set_input_source(p_scan)
    return g


def _same_linearize(a, b, T, n):
    ra, rb = a.evaluate_cost(T), b.evaluate_cost(T)
    pa, pb = a.get_planes(n), b.get_planes(n)
    assert np.array_equal(np.isnan(pa[:, 0]), np.isnan(pb[:, 0]))
    ok = ~np.isnan(pa[:, 0])
    assert np.array_equal(pa[ok], pb[ok])
    assert ra[3] == rb[3] and ra[0] == rb[0] and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2])


@pytest.mark.parametrize("nn", [1, 7, 19, 27])
@pytest.mark.parametrize("sort_source", [0, 1])
def test_counted_equals_default_linearize(pcm, synth, nn, sort_source):
    p = synth.make_pair(3, 12000, 120000)
    n = len(p.scan)
    a = _reg(pcm, p.submap, p.scan, num_neighbors=nn, sort_source=sort_source, flags=COUNTED)
    b = _reg(pcm, p.submap, p.scan, num_neighbors=nn, sort_source=sort_source)
    for T in (p.T_gt, p.guess.astype(np.float64)):
        if sort_source:   # the device order of the scan is fixed by the first align's guess: same for both objects
            a.align(p.guess); b.align(p.guess)
        _same_linearize(a, b, T, n)


@pytest.mark.parametrize("optimizer", ["GN", "LM"])
def test_kernels_give_identical_aligns(pcm, synth, optimizer):
    """Same poses bit for bit with the default kernel; single and batched; ragged scans whose last tile is partial."""
    pairs = [synth.make_pair(40 + i, 5000 + 2500 * i, 50000 + 20000 * i) for i in range(4)]
    guesses = np.stack([p.guess for p in pairs])
    res = {}
    for flags in (COUNTED, 0):
        regs = [_reg(pcm, p.submap, p.scan, optimizer, flags=flags) for p in pairs]
        res[flags] = (pcm.align_batch(regs, guesses), [g.align(p.guess) for g, p in zip(regs, pairs)])
    for k in range(len(pairs)):
        base = res[0][0][k]
        for r in (res[COUNTED][0][k], res[COUNTED][1][k]):
            assert np.array_equal(r.T64, base.T64) and np.array_equal(r.H, base.H)
            assert r.iterations == base.iterations and r.num_inliers == base.num_inliers and r.num_linearize == base.num_linearize
            assert r.num_compute_error == base.num_compute_error and r.cost == base.cost


def test_partial_tiles_with_dead_lanes_in_every_position(pcm, synth):
    """Scan sizes that leave the last tile with 1 ... 255 live lanes: lanes 0..26 of a wave hand the cell offsets to the others
    through v_readlane whether or not they hold a query point themselves (a first form read registers dead lanes never wrote)."""
    p = synth.make_pair(44, 2048, 30000)
    for n in (257, 264, 300, 330, 383, 449, 511, 513, 1000, 2047):
        a = _reg(pcm, p.submap, p.scan[:n], sort_source=0, flags=COUNTED)
        b = _reg(pcm, p.submap, p.scan[:n], sort_source=0)
        _same_linearize(a, b, p.T_gt, n)


@pytest.mark.parametrize("m_map,res", [(130000, 0.5), (340000, 0.5), (60000, 2.0)])
def test_dense_voxels(pcm, synth, m_map, res):
    """Voxels with tens to hundreds of points (the batch loop behind the first four candidates of a cell) and voxels with more than
    255 points (the tile searches the global structures): same planes and sums as the default kernel and the global path, oracle parity."""
    from oracle import Oracle
    sc, sm, T = synth.corner_scene(3000, m_map, seed=5, noise=0.01)
    n = len(sc)
    a = _reg(pcm, sm, sc, voxel_resolution=res, flags=COUNTED)
    b = _reg(pcm, sm, sc, voxel_resolution=res)
    c = _reg(pcm, sm, sc, voxel_resolution=res, flags=NO_LDS)
    o = Oracle("P2PLANE", "GN", voxel_resolution=res, num_neighbors=27); o.set_input_target(sm); o.set_input_source(sc)
    G = T.copy(); G[:3, 3] += [0.03, -0.02, 0.04]
    for X in (T, G):
        _same_linearize(a, b, X, n)
        _same_linearize(a, c, X, n)
        c0, H0, b0 = o.linearize(X)
        c1, H1, b1, inl = a.evaluate_cost(X)
        assert inl == o.num_inliers and rel_err(H1, H0) < HB_RTOL and rel_err(b1, b0) < HB_RTOL
    ra, rb = a.align(G.astype(np.float32)), b.align(G.astype(np.float32))
    assert np.array_equal(ra.T64, rb.T64) and ra.iterations == rb.iterations
    dt, dr = pose_error(T, ra.T64)
    assert dt < 5e-3 and dr < 5e-3


def test_sparse_and_ragged_tiles(pcm, synth):
    """Few neighbours (3- and 4-point double-precision fits queue from the back of the fit table), empty neighbourhoods, a scan
    that is not a multiple of the tile size, lanes outside the key range."""
    from oracle import Oracle
    p = synth.make_pair(7, 3001, 9000, density=1.5)   # a sparse map (1.5 points per square metre): many points with fewer than five neighbours
    sc = p.scan.copy()
    sc[5, :3] = [1e7, -1e7, 1e7]         # outside the voxel key range
    sc[6, :3] = np.nan
    n = len(sc)
    for nn in (7, 27):
        a = _reg(pcm, p.submap, sc, num_neighbors=nn, sort_source=0, flags=COUNTED)
        b = _reg(pcm, p.submap, sc, num_neighbors=nn, sort_source=0)
        o = Oracle("P2PLANE", "GN", voxel_resolution=0.5, num_neighbors=nn); o.set_input_target(p.submap); o.set_input_source(sc)
        for X in (p.T_gt, p.guess.astype(np.float64)):
            _same_linearize(a, b, X, n)
            o.linearize(X); a.evaluate_cost(X)
            po, so = o.get_planes(n)
            pg = a.get_planes(n)
            sg = ~np.isnan(pg[:, 0])
            assert np.array_equal(so, sg) and np.array_equal(po[so], pg[sg])
