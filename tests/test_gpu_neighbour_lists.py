"""k_linearize_lists (PCM_FLAG_NEIGHBOUR_LISTS: the candidate list of every voxel a query can fall into, built with the map in the
reference's visit order, jueying_lio ivox3d.h:132-235; the pass walks one flat run per query) against the default search kernel
k_linearize and the oracle: the same candidates in the same order into the same 5-best list, hence bit-identical planes, sums and
poses.  Run on the MI355X box with ``-m gpu``.
"""
import numpy as np
import pytest

from helpers import HB_RTOL, POSE_TOL_M, POSE_TOL_RAD, pose_error, rel_err

pytestmark = pytest.mark.gpu

LISTS, NO_LDS, TILE = 16, 1, 64   # TILE = PCM_FLAG_NO_NEIGHBOUR_LISTS: the tile kernel whatever the number of registrations


def _reg(pcm, p_map, p_scan, optimizer="GN", **kw):
    kw.setdefault("flags", TILE)
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
def test_lists_equal_default_linearize(pcm, synth, nn, sort_source):
    p = synth.make_pair(3, 12000, 120000)
    n = len(p.scan)
    a = _reg(pcm, p.submap, p.scan, num_neighbors=nn, sort_source=sort_source, flags=LISTS)
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
    for flags in (LISTS, TILE):
        regs = [_reg(pcm, p.submap, p.scan, optimizer, flags=flags) for p in pairs]
        res[flags] = (pcm.align_batch(regs, guesses), [g.align(p.guess) for g, p in zip(regs, pairs)])
    for k in range(len(pairs)):
        base = res[TILE][0][k]
        for r in (res[LISTS][0][k], res[LISTS][1][k]):
            assert np.array_equal(r.T64, base.T64) and np.array_equal(r.H, base.H)
            assert r.iterations == base.iterations and r.num_inliers == base.num_inliers and r.num_linearize == base.num_linearize
            assert r.num_compute_error == base.num_compute_error and r.cost == base.cost


def test_partial_tiles_with_dead_lanes_in_every_position(pcm, synth):
    """Scan sizes that leave the last tile with 1 ... 255 live lanes."""
    p = synth.make_pair(44, 2048, 30000)
    for n in (257, 264, 300, 330, 383, 449, 511, 513, 1000, 2047):
        a = _reg(pcm, p.submap, p.scan[:n], sort_source=0, flags=LISTS)
        b = _reg(pcm, p.submap, p.scan[:n], sort_source=0)
        _same_linearize(a, b, p.T_gt, n)


@pytest.mark.parametrize("m_map,res", [(130000, 0.5), (340000, 0.5), (60000, 2.0)])
def test_dense_voxels(pcm, synth, m_map, res):
    """Voxels with tens to hundreds of points (lists of thousands of candidates): same planes and sums as the default kernel and
    the global path, oracle parity."""
    from oracle import Oracle
    sc, sm, T = synth.corner_scene(3000, m_map, seed=5, noise=0.01)
    n = len(sc)
    a = _reg(pcm, sm, sc, voxel_resolution=res, flags=LISTS)
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
    """Few neighbours (3- and 4-point double-precision fits), empty neighbourhoods (voxels without a list), a scan that is not a
    multiple of the tile size, lanes outside the key range."""
    from oracle import Oracle
    p = synth.make_pair(7, 3001, 9000, density=1.5)   # a sparse map (1.5 points per square metre): many points with fewer than five neighbours
    sc = p.scan.copy()
    sc[5, :3] = [1e7, -1e7, 1e7]         # outside the voxel key range
    sc[6, :3] = np.nan
    n = len(sc)
    for nn in (7, 27):
        a = _reg(pcm, p.submap, sc, num_neighbors=nn, sort_source=0, flags=LISTS)
        b = _reg(pcm, p.submap, sc, num_neighbors=nn, sort_source=0)
        o = Oracle("P2PLANE", "GN", voxel_resolution=0.5, num_neighbors=nn); o.set_input_target(p.submap); o.set_input_source(sc)
        for X in (p.T_gt, p.guess.astype(np.float64)):
            _same_linearize(a, b, X, n)
            o.linearize(X); a.evaluate_cost(X)
            po, so = o.get_planes(n)
            pg = a.get_planes(n)
            sg = ~np.isnan(pg[:, 0])
            assert np.array_equal(so, sg) and np.array_equal(po[so], pg[sg])


def test_lists_follow_the_target_and_the_neighbourhood(pcm, synth):
    """A new target or another neighbourhood rebuilds the lists; a target that grows drops them."""
    p, q = synth.make_pair(11, 6000, 60000), synth.make_pair(12, 6000, 70000)
    a = _reg(pcm, p.submap, p.scan, flags=LISTS)
    b = _reg(pcm, p.submap, p.scan)
    _same_linearize(a, b, p.T_gt, len(p.scan))
    for g in (a, b):
        g.set_input_target(q.submap); g.set_input_source(q.scan)
    _same_linearize(a, b, q.T_gt, len(q.scan))
    for g in (a, b):   # another neighbourhood on the same objects: the lists are rebuilt for it
        g.set_num_neighbors(7)
    _same_linearize(a, b, q.T_gt, len(q.scan))
    # a target that grows keeps the tile kernel from then on; same results either way
    extra = p.submap[:5000].copy(); extra[:, :3] += 0.01
    for g in (a, b):
        g.target_insert(extra)
    _same_linearize(a, b, q.T_gt, len(q.scan))


def test_default_builds_the_lists_at_the_second_registration(pcm, synth):
    """No flag: the first registration against a target runs the tile kernel, the lists are built at the second; results equal."""
    p = synth.make_pair(13, 9000, 90000)
    a = _reg(pcm, p.submap, p.scan, flags=0)
    b = _reg(pcm, p.submap, p.scan)
    ra = [a.align(p.guess) for _ in range(3)]
    rb = [b.align(p.guess) for _ in range(3)]
    for x, y in zip(ra, rb):
        assert np.array_equal(x.T64, y.T64) and x.iterations == y.iterations and x.num_inliers == y.num_inliers and x.cost == y.cost
