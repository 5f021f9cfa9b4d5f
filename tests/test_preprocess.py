"""Scan pre-processing: IMU motion compensation of a scan."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation


def _frame(n=6000, first_ms=7.0, seed=0):
    """A 100 ms frame: 11 propagated IMU poses (every 10 ms, mildly accelerating turn), points sorted by time.  The first
    point is sampled after the first IMU poses, which triggers the reference loop's repeated visit of that point."""
    rng = np.random.default_rng(seed)
    K = 11
    poses = np.zeros((K, 22))
    acc = np.array([0.3, -0.2, 0.11]); gyr = np.array([0.2, -0.3, 0.6]); dt = 0.01
    R = Rotation.from_euler("xyz", [0.02, -0.01, 0.3]); vel = np.array([5.0, 0.5, -0.1]); pos = np.zeros(3)
    for k in range(K):                                       # a self-consistent propagation: constant body rate and acceleration
        poses[k, 0] = dt * k
        poses[k, 1:4] = acc; poses[k, 4:7] = gyr; poses[k, 7:10] = vel; poses[k, 10:13] = pos
        poses[k, 13:22] = R.as_matrix().ravel()
        pos = pos + vel * dt + 0.5 * acc * dt * dt
        vel = vel + acc * dt
        R = R * Rotation.from_rotvec(gyr * dt)
    pts = np.zeros((n, 12), np.float32)                     # PointXYZINormal: 48 bytes, curvature = column 10
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts[:, :3] = (d * rng.uniform(2.0, 60.0, (n, 1))).astype(np.float32)
    times = np.sort(rng.uniform(first_ms, 100.0, n)).astype(np.float32)
    times[0] = first_ms
    times[50] = 30.0                                         # exactly on an IMU stamp: belongs to the segment before it
    pts[:, 10] = np.sort(times)
    rot_end = Rotation.from_matrix(poses[-1, 13:22].reshape(3, 3)).as_quat()
    state = dict(rot_xyzw=rot_end, pos=poses[-1, 10:13], off_R_xyzw=Rotation.from_euler("xyz", [0.01, -0.02, 0.03]).as_quat(), off_T=[0.05, -0.02, 0.1])
    return pts, poses, state


def test_undistort_oracle_properties():
    """orc_preprocess.c: a point sampled at the frame end with the end pose is unchanged; with zero motion nothing moves;
    the first point takes the extra visits the reference loop gives it."""
    from oracle.loader import undistort
    pts, poses, st = _frame()
    still = poses.copy(); still[:, 1:13] = 0.0; still[:, 13:22] = np.eye(3).ravel()
    st0 = dict(rot_xyzw=[0, 0, 0, 1], pos=[0, 0, 0], off_R_xyzw=st["off_R_xyzw"], off_T=st["off_T"])
    a = pts.copy(); undistort(a, 10, still, **st0)
    assert np.abs(a[:, :3] - pts[:, :3]).max() < 2e-5        # zero motion: only double -> float rounding of the round trip
    b = pts.copy(); undistort(b, 10, poses, **st)
    moved = np.linalg.norm(b[:, :3] - pts[:, :3], axis=1)
    assert moved[-1] < 1e-3 and moved[1] > 0.1               # a point sampled at the frame end stays, an early one moves by the ego-motion
    assert np.array_equal(b[:, 3:], pts[:, 3:])              # only x, y, z are rewritten
    # the first point is visited once per segment below its own (here: 7 ms -> segments starting at 0 ms only: one visit)
    c = pts.copy(); c[0, 10] = 35.0; c[1:, 10] = np.maximum(c[1:, 10], 35.0)
    ref = c.copy(); undistort(ref, 10, poses, **st)
    dummy = np.zeros((1, 12), np.float32); dummy[0, :3] = 1.0; dummy[0, 10] = 34.9       # another point in front: c[0] becomes ordinary
    single = np.ascontiguousarray(np.vstack([dummy, c]))
    undistort(single, 10, poses, **st)
    assert not np.allclose(single[1, :3], ref[0, :3], atol=1e-4)      # as an ordinary point it gets ONE compensation, as first point four
    assert np.array_equal(single[2:, :3], ref[1:, :3])


@pytest.mark.gpu
def test_undistort_matches_oracle(pcm):
    from oracle.loader import undistort
    for first_ms, seed in ((7.0, 0), (35.0, 1), (0.0, 2)):
        pts, poses, st = _frame(first_ms=first_ms, seed=seed)
        a, b = pts.copy(), pts.copy()
        undistort(a, 10, poses, **st)
        reg = pcm.P2PlaneRegistration(0)
        reg.undistort(b, 10, poses, **st)
        assert np.array_equal(a[:, 3:], b[:, 3:])
        ulp = np.spacing(np.maximum(np.abs(a[:, :3]), 1.0).astype(np.float32))
        assert (np.abs(a[:, :3] - b[:, :3]) <= 2 * ulp).all()          # device sin/cos vs libm: at most the last float bit or two
        assert (a[:, :3] == b[:, :3]).mean() > 0.99


def _scan_for_downsample(seed=3, n=20000):
    rng = np.random.default_rng(seed)
    pts = np.zeros((n, 12), np.float32)
    pts[:, :3] = rng.normal(0, [8.0, 6.0, 1.5], (n, 3)).astype(np.float32)
    pts[:, 8] = rng.uniform(0, 255, n).astype(np.float32)          # intensity
    pts[:, 10] = np.sort(rng.uniform(0, 100, n)).astype(np.float32)   # curvature (time)
    pts[5, 0] = np.nan                                               # non-finite points are dropped
    return pts


def test_voxel_downsample_oracle():
    """orc_voxel_downsample: one centroid per occupied leaf, increasing leaf-index order, all fields averaged."""
    from oracle.loader import voxel_downsample
    pts = _scan_for_downsample()
    leaf = 0.5
    out = voxel_downsample(pts, leaf)
    ok = np.isfinite(pts[:, :3]).all(axis=1)
    p = pts[ok]
    inv = np.float32(1.0) / np.float32(leaf)
    ijk = np.floor(p[:, :3] * inv).astype(np.int64)
    ijk -= np.floor(p[:, :3].min(axis=0) * inv).astype(np.int64)
    div = ijk.max(axis=0) + 1
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    uniq = np.unique(idx)
    assert len(out) == len(uniq)
    for j in (0, len(uniq) // 2, len(uniq) - 1):
        assert np.allclose(out[j], p[idx == uniq[j]].astype(np.float64).mean(axis=0), rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_voxel_downsample_matches_oracle(pcm):
    from oracle.loader import voxel_downsample
    reg = pcm.P2PlaneRegistration(0)
    for seed, leaf in ((3, 0.5), (4, 0.2), (5, 1.0)):
        pts = _scan_for_downsample(seed)
        a = voxel_downsample(pts, leaf)
        b = reg.voxel_downsample(pts, leaf)
        # double sums on both sides (sequential in the oracle, 64 interleaved partial sums + a fixed tree on the device):
        # the float results agree except where a double rounding difference crosses a float rounding boundary
        assert a.shape == b.shape
        ulp = np.spacing(np.maximum(np.abs(a), 1e-3).astype(np.float32))
        assert (np.abs(a - b) <= ulp).all() and (a == b).mean() > 0.9999
    with pytest.raises(pcm.PcmError):
        far = _scan_for_downsample(6); far[0, :3] = 1e7
        reg.voxel_downsample(far, 0.001)                           # index overflow, as PCL refuses it


# ---- PointCloudPreprocess::AviaHandler (pointcloud_preprocess.cc:44-88) ------------------------------------------------------
def _livox_msg(n=20000, seed=3):
    """A livox CustomMsg-like frame: mostly good returns, with noise tags, lines beyond num_scans, exact repeats of the previous
    point (the duplicate test), points inside the blind radius that differ only in z, and an all-zero return."""
    from oracle.loader import LIVOX_POINT
    rng = np.random.default_rng(seed)
    a = np.zeros(n, LIVOX_POINT)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    p = (d * rng.uniform(0.5, 80.0, (n, 1))).astype(np.float32)
    a["x"], a["y"], a["z"] = p[:, 0], p[:, 1], p[:, 2]
    a["offset_time"] = np.sort(rng.integers(0, 100_000_000, n)).astype(np.uint32)       # ns within a 100 ms frame
    a["reflectivity"] = rng.integers(0, 256, n)
    a["tag"] = rng.choice([0x00, 0x10, 0x20, 0x30, 0x11, 0x05, 0x25], n, p=[0.3, 0.4, 0.05, 0.05, 0.1, 0.05, 0.05])
    a["line"] = rng.integers(0, 8, n)                                                  # num_scans = 6: lines 6, 7 are dropped
    if n > 200:
        rep = rng.choice(np.arange(2, n), n // 20, replace=False)                      # exact repeats of the previous raw point
        for k in ("x", "y", "z"):
            a[k][rep] = a[k][rep - 1]
        near = rng.choice(np.arange(2, n), n // 20, replace=False)                     # same x, y as the previous point, tiny range
        a["x"][near] = a["x"][near - 1]; a["y"][near] = a["y"][near - 1]; a["z"][near] = a["z"][near - 1] + np.float32(0.001)
        a["x"][100] = a["y"][100] = a["z"][100] = 0.0
    return a


def _avia_reference(a, num_scans, filt, blind):
    """The handler written out in numpy float32/float64, serial reading of the par_unseq loop."""
    n = len(a)
    i = np.arange(n)
    passes = (a["line"] < num_scans) & (((a["tag"] & 0x30) == 0x10) | ((a["tag"] & 0x30) == 0x00)) & (i % filt == 0) & (i >= 1)
    prev = np.zeros((n, 3), np.float32)
    cp = np.r_[False, passes[:-1]]
    xyz = np.stack([a["x"], a["y"], a["z"]], 1)
    prev[cp] = np.r_[np.zeros((1, 3), np.float32), xyz[:-1]][cp]
    d = np.abs(xyz - prev).astype(np.float64)
    r2 = (xyz[:, 0] * xyz[:, 0] + xyz[:, 1] * xyz[:, 1] + xyz[:, 2] * xyz[:, 2]).astype(np.float64)
    keep = passes & ((d[:, 0] > 1e-7) | (d[:, 1] > 1e-7) | ((d[:, 2] > 1e-7) & (r2 > blind * blind)))
    out = np.zeros((int(keep.sum()), 12), np.float32)
    out[:, :3] = xyz[keep]; out[:, 3] = 1.0
    out[:, 8] = a["reflectivity"][keep].astype(np.float32)
    out[:, 9] = a["offset_time"][keep].astype(np.float32) / np.float32(1000000)
    return out


@pytest.mark.parametrize("filt,blind", [(1, 0.01), (2, 0.1), (3, 4.0)])      # header default; config/livox.yaml; config/horizon.yaml
def test_livox_filter_oracle_against_a_numpy_statement(filt, blind):
    from oracle.loader import livox_filter
    a = _livox_msg()
    got = livox_filter(a, 6, filt, blind)
    want = _avia_reference(a, 6, filt, blind)
    assert 0.2 * len(a) / filt < len(got) < len(a) and got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if filt == 1:   # the reference's precedence: a point inside the blind radius survives unless it differs from its predecessor in z only
        assert (np.linalg.norm(livox_filter(a, 6, 1, 4.0)[:, :3], axis=1) < 4.0).sum() > 100


def test_livox_filter_tiny_inputs():
    from oracle.loader import livox_filter, LIVOX_POINT
    assert len(livox_filter(np.zeros(0, LIVOX_POINT))) == 0 and len(livox_filter(_livox_msg(1))) == 0     # index 0 is never taken (:53-56)


@pytest.mark.gpu
@pytest.mark.parametrize("filt,blind", [(1, 0.01), (2, 0.1), (3, 4.0)])
def test_gpu_livox_filter_equals_the_oracle(pcm, filt, blind):
    from oracle.loader import livox_filter
    g = pcm.P2PlaneRegistration(0)
    for n in (20000, 257, 2, 1):
        a = _livox_msg(n)
        want = livox_filter(a, 6, filt, blind)
        got = g.livox_filter(a, 6, filt, blind)
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
