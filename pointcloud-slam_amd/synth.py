"""Seeded synthetic scan / submap pairs for the registration hot path.

Inputs follow SURVEY.md §8(d): an analytic scene (ground, perimeter walls,
axis-aligned boxes, vertical cylinders), an area-uniform *submap* sample of it
with N(0, 0.01 m) normal noise, and a *Livox-shaped scan*: a non-repetitive
rosette ray pattern (6 lines, 70.4 x 77.2 deg FoV, 100 ms frame, per-point
``offset_time``; message layout of
``jueying_lio/thirdparty/livox_ros_driver/msg/CustomPoint.msg``) ray-cast from
a ground-truth sensor pose, with N(0, 0.02 m) range noise and the ``blind``
(0.1 m, ``jueying_lio/config/livox.yaml:9``) cut.  The scene is scaled so the
submap has a LIO-like density (a few points per 0.5 m voxel), because a 1 M
point map of a single room would put hundreds of points in every voxel, which
no voxel-filtered LIO map (``laser_mapping.cc:525-583``) ever has.

Everything is numpy on the host; no file or network access.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

FOV_H_DEG = 70.4
FOV_V_DEG = 77.2
NUM_LINES = 6          # config/livox.yaml:8 scan_line
BLIND = 0.1            # config/livox.yaml:9
FRAME_MS = 100.0


@dataclasses.dataclass
class Scene:
    lx: float
    ly: float
    wall_h: float
    boxes: np.ndarray      # (B, 6) xmin,ymin,zmin,xmax,ymax,zmax
    cyls: np.ndarray       # (C, 4) cx, cy, radius, height

    def area_parts(self):
        """(name, area) of every sampled surface piece."""
        parts = [("ground", self.lx * self.ly)]
        for w in range(4):
            length = self.lx if w < 2 else self.ly
            parts.append((f"wall{w}", length * self.wall_h))
        for b in self.boxes:
            dx, dy, dz = b[3] - b[0], b[4] - b[1], b[5] - b[2]
            parts += [("bx", dy * dz), ("bx", dy * dz), ("by", dx * dz), ("by", dx * dz), ("bz", dx * dy)]
        for c in self.cyls:
            parts.append(("cyl", 2.0 * math.pi * c[2] * c[3]))
        return parts


def make_scene(seed: int, scale: float, n_boxes: int = 160, n_cyls: int = 30) -> Scene:
    rng = np.random.default_rng(seed)
    lx, ly = 4.0 * scale, 3.0 * scale
    wall_h = 0.1 * scale + 2.0
    boxes = []
    for _ in range(n_boxes):
        # many small "buildings / containers / vehicles": vertical structure within
        # ~10-20 m of any sensor position keeps yaw and the horizontal translation
        # observable (a scan that sees only ground is degenerate for any ICP)
        sx, sy = rng.uniform(0.04, 0.16, 2) * scale
        sz = rng.uniform(0.03, 0.15) * scale + 0.5
        cx = rng.uniform(0.05 * lx + sx / 2, 0.95 * lx - sx / 2)
        cy = rng.uniform(0.05 * ly + sy / 2, 0.95 * ly - sy / 2)
        boxes.append([cx - sx / 2, cy - sy / 2, 0.0, cx + sx / 2, cy + sy / 2, sz])
    cyls = []
    for _ in range(n_cyls):
        r = rng.uniform(0.004, 0.012) * scale + 0.1
        cyls.append([rng.uniform(0.1 * lx, 0.9 * lx), rng.uniform(0.1 * ly, 0.9 * ly), r, rng.uniform(0.05, 0.15) * scale + 1.0])
    return Scene(lx, ly, wall_h, np.asarray(boxes, dtype=np.float64).reshape(-1, 6), np.asarray(cyls, dtype=np.float64).reshape(-1, 4))


def scene_for_points(seed: int, m_points: int, density: float = 8.0) -> Scene:
    """Scene scaled so that ``m_points`` area-uniform samples give ``density`` pts/m^2."""
    unit = make_scene(seed, 1.0)
    a_unit = sum(a for _, a in unit.area_parts())
    # area grows ~quadratically with scale (the +const terms make it approximate): fixed-point iterate
    scale = math.sqrt(m_points / (density * a_unit))
    for _ in range(8):
        a = sum(a for _, a in make_scene(seed, scale).area_parts())
        scale *= math.sqrt(m_points / (density * a))
    return make_scene(seed, scale)


def sample_submap(scene: Scene, m_points: int, seed: int, noise: float = 0.01) -> np.ndarray:
    """(M,4) float32 {x,y,z,1}: area-uniform surface samples + normal noise."""
    rng = np.random.default_rng(seed)
    parts = scene.area_parts()
    areas = np.array([a for _, a in parts])
    counts = rng.multinomial(m_points, areas / areas.sum())
    out = np.empty((m_points, 3), dtype=np.float64)
    pos = 0
    k = 0

    def put(pts, normal_axis=None, normals=None):
        nonlocal pos
        n = pts.shape[0]
        eps = rng.normal(0.0, noise, n)
        if normals is None:
            pts[:, normal_axis] += eps
        else:
            pts += normals * eps[:, None]
        out[pos:pos + n] = pts
        pos += n

    # ground
    n = counts[k]; k += 1
    put(np.stack([rng.uniform(0, scene.lx, n), rng.uniform(0, scene.ly, n), np.zeros(n)], 1), 2)
    # walls: y=0, y=ly, x=0, x=lx
    for w in range(4):
        n = counts[k]; k += 1
        z = rng.uniform(0, scene.wall_h, n)
        if w < 2:
            put(np.stack([rng.uniform(0, scene.lx, n), np.full(n, 0.0 if w == 0 else scene.ly), z], 1), 1)
        else:
            put(np.stack([np.full(n, 0.0 if w == 2 else scene.lx), rng.uniform(0, scene.ly, n), z], 1), 0)
    for b in scene.boxes:
        for f in range(5):
            n = counts[k]; k += 1
            x = rng.uniform(b[0], b[3], n); y = rng.uniform(b[1], b[4], n); z = rng.uniform(b[2], b[5], n)
            if f == 0: x[:] = b[0]
            elif f == 1: x[:] = b[3]
            elif f == 2: y[:] = b[1]
            elif f == 3: y[:] = b[4]
            else: z[:] = b[5]
            put(np.stack([x, y, z], 1), 0 if f < 2 else (1 if f < 4 else 2))
    for c in scene.cyls:
        n = counts[k]; k += 1
        th = rng.uniform(0, 2 * math.pi, n)
        nrm = np.stack([np.cos(th), np.sin(th), np.zeros(n)], 1)
        put(np.stack([c[0] + c[2] * np.cos(th), c[1] + c[2] * np.sin(th), rng.uniform(0, c[3], n)], 1), normals=nrm)
    assert pos == m_points
    rng.shuffle(out, axis=0)
    res = np.ones((m_points, 4), dtype=np.float32)
    res[:, :3] = out.astype(np.float32)
    return res


def rot_xyz(rx: float, ry: float, rz: float) -> np.ndarray:
    cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def sensor_pose(scene: Scene, seed: int) -> np.ndarray:
    """Ground-truth world<-body pose (4x4 float64) of a sensor standing in free space."""
    rng = np.random.default_rng(seed)
    for _ in range(1000):
        x = rng.uniform(0.2 * scene.lx, 0.8 * scene.lx)
        y = rng.uniform(0.2 * scene.ly, 0.8 * scene.ly)
        m = 1.0
        inside = np.any((scene.boxes[:, 0] - m < x) & (x < scene.boxes[:, 3] + m) & (scene.boxes[:, 1] - m < y) & (y < scene.boxes[:, 4] + m))
        if len(scene.cyls):
            inside |= bool(np.any(np.hypot(scene.cyls[:, 0] - x, scene.cyls[:, 1] - y) < scene.cyls[:, 2] + m))
        if not inside:
            break
    T = np.eye(4)
    T[:3, :3] = rot_xyz(rng.uniform(-0.03, 0.03), rng.uniform(-0.03, 0.03), rng.uniform(-math.pi, math.pi))
    T[:3, 3] = [x, y, rng.uniform(1.2, 1.8)]
    return T


def rosette_dirs(n: int, seed: int):
    """Unit ray directions (body frame, x forward) of a Livox-like non-repetitive
    rosette + per-point offset_time [ns], line id."""
    rng = np.random.default_rng(seed)
    k = np.arange(n)
    t = (k + rng.uniform(0, 1)) / n                      # fraction of the 100 ms frame
    line = (k % NUM_LINES).astype(np.uint8)
    phi = 2.0 * math.pi * (17.0 * t + 0.013 * line) + rng.uniform(0, 2 * math.pi)
    rho = np.cos(math.pi * math.sqrt(2.0) * 23.0 * t + 0.31 * line)   # irrational petal ratio
    az = np.deg2rad(FOV_H_DEG / 2) * rho * np.cos(phi)
    el = np.deg2rad(FOV_V_DEG / 2) * rho * np.sin(phi)
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], 1)
    offset_time = (t * FRAME_MS * 1e6).astype(np.uint32)
    return d, offset_time, line


def raycast(scene: Scene, o: np.ndarray, d: np.ndarray, max_range: float) -> np.ndarray:
    """Range of the first hit for rays o + t d (d unit, (N,3)); inf when none."""
    n = d.shape[0]
    best = np.full(n, np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        # ground z = 0
        t = -o[2] / d[:, 2]
        x = o[0] + t * d[:, 0]; y = o[1] + t * d[:, 1]
        ok = (t > 0) & (x >= 0) & (x <= scene.lx) & (y >= 0) & (y <= scene.ly)
        best = np.where(ok & (t < best), t, best)
        # walls
        for w, (axis, val) in enumerate([(1, 0.0), (1, scene.ly), (0, 0.0), (0, scene.lx)]):
            t = (val - o[axis]) / d[:, axis]
            other = 1 - axis
            u = o[other] + t * d[:, other]
            z = o[2] + t * d[:, 2]
            lim = scene.lx if other == 0 else scene.ly
            ok = (t > 0) & (u >= 0) & (u <= lim) & (z >= 0) & (z <= scene.wall_h)
            best = np.where(ok & (t < best), t, best)
        inv = 1.0 / d
        for b in scene.boxes:
            t0 = (b[:3] - o) * inv
            t1 = (b[3:] - o) * inv
            tn = np.nanmax(np.minimum(t0, t1), axis=1)
            tf = np.nanmin(np.maximum(t0, t1), axis=1)
            ok = (tn <= tf) & (tn > 0)
            best = np.where(ok & (tn < best), tn, best)
        for c in scene.cyls:
            ox, oy = o[0] - c[0], o[1] - c[1]
            a = d[:, 0] ** 2 + d[:, 1] ** 2
            bq = 2 * (ox * d[:, 0] + oy * d[:, 1])
            cq = ox * ox + oy * oy - c[2] ** 2
            disc = bq * bq - 4 * a * cq
            t = (-bq - np.sqrt(np.maximum(disc, 0))) / (2 * a)
            z = o[2] + t * d[:, 2]
            ok = (disc > 0) & (t > 0) & (z >= 0) & (z <= c[3])
            best = np.where(ok & (t < best), t, best)
    best[best > max_range] = np.inf
    return best


def livox_scan(scene: Scene, T_wb: np.ndarray, n_points: int, seed: int, max_range: float = 150.0,
               range_noise: float = 0.02, point_filter_num: int = 2):
    """(N,4) float32 body-frame scan {x,y,z,1} + dict of CustomPoint-like extras.

    ``point_filter_num`` (config/livox.yaml:40): only every n-th valid return is kept.
    """
    rng = np.random.default_rng(seed)
    want = n_points
    pts, times, lines = [], [], []
    got = 0
    rounds = 0
    while got < want:
        nray = int((want - got) * 1.5 * point_filter_num) + 1024
        d, ot, ln = rosette_dirs(nray, seed * 7919 + rounds)
        dw = d @ T_wb[:3, :3].T
        r = raycast(scene, T_wb[:3, 3], dw, max_range)
        r = r + rng.normal(0.0, range_noise, nray)
        ok = np.isfinite(r) & (r > BLIND)
        idx = np.nonzero(ok)[0][::point_filter_num]
        pts.append(d[idx] * r[idx, None]); times.append(ot[idx]); lines.append(ln[idx])
        got += len(idx)
        rounds += 1
        if rounds > 20:
            raise RuntimeError("scan generation did not reach the requested size")
    p = np.concatenate(pts)[:want]
    order = np.argsort(np.concatenate(times)[:want], kind="stable")
    out = np.ones((want, 4), dtype=np.float32)
    out[:, :3] = p[order].astype(np.float32)
    extras = {"offset_time": np.concatenate(times)[:want][order], "line": np.concatenate(lines)[:want][order],
              "reflectivity": np.full(want, 100, np.uint8), "tag": np.full(want, 0x10, np.uint8)}
    return out, extras


def perturb_pose(T: np.ndarray, seed: int, dt: float = 0.2, drot_deg: float = 3.0) -> np.ndarray:
    """Initial guess: ground truth perturbed by U(-dt,dt) m and U(-drot,drot) deg per
    axis, applied in the SENSOR frame (T @ D): an odometry-style error about the
    sensor, not a rotation about the far-away world origin."""
    rng = np.random.default_rng(seed)
    a = np.deg2rad(rng.uniform(-drot_deg, drot_deg, 3))
    D = np.eye(4)
    D[:3, :3] = rot_xyz(*a)
    D[:3, 3] = rng.uniform(-dt, dt, 3)
    return T @ D


@dataclasses.dataclass
class Pair:
    scan: np.ndarray      # (N,4) f32 body frame
    submap: np.ndarray    # (M,4) f32 world frame
    T_gt: np.ndarray      # (4,4) f64 world<-body
    guess: np.ndarray     # (4,4) f32 row-major initial guess
    extras: dict


def make_pair(pair_id: int, n_scan: int, m_map: int, density: float = 8.0) -> Pair:
    """SURVEY §8(d) seeds: scene 1234+pair_id, pose 99+pair_id."""
    scene = scene_for_points(1234 + pair_id, m_map, density)
    submap = sample_submap(scene, m_map, 4321 + pair_id)
    T = sensor_pose(scene, 77 + pair_id)
    scan, extras = livox_scan(scene, T, n_scan, 555 + pair_id)
    guess = perturb_pose(T, 99 + pair_id).astype(np.float32)
    return Pair(scan, submap, T, guess, extras)


def corner_scene(n_scan: int, m_map: int, seed: int = 0, noise: float = 0.0, scan_margin: float = 0.5):
    """Known-answer scene: three orthogonal 10 m faces meeting at (3, 4, -2).
    Returns (scan_body, submap_world, T_gt); the scan is an exact rigid copy of
    fresh samples of the same planes.  The corner is kept away from the origin
    because the reference's plane model ``n.p = -1`` (common_lib.h:199-208)
    cannot represent a plane through the origin."""
    rng = np.random.default_rng(seed)

    def samp(n, margin=0.0):
        f = rng.integers(0, 3, n)
        p = rng.uniform(margin, 10.0 - margin, (n, 3))
        p[np.arange(n), f] = 0.0
        if noise > 0:
            p[np.arange(n), f] += rng.normal(0, noise, n)
        return p + np.array([3.0, 4.0, -2.0])

    m = samp(m_map)
    s_world = samp(n_scan, scan_margin)   # keep scan points off the edges: a 5-NN set
    # straddling two faces still passes the 0.1 m plane test (common_lib.h:235-241)
    T = np.eye(4)
    T[:3, :3] = rot_xyz(0.02, -0.015, 0.03)
    T[:3, 3] = [0.05, -0.04, 0.03]
    Ti = np.linalg.inv(T)
    s_body = s_world @ Ti[:3, :3].T + Ti[:3, 3]
    sm = np.ones((m_map, 4), np.float32); sm[:, :3] = m
    sc = np.ones((n_scan, 4), np.float32); sc[:, :3] = s_body
    return sc, sm, T


LIVOX_POINT = np.dtype([("offset_time", "<u4"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("reflectivity", "u1"), ("tag", "u1"), ("line", "u1"), ("pad", "u1")])


def custom_msg(scan: np.ndarray, extras: dict, frame_ns: float = 99e6) -> np.ndarray:
    """The points of a livox_ros_driver::CustomMsg (20-byte records, jueying_lio/thirdparty/livox_ros_driver/msg/CustomPoint.msg)
    for a synthetic scan of livox_scan(): time-ordered offset_time in ns inside one frame, 6 lines, valid tags."""
    n = len(scan)
    a = np.zeros(n, LIVOX_POINT)
    a["x"], a["y"], a["z"] = scan[:, 0], scan[:, 1], scan[:, 2]
    t = np.asarray(extras["offset_time"], np.float64)
    a["offset_time"] = np.clip((t - t.min()) / max(1e-12, float(t.max() - t.min())) * frame_ns, 0, frame_ns).astype(np.uint32)
    a["reflectivity"] = extras["reflectivity"]
    a["tag"] = extras["tag"]
    a["line"] = np.asarray(extras["line"]) % 6
    return a
