"""Shared helpers for the parity tests."""
import numpy as np


def pose_error(Ta, Tb):
    """(translation [m], rotation [rad]) between two 4x4 poses."""
    D = np.linalg.inv(np.asarray(Ta, np.float64)) @ np.asarray(Tb, np.float64)
    c = np.clip((np.trace(D[:3, :3]) - 1.0) / 2.0, -1.0, 1.0)
    # small-angle safe: use the skew part for tiny rotations
    s = 0.5 * np.linalg.norm([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]])
    ang = np.arctan2(s, c)
    return float(np.linalg.norm(D[:3, 3])), float(ang)


# parity tolerance stated by BASELINE.json north_star: 1e-4 m / 1e-4 rad
POSE_TOL_M = 1e-4
POSE_TOL_RAD = 1e-4
# normal equations: relative tolerance of SURVEY.md §7 step 4
HB_RTOL = 1e-5


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
