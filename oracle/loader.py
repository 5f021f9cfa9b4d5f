"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MODEL = {"P2PLANE": 0, "GICP": 1, "VGICP": 2, "NDT_P2D": 3, "NDT_D2D": 4, "NDT_OMP": 5, "VGICP_CUDA": 6}
OPT = {"GN": 0, "LM": 1}
REG = {"NONE": 0, "MIN_EIG": 1, "NORMALIZED_MIN_EIG": 2, "PLANE": 3, "FROBENIUS": 4, "PCLOMP": 5}


class OracleConfig(C.Structure):
    _fields_ = [("model", C.c_int), ("optimizer", C.c_int), ("max_iterations", C.c_int),
                ("rotation_eps", C.c_double), ("translation_eps", C.c_double),
                ("lm_max_iterations", C.c_int), ("lm_init_lambda_factor", C.c_double),
                ("voxel_resolution", C.c_double), ("num_neighbors", C.c_int), ("knn", C.c_int),
                ("min_knn", C.c_int), ("max_range", C.c_double), ("plane_threshold", C.c_double),
                ("max_corr_dist", C.c_double), ("k_correspondences", C.c_int),
                ("regularization", C.c_int), ("num_threads", C.c_int), ("map_capacity", C.c_long),
                ("ndt_step_size", C.c_double), ("ndt_outlier_ratio", C.c_double), ("voxel_mode", C.c_int),
                ("rbf_kernel_width", C.c_double), ("rbf_max_dist", C.c_double)]


class LioState(C.Structure):
    _fields_ = [("rot", C.c_double * 4), ("pos", C.c_double * 3), ("off_R", C.c_double * 4), ("off_T", C.c_double * 3)]


class OracleResult(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("T64", C.c_double * 16), ("H", C.c_double * 36),
                ("cost", C.c_double), ("iterations", C.c_int), ("converged", C.c_int),
                ("num_linearize", C.c_int), ("num_compute_error", C.c_int), ("num_inliers", C.c_int)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libpcm_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h", ".cpp"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(OracleConfig)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_default_config.argtypes = [C.POINTER(OracleConfig)]
        for f in (L.orc_set_target, L.orc_set_source):
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long]
        L.orc_swap_source_and_target.argtypes = [C.c_void_p]
        L.orc_linearize.restype = C.c_double
        L.orc_linearize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_compute_error.restype = C.c_double
        L.orc_compute_error.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_num_inliers.argtypes = [C.c_void_p]
        L.orc_get_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
        L.orc_obs_model.argtypes = [C.c_void_p, C.POINTER(LioState), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.orc_set_neighbor_radius.argtypes = [C.c_void_p, C.c_double]
        L.orc_set_neighbor_radius.restype = None
        L.orc_set_knn_order.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_knn_order.restype = None
        L.orc_set_lio_reference_semantics.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_lio_reference_semantics.restype = None
        L.orc_get_lio_members.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
        L.orc_target_insert.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long]
        L.orc_map_incremental.argtypes = [C.c_void_p, C.POINTER(LioState), C.c_double, C.c_int, C.POINTER(C.c_long)]
        L.orc_target_size.restype = C.c_long
        L.orc_target_size.argtypes = [C.c_void_p]
        L.orc_target_voxels.restype = C.c_long
        L.orc_target_voxels.argtypes = [C.c_void_p]
        L.orc_get_target.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_align.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(OracleResult)]
        L.orc_set_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_trace_count.argtypes = [C.c_void_p]
        L.orc_test_so3_exp.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_test_ldlt6_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_test_esti_plane.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p]
        L.orc_test_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_test_voxel_key.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_test_gauss_voxel.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.orc_pclndt_derivatives.restype = C.c_double
        L.orc_pclndt_derivatives.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_pclndt_hessian.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_pclndt_score.restype = C.c_double
        L.orc_pclndt_score.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_pclndt_leaf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.orc_pclndt_pose.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_pclndt_euler.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_pclndt_svd_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_fitness_score.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
        L.orc_fitness_score.restype = C.c_double
        L.orc_test_knn_exact.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_test_covariances.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_test_covariances_f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _LIB = L
    return _LIB


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] >= 3
    return a


class Oracle:
    """CPU oracle registration object (mirrors the pcl::Registration call order)."""

    def __init__(self, model="P2PLANE", optimizer="LM", **kw):
        L = lib()
        cfg = OracleConfig()
        L.orc_default_config(C.byref(cfg))
        cfg.model = MODEL[model]
        cfg.optimizer = OPT[optimizer]
        if "regularization" in kw:
            kw["regularization"] = REG[kw["regularization"]] if isinstance(kw["regularization"], str) else kw["regularization"]
        for k, v in kw.items():
            if not hasattr(cfg, k):
                raise KeyError(k)
            setattr(cfg, k, v)
        self.cfg = cfg
        self._h = L.orc_create(C.byref(cfg))
        self._keep = {}
        self._trace = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h)
            self._h = None

    def set_input_target(self, pts):
        a = _f32(pts); self._keep["t"] = a
        lib().orc_set_target(self._h, a.ctypes.data, a.shape[0], a.shape[1])

    def set_input_source(self, pts):
        a = _f32(pts); self._keep["s"] = a
        lib().orc_set_source(self._h, a.ctypes.data, a.shape[0], a.shape[1])

    def swap_source_and_target(self):
        lib().orc_swap_source_and_target(self._h)

    def linearize(self, T):
        T = np.ascontiguousarray(T, dtype=np.float64)
        H = np.zeros((6, 6)); b = np.zeros(6)
        cost = lib().orc_linearize(self._h, T.ctypes.data, H.ctypes.data, b.ctypes.data)
        return cost, H, b

    def compute_error(self, T):
        T = np.ascontiguousarray(T, dtype=np.float64)
        return lib().orc_compute_error(self._h, T.ctypes.data)

    @property
    def num_inliers(self):
        return lib().orc_num_inliers(self._h)

    def get_planes(self, n):
        pl = np.zeros((n, 4), np.float32); sel = np.zeros(n, np.uint8)
        if lib().orc_get_planes(self._h, pl.ctypes.data, sel.ctypes.data, n) != 0:
            raise RuntimeError("orc_get_planes")
        return pl, sel.astype(bool)

    def obs_model(self, rot_xyzw, pos, off_R_xyzw, off_T, extrinsic_est_en=False, converge=True):
        """ObsModel + IEKF reduction -> (HTH 12x12, HTh 12, n_eff, sum_h2); n_eff == 0 <=> not valid."""
        st = LioState()
        st.rot[:] = list(rot_xyzw); st.pos[:] = list(pos); st.off_R[:] = list(off_R_xyzw); st.off_T[:] = list(off_T)
        HTH = np.zeros((12, 12)); HTh = np.zeros(12); n = C.c_int(); s2 = C.c_double()
        lib().orc_obs_model(self._h, C.byref(st), int(extrinsic_est_en), int(converge), HTH.ctypes.data, HTh.ctypes.data, C.byref(n), C.byref(s2))
        return HTH, HTh, n.value, s2.value

    def set_covariances(self, covs, target=False):
        """setSourceCovariances / setTargetCovariances: (N,3,3) float64, input order."""
        L = lib()
        L.orc_set_covariances.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long]
        a = np.ascontiguousarray(covs, np.float64).reshape(-1, 9)
        if L.orc_set_covariances(self._h, int(bool(target)), a.ctypes.data, a.shape[0]) != 0:
            raise RuntimeError("orc_set_covariances: size mismatch")

    def gicp_bfgs_correspondences(self, transformation, guess):
        """pclomp GICP correspondence step: (idx_src, idx_tgt, mahalanobis (m,3,3) float32) in source order."""
        L = lib()
        L.orc_gicp_bfgs_correspondences.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_long)]
        n = self._keep["s"].shape[0]
        T = np.ascontiguousarray(transformation, np.float32); G = np.ascontiguousarray(guess, np.float32)
        isrc = np.zeros(n, np.int32); itgt = np.zeros(n, np.int32); M = np.zeros((n, 9), np.float32); m = C.c_long()
        if L.orc_gicp_bfgs_correspondences(self._h, T.ctypes.data, G.ctypes.data, isrc.ctypes.data, itgt.ctypes.data, M.ctypes.data, C.byref(m)) != 0:
            raise RuntimeError("orc_gicp_bfgs_correspondences")
        return isrc[:m.value].copy(), itgt[:m.value].copy(), M[:m.value].reshape(-1, 3, 3).copy()

    def set_neighbor_radius(self, radius):
        """NeighborSearchMethod::DIRECT_RADIUS of the CUDA-core models (radius in voxels; 0 = off)."""
        lib().orc_set_neighbor_radius(self._h, float(radius))

    def set_knn_order(self, order="ascending"):
        """Row order of the neighbours handed to esti_plane: "ascending" (default, the HIP kernels' order) or "libstdcxx"
        (what the reference's three std::nth_element calls leave with the container's libstdc++)."""
        lib().orc_set_knn_order(self._h, {"ascending": 0, "libstdcxx": 1}[order])

    def set_lio_reference_semantics(self, on=True):
        """Keep residuals_/point_selected_surf_/plane_coef_ across calls and frames as LaserMapping's members do."""
        lib().orc_set_lio_reference_semantics(self._h, int(bool(on)))

    def get_lio_members(self, n):
        pl = np.zeros((n, 4), np.float32); res = np.zeros(n, np.float32); sel = np.zeros(n, np.uint8)
        if lib().orc_get_lio_members(self._h, pl.ctypes.data, res.ctypes.data, sel.ctypes.data, n) != 0:
            raise RuntimeError("orc_get_lio_members")
        return pl, res, sel.astype(bool)

    def target_insert(self, pts):
        a = _f32(pts)
        lib().orc_target_insert(self._h, a.ctypes.data, a.shape[0], a.shape[1])

    def map_incremental(self, rot_xyzw, pos, off_R_xyzw, off_T, filter_size_map, ekf_inited=True):
        st = LioState()
        st.rot[:] = list(rot_xyzw); st.pos[:] = list(pos); st.off_R[:] = list(off_R_xyzw); st.off_T[:] = list(off_T)
        n = C.c_long()
        lib().orc_map_incremental(self._h, C.byref(st), float(filter_size_map), int(ekf_inited), C.byref(n))
        return n.value

    def get_target(self):
        n = lib().orc_target_size(self._h)
        out = np.zeros((n, 3), np.float32)
        if n:
            lib().orc_get_target(self._h, out.ctypes.data)
        return out

    # -- pclomp NDT hooks (orc_pclndt.c) ------------------------------------
    def ndt_derivatives(self, p, compute_hessian=True):
        p = np.ascontiguousarray(p, dtype=np.float64)
        g = np.zeros(6); H = np.zeros((6, 6))
        score = lib().orc_pclndt_derivatives(self._h, p.ctypes.data, int(compute_hessian), g.ctypes.data, H.ctypes.data)
        return score, g, H

    def ndt_hessian(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        H = np.zeros((6, 6))
        lib().orc_pclndt_hessian(self._h, p.ctypes.data, H.ctypes.data)
        return H

    def ndt_score(self, T):
        T = np.ascontiguousarray(T, dtype=np.float32)
        return lib().orc_pclndt_score(self._h, T.ctypes.data)

    def ndt_leaf(self, pt):
        pt = np.ascontiguousarray(pt, dtype=np.float32)
        mean = np.zeros(3); icov = np.zeros((3, 3)); n = C.c_int(0)
        ok = lib().orc_pclndt_leaf(self._h, pt.ctypes.data, mean.ctypes.data, icov.ctypes.data, C.byref(n))
        return (mean, icov, n.value) if ok else None

    def fitness_score(self, T, max_range=float(np.finfo(np.float64).max)):
        """pcl::Registration::getFitnessScore(max_range) under the float pose T."""
        T = np.ascontiguousarray(T, dtype=np.float32)
        return lib().orc_fitness_score(self._h, T.ctypes.data, float(max_range))

    def knn_exact(self, q, k):
        q = np.ascontiguousarray(q, dtype=np.float32)
        idx = np.zeros(k, np.int32); d2 = np.zeros(k, np.float32)
        m = lib().orc_test_knn_exact(self._h, q.ctypes.data, k, idx.ctypes.data, d2.ctypes.data)
        return idx[:m], d2[:m]

    def covariances(self, target=False):
        n = self._keep["t" if target else "s"].shape[0]
        if self.cfg.model == MODEL["VGICP_CUDA"]:      # float CUDA-core semantics
            outf = np.zeros((n, 3, 3), np.float32)
            lib().orc_test_covariances_f(self._h, 1 if target else 0, outf.ctypes.data)
            return outf.astype(np.float64)
        out = np.zeros((n, 3, 3))
        lib().orc_test_covariances(self._h, 1 if target else 0, out.ctypes.data)
        return out

    @property
    def target_voxels(self):
        return lib().orc_target_voxels(self._h)

    def enable_trace(self, max_records=256):
        self._trace = np.zeros((max_records, 43))
        lib().orc_set_trace(self._h, self._trace.ctypes.data, max_records)

    def trace(self):
        n = lib().orc_trace_count(self._h)
        return self._trace[:n].copy()

    def align(self, guess=None):
        g = np.eye(4, dtype=np.float32) if guess is None else np.ascontiguousarray(guess, dtype=np.float32)
        res = OracleResult()
        rc = lib().orc_align(self._h, g.ctypes.data, C.byref(res))
        if rc != 0:
            raise RuntimeError("oracle align failed: %d" % rc)
        return res


def undistort(points, time_index, poses, rot_xyzw, pos, off_R_xyzw, off_T):
    """orc_undistort on an (N,F) float32 array, in place (poses: (K,22) float64 Pose6D rows)."""
    L = lib()
    assert points.dtype == np.float32 and points.flags["C_CONTIGUOUS"]
    poses = np.ascontiguousarray(poses, dtype=np.float64)
    st = LioState()
    st.rot[:] = list(map(float, rot_xyzw)); st.pos[:] = list(map(float, pos))
    st.off_R[:] = list(map(float, off_R_xyzw)); st.off_T[:] = list(map(float, off_T))
    L.orc_undistort.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_int, C.POINTER(LioState)]
    L.orc_undistort.restype = None
    L.orc_undistort(points.ctypes.data, points.shape[0], points.shape[1], int(time_index), poses.ctypes.data, poses.shape[0], C.byref(st))
    return points


def voxel_downsample(points, leaf):
    """orc_voxel_downsample on an (N,F) float32 array -> (M,F) float32."""
    L = lib()
    points = np.ascontiguousarray(points, dtype=np.float32)
    out = np.zeros_like(points)
    L.orc_voxel_downsample.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_float, C.c_void_p]
    L.orc_voxel_downsample.restype = C.c_long
    m = L.orc_voxel_downsample(points.ctypes.data, points.shape[0], points.shape[1], float(leaf), out.ctypes.data)
    if m < 0:
        raise OverflowError("leaf size too small for the extent of the cloud")
    return out[:m].copy()


def result_T(res) -> np.ndarray:
    return np.array(res.T64[:], dtype=np.float64).reshape(4, 4)


def gicp_bfgs_fdf(src, tgt, idx_src, idx_tgt, maha, base_T, x, mode=2):
    """orc_gicp_bfgs_fdf: (f, g[6]) of pclomp's GICP-BFGS functor.  src/tgt: (N,F) float32 (x y z first); maha: (N_src,16)
    float32, column-major Matrix4f per SOURCE point; base_T: 4x4; x: 6 doubles; mode 0 = f, 1 = df, 2 = fdf."""
    L = lib()
    src = np.ascontiguousarray(src, np.float32); tgt = np.ascontiguousarray(tgt, np.float32)
    assert src.shape[1] == tgt.shape[1]
    idx_src = np.ascontiguousarray(idx_src, np.int32); idx_tgt = np.ascontiguousarray(idx_tgt, np.int32)
    maha = np.ascontiguousarray(maha, np.float32).reshape(-1, 16)
    base = np.ascontiguousarray(base_T, np.float32).reshape(16); xx = np.ascontiguousarray(x, np.float64).reshape(6)
    f = C.c_double(0.0); g = np.zeros(6, np.float64)
    L.orc_gicp_bfgs_fdf.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.POINTER(C.c_double), C.c_void_p]
    L.orc_gicp_bfgs_fdf.restype = C.c_int
    rc = L.orc_gicp_bfgs_fdf(src.ctypes.data, tgt.ctypes.data, src.shape[1], idx_src.ctypes.data, idx_tgt.ctypes.data, len(idx_src), maha.ctypes.data,
                             base.ctypes.data, xx.ctypes.data, int(mode), C.byref(f), g.ctypes.data)
    if rc != 0:
        raise ValueError("orc_gicp_bfgs_fdf: %d" % rc)
    return f.value, g


def gicp_bfgs_apply_state(base_T, x):
    L = lib()
    base = np.ascontiguousarray(base_T, np.float32).reshape(16); xx = np.ascontiguousarray(x, np.float64).reshape(6)
    T = np.zeros(16, np.float32)
    L.orc_gicp_bfgs_apply_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_gicp_bfgs_apply_state.restype = None
    L.orc_gicp_bfgs_apply_state(base.ctypes.data, xx.ctypes.data, T.ctypes.data)
    return T.reshape(4, 4)


LIVOX_POINT = np.dtype([("offset_time", "<u4"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("reflectivity", "u1"), ("tag", "u1"), ("line", "u1"), ("pad", "u1")])


def livox_filter(msg_points, num_scans=6, point_filter_num=1, blind=0.01):
    """PointCloudPreprocess::AviaHandler: structured array of LIVOX_POINT -> (m, 12) float32 PointXYZINormal records."""
    L = lib()
    L.orc_livox_filter.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_double, C.c_void_p]
    L.orc_livox_filter.restype = C.c_long
    a = np.ascontiguousarray(msg_points, dtype=LIVOX_POINT)
    out = np.zeros((max(len(a), 1), 12), np.float32)
    m = L.orc_livox_filter(a.ctypes.data, len(a), int(num_scans), int(point_filter_num), float(blind), out.ctypes.data)
    return out[:m].copy()
