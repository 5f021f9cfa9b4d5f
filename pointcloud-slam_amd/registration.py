"""Host-side mirror of the reference's registration operator surface.

Same names, argument meaning and call order as the reference's pybind module
``pygicp`` (/root/reference/src/pointcloud_match/fast_gicp/src/python/main.cpp:135-215:
``set_input_target/source``, ``align``, ``swap_source_and_target``,
``get_final_transformation``, ``get_final_hessian`` ...) and the
``pcl::Registration`` setters it wraps, so the parity tests read like the
reference's gtest (src/test/gicp_test.cpp:147-201).  All compute goes through
the C ABI of include/pcm_amd.h; nothing here touches oracle/.
"""
from __future__ import annotations

import ctypes as C
import dataclasses

import numpy as np

from . import capi


@dataclasses.dataclass
class RegistrationResult:
    T: np.ndarray            # (4,4) float32 final_transformation_
    T64: np.ndarray          # (4,4) float64 pose before the float cast
    H: np.ndarray            # (6,6) final_hessian_
    cost: float
    iterations: int
    converged: bool
    num_linearize: int
    num_compute_error: int
    num_inliers: int
    status: int


def _result(r: capi.PcmResult) -> RegistrationResult:
    return RegistrationResult(np.array(r.T[:], np.float32).reshape(4, 4), np.array(r.T64[:]).reshape(4, 4),
                              np.array(r.H[:]).reshape(6, 6), r.cost, r.iterations, bool(r.converged),
                              r.num_linearize, r.num_compute_error, r.num_inliers, r.status)


def _points(a):
    """Accept (N,>=3) float32 host arrays or torch CUDA tensors; returns (ptr, n, stride, memory, keepalive)."""
    if hasattr(a, "data_ptr") and hasattr(a, "is_cuda"):
        if a.dtype.__str__() != "torch.float32" or a.dim() != 2 or a.shape[1] < 3 or not a.is_contiguous():
            raise ValueError("expected a contiguous (N,>=3) float32 tensor")
        mem = capi.MEM_DEVICE if a.is_cuda else capi.MEM_HOST
        return a.data_ptr(), a.shape[0], a.shape[1] * 4, mem, a
    arr = np.ascontiguousarray(a, dtype=np.float32)
    if arr.ndim != 2 or arr.shape[1] < 3:
        raise ValueError("expected an (N,>=3) float32 array")
    return arr.ctypes.data, arr.shape[0], arr.shape[1] * 4, capi.MEM_HOST, arr


class Registration:
    """One registration object bound to a HIP device (= one ``pcm_ctx``)."""

    model = "P2PLANE"
    defaults = {}

    def __init__(self, device: int = 0, model: str = None, **params):
        self._L = capi.load_library()
        cfg = capi.PcmConfig()
        self._L.pcm_default_config(C.byref(cfg))
        if model is not None:
            self.model = model
        cfg.model = capi.MODEL[self.model]
        for k, v in self.defaults.items():
            setattr(cfg, k, v)
        self._cfg = cfg
        self._h = self._L.pcm_create(device, C.byref(cfg))
        if not self._h:
            raise capi.PcmError(-3, "pcm_create failed")
        self._keep = {}
        self._last = None
        self._set(**params)

    # -- plumbing ---------------------------------------------------------
    def _check(self, rc, allow=(capi.PCM_OK,)):
        if rc not in allow:
            raise capi.PcmError(rc, (self._L.pcm_last_error(self._h) or b"").decode())

    def _set(self, **kw):
        for k, v in kw.items():
            if k == "optimizer" and isinstance(v, str):
                v = capi.OPTIMIZER[v]
            if k == "regularization" and isinstance(v, str):
                v = capi.REGULARIZATION[v]
            if not hasattr(self._cfg, k):
                raise KeyError(k)
            setattr(self._cfg, k, v)
        self._check(self._L.pcm_set_config(self._h, C.byref(self._cfg)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.pcm_destroy(h)

    @property
    def handle(self):
        return self._h

    @property
    def config(self):
        """The object's pcm_config (read-only view; change it through the setters)."""
        return self._cfg

    # -- pcl::Registration / LsqRegistration setters ------------------------
    def set_max_iterations(self, n): self._set(max_iterations=int(n))            # setMaximumIterations
    def set_transformation_epsilon(self, e): self._set(translation_eps=float(e))   # setTransformationEpsilon
    def set_rotation_epsilon(self, e): self._set(rotation_eps=float(e))            # lsq_registration_impl.hpp:27-29
    def set_initial_lambda_factor(self, f): self._set(lm_init_lambda_factor=float(f))  # :32-34
    def set_optimizer(self, name): self._set(optimizer=name)                       # lsq_optimizer_type_
    def set_resolution(self, r): self._set(voxel_resolution=float(r))              # fast_vgicp_impl.hpp:28-30
    def set_num_neighbors(self, n): self._set(num_neighbors=int(n))                # setNeighborSearchMethod / ivox_nearby_type

    def set_neighbor_search_method(self, method: str, radius: float = -1.0):
        """pygicp's set_neighbor_search_method (src/python/main.cpp:195-212): DIRECT1 / DIRECT7 / DIRECT27, or DIRECT_RADIUS with the radius in
        voxels for the CUDA-core models."""
        if method == "DIRECT_RADIUS":
            self._set(neighbor_search_radius=float(radius))
        else:
            self._set(neighbor_search_radius=0.0, num_neighbors={"DIRECT1": 1, "DIRECT7": 7, "DIRECT27": 27}[method])

    def set_max_correspondence_distance(self, d): self._set(max_corr_dist=float(d))   # corr_dist_threshold_ (GICP family); the point-to-plane search radius is set_max_range
    def set_max_range(self, r): self._set(max_range=float(r))                        # IVox max_range (ivox3d.h:80)
    def set_num_threads(self, n): pass                                              # setNumThreads: no meaning on the GPU
    def set_stream(self, hip_stream: int): self._check(self._L.pcm_set_stream(self._h, hip_stream))
    def set_profiling(self, flags: int): self._check(self._L.pcm_set_profiling(self._h, int(flags)))

    # -- inputs -------------------------------------------------------------
    def set_input_target(self, cloud, tag: int = 0):
        ptr, n, stride, mem, keep = _points(cloud)
        self._check(self._L.pcm_set_target(self._h, ptr, n, stride, mem, tag))

    def set_input_source(self, cloud, tag: int = 0):
        ptr, n, stride, mem, keep = _points(cloud)
        self._check(self._L.pcm_set_source(self._h, ptr, n, stride, mem, tag))

    def swap_source_and_target(self): self._check(self._L.pcm_swap_source_and_target(self._h))
    def clear_source(self): self._check(self._L.pcm_clear_source(self._h))
    def clear_target(self): self._check(self._L.pcm_clear_target(self._h))

    # -- compute ------------------------------------------------------------
    def align(self, initial_guess=None) -> RegistrationResult:
        g = np.eye(4, dtype=np.float32) if initial_guess is None else np.ascontiguousarray(initial_guess, dtype=np.float32)
        res = capi.PcmResult()
        self._check(self._L.pcm_align(self._h, g.ctypes.data, C.byref(res)), allow=(capi.PCM_OK, capi.PCM_ERR_NOT_CONVERGED))
        self._last = _result(res)
        return self._last

    def evaluate_cost(self, T):
        """LsqRegistration::evaluateCost -> (cost, H, b, num_inliers)  (lsq_registration_impl.hpp:46-49)."""
        T = np.ascontiguousarray(T, dtype=np.float64)
        H = np.zeros((6, 6)); b = np.zeros(6)
        cost = C.c_double(); inl = C.c_int32()
        self._check(self._L.pcm_linearize(self._h, T.ctypes.data, H.ctypes.data, b.ctypes.data, C.byref(cost), C.byref(inl)))
        return cost.value, H, b, inl.value

    linearize = evaluate_cost

    def compute_error(self, T) -> float:
        T = np.ascontiguousarray(T, dtype=np.float64)
        cost = C.c_double()
        self._check(self._L.pcm_compute_error(self._h, T.ctypes.data, C.byref(cost)))
        return cost.value

    def obs_model(self, rot_xyzw, pos, off_R_xyzw, off_T, extrinsic_est_en=False, converge=True):
        """jueying_lio's ObsModel + the IEKF reduction (pcm_obs_model):
        returns (HTH 12x12, HTh 12, n_eff, sum_h2, valid)."""
        st = capi.PcmLioState()
        st.rot[:] = list(rot_xyzw); st.pos[:] = list(pos); st.off_R[:] = list(off_R_xyzw); st.off_T[:] = list(off_T)
        out = capi.PcmObsResult()
        self._check(self._L.pcm_obs_model(self._h, C.byref(st), int(extrinsic_est_en), int(converge), C.byref(out)))
        return np.array(out.HTH[:]).reshape(12, 12), np.array(out.HTh[:]), out.n_eff, out.sum_h2, bool(out.valid)

    def get_lio_members(self, n: int):
        """(residuals_, point_selected_surf_) as the last obs_model left them (reference-semantics mode only)."""
        res = np.zeros(n, np.float32); sel = np.zeros(n, np.uint8)
        self._check(self._L.pcm_get_lio_members(self._h, res.ctypes.data, sel.ctypes.data, n))
        return res, sel.astype(bool)

    def target_insert(self, cloud):
        """IVox::AddPoints: append points to the sliding submap (LRU beyond map_capacity voxels)."""
        ptr, n, stride, mem, keep = _points(cloud)
        self._check(self._L.pcm_target_insert(self._h, ptr, n, stride, mem))

    def map_incremental(self, rot_xyzw, pos, off_R_xyzw, off_T, filter_size_map: float, ekf_inited: bool = True) -> int:
        """LaserMapping::MapIncremental with the add-filter; returns the number of points inserted."""
        st = capi.PcmLioState()
        st.rot[:] = list(rot_xyzw); st.pos[:] = list(pos); st.off_R[:] = list(off_R_xyzw); st.off_T[:] = list(off_T)
        n = C.c_size_t()
        self._check(self._L.pcm_map_incremental(self._h, C.byref(st), C.c_float(filter_size_map), int(ekf_inited), C.byref(n)))
        return n.value

    def lio_frame_begin(self, msg_points, poses=None, rot_xyzw=(0, 0, 0, 1.0), pos=(0, 0, 0), off_R_xyzw=(0, 0, 0, 1.0), off_T=(0, 0, 0),
                        num_scans: int = 6, point_filter_num: int = 2, blind: float = 0.1, leaf_size: float = 0.5) -> int:
        """Front end of one LaserMapping::Run frame on the device (pcm_lio_frame_begin): raw livox CustomPoint records (20-byte
        structured array; host array or a CUDA uint8 tensor) -> driver-message filter -> motion compensation (poses: (K,22) Pose6D rows,
        None = none) -> voxel-grid down-sampling -> source of this object.  Returns the number of scan points."""
        mem, ptr, n = capi.MEM_HOST, None, 0
        if hasattr(msg_points, "data_ptr"):
            assert msg_points.is_cuda and msg_points.element_size() * msg_points.numel() % 20 == 0
            mem, ptr, n = capi.MEM_DEVICE, msg_points.data_ptr(), msg_points.element_size() * msg_points.numel() // 20
            self._keep_frame = msg_points
        else:
            a = np.ascontiguousarray(msg_points)
            assert a.dtype.itemsize == 20
            ptr, n = a.ctypes.data, len(a)
            self._keep_frame = a
        prm = capi.PcmLioFrameParams(int(num_scans), int(point_filter_num), float(blind), float(leaf_size), 0)
        st = capi.PcmLioState()
        st.rot[:] = list(map(float, rot_xyzw)); st.pos[:] = list(map(float, pos)); st.off_R[:] = list(map(float, off_R_xyzw)); st.off_T[:] = list(map(float, off_T))
        pp, npose = None, 0
        if poses is not None:
            pa = np.ascontiguousarray(poses, dtype=np.float64)
            assert pa.ndim == 2 and pa.shape[1] == 22
            pp, npose = pa.ctypes.data, pa.shape[0]
        m = C.c_size_t()
        self._check(self._L.pcm_lio_frame_begin(self._h, C.c_void_p(ptr), C.c_size_t(n), mem, C.byref(prm), C.c_void_p(pp), int(npose), C.byref(st), C.byref(m)))
        return m.value

    def lio_frame_end(self, rot_xyzw, pos, off_R_xyzw, off_T, filter_size_map: float, ekf_inited: bool = True) -> int:
        """Back end of the frame (pcm_lio_frame_end = MapIncremental with the updated state); returns the points inserted."""
        return self.map_incremental(rot_xyzw, pos, off_R_xyzw, off_T, filter_size_map, ekf_inited)

    def get_source(self) -> np.ndarray:
        n = C.c_size_t()
        self._check(self._L.pcm_get_source(self._h, None, 0, C.byref(n)))
        out = np.zeros((n.value, 3), np.float32)
        if n.value:
            self._check(self._L.pcm_get_source(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def ndt_derivatives(self, p, hessian="float"):
        """pclomp NDT: (score, gradient, Hessian) at the pose vector p = (x, y, z, roll, pitch, yaw)
        (computeDerivatives, ndt_omp_impl.hpp:168-267); hessian = "float" | None | "double" (computeHessian :498-559)."""
        p = np.ascontiguousarray(p, dtype=np.float64)
        g = np.zeros(6); H = np.zeros((6, 6)); score = C.c_double()
        mode = {"float": 0, None: 1, "double": 2}[hessian]
        self._check(self._L.pcm_ndt_derivatives(self._h, p.ctypes.data, mode, C.byref(score), g.ctypes.data, H.ctypes.data))
        return score.value, g, H

    def ndt_score(self, T) -> float:
        """pclomp NDT calculateScore of the source transformed by T (ndt_omp_impl.hpp:835-880)."""
        T = np.ascontiguousarray(T, dtype=np.float32)
        s = C.c_double()
        self._check(self._L.pcm_ndt_score(self._h, T.ctypes.data, C.byref(s)))
        return s.value

    def get_covariances(self, target: bool = False) -> np.ndarray:
        """(N,3,3) regularised covariances of the source (or target) cloud, input order
        (FastGICP::getSourceCovariances / getTargetCovariances, fast_gicp.hpp:64-70)."""
        n = C.c_size_t()
        self._check(self._L.pcm_get_covariances(self._h, int(bool(target)), None, 0, C.byref(n)))
        out = np.zeros((n.value, 3, 3), np.float64)
        if n.value:
            self._check(self._L.pcm_get_covariances(self._h, int(bool(target)), out.ctypes.data, n.value, C.byref(n)))
        return out

    def set_covariances(self, covs, target: bool = False):
        """setSourceCovariances / setTargetCovariances (fast_gicp_impl.hpp:93-100): (N,3,3) or (N,4,4) float64, input order."""
        a = np.ascontiguousarray(covs, np.float64)
        elems = a.shape[1] * a.shape[2]
        self._check(self._L.pcm_set_covariances(self._h, int(bool(target)), a.ctypes.data, a.shape[0], elems))

    def set_correspondence_randomness(self, k): self._set(k_correspondences=int(k))   # setCorrespondenceRandomness  fast_gicp_impl.hpp:61-63
    def set_regularization_method(self, m): self._set(regularization=m)               # setRegularizationMethod      :66-68

    def undistort(self, points, time_index, poses, rot_xyzw, pos, off_R_xyzw, off_T):
        """Motion compensation of a scan into its frame-end pose, in place (ImuProcess::UndistortPcl's backward propagation,
        jueying_lio/include/imu_processing.hpp:245-285).  points: (N,F) float32 (x,y,z first, time [ms] in column time_index,
        sorted by time); poses: (K,22) float64 rows = Pose6D (offset_time, acc, gyr, vel, pos, rot row-major)."""
        assert points.dtype == np.float32 and points.flags["C_CONTIGUOUS"] and points.ndim == 2
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        assert poses.ndim == 2 and poses.shape[1] == 22
        st = capi.PcmLioState()
        st.rot[:] = list(map(float, rot_xyzw)); st.pos[:] = list(map(float, pos))
        st.off_R[:] = list(map(float, off_R_xyzw)); st.off_T[:] = list(map(float, off_T))
        self._check(self._L.pcm_undistort(self._h, points.ctypes.data, points.shape[0], points.strides[0], 4 * int(time_index), capi.MEM_HOST,
                                          poses.ctypes.data, poses.shape[0], C.byref(st)))
        return points

    def voxel_downsample(self, points, leaf_size) -> np.ndarray:
        """pcl::VoxelGrid down-sampling of an (N,F) float32 scan -> (M,F) centroids in leaf-index order
        (voxel_scan_.filter(), jueying_lio/src/laser_mapping.cc:323-328)."""
        points = np.ascontiguousarray(points, dtype=np.float32)
        out = np.zeros_like(points)
        m = C.c_size_t()
        self._check(self._L.pcm_voxel_downsample(self._h, points.ctypes.data, points.shape[0], points.strides[0], capi.MEM_HOST, float(leaf_size),
                                                 out.ctypes.data, out.shape[0], C.byref(m)))
        return out[:m.value].copy()

    def gicp_bfgs_set_correspondences(self, src, tgt, idx_src, idx_tgt, mahalanobis):
        """Pack the correspondence set of one outer GICP-BFGS iteration (pclomp gicp_omp_impl.hpp:199-203): src/tgt (N,F) float32
        clouds (x y z first), index pairs, mahalanobis_ as (N_src,16) float32 (column-major Matrix4f per source point)."""
        src = np.ascontiguousarray(src, np.float32); tgt = np.ascontiguousarray(tgt, np.float32)
        if src.ndim != 2 or tgt.ndim != 2 or src.shape[1] != tgt.shape[1] or src.shape[1] < 3:
            raise ValueError("src/tgt: (N,F>=3) float32 with equal record sizes")
        idx_src = np.ascontiguousarray(idx_src, np.int32); idx_tgt = np.ascontiguousarray(idx_tgt, np.int32)
        maha = np.ascontiguousarray(mahalanobis, np.float32).reshape(-1, 16)
        if len(idx_src) != len(idx_tgt) or maha.shape[0] != src.shape[0]:
            raise ValueError("one index pair per correspondence, one Mahalanobis matrix per source point")
        self._check(self._L.pcm_gicp_bfgs_set_correspondences(self._h, src.ctypes.data, src.shape[0], tgt.ctypes.data, tgt.shape[0], src.strides[0],
                                                               idx_src.ctypes.data, idx_tgt.ctypes.data, len(idx_src), maha.ctypes.data, capi.MEM_HOST))

    def gicp_bfgs_fdf(self, base_T, x, mode: int = 2):
        """(f, g) of pclomp's OptimizationFunctorWithIndices at x (gicp_omp_impl.hpp:246-365); mode 0 = operator(), 1 = df, 2 = fdf."""
        base = np.ascontiguousarray(base_T, np.float32).reshape(16); xx = np.ascontiguousarray(x, np.float64).reshape(6)
        f = C.c_double(float("nan")); g = np.full(6, np.nan)
        self._check(self._L.pcm_gicp_bfgs_fdf(self._h, base.ctypes.data, xx.ctypes.data, int(mode), C.byref(f), g.ctypes.data))
        return f.value, g

    def gicp_bfgs_update_correspondences(self, transformation, guess) -> int:
        """pclomp GICP's correspondence step on the device (gicp_omp_impl.hpp:405-472); the pairs become the functor's record set."""
        T = np.ascontiguousarray(transformation, np.float32).reshape(16); G = np.ascontiguousarray(guess, np.float32).reshape(16)
        m = C.c_size_t()
        self._check(self._L.pcm_gicp_bfgs_update_correspondences(self._h, T.ctypes.data, G.ctypes.data, C.byref(m)))
        self._bfgs_m = m.value
        return m.value

    def gicp_bfgs_get_correspondences(self):
        """(idx_src, idx_tgt, mahalanobis (m,3,3) float32) of the last device-side correspondence step."""
        m = self._bfgs_m
        isrc = np.zeros(m, np.int32); itgt = np.zeros(m, np.int32); M = np.zeros((m, 9), np.float32)
        self._check(self._L.pcm_gicp_bfgs_get_correspondences(self._h, isrc.ctypes.data, itgt.ctypes.data, M.ctypes.data, m))
        return isrc, itgt, M.reshape(-1, 3, 3)

    def livox_filter(self, msg_points, num_scans: int = 6, point_filter_num: int = 1, blind: float = 0.01) -> np.ndarray:
        """PointCloudPreprocess::AviaHandler (pointcloud_preprocess.cc:44-88): livox CustomPoint records (20-byte structured array:
        offset_time u4, x y z f4, reflectivity tag line u1, pad) -> (m, 12) float32 pcl::PointXYZINormal records in input order."""
        a = np.ascontiguousarray(msg_points)
        assert a.dtype.itemsize == 20
        out = np.zeros((max(len(a), 1), 12), np.float32)
        m = C.c_size_t()
        self._check(self._L.pcm_livox_filter(self._h, a.ctypes.data, len(a), capi.MEM_HOST, int(num_scans), int(point_filter_num), float(blind), out.ctypes.data, len(out), C.byref(m)))
        return out[:m.value].copy()

    def get_target(self) -> np.ndarray:
        """(M,3) current target points in insertion order."""
        n = C.c_size_t()
        self._check(self._L.pcm_get_target(self._h, None, 0, C.byref(n)))
        out = np.zeros((n.value, 3), np.float32)
        if n.value:
            self._check(self._L.pcm_get_target(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def get_planes(self, n: int) -> np.ndarray:
        """(n,4) planes fitted by the last evaluate_cost (NaN row = point not selected)."""
        out = np.zeros((n, 4), np.float32)
        self._check(self._L.pcm_get_planes(self._h, out.ctypes.data, n))
        return out

    def get_fitness_score(self, max_range: float = float(np.finfo(np.float64).max), T=None) -> float:
        """pcl::Registration::getFitnessScore(max_range) (pygicp get_fitness_score, main.cpp:169-215): mean squared distance of
        the source points under the final transformation (or T) to their exact nearest target points, on the device."""
        if T is None:
            T = self._last.T
        T = np.ascontiguousarray(T, dtype=np.float32)
        s = C.c_double()
        self._check(self._L.pcm_fitness_score(self._h, T.ctypes.data, float(max_range), C.byref(s)))
        return s.value

    def get_final_transformation(self): return self._last.T
    def get_final_hessian(self): return self._last.H
    def has_converged(self): return self._last.converged

    def stats(self) -> dict:
        s = capi.PcmStats()
        self._check(self._L.pcm_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in capi.PcmStats._fields_ if k != "reserved"}

    def phase_cycles(self):
        out = (C.c_uint64 * 8)()
        self._check(self._L.pcm_debug_phase_cycles(self._h, out))
        return list(out)

    def reset_stats(self): self._check(self._L.pcm_reset_stats(self._h))


class P2PlaneRegistration(Registration):
    """Point-to-plane scan-to-submap ICP with jueying_lio's matcher semantics
    (5-NN in the voxel hash -> plane fit -> n.p+d; laser_mapping.cc:592-701)
    under fast_gicp's GN/LM loop."""
    model = "P2PLANE"


class GicpRegistration(Registration):
    """Generalized ICP with FastGICP's semantics (fast_gicp/include/fast_gicp/gicp/impl/fast_gicp_impl.hpp):
    20-NN covariances regularised to planes, exact nearest-neighbour correspondences, distribution-to-
    distribution cost in double.  ``voxel_resolution`` only sizes the search grid (results do not depend on it)."""
    model = "GICP"
    defaults = {"voxel_resolution": 0.5}


class VgicpRegistration(Registration):
    """Voxelized GICP with FastVGICP's CPU semantics (impl/fast_vgicp_impl.hpp, fast_vgicp_voxel.hpp):
    additive voxel distributions at resolution 1.0, DIRECT1 neighbourhood by default (:22-25)."""
    model = "VGICP"
    defaults = {"voxel_resolution": 1.0, "num_neighbors": 1}


class PclNdtRegistration(Registration):
    """pclomp::NormalDistributionsTransform (pointcloud_match/ndt_omp/include/pclomp/ndt_omp_impl.hpp):
    Newton step with the analytic Hessian + More-Thuente line search on VoxelGridCovariance leaves;
    defaults of that class: resolution 1.0, step 0.1, outlier ratio 0.55, epsilon 0.1, 35 iterations, DIRECT7 (:48,60-63).
    ``num_neighbors``: 0 = KDTREE (radius search over the leaf centroids), 1 / 7 / 27 = DIRECT1 / DIRECT7 / DIRECT26."""
    model = "NDT_OMP"
    defaults = {"voxel_resolution": 1.0, "num_neighbors": 7, "max_iterations": 35, "translation_eps": 0.1}

    def set_step_size(self, s): self._set(ndt_step_size=float(s))                  # setStepSize       ndt_omp.h
    def set_outlier_ratio(self, r): self._set(ndt_outlier_ratio=float(r))          # setOutlierRatio   ndt_omp.h:188


class VgicpCudaRegistration(Registration):
    """FastVGICPCuda's float core (fast_gicp/src/fast_gicp/cuda/*.cu behind impl/fast_vgicp_cuda_impl.hpp): float 20-NN
    covariances and voxel distributions, D2D cost with w = sqrt(n); resolution 1.0, DIRECT1, PLANE (:24-27)."""
    model = "VGICP_CUDA"
    defaults = {"voxel_resolution": 1.0, "num_neighbors": 1}

    def set_nearest_neighbor_search_method(self, method: str):
        """setNearestNeighborSearchMethod (fast_vgicp_cuda_impl.hpp:64-66): "CPU_PARALLEL_KDTREE" / "GPU_BRUTEFORCE" (exact kNN
        covariances) or "GPU_RBF_KERNEL" (cuda/covariance_estimation_rbf.cu)."""
        self._set(covariance_method=1 if method == "GPU_RBF_KERNEL" else 0)

    def set_kernel_width(self, kernel_width: float, max_dist: float = -1.0):
        """setKernelWidth (fast_vgicp_cuda_impl.hpp:46-52): max_dist defaults to 5 x kernel_width."""
        self._set(rbf_kernel_width=float(kernel_width), rbf_max_dist=float(kernel_width * 5.0 if max_dist <= 0 else max_dist))


class NdtRegistration(Registration):
    """NDT on Gaussian voxels with the reference NDTCuda's semantics
    (fast_gicp/include/fast_gicp/ndt/ndt_cuda.hpp:21-71, src/fast_gicp/cuda/ndt_cuda.cu):
    D2D distance mode, DIRECT7 neighbourhood and resolution 1.0 by default (ndt_cuda.cu:15-22)."""
    model = "NDT_D2D"
    defaults = {"voxel_resolution": 1.0, "num_neighbors": 7}

    def set_distance_mode(self, mode: str):      # setDistanceMode(NDTDistanceMode)
        self._set(model=capi.MODEL["NDT_" + mode.upper()])


def align_batch(regs, guesses, device_out=None):
    """Align a batch of independent registration objects in lock-step launches
    (pcm_align_batch).  ``device_out``: optional device pointer (int) receiving
    the packed ``pcm_result`` records, e.g. a tensor handed to an RCCL gather."""
    L = capi.load_library()
    n = len(regs)
    g = np.ascontiguousarray(guesses, dtype=np.float32).reshape(n, 16)
    arr = (C.c_void_p * n)(*[r.handle for r in regs])
    out = (capi.PcmResult * n)()
    rc = L.pcm_align_batch(arr, n, g.ctypes.data, out, device_out)
    if rc not in (capi.PCM_OK, capi.PCM_ERR_NOT_CONVERGED):
        raise capi.PcmError(rc, (L.pcm_last_error(regs[0].handle) or b"").decode())
    res = [_result(out[i]) for i in range(n)]
    for r, x in zip(regs, res):
        r._last = x
    return res
