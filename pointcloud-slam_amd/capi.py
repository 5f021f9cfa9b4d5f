"""ctypes binding of libpcm_amd.so (the C ABI declared in include/pcm_amd.h)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# every symbol include/pcm_amd.h declares (tests check the .so exports each one)
SYMBOLS = [
    "pcm_abi_version", "pcm_default_config", "pcm_create", "pcm_destroy", "pcm_last_error",
    "pcm_get_config", "pcm_set_config", "pcm_set_stream", "pcm_set_target", "pcm_set_source",
    "pcm_swap_source_and_target", "pcm_clear_source", "pcm_clear_target", "pcm_align",
    "pcm_linearize", "pcm_compute_error", "pcm_get_planes", "pcm_get_lio_members", "pcm_obs_model", "pcm_target_insert", "pcm_map_incremental", "pcm_get_target", "pcm_get_covariances", "pcm_set_covariances", "pcm_ndt_derivatives", "pcm_ndt_score", "pcm_fitness_score", "pcm_undistort", "pcm_voxel_downsample", "pcm_livox_filter", "pcm_gicp_bfgs_set_correspondences", "pcm_gicp_bfgs_fdf", "pcm_gicp_bfgs_update_correspondences", "pcm_gicp_bfgs_get_correspondences", "pcm_align_batch", "pcm_set_profiling", "pcm_debug_phase_cycles",
    "pcm_get_stats", "pcm_reset_stats", "pcm_lio_frame_begin", "pcm_lio_frame_end", "pcm_get_source",
]

PCM_ABI_VERSION = 3   # include/pcm_amd.h
PCM_OK = 0
PCM_FLAG_NO_LDS_STAGING = 1
PCM_FLAG_FUSED_STEP = 2
PCM_FLAG_COUNTED_SEARCH = 8           # k_linearize_counted instead of k_linearize (A/B; same results, measured slower)
PCM_FLAG_NEIGHBOUR_LISTS = 16           # static targets: per-voxel candidate lists built with the map (default: from the 2nd registration on), same results
PCM_FLAG_NO_NEIGHBOUR_LISTS = 64        # never: the tile kernel serves every pass
PCM_FLAG_REFERENCE_KNN_ORDER = 32      # neighbours in the order of libstdc++'s std::nth_element (the reference's), slower kernel
PCM_FLAG_LIO_REFERENCE_SEMANTICS = 4   # pcm_obs_model keeps LaserMapping's per-point members across calls and scans
PCM_ERR_NOT_CONVERGED = -6
PCM_ERR_INTERNAL = -7
MEM_HOST, MEM_DEVICE = 0, 1
MODEL = {"P2PLANE": 0, "GICP": 1, "VGICP": 2, "NDT_P2D": 3, "NDT_D2D": 4, "NDT_OMP": 5, "VGICP_CUDA": 6}
OPTIMIZER = {"GN": 0, "LM": 1}
REGULARIZATION = {"NONE": 0, "MIN_EIG": 1, "NORMALIZED_MIN_EIG": 2, "PLANE": 3, "FROBENIUS": 4, "PCLOMP": 5}


class PcmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("pcm error %d: %s" % (code, msg))
        self.code = code


class PcmConfig(C.Structure):
    _fields_ = [("model", C.c_int32), ("optimizer", C.c_int32), ("max_iterations", C.c_int32),
                ("lm_max_iterations", C.c_int32), ("rotation_eps", C.c_double),
                ("translation_eps", C.c_double), ("lm_init_lambda_factor", C.c_double),
                ("voxel_resolution", C.c_float), ("num_neighbors", C.c_int32), ("knn", C.c_int32),
                ("min_knn", C.c_int32), ("max_range", C.c_float), ("plane_threshold", C.c_float),
                ("max_corr_dist", C.c_float), ("k_correspondences", C.c_int32),
                ("regularization", C.c_int32), ("sort_source", C.c_int32), ("flags", C.c_int32),
                ("map_capacity", C.c_int32), ("ndt_step_size", C.c_float), ("ndt_outlier_ratio", C.c_float),
                ("batch_window", C.c_int32), ("voxel_mode", C.c_int32), ("neighbor_search_radius", C.c_float),
                ("covariance_method", C.c_int32), ("rbf_kernel_width", C.c_float), ("rbf_max_dist", C.c_float)]


class PcmResult(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("T64", C.c_double * 16), ("H", C.c_double * 36),
                ("cost", C.c_double), ("iterations", C.c_int32), ("converged", C.c_int32),
                ("num_linearize", C.c_int32), ("num_compute_error", C.c_int32),
                ("num_inliers", C.c_int32), ("status", C.c_int32)]


class PcmLioState(C.Structure):
    _fields_ = [("rot", C.c_double * 4), ("pos", C.c_double * 3), ("off_R", C.c_double * 4), ("off_T", C.c_double * 3)]


class PcmObsResult(C.Structure):
    _fields_ = [("HTH", C.c_double * 144), ("HTh", C.c_double * 12), ("sum_h2", C.c_double), ("n_eff", C.c_int32), ("valid", C.c_int32)]


class PcmLioFrameParams(C.Structure):
    _fields_ = [("num_scans", C.c_int32), ("point_filter_num", C.c_int32), ("blind", C.c_double), ("leaf_size", C.c_float), ("reserved", C.c_int32)]


class PcmStats(C.Structure):
    _fields_ = [("linearize_launches", C.c_uint64), ("point_passes", C.c_uint64),
                ("candidates", C.c_uint64), ("slots_probed", C.c_uint64), ("linearize_ms", C.c_double),
                ("target_voxels", C.c_uint64), ("target_slots", C.c_uint64), ("tiles", C.c_uint64),
                ("tiles_lds_grid", C.c_uint64), ("tiles_lds_points", C.c_uint64), ("residual_ms", C.c_double),
                ("timed_launches", C.c_uint64), ("timed_pair_slots", C.c_uint64), ("launched_pair_slots", C.c_uint64),
                ("lru_batch_hazards", C.c_uint64)]


def library_path() -> str:
    """The in-tree build; PCM_AMD_LIBRARY names another build of the same ABI (A/B measurements of two builds on one box)."""
    return os.environ.get("PCM_AMD_LIBRARY") or os.path.join(_HERE, "libpcm_amd.so")


def build_library(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    so = library_path()
    src_dir = os.path.join(_HERE, "csrc")
    inc = os.path.join(os.path.dirname(_HERE), "include", "pcm_amd.h")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".hip", ".h"))] + [inc]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", src_dir, "-s", "-j4"])
    return so


def load_library():
    """Load libpcm_amd.so.  Raises (never falls back) when it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # One HIP runtime per process: the torch wheel bundles its own libamdhip64.so.7
    # (same SONAME as /opt/rocm's).  If libpcm_amd.so pulled in the system copy
    # first, a later `import torch` would load a second runtime that finds no GPU;
    # importing torch first lets the dynamic loader reuse torch's copy for us.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    so = library_path()
    if not os.path.exists(so):
        raise PcmError(-3, "libpcm_amd.so is not built (run __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(so)
    vp, i32, u64, sz = C.c_void_p, C.c_int, C.c_uint64, C.c_size_t
    L.pcm_abi_version.restype = i32
    L.pcm_default_config.argtypes = [C.POINTER(PcmConfig)]
    L.pcm_default_config.restype = None
    L.pcm_create.argtypes = [i32, C.POINTER(PcmConfig)]
    L.pcm_create.restype = vp
    L.pcm_destroy.argtypes = [vp]
    L.pcm_destroy.restype = None
    L.pcm_last_error.argtypes = [vp]
    L.pcm_last_error.restype = C.c_char_p
    L.pcm_get_config.argtypes = [vp, C.POINTER(PcmConfig)]
    L.pcm_set_config.argtypes = [vp, C.POINTER(PcmConfig)]
    L.pcm_set_stream.argtypes = [vp, vp]
    for f in (L.pcm_set_target, L.pcm_set_source):
        f.argtypes = [vp, vp, sz, sz, i32, u64]
    for f in (L.pcm_swap_source_and_target, L.pcm_clear_source, L.pcm_clear_target, L.pcm_reset_stats):
        f.argtypes = [vp]
    L.pcm_align.argtypes = [vp, vp, C.POINTER(PcmResult)]
    L.pcm_linearize.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.pcm_compute_error.argtypes = [vp, vp, C.POINTER(C.c_double)]
    L.pcm_get_planes.argtypes = [vp, vp, sz]
    L.pcm_get_lio_members.argtypes = [vp, vp, vp, sz]
    L.pcm_obs_model.argtypes = [vp, C.POINTER(PcmLioState), i32, i32, C.POINTER(PcmObsResult)]
    L.pcm_target_insert.argtypes = [vp, vp, sz, sz, i32]
    L.pcm_map_incremental.argtypes = [vp, C.POINTER(PcmLioState), C.c_float, i32, C.POINTER(sz)]
    L.pcm_get_target.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.pcm_get_covariances.argtypes = [vp, C.c_int, vp, sz, C.POINTER(sz)]
    L.pcm_livox_filter.argtypes = [vp, vp, sz, C.c_int, C.c_int, C.c_int, C.c_double, vp, sz, C.POINTER(sz)]
    L.pcm_set_covariances.argtypes = [vp, C.c_int, vp, sz, C.c_int]
    L.pcm_ndt_derivatives.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_double), vp, vp]
    L.pcm_ndt_score.argtypes = [vp, vp, C.POINTER(C.c_double)]
    L.pcm_fitness_score.argtypes = [vp, vp, C.c_double, C.POINTER(C.c_double)]
    L.pcm_voxel_downsample.argtypes = [vp, vp, sz, sz, C.c_int, C.c_float, vp, sz, C.POINTER(sz)]
    L.pcm_gicp_bfgs_set_correspondences.argtypes = [vp, vp, sz, vp, sz, sz, vp, vp, sz, vp, C.c_int]
    L.pcm_gicp_bfgs_fdf.argtypes = [vp, vp, vp, C.c_int, C.POINTER(C.c_double), vp]
    L.pcm_gicp_bfgs_update_correspondences.argtypes = [vp, vp, vp, C.POINTER(sz)]
    L.pcm_gicp_bfgs_get_correspondences.argtypes = [vp, vp, vp, vp, sz]
    L.pcm_undistort.argtypes = [vp, vp, sz, sz, sz, C.c_int, vp, C.c_int, C.POINTER(PcmLioState)]
    L.pcm_align_batch.argtypes = [C.POINTER(vp), i32, vp, vp, vp]
    L.pcm_set_profiling.argtypes = [vp, i32]
    L.pcm_debug_phase_cycles.argtypes = [vp, vp]
    L.pcm_get_stats.argtypes = [vp, C.POINTER(PcmStats)]
    L.pcm_lio_frame_begin.argtypes = [vp, vp, sz, C.c_int, C.POINTER(PcmLioFrameParams), vp, C.c_int, C.POINTER(PcmLioState), C.POINTER(sz)]
    L.pcm_lio_frame_end.argtypes = [vp, C.POINTER(PcmLioState), C.c_float, i32, C.POINTER(sz)]
    L.pcm_get_source.argtypes = [vp, vp, sz, C.POINTER(sz)]
    _LIB = L
    return L
