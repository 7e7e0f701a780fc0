"""ctypes binding of libaar.so -- the C ABI declared in include/aar.h.

This is test / benchmark plumbing: the product is the C-ABI library (HIP kernels for gfx950) and the C++
MultiCamMapper mirror in automatic-ar_amd/host/.  Nothing here computes; every numeric call lands in a
HIP kernel and raises AarError(AAR_ERR_NO_DEVICE) on a machine without a GPU -- there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AAR_LIB") or os.path.join(os.path.dirname(_HERE), "libaar.so")   # AAR_LIB: A/B runs of two builds

AAR_OK = 0
AAR_ERR_INVALID, AAR_ERR_NO_DEVICE, AAR_ERR_HIP, AAR_ERR_UNSUPPORTED = -1, -2, -3, -4
AAR_ERR_NUMERIC, AAR_ERR_IO, AAR_ERR_COMM = -5, -6, -7
RES_F32, RES_F64 = 0, 1
COMM_ID_BYTES = 128

# every symbol include/aar.h declares (tests/test_capi_symbols.py checks the header against this list and the .so)
SYMBOLS = [
    "aar_last_error", "aar_dataset_free", "aar_dataset_full_len", "aar_synth_default", "aar_synth_generate",
    "aar_solution_read", "aar_solution_write", "aar_solution_write_yaml", "aar_detections_write",
    "aar_rodrigues_vec2mat", "aar_rodrigues_mat2vec", "aar_plan_shards", "aar_comm_make_id", "aar_comm_create",
    "aar_comm_destroy", "aar_problem_desc_from_dataset", "aar_problem_create", "aar_problem_destroy",
    "aar_problem_full_len", "aar_problem_num_vars", "aar_problem_local_obs", "aar_eval_residuals",
    "aar_eval_normal_equations", "aar_eval_damped_step", "aar_lm_default_params", "aar_lm_init", "aar_lm_step",
    "aar_lm_get_solution", "aar_lm_solve", "aar_get_stage_times", "aar_reproj_stats", "aar_device_count",
    "aar_device_synchronize", "aar_set_kernel_profiling", "aar_get_kernel_times", "aar_kernel_name",
    "aar_problem_set_huber_delta", "aar_problem_get_huber_delta", "aar_track", "aar_cam_config_read",
    "aar_undistort_points", "aar_local_group_create", "aar_local_group_destroy", "aar_comm_create_local",
    "aar_cam_configs_read", "aar_detections_read", "aar_detections_free", "aar_subseqs_read", "aar_ippe_square",
    "aar_vote_transforms", "aar_init_default_params", "aar_initializer_run", "aar_initializer_object_poses",
    "aar_comm_get_stats", "aar_lm_set_step_callback", "aar_lm_set_stop_function", "aar_problem_extract_z", "aar_problem_merge_z",
    "aar_solution_read_ex", "aar_cam_configs_read_ex", "aar_set_stage_timers", "aar_problem_pcg_iterations",
    "aar_solver_default_options", "aar_problem_create_ex", "aar_problem_get_solver_stats", "aar_problem_set_test_hook",
]
NUM_KERNELS = 17
SOLVER_DIRECT, SOLVER_PCG, SOLVER_SPCG, SOLVER_AUTO = 0, 1, 2, 3
TEST_HOOK_SPCG_DROP = 1
ENV_SOLVER, ENV_DETERMINISTIC, ENV_PCG_ETA, ENV_PCG_MAX_IT = 1, 2, 4, 8
SOLVERS = {"direct": SOLVER_DIRECT, "pcg": SOLVER_PCG, "spcg": SOLVER_SPCG, "auto": SOLVER_AUTO}
SOLVER_NAMES = {v: k for k, v in SOLVERS.items()}


class AarError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("aar error %d: %s" % (code, msg))
        self.code = code


class CDataset(C.Structure):
    _fields_ = [
        ("num_cams", C.c_int32), ("num_markers", C.c_int32), ("num_frames", C.c_int32),
        ("root_cam", C.c_int32), ("root_marker", C.c_int32),
        ("cam_ids", C.POINTER(C.c_int32)), ("marker_ids", C.POINTER(C.c_int32)), ("frame_ids", C.POINTER(C.c_int32)),
        ("image_sizes", C.POINTER(C.c_int32)), ("cam_mats", C.POINTER(C.c_double)), ("dist_coeffs", C.POINTER(C.c_double)),
        ("marker_size", C.c_double), ("num_obs", C.c_int64),
        ("obs_frame", C.POINTER(C.c_int32)), ("obs_cam", C.POINTER(C.c_int32)), ("obs_marker", C.POINTER(C.c_int32)),
        ("obs_uv", C.POINTER(C.c_float)), ("x_full", C.POINTER(C.c_double)), ("x_truth", C.POINTER(C.c_double)),
        ("optimize_cam_poses", C.c_int32), ("optimize_marker_poses", C.c_int32),
        ("optimize_object_poses", C.c_int32), ("optimize_cam_intrinsics", C.c_int32),
    ]


class CSynthDesc(C.Structure):
    _fields_ = [
        ("num_cams", C.c_int32), ("num_markers", C.c_int32), ("num_frames", C.c_int32), ("seed", C.c_uint64),
        ("marker_size", C.c_double), ("noise_px", C.c_double), ("init_rot_sigma", C.c_double),
        ("init_trans_sigma", C.c_double), ("init_scale", C.c_double), ("cam_arc_deg", C.c_double),
        ("min_view_cos", C.c_double),
    ]


class CProblemDesc(C.Structure):
    _fields_ = [
        ("num_cams", C.c_int32), ("num_markers", C.c_int32), ("num_frames", C.c_int32),
        ("root_cam", C.c_int32), ("root_marker", C.c_int32),
        ("cam_mats", C.POINTER(C.c_double)), ("marker_size", C.c_double), ("num_obs", C.c_int64),
        ("obs_frame", C.POINTER(C.c_int32)), ("obs_cam", C.POINTER(C.c_int32)), ("obs_marker", C.POINTER(C.c_int32)),
        ("obs_uv", C.POINTER(C.c_float)),
        ("optimize_cam_poses", C.c_int32), ("optimize_marker_poses", C.c_int32), ("optimize_object_poses", C.c_int32),
        ("optimize_cam_intrinsics", C.c_int32),
        ("residual_mode", C.c_int32), ("with_huber", C.c_int32), ("device_id", C.c_int32), ("comm", C.c_void_p),
    ]


class CSolverOptions(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("solver", C.c_int32), ("deterministic", C.c_int32), ("pcg_max_it", C.c_int32),
                ("pcg_eta", C.c_double), ("pcg_eta_loose", C.c_double), ("pcg_eta_switch", C.c_double), ("pcg_abs_tol", C.c_double)]


class CSolverStats(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("solver", C.c_int32), ("deterministic", C.c_int32), ("last_iterations", C.c_int32),
                ("total_iterations", C.c_int64), ("solves", C.c_int64), ("fallbacks", C.c_int64), ("pcg_eta", C.c_double),
                ("pcg_max_it", C.c_int32), ("env_overrides", C.c_int32), ("same_xcd_solves", C.c_int64), ("pcg_eta_loose", C.c_double),
                ("pcg_eta_switch", C.c_double), ("pcg_abs_tol", C.c_double)]


class CLmParams(C.Structure):
    _fields_ = [
        ("max_iters", C.c_int32), ("min_error", C.c_double), ("min_step_error_diff", C.c_double),
        ("min_average_step_error_diff", C.c_double), ("tau", C.c_double), ("verbose", C.c_int32),
    ]


class CLmIter(C.Structure):
    _fields_ = [("err", C.c_double), ("mu", C.c_double), ("gain", C.c_double), ("delta_norm", C.c_double),
                ("accepted", C.c_int32), ("tries", C.c_int32)]


class CLmReport(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32), ("stop_code", C.c_int32), ("initial_err", C.c_double), ("final_err", C.c_double),
        ("final_mu", C.c_double), ("solve_seconds", C.c_double), ("trial_points", C.c_int64),
        ("trace", C.POINTER(CLmIter)), ("trace_cap", C.c_int32),
    ]


class CCommStats(C.Structure):
    _fields_ = [("world_size", C.c_int32), ("rank", C.c_int32), ("ranks_seen", C.c_int32), ("allreduce_calls", C.c_int64),
                ("allreduce_bytes", C.c_int64), ("system_allreduce_bytes", C.c_int64)]


class CStageTimes(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("unpack", "jacobian_normal_eq", "schur", "chol", "backsub", "residual",
                                          "control", "allreduce", "total")] + [("launches", C.c_int64)]


_lib = None
STEP_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int64)
STOP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)


class CCamModel(C.Structure):
    _fields_ = [("K", C.c_double * 9), ("dist", C.c_double * 12), ("n_dist", C.c_int32), ("width", C.c_int32),
                ("height", C.c_int32)]


class CDetections(C.Structure):
    _fields_ = [("num_cams", C.c_int32), ("num_frames", C.c_int32), ("num_det", C.c_int64),
                ("det_frame", C.POINTER(C.c_int32)), ("det_cam", C.POINTER(C.c_int32)), ("det_id", C.POINTER(C.c_int32)),
                ("det_uv", C.POINTER(C.c_float))]


class CInitParams(C.Structure):
    _fields_ = [("marker_size", C.c_double), ("threshold", C.c_double), ("min_detections", C.c_int32),
                ("n_excluded", C.c_int32), ("excluded_cams", C.POINTER(C.c_int32)), ("device_id", C.c_int32)]


def lib():
    """Load libaar.so (built by __graft_entry__.build() / `make -C automatic-ar_amd`).  Fails loudly when missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libaar.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    L.aar_last_error.restype = C.c_char_p
    L.aar_dataset_full_len.restype = C.c_int64
    L.aar_dataset_full_len.argtypes = [C.POINTER(CDataset)]
    L.aar_dataset_free.argtypes = [C.POINTER(CDataset)]
    L.aar_dataset_free.restype = None
    L.aar_synth_default.argtypes = [C.POINTER(CSynthDesc), C.c_int32]
    L.aar_synth_default.restype = None
    L.aar_synth_generate.argtypes = [C.POINTER(CSynthDesc), C.POINTER(C.POINTER(CDataset))]
    L.aar_solution_read.argtypes = [C.c_char_p, C.POINTER(C.POINTER(CDataset))]
    L.aar_solution_read_ex.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.POINTER(CDataset))]
    for n in ("aar_solution_write", "aar_solution_write_yaml", "aar_detections_write"):
        getattr(L, n).argtypes = [C.c_char_p, C.POINTER(CDataset)]
    L.aar_rodrigues_vec2mat.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.aar_rodrigues_vec2mat.restype = None
    L.aar_rodrigues_mat2vec.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.aar_rodrigues_mat2vec.restype = None
    L.aar_plan_shards.argtypes = [C.c_int32, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int32)]
    L.aar_comm_make_id.argtypes = [C.c_char_p]
    L.aar_comm_create.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.aar_comm_destroy.argtypes = [C.c_void_p]
    L.aar_comm_get_stats.argtypes = [C.c_void_p, C.POINTER(CCommStats)]
    L.aar_comm_destroy.restype = None
    L.aar_problem_desc_from_dataset.argtypes = [C.POINTER(CDataset), C.POINTER(CProblemDesc)]
    L.aar_problem_desc_from_dataset.restype = None
    L.aar_problem_create.argtypes = [C.POINTER(CProblemDesc), C.POINTER(C.c_void_p)]
    L.aar_problem_create_ex.argtypes = [C.POINTER(CProblemDesc), C.POINTER(CSolverOptions), C.POINTER(C.c_void_p)]
    L.aar_solver_default_options.argtypes = [C.POINTER(CSolverOptions)]
    L.aar_solver_default_options.restype = None
    L.aar_problem_get_solver_stats.argtypes = [C.c_void_p, C.POINTER(CSolverStats)]
    L.aar_problem_set_test_hook.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.aar_problem_destroy.argtypes = [C.c_void_p]
    L.aar_problem_destroy.restype = None
    for n in ("aar_problem_full_len", "aar_problem_num_vars", "aar_problem_local_obs"):
        getattr(L, n).argtypes = [C.c_void_p]
        getattr(L, n).restype = C.c_int64
    dp = C.POINTER(C.c_double)
    L.aar_eval_residuals.argtypes = [C.c_void_p, dp, dp, dp]
    L.aar_eval_normal_equations.argtypes = [C.c_void_p, dp, dp, dp, dp]
    L.aar_eval_damped_step.argtypes = [C.c_void_p, dp, C.c_double, dp]
    L.aar_lm_default_params.argtypes = [C.POINTER(CLmParams)]
    L.aar_lm_default_params.restype = None
    L.aar_lm_init.argtypes = [C.c_void_p, dp, C.POINTER(CLmParams)]
    L.aar_lm_step.argtypes = [C.c_void_p, C.POINTER(CLmIter)]
    L.aar_lm_get_solution.argtypes = [C.c_void_p, dp, dp]
    L.aar_lm_solve.argtypes = [C.c_void_p, dp, C.POINTER(CLmParams), C.POINTER(CLmReport)]
    L.aar_get_stage_times.argtypes = [C.c_void_p, C.POINTER(CStageTimes)]
    L.aar_set_stage_timers.argtypes = [C.c_void_p, C.c_int]
    L.aar_problem_pcg_iterations.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.aar_reproj_stats.argtypes = [C.c_void_p, dp, dp, dp]
    L.aar_set_kernel_profiling.argtypes = [C.c_void_p, C.c_int]
    L.aar_get_kernel_times.argtypes = [C.c_void_p, dp, C.POINTER(C.c_int64)]
    L.aar_kernel_name.argtypes = [C.c_int]
    L.aar_kernel_name.restype = C.c_char_p
    L.aar_track.argtypes = [C.c_void_p, dp, C.POINTER(CLmParams), C.POINTER(C.c_int32), dp]
    L.aar_local_group_create.argtypes = [C.c_int32, C.POINTER(C.c_void_p)]
    L.aar_local_group_destroy.argtypes = [C.c_void_p]
    L.aar_local_group_destroy.restype = None
    L.aar_comm_create_local.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.aar_cam_config_read.argtypes = [C.c_char_p, dp, dp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.aar_undistort_points.argtypes = [dp, dp, C.c_int32, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32]
    ip, lp, fp = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_float)
    L.aar_cam_configs_read.argtypes = [C.c_char_p, C.POINTER(C.POINTER(CCamModel)), ip]
    L.aar_cam_configs_read_ex.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.POINTER(CCamModel)), ip]
    L.aar_detections_read.argtypes = [C.c_char_p, ip, C.c_int32, C.POINTER(C.POINTER(CDetections))]
    L.aar_detections_free.argtypes = [C.POINTER(CDetections)]
    L.aar_detections_free.restype = None
    L.aar_subseqs_read.argtypes = [C.c_char_p, C.POINTER(ip), ip]
    L.aar_ippe_square.argtypes = [C.c_double, C.POINTER(CCamModel), C.c_int64, fp, dp, dp, dp, dp, C.c_int32]
    L.aar_vote_transforms.argtypes = [C.c_double, C.c_int64, lp, dp, dp, dp, lp, dp, dp, C.c_int32]
    L.aar_init_default_params.argtypes = [C.POINTER(CInitParams)]
    L.aar_init_default_params.restype = None
    L.aar_initializer_run.argtypes = [C.POINTER(CDetections), C.POINTER(CCamModel), C.c_int32, C.POINTER(CInitParams),
                                      C.POINTER(C.POINTER(CDataset))]
    L.aar_initializer_object_poses.argtypes = [C.POINTER(CDataset), C.POINTER(CDetections), C.POINTER(CCamModel), C.c_int32,
                                               C.POINTER(CInitParams), C.POINTER(C.POINTER(CDataset))]
    L.aar_lm_set_step_callback.argtypes = [C.c_void_p, STEP_CB, C.c_void_p, C.c_int32]
    L.aar_lm_set_stop_function.argtypes = [C.c_void_p, STOP_FN, C.c_void_p]
    L.aar_problem_extract_z.argtypes = [C.c_void_p, dp, dp]
    L.aar_problem_merge_z.argtypes = [C.c_void_p, dp, dp]
    L.aar_problem_set_huber_delta.argtypes = [C.c_void_p, C.c_float]
    L.aar_problem_get_huber_delta.argtypes = [C.c_void_p]
    L.aar_problem_get_huber_delta.restype = C.c_float
    _lib = L
    return L


def _check(rc):
    if rc != AAR_OK:
        raise AarError(rc, lib().aar_last_error().decode("utf-8", "replace"))


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _np(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class Dataset:
    """numpy copy of an aar_dataset (the content of a `.solution` file)."""

    FIELDS = ("cam_ids", "marker_ids", "frame_ids", "image_sizes", "cam_mats", "dist_coeffs", "obs_frame", "obs_cam",
              "obs_marker", "obs_uv", "x_full", "x_truth")

    def __init__(self, cptr=None):
        if cptr is None:
            return
        d = cptr.contents
        self.num_cams, self.num_markers, self.num_frames = d.num_cams, d.num_markers, d.num_frames
        self.root_cam, self.root_marker = d.root_cam, d.root_marker
        self.marker_size = d.marker_size
        self.num_obs = d.num_obs
        Cn, M, F, N = d.num_cams, d.num_markers, d.num_frames, d.num_obs
        full = 6 * (Cn - 1) + 6 * (M - 1) + 6 * F
        self.cam_ids = _np(d.cam_ids, Cn, np.int32)
        self.marker_ids = _np(d.marker_ids, M, np.int32)
        self.frame_ids = _np(d.frame_ids, F, np.int32)
        self.image_sizes = _np(d.image_sizes, 2 * Cn, np.int32).reshape(Cn, 2)
        self.cam_mats = _np(d.cam_mats, 9 * Cn, np.float64).reshape(Cn, 9)
        self.dist_coeffs = _np(d.dist_coeffs, 5 * Cn, np.float64).reshape(Cn, 5)
        self.obs_frame = _np(d.obs_frame, N, np.int32)
        self.obs_cam = _np(d.obs_cam, N, np.int32)
        self.obs_marker = _np(d.obs_marker, N, np.int32)
        self.obs_uv = _np(d.obs_uv, 8 * N, np.float32).reshape(N, 8)
        self.x_full = _np(d.x_full, full, np.float64)
        self.x_truth = _np(d.x_truth, full, np.float64) if d.x_truth else None
        self.optimize_cam_poses = bool(d.optimize_cam_poses)
        self.optimize_marker_poses = bool(d.optimize_marker_poses)
        self.optimize_object_poses = bool(d.optimize_object_poses)
        self.optimize_cam_intrinsics = bool(d.optimize_cam_intrinsics)

    @property
    def full_len(self):
        return 6 * (self.num_cams - 1) + 6 * (self.num_markers - 1) + 6 * self.num_frames

    def num_vars(self):
        n = 0
        if self.optimize_cam_poses:
            n += 6 * (self.num_cams - 1)
        if self.optimize_marker_poses:
            n += 6 * (self.num_markers - 1)
        if self.optimize_object_poses:
            n += 6 * self.num_frames
        return n

    def as_c(self):
        """A CDataset whose pointers alias this object's arrays (keep `self` alive while it is used)."""
        d = CDataset()
        d.num_cams, d.num_markers, d.num_frames = self.num_cams, self.num_markers, self.num_frames
        d.root_cam, d.root_marker = self.root_cam, self.root_marker
        d.marker_size = self.marker_size
        d.num_obs = self.num_obs
        self._keep = {}
        for name, ct, dt in (("cam_ids", C.c_int32, np.int32), ("marker_ids", C.c_int32, np.int32),
                             ("frame_ids", C.c_int32, np.int32), ("image_sizes", C.c_int32, np.int32),
                             ("cam_mats", C.c_double, np.float64), ("dist_coeffs", C.c_double, np.float64),
                             ("obs_frame", C.c_int32, np.int32), ("obs_cam", C.c_int32, np.int32),
                             ("obs_marker", C.c_int32, np.int32), ("obs_uv", C.c_float, np.float32),
                             ("x_full", C.c_double, np.float64)):
            a = np.ascontiguousarray(getattr(self, name), dtype=dt)
            self._keep[name] = a
            setattr(d, name, a.ctypes.data_as(C.POINTER(ct)))
        d.x_truth = None
        d.optimize_cam_poses = int(self.optimize_cam_poses)
        d.optimize_marker_poses = int(self.optimize_marker_poses)
        d.optimize_object_poses = int(self.optimize_object_poses)
        d.optimize_cam_intrinsics = int(self.optimize_cam_intrinsics)
        return d


def synth_desc(config_index, **over):
    sd = CSynthDesc()
    lib().aar_synth_default(C.byref(sd), config_index)
    for k, v in over.items():
        setattr(sd, k, v)
    return sd


def synth(config_index, **over):
    """Deterministic synthetic data set; config_index 2..5 = BASELINE.json configs[1..4]."""
    sd = synth_desc(config_index, **over)
    p = C.POINTER(CDataset)()
    _check(lib().aar_synth_generate(C.byref(sd), C.byref(p)))
    try:
        return Dataset(p)
    finally:
        lib().aar_dataset_free(p)


def solution_read(path, reference_indexing=False):
    p = C.POINTER(CDataset)()
    _check(lib().aar_solution_read_ex(path.encode(), 1 if reference_indexing else 0, C.byref(p)))
    try:
        return Dataset(p)
    finally:
        lib().aar_dataset_free(p)


def solution_write(path, ds):
    c = ds.as_c()
    _check(lib().aar_solution_write(path.encode(), C.byref(c)))


def solution_write_yaml(path, ds):
    c = ds.as_c()
    _check(lib().aar_solution_write_yaml(path.encode(), C.byref(c)))


def detections_write(path, ds):
    c = ds.as_c()
    _check(lib().aar_detections_write(path.encode(), C.byref(c)))


def rodrigues_vec2mat(w):
    w = np.ascontiguousarray(w, dtype=np.float64)
    R = np.zeros(9)
    lib().aar_rodrigues_vec2mat(_dptr(w), _dptr(R))
    return R.reshape(3, 3)


def rodrigues_mat2vec(R):
    R = np.ascontiguousarray(R, dtype=np.float64).reshape(9)
    w = np.zeros(3)
    lib().aar_rodrigues_mat2vec(_dptr(R), _dptr(w))
    return w


MAX_DIST = 12


def cam_config_read(path):
    """calib.{xml,yml,yaml} of one camera -> (K 3x3, dist (n,), (width, height)) -- libs/cam_config.cpp:52-80."""
    K = np.zeros(9)
    dist = np.zeros(MAX_DIST)
    n, w, h = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    _check(lib().aar_cam_config_read(path.encode(), _dptr(K), _dptr(dist), C.byref(n), C.byref(w), C.byref(h)))
    return K.reshape(3, 3), dist[: n.value].copy(), (w.value, h.value)


def undistort_points(K, dist, uv, device=0):
    """MultiCamMapper::remove_distortions for one camera's corners (libs/multicam_mapper.cpp:554-578) on the GPU."""
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    dist = np.ascontiguousarray(dist, dtype=np.float64).reshape(-1)
    uv = np.ascontiguousarray(uv, dtype=np.float32)
    out = np.empty_like(uv)
    fp = C.POINTER(C.c_float)
    _check(lib().aar_undistort_points(_dptr(K), _dptr(dist) if len(dist) else None, len(dist), uv.size // 2,
                                      uv.ctypes.data_as(fp), out.ctypes.data_as(fp), device))
    return out


# ---- Initializer (libs/initializer.cpp) ----
def cam_models(Ks, dists, sizes=None):
    """array of aar_cam_model, one per camera slot"""
    arr = (CCamModel * max(len(Ks), 1))()
    for i, (K, d) in enumerate(zip(Ks, dists)):
        K = np.asarray(K, dtype=np.float64).reshape(9)
        d = np.asarray(d, dtype=np.float64).reshape(-1)
        for j in range(9):
            arr[i].K[j] = K[j]
        for j in range(MAX_DIST):
            arr[i].dist[j] = d[j] if j < len(d) else 0.0
        arr[i].n_dist = len(d)
        if sizes is not None:
            arr[i].width, arr[i].height = int(sizes[i][0]), int(sizes[i][1])
    return arr


def cam_configs_read(folder, readdir_order=False):
    """CamConfig::read_cam_configs: list of (K 3x3, dist, (w, h)) in ascending sub-directory order (or readdir order, as the reference)"""
    p = C.POINTER(CCamModel)()
    n = C.c_int32(0)
    _check(lib().aar_cam_configs_read_ex(folder.encode(), 1 if readdir_order else 0, C.byref(p), C.byref(n)))
    out = []
    for i in range(n.value):
        m = p[i]
        out.append((np.array(m.K[:]).reshape(3, 3), np.array(m.dist[: m.n_dist]), (m.width, m.height)))
    C.CDLL(None).free(p)
    return out


class Detections:
    """numpy copy of an aar_detections (the content of an aruco.detections file)"""

    def __init__(self, num_cams, num_frames, det_frame, det_cam, det_id, det_uv):
        self.num_cams, self.num_frames = int(num_cams), int(num_frames)
        self.det_frame = np.ascontiguousarray(det_frame, dtype=np.int32)
        self.det_cam = np.ascontiguousarray(det_cam, dtype=np.int32)
        self.det_id = np.ascontiguousarray(det_id, dtype=np.int32)
        self.det_uv = np.ascontiguousarray(det_uv, dtype=np.float32).reshape(-1, 8)

    def as_c(self):
        c = CDetections()
        c.num_cams, c.num_frames, c.num_det = self.num_cams, self.num_frames, len(self.det_frame)
        ip = C.POINTER(C.c_int32)
        c.det_frame = self.det_frame.ctypes.data_as(ip)
        c.det_cam = self.det_cam.ctypes.data_as(ip)
        c.det_id = self.det_id.ctypes.data_as(ip)
        c.det_uv = self.det_uv.ctypes.data_as(C.POINTER(C.c_float))
        return c


def detections_read(path, subseqs=None):
    """Initializer::read_detections_file (libs/initializer.cpp:316-362)"""
    p = C.POINTER(CDetections)()
    ss = np.ascontiguousarray(subseqs if subseqs is not None else [], dtype=np.int32)
    _check(lib().aar_detections_read(path.encode(), ss.ctypes.data_as(C.POINTER(C.c_int32)) if len(ss) else None, len(ss),
                                     C.byref(p)))
    try:
        d = p.contents
        n = d.num_det
        return Detections(d.num_cams, d.num_frames, _np(d.det_frame, n, np.int32), _np(d.det_cam, n, np.int32),
                          _np(d.det_id, n, np.int32), _np(d.det_uv, 8 * n, np.float32))
    finally:
        lib().aar_detections_free(p)


def subseqs_read(path):
    p = C.POINTER(C.c_int32)()
    n = C.c_int32(0)
    _check(lib().aar_subseqs_read(path.encode(), C.byref(p), C.byref(n)))
    out = _np(p, n.value, np.int32)
    C.CDLL(None).free(p)
    return out


def ippe_square(marker_size, K, dist, uv, device=0):
    """aruco::solvePnP_ for n markers of one camera on the GPU: (T1[n,4,4], e1[n], T2[n,4,4], e2[n])"""
    uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 8)
    n = len(uv)
    cams = cam_models([K], [dist])
    T1 = np.zeros((max(n, 1), 16)); T2 = np.zeros((max(n, 1), 16)); e1 = np.zeros(max(n, 1)); e2 = np.zeros(max(n, 1))
    _check(lib().aar_ippe_square(float(marker_size), cams, n, uv.ctypes.data_as(C.POINTER(C.c_float)), _dptr(T1), _dptr(e1),
                                 _dptr(T2), _dptr(e2), device))
    return T1[:n].reshape(n, 4, 4), e1[:n], T2[:n].reshape(n, 4, 4), e2[:n]


def vote_transforms(marker_size, set_begin, T, T1inv, T2inv, device=0):
    """Initializer::find_best_transformation for several candidate sets on the GPU: (best[s], weight[s], cost[n])"""
    sb = np.ascontiguousarray(set_begin, dtype=np.int64)
    ns = len(sb) - 1
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(-1, 16)
    A = np.ascontiguousarray(T1inv, dtype=np.float64).reshape(-1, 16)
    B = np.ascontiguousarray(T2inv, dtype=np.float64).reshape(-1, 16)
    n = len(T)
    best = np.zeros(max(ns, 1), dtype=np.int64); weight = np.zeros(max(ns, 1)); cost = np.zeros(max(n, 1))
    lp = C.POINTER(C.c_int64)
    _check(lib().aar_vote_transforms(float(marker_size), ns, sb.ctypes.data_as(lp), _dptr(T), _dptr(A), _dptr(B),
                                     best.ctypes.data_as(lp), _dptr(weight), _dptr(cost), device))
    return best[:ns], weight[:ns], cost[:n]


def initializer_run(det, Ks, dists, marker_size, sizes=None, excluded=(), threshold=2.0, min_detections=2, device=0,
                    solution=None):
    """Initializer(...) + MultiCamMapper(Initializer&): the data set the reference writes as initial.solution.
    With `solution` (a Dataset): apps/track.cpp's use -- its cameras / markers stay fixed, only the object poses of the
    detections' frames are initialised (aar_initializer_object_poses)."""
    prm = CInitParams()
    lib().aar_init_default_params(C.byref(prm))
    prm.marker_size, prm.threshold, prm.min_detections, prm.device_id = marker_size, threshold, min_detections, device
    ex = np.ascontiguousarray(list(excluded), dtype=np.int32)
    prm.n_excluded = len(ex)
    prm.excluded_cams = ex.ctypes.data_as(C.POINTER(C.c_int32)) if len(ex) else None
    cams = cam_models(Ks, dists, sizes)
    c = det.as_c()
    p = C.POINTER(CDataset)()
    if solution is None:
        _check(lib().aar_initializer_run(C.byref(c), cams, len(Ks), C.byref(prm), C.byref(p)))
    else:
        sc = solution.as_c()
        _check(lib().aar_initializer_object_poses(C.byref(sc), C.byref(c), cams, len(Ks), C.byref(prm), C.byref(p)))
    try:
        return Dataset(p)
    finally:
        lib().aar_dataset_free(p)


def plan_shards(obs_per_frame, world):
    c = np.ascontiguousarray(obs_per_frame, dtype=np.int64)
    begin = np.zeros(world + 1, dtype=np.int32)
    _check(lib().aar_plan_shards(len(c), c.ctypes.data_as(C.POINTER(C.c_int64)), world,
                                 begin.ctypes.data_as(C.POINTER(C.c_int32))))
    return begin


def device_count():
    return lib().aar_device_count()


def lm_default_params(**over):
    p = CLmParams()
    lib().aar_lm_default_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


class Comm:
    """RCCL communicator of one rank (multi-GPU)."""

    @staticmethod
    def make_id():
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _check(lib().aar_comm_make_id(buf))
        return buf.raw

    def __init__(self, uid, world, rank, device):
        self.handle = C.c_void_p()
        self.world, self.rank = world, rank
        _check(lib().aar_comm_create(uid, world, rank, device, C.byref(self.handle)))

    @classmethod
    def local(cls, group, rank, device=0):
        """Rank `rank` of an in-process LocalGroup (all ranks on one GPU, one host thread each)."""
        self = cls.__new__(cls)
        self.handle = C.c_void_p()
        self.world, self.rank = group.world, rank
        _check(lib().aar_comm_create_local(group.handle, rank, device, C.byref(self.handle)))
        return self

    def stats(self):
        st = CCommStats()
        _check(lib().aar_comm_get_stats(self.handle, C.byref(st)))
        return {n: getattr(st, n) for n, _ in CCommStats._fields_}

    def close(self):
        if self.handle:
            lib().aar_comm_destroy(self.handle)
            self.handle = C.c_void_p()


class LocalGroup:
    """aar_local_group: the in-process stand-in for an RCCL communicator (tests / bring-up on a 1-GPU box)."""

    def __init__(self, world):
        self.world = world
        self.handle = C.c_void_p()
        _check(lib().aar_local_group_create(world, C.byref(self.handle)))

    def close(self):
        if self.handle:
            lib().aar_local_group_destroy(self.handle)
            self.handle = C.c_void_p()


class Problem:
    """aar_problem: the bundle-adjustment problem resident on one GPU."""

    def __init__(self, ds, residual_mode=RES_F32, device=0, comm=None, optimize=None, with_huber=False, intrinsics=False,
                 solver=None, deterministic=None, pcg_eta=None, pcg_max_it=None, pcg_eta_loose=None, pcg_eta_switch=None, pcg_abs_tol=None):
        """intrinsics=True: Config::optimize_cam_intrinsics -- every vector ends with 9 per camera (x_with_intrinsics builds one)
        solver ("direct" | "spcg" | "pcg" | "auto"), deterministic, pcg_eta, pcg_max_it, pcg_eta_loose, pcg_eta_switch: aar_solver_options
        (None = the library's default: solver AUTO -- direct for one tile of unknowns, SPCG wherever it fits, PCG for many entities per frame x many frames -- with
        one pose-grade forcing term and an absolute tolerance; a forcing SEQUENCE only when pcg_eta_loose is given)"""
        self.ds = ds
        self._cds = ds.as_c()
        d = CProblemDesc()
        lib().aar_problem_desc_from_dataset(C.byref(self._cds), C.byref(d))
        if optimize is not None:
            d.optimize_cam_poses, d.optimize_marker_poses, d.optimize_object_poses = [int(b) for b in optimize]
        d.optimize_cam_intrinsics = int(intrinsics)
        d.residual_mode = residual_mode
        d.with_huber = int(with_huber)
        d.device_id = device
        d.comm = comm.handle if comm is not None else None
        self.handle = C.c_void_p()
        if all(v is None for v in (solver, deterministic, pcg_eta, pcg_max_it, pcg_eta_loose, pcg_eta_switch, pcg_abs_tol)):
            _check(lib().aar_problem_create(C.byref(d), C.byref(self.handle)))
        else:
            so = CSolverOptions()
            lib().aar_solver_default_options(C.byref(so))
            if solver is not None:
                so.solver = SOLVERS[solver] if isinstance(solver, str) else int(solver)
            if deterministic is not None:
                so.deterministic = int(bool(deterministic))
            if pcg_eta is not None:
                so.pcg_eta = float(pcg_eta)
            if pcg_max_it is not None:
                so.pcg_max_it = int(pcg_max_it)
            if pcg_eta_loose is not None:
                so.pcg_eta_loose = float(pcg_eta_loose)
            if pcg_eta_switch is not None:
                so.pcg_eta_switch = float(pcg_eta_switch)
            if pcg_abs_tol is not None:
                so.pcg_abs_tol = float(pcg_abs_tol)
            _check(lib().aar_problem_create_ex(C.byref(d), C.byref(so), C.byref(self.handle)))
        self.full_len = lib().aar_problem_full_len(self.handle)
        self.num_vars = lib().aar_problem_num_vars(self.handle)
        self.local_obs = lib().aar_problem_local_obs(self.handle)

    def close(self):
        if self.handle:
            lib().aar_problem_destroy(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _x(self, x_full):
        x = np.ascontiguousarray(x_full, dtype=np.float64)
        assert x.shape == (self.full_len,)
        return x

    def x_with_intrinsics(self, x_pose):
        """pose vector + the data set's (fx cx fy cy d0..d4) per camera: the `.solution` vector (libs/multicam_mapper.cpp:1085-1089)"""
        K = np.asarray(self.ds.cam_mats, dtype=np.float64).reshape(-1, 9)
        d = np.asarray(self.ds.dist_coeffs, dtype=np.float64).reshape(-1, 5)
        intr = np.concatenate([np.stack([K[:, 0], K[:, 2], K[:, 4], K[:, 5]], axis=1), d], axis=1).reshape(-1)
        return np.concatenate([np.asarray(x_pose, dtype=np.float64), intr])

    def eval_residuals(self, x_full, want_vector=True):
        x = self._x(x_full)
        ss = C.c_double()
        r = np.zeros(8 * self.ds.num_obs) if want_vector else None
        _check(lib().aar_eval_residuals(self.handle, _dptr(x), _dptr(r) if want_vector else None, C.byref(ss)))
        return r, ss.value

    def eval_normal_equations(self, x_full):
        x = self._x(x_full)
        P = self.num_vars
        H = np.zeros((P, P))
        B = np.zeros(P)
        ss = C.c_double()
        _check(lib().aar_eval_normal_equations(self.handle, _dptr(x), _dptr(H), _dptr(B), C.byref(ss)))
        return H, B, ss.value

    def eval_damped_step(self, x_full, mu):
        x = self._x(x_full)
        delta = np.zeros(self.num_vars)
        _check(lib().aar_eval_damped_step(self.handle, _dptr(x), mu, _dptr(delta)))
        return delta

    def reproj_stats(self, x_full):
        x = self._x(x_full)
        rmse, ss = C.c_double(), C.c_double()
        _check(lib().aar_reproj_stats(self.handle, _dptr(x), C.byref(rmse), C.byref(ss)))
        return rmse.value, ss.value

    def lm_init(self, x_full, params=None):
        x = self._x(x_full)
        _check(lib().aar_lm_init(self.handle, _dptr(x), C.byref(params) if params is not None else None))

    def lm_step(self):
        it = CLmIter()
        _check(lib().aar_lm_step(self.handle, C.byref(it)))
        return dict(err=it.err, mu=it.mu, gain=it.gain, delta_norm=it.delta_norm, accepted=it.accepted, tries=it.tries)

    def lm_get_solution(self):
        x = np.array(self.ds.x_full, dtype=np.float64)
        err = C.c_double()
        _check(lib().aar_lm_get_solution(self.handle, _dptr(x), C.byref(err)))
        return x, err.value

    def lm_solve(self, x_full, params=None, trace_cap=256):
        x = np.array(self._x(x_full), dtype=np.float64)
        rep = CLmReport()
        tr = (CLmIter * trace_cap)()
        rep.trace = tr
        rep.trace_cap = trace_cap
        _check(lib().aar_lm_solve(self.handle, _dptr(x), C.byref(params) if params is not None else None, C.byref(rep)))
        n = min(rep.iterations, trace_cap)
        trace = [dict(err=tr[i].err, mu=tr[i].mu, gain=tr[i].gain, delta_norm=tr[i].delta_norm,
                      accepted=tr[i].accepted, tries=tr[i].tries) for i in range(n)]
        report = dict(iterations=rep.iterations, stop_code=rep.stop_code, initial_err=rep.initial_err,
                      final_err=rep.final_err, final_mu=rep.final_mu, solve_seconds=rep.solve_seconds,
                      trial_points=rep.trial_points, trace=trace)
        return x, report

    def track(self, x_full, params=None):
        """MultiCamMapper::track() for every frame at once: returns (x_full with refined frame poses, iterations[F], err[F])."""
        x = np.array(self._x(x_full), dtype=np.float64)
        it = np.zeros(self.ds.num_frames, dtype=np.int32)
        err = np.zeros(self.ds.num_frames)
        _check(lib().aar_track(self.handle, _dptr(x), C.byref(params) if params is not None else None,
                               it.ctypes.data_as(C.POINTER(C.c_int32)), _dptr(err)))
        return x, it, err

    def set_step_callback(self, fn, want_z=True):
        """SparseLevMarq::setStepCallBackFunc: fn(z) after every step (z = numpy copy of curr_z, or None when want_z is False)."""
        if fn is None:
            self._step_cb = None
            _check(lib().aar_lm_set_step_callback(self.handle, C.cast(None, STEP_CB), None, 0))
            return
        def tramp(ctx, zp, n):
            fn(np.ctypeslib.as_array(zp, shape=(n,)).copy() if zp else None)
        self._step_cb = STEP_CB(tramp)          # keep the trampoline alive
        _check(lib().aar_lm_set_step_callback(self.handle, self._step_cb, None, int(want_z)))

    def set_stop_function(self, fn):
        """SparseLevMarq::setStopFunction: fn(z) -> True stops the loop (no iteration cap while it is set)."""
        if fn is None:
            self._stop_fn = None
            _check(lib().aar_lm_set_stop_function(self.handle, C.cast(None, STOP_FN), None))
            return
        def tramp(ctx, zp, n):
            return 1 if fn(np.ctypeslib.as_array(zp, shape=(n,)).copy()) else 0
        self._stop_fn = STOP_FN(tramp)
        _check(lib().aar_lm_set_stop_function(self.handle, self._stop_fn, None))

    def extract_z(self, x_full):
        x = self._x(x_full)
        z = np.zeros(self.num_vars)
        _check(lib().aar_problem_extract_z(self.handle, _dptr(x), _dptr(z)))
        return z

    def set_huber_delta(self, delta):
        _check(lib().aar_problem_set_huber_delta(self.handle, float(delta)))

    def get_huber_delta(self):
        return float(lib().aar_problem_get_huber_delta(self.handle))

    def set_kernel_profiling(self, on):
        _check(lib().aar_set_kernel_profiling(self.handle, int(on)))

    def kernel_times(self):
        """{kernel name: (total seconds, launches)} accumulated since profiling was switched on."""
        sec = np.zeros(NUM_KERNELS)
        cnt = np.zeros(NUM_KERNELS, dtype=np.int64)
        _check(lib().aar_get_kernel_times(self.handle, _dptr(sec), cnt.ctypes.data_as(C.POINTER(C.c_int64))))
        return {lib().aar_kernel_name(i).decode(): (float(sec[i]), int(cnt[i])) for i in range(NUM_KERNELS)}

    def pcg_iterations(self):
        """(CG iterations of the last damped solve, running total) of the pcg / spcg solvers; zeros for direct"""
        out = (C.c_int32 * 2)()
        _check(lib().aar_problem_pcg_iterations(self.handle, out))
        return int(out[0]), int(out[1])

    def solver_stats(self):
        """aar_problem_get_solver_stats: the solver the problem runs with (AUTO resolved) and what its inner CG has done so far"""
        st = CSolverStats()
        st.struct_size = C.sizeof(CSolverStats)
        _check(lib().aar_problem_get_solver_stats(self.handle, C.byref(st)))
        return dict(solver=SOLVER_NAMES[st.solver], deterministic=bool(st.deterministic), last_iterations=st.last_iterations,
                    total_iterations=st.total_iterations, solves=st.solves, fallbacks=st.fallbacks, pcg_eta=st.pcg_eta, pcg_max_it=st.pcg_max_it,
                    same_xcd_solves=st.same_xcd_solves, pcg_eta_loose=st.pcg_eta_loose, pcg_eta_switch=st.pcg_eta_switch, pcg_abs_tol=st.pcg_abs_tol, env_overrides=st.env_overrides)

    def set_test_hook(self, hook, value):
        """aar_problem_set_test_hook (testing only): fault injection for the solvers' fall-back paths"""
        _check(lib().aar_problem_set_test_hook(self.handle, int(hook), int(value)))

    def set_stage_timers(self, on):
        _check(lib().aar_set_stage_timers(self.handle, int(on)))

    def stage_times(self):
        t = CStageTimes()
        _check(lib().aar_get_stage_times(self.handle, C.byref(t)))
        return {n: getattr(t, n) for n, _ in CStageTimes._fields_}
