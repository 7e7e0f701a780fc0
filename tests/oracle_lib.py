"""ctypes binding of the CPU checker (oracle/liboracle.so, oracle/_ref/libref_lm.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (automatic-ar_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_lm.so")

RES_F32, RES_F64 = 0, 1
JAC_NUMERIC_F32, JAC_NUMERIC_F64, JAC_ANALYTIC, JAC_TRACK = 0, 1, 2, 3


class OrcProblem(C.Structure):
    _fields_ = [
        ("num_cams", C.c_int32), ("num_markers", C.c_int32), ("num_frames", C.c_int32),
        ("root_cam", C.c_int32), ("root_marker", C.c_int32),
        ("K", C.POINTER(C.c_double)), ("marker_size", C.c_double), ("num_obs", C.c_int64),
        ("obs_frame", C.POINTER(C.c_int32)), ("obs_cam", C.POINTER(C.c_int32)), ("obs_marker", C.POINTER(C.c_int32)),
        ("obs_uv", C.POINTER(C.c_float)),
        ("opt_cams", C.c_int32), ("opt_markers", C.c_int32), ("opt_frames", C.c_int32),
        ("with_huber", C.c_int32), ("huber_delta", C.c_float),
        ("opt_intrinsics", C.c_int32), ("dist", C.POINTER(C.c_double)),
    ]


class OrcLmParams(C.Structure):
    _fields_ = [("max_iters", C.c_int32), ("min_error", C.c_double), ("min_step_error_diff", C.c_double),
                ("min_average_step_error_diff", C.c_double), ("tau", C.c_double), ("huber_fixed", C.c_int32)]


class OrcLmIter(C.Structure):
    _fields_ = [("err", C.c_double), ("mu", C.c_double), ("gain", C.c_double), ("delta_norm", C.c_double),
                ("accepted", C.c_int32), ("tries", C.c_int32)]


class OrcCamModel(C.Structure):
    _fields_ = [("K", C.c_double * 9), ("dist", C.c_double * 12), ("n_dist", C.c_int32)]


def cam_models(Ks, dists):
    """array of orc_cam_model from per-camera K (3x3) and distortion vectors"""
    arr = (OrcCamModel * len(Ks))()
    for i, (K, d) in enumerate(zip(Ks, dists)):
        K = np.asarray(K, dtype=np.float64).reshape(9)
        d = np.asarray(d, dtype=np.float64).reshape(-1)
        for j in range(9):
            arr[i].K[j] = K[j]
        for j in range(12):
            arr[i].dist[j] = d[j] if j < len(d) else 0.0
        arr[i].n_dist = len(d)
    return arr


def build_oracle(with_ref=True):
    """Compile the checker (gcc, seconds).  The _ref target needs /root/reference and is skipped without it."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])
    if with_ref and os.path.exists("/root/reference/libs/sparselevmarq.h"):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_orc = None
_ref = None


def oracle():
    global _orc
    if _orc is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle(with_ref=False)
        L = C.CDLL(ORACLE_SO)
        pp = C.POINTER(OrcProblem)
        L.orc_full_len.restype = C.c_int64
        L.orc_full_len.argtypes = [pp]
        L.orc_num_vars.restype = C.c_int64
        L.orc_num_vars.argtypes = [pp]
        L.orc_extract_z.argtypes = [pp, _dp, _dp]
        L.orc_get_intrinsics.argtypes = [pp, _dp, _dp, _dp]
        L.orc_jac_capacity.restype = C.c_int64
        L.orc_jac_capacity.argtypes = [pp]
        L.orc_merge_z.argtypes = [pp, _dp, _dp, _dp]
        L.orc_rodrigues_vec2mat.argtypes = [_dp, _dp]
        L.orc_rodrigues_mat2vec.argtypes = [_dp, _dp]
        L.orc_residuals.argtypes = [pp, _dp, _dp, C.c_int, _dp]
        L.orc_jacobian.restype = C.c_int64
        L.orc_jacobian.argtypes = [pp, _dp, _dp, C.c_int, _ip, _ip, _dp]
        L.orc_normal_equations_dense.argtypes = [pp, _dp, _dp, C.c_int, C.c_int, _dp, _dp]
        L.orc_damped_solve.argtypes = [pp, _dp, _dp, C.c_int, C.c_int, C.c_double, _dp]
        L.orc_lm_solve.restype = C.c_double
        L.orc_lm_solve.argtypes = [pp, _dp, _dp, C.POINTER(OrcLmParams), C.c_int, C.c_int, C.POINTER(OrcLmIter),
                                   C.c_int32, C.POINTER(C.c_int32), C.c_int32]
        L.orc_reproj_stats.argtypes = [pp, _dp, _dp, _dp, _dp, _dp]
        _fp = C.POINTER(C.c_float)
        L.orc_undistort_points.argtypes = [_dp, _dp, C.c_int, C.c_int64, _fp, _fp]
        L.orc_distort_points.argtypes = [_dp, _dp, C.c_int, C.c_int64, _dp, _dp]
        cm = C.POINTER(OrcCamModel)
        L.orc_undistort_normalized.argtypes = [cm, C.c_int64, _fp, _fp]
        L.orc_ippe_square.argtypes = [C.c_float, cm, C.c_int64, _fp, _dp, _dp, _dp, _dp]
        L.orc_vote.restype = C.c_int64
        L.orc_vote.argtypes = [C.c_double, C.c_int64, _dp, _dp, _dp, _dp, _dp]
        L.orc_inv4.argtypes = [_dp, _dp]
        L.orc_init_run.restype = C.c_void_p
        L.orc_init_run.argtypes = [C.c_int32, C.c_int32, C.c_int64, _ip, _ip, _ip, _fp, C.c_double, cm, _ip, C.c_int32,
                                   C.c_double, C.c_int32]
        L.orc_init_object_poses.restype = C.c_void_p
        L.orc_init_object_poses.argtypes = [C.c_int32, C.c_int32, C.c_int64, _ip, _ip, _ip, _fp, C.c_double, cm, C.c_int32, _ip,
                                            _dp, C.c_int32, _ip, _dp, C.c_double, C.c_int32]
        L.orc_init_counts.argtypes = [C.c_void_p, _ip]
        L.orc_init_get.argtypes = [C.c_void_p, _ip, _dp, _ip, _dp, _ip, _dp, _ip]
        L.orc_init_free.argtypes = [C.c_void_p]
        _orc = L
    return _orc


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    """The REAL reference solver (libs/sparselevmarq.h + Eigen) compiled into oracle/_ref/."""
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        pp = C.POINTER(OrcProblem)
        L.ref_lm_solve.restype = C.c_double
        L.ref_lm_solve.argtypes = [pp, _dp, _dp, C.POINTER(OrcLmParams), C.c_int, C.c_int, C.POINTER(OrcLmIter),
                                   C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32]
        L.ref_track_solve.restype = C.c_double
        L.ref_track_solve.argtypes = [pp, _dp, _dp, C.POINTER(OrcLmParams), C.POINTER(C.c_int32), C.c_int32]
        L.ref_damped_solve.argtypes = [C.c_int64, C.c_int64, C.c_int64, _ip, _ip, _dp, _dp, C.c_double, _dp, _dp, _dp]
        _ref = L
    return _ref


def _d(a):
    return a.ctypes.data_as(_dp)


def mapper_params(**over):
    """LM parameters as MultiCamMapper::init installs them (libs/multicam_mapper.cpp:326-330)."""
    p = OrcLmParams(10000, 1e-5, 0.0, 1e-4, 1.0, 0)
    for k, v in over.items():
        setattr(p, k, v)
    return p


class Oracle:
    """One problem bound to the oracle.  `ds` is an aar.Dataset-like object (numpy fields)."""

    def __init__(self, ds, optimize=(True, True, True), with_huber=False, huber_delta=10.0, intrinsics=False):
        self.ds = ds
        self.dist = np.ascontiguousarray(getattr(ds, "dist_coeffs", np.zeros((ds.num_cams, 5))), dtype=np.float64).reshape(-1)
        self.K = np.ascontiguousarray(ds.cam_mats, dtype=np.float64).reshape(-1)
        self.of = np.ascontiguousarray(ds.obs_frame, dtype=np.int32)
        self.oc = np.ascontiguousarray(ds.obs_cam, dtype=np.int32)
        self.om = np.ascontiguousarray(ds.obs_marker, dtype=np.int32)
        self.uv = np.ascontiguousarray(ds.obs_uv, dtype=np.float32).reshape(-1)
        p = OrcProblem()
        p.num_cams, p.num_markers, p.num_frames = ds.num_cams, ds.num_markers, ds.num_frames
        p.root_cam, p.root_marker = ds.root_cam, ds.root_marker
        p.K = _d(self.K)
        p.marker_size = ds.marker_size
        p.num_obs = len(self.of)
        p.obs_frame = self.of.ctypes.data_as(_ip)
        p.obs_cam = self.oc.ctypes.data_as(_ip)
        p.obs_marker = self.om.ctypes.data_as(_ip)
        p.obs_uv = self.uv.ctypes.data_as(C.POINTER(C.c_float))
        p.opt_cams, p.opt_markers, p.opt_frames = [int(b) for b in optimize]
        p.with_huber = int(with_huber)
        p.huber_delta = huber_delta
        p.opt_intrinsics = int(intrinsics)
        p.dist = _d(self.dist)
        self.p = p
        self.N = int(p.num_obs)
        self.full_len = oracle().orc_full_len(C.byref(p))
        self.num_vars = oracle().orc_num_vars(C.byref(p))

    def extract_z(self, x_full):
        x = np.ascontiguousarray(x_full, dtype=np.float64)
        z = np.zeros(self.num_vars)
        oracle().orc_extract_z(C.byref(self.p), _d(x), _d(z))
        return z

    def merge_z(self, x_full, z):
        x = np.ascontiguousarray(x_full, dtype=np.float64)
        z = np.ascontiguousarray(z, dtype=np.float64)
        out = np.zeros(self.full_len)
        oracle().orc_merge_z(C.byref(self.p), _d(x), _d(z), _d(out))
        return out

    def intrinsics(self, z):
        """(K [C,3,3], dist [C,5]) as intrinsics_vec2mats rebuilds them from z (the data set's when z holds none)"""
        z = np.ascontiguousarray(z, dtype=np.float64)
        K = np.zeros(9 * self.ds.num_cams); d = np.zeros(5 * self.ds.num_cams)
        oracle().orc_get_intrinsics(C.byref(self.p), _d(z), _d(K), _d(d))
        return K.reshape(-1, 3, 3), d.reshape(-1, 5)

    def residuals(self, x_full, z=None, res_mode=RES_F32):
        x = np.ascontiguousarray(x_full, dtype=np.float64)
        z = self.extract_z(x) if z is None else np.ascontiguousarray(z, dtype=np.float64)
        r = np.zeros(8 * self.N)
        oracle().orc_residuals(C.byref(self.p), _d(x), _d(z), res_mode, _d(r))
        return r

    def jacobian(self, x_full, z=None, jac_mode=JAC_ANALYTIC):
        x = np.ascontiguousarray(x_full, dtype=np.float64)
        z = self.extract_z(x) if z is None else np.ascontiguousarray(z, dtype=np.float64)
        cap = oracle().orc_jac_capacity(C.byref(self.p))
        rows = np.zeros(cap, dtype=np.int32)
        cols = np.zeros(cap, dtype=np.int32)
        vals = np.zeros(cap)
        n = oracle().orc_jacobian(C.byref(self.p), _d(x), _d(z), jac_mode, rows.ctypes.data_as(_ip),
                                  cols.ctypes.data_as(_ip), _d(vals))
        return rows[:n].copy(), cols[:n].copy(), vals[:n].copy()

    def normal_equations(self, x_full, z=None, jac_mode=JAC_ANALYTIC, res_mode=RES_F64):
        x = np.ascontiguousarray(x_full, dtype=np.float64)
        z = self.extract_z(x) if z is None else np.ascontiguousarray(z, dtype=np.float64)
        P = self.num_vars
        H = np.zeros((P, P))
        B = np.zeros(P)
        oracle().orc_normal_equations_dense(C.byref(self.p), _d(x), _d(z), jac_mode, res_mode, _d(H), _d(B))
        return H, B

    def damped_solve(self, x_full, mu, z=None, jac_mode=JAC_ANALYTIC, res_mode=RES_F64):
        x = np.ascontiguousarray(x_full, dtype=np.float64)
        z = self.extract_z(x) if z is None else np.ascontiguousarray(z, dtype=np.float64)
        d = np.zeros(self.num_vars)
        rc = oracle().orc_damped_solve(C.byref(self.p), _d(x), _d(z), jac_mode, res_mode, mu, _d(d))
        assert rc == 0
        return d

    def _lm(self, fn, x_full, params, jac_mode, res_mode, threads, extra=()):
        x = np.ascontiguousarray(x_full, dtype=np.float64)
        z = self.extract_z(x)
        prm = params if params is not None else mapper_params()
        cap = 512
        tr = (OrcLmIter * cap)()
        n = C.c_int32()
        err = fn(C.byref(self.p), _d(x), _d(z), C.byref(prm), jac_mode, res_mode, tr, cap, C.byref(n), threads, *extra)
        trace = [dict(err=tr[i].err, mu=tr[i].mu, gain=tr[i].gain, delta_norm=tr[i].delta_norm,
                      accepted=tr[i].accepted, tries=tr[i].tries) for i in range(min(n.value, cap))]
        return self.merge_z(x, z), dict(final_err=err, iterations=n.value, trace=trace, z=z)

    def lm_solve(self, x_full, params=None, jac_mode=JAC_NUMERIC_F32, res_mode=RES_F32, threads=0):
        """The restated LM loop + own sparse LDL^T (the "port")."""
        return self._lm(oracle().orc_lm_solve, x_full, params, jac_mode, res_mode, threads)

    def ref_lm_solve(self, x_full, params=None, jac_mode=JAC_NUMERIC_F32, res_mode=RES_F32, threads=0, use_omp_mult=True):
        """The real ucoslam::SparseLevMarq<double>::solve driving the restated callbacks."""
        return self._lm(ref().ref_lm_solve, x_full, params, jac_mode, res_mode, threads, (int(use_omp_mult),))

    def reproj_stats(self, x_full):
        x = np.ascontiguousarray(x_full, dtype=np.float64)
        z = self.extract_z(x)
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        oracle().orc_reproj_stats(C.byref(self.p), _d(x), _d(z), C.byref(a), C.byref(b), C.byref(c))
        return dict(rmse=a.value, mean_dist=b.value, sum_sq=c.value)


def frame_subproblem(ds, f):
    """The single-frame data set MultiCamMapper::init(object_poses, fcm) leaves for track() (libs/multicam_mapper.cpp:272-279)."""
    import copy
    sub = copy.copy(ds)
    keep = np.asarray(ds.obs_frame) == f
    sub.obs_frame = np.zeros(int(keep.sum()), dtype=np.int32)
    sub.obs_cam = np.asarray(ds.obs_cam)[keep]
    sub.obs_marker = np.asarray(ds.obs_marker)[keep]
    sub.obs_uv = np.asarray(ds.obs_uv)[keep]
    sub.num_obs = int(keep.sum())
    sub.num_frames = 1
    sub.frame_ids = np.asarray(ds.frame_ids)[f:f + 1]
    ns = 6 * (ds.num_cams - 1) + 6 * (ds.num_markers - 1)
    sub.x_full = np.concatenate([np.asarray(ds.x_full)[:ns], np.asarray(ds.x_full)[ns + 6 * f: ns + 6 * f + 6]])
    sub.x_truth = None
    return sub


def track_frames(ds, x_full, with_huber=False, huber_delta=10.0, use_ref=False, threads=1, params=None):
    """track() frame by frame on the CPU: the restated LM with the calcDerivates Jacobian (port) or, with use_ref, the real
    solver's own solve(z, f).  Returns (x_full with refined frame poses, iterations[F], err[F])."""
    x = np.array(x_full, dtype=np.float64)
    ns = 6 * (ds.num_cams - 1) + 6 * (ds.num_markers - 1)
    its = np.zeros(ds.num_frames, dtype=np.int32)
    errs = np.zeros(ds.num_frames)
    prm = params if params is not None else mapper_params(huber_fixed=1)
    dsx = type("D", (), {})()
    dsx.__dict__.update(ds.__dict__)
    dsx.x_full = x
    for f in range(ds.num_frames):
        sub = frame_subproblem(dsx, f)
        o = Oracle(sub, optimize=(False, False, True), with_huber=with_huber, huber_delta=huber_delta)
        if use_ref:
            z = np.ascontiguousarray(sub.x_full[ns:ns + 6])
            xs = np.ascontiguousarray(sub.x_full)
            n = C.c_int32()
            e = ref().ref_track_solve(C.byref(o.p), _d(xs), _d(z), C.byref(prm), C.byref(n), threads)
            x[ns + 6 * f: ns + 6 * f + 6] = z
            its[f], errs[f] = n.value, e
        else:
            xs, rep = o.lm_solve(sub.x_full, params=prm, jac_mode=JAC_TRACK, res_mode=RES_F64, threads=threads)
            x[ns + 6 * f: ns + 6 * f + 6] = xs[ns:ns + 6]
            its[f], errs[f] = rep["iterations"], rep["final_err"]
    return x, its, errs


def ref_damped_solve(n_rows, P, rows, cols, vals, r, mu, want_dense=True):
    """Eigen: JtJ = Jt*J, B = -Jt*r, delta = SimplicialLDLT(JtJ + mu I).solve(B)."""
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    r = np.ascontiguousarray(r, dtype=np.float64)
    H = np.zeros((P, P)) if want_dense else None
    B = np.zeros(P)
    d = np.zeros(P)
    rc = ref().ref_damped_solve(n_rows, P, len(vals), rows.ctypes.data_as(_ip), cols.ctypes.data_as(_ip), _d(vals),
                                _d(r), mu, _d(H) if want_dense else None, _d(B), _d(d))
    assert rc == 0
    return H, B, d


def rodrigues_vec2mat(w):
    w = np.ascontiguousarray(w, dtype=np.float64)
    R = np.zeros(9)
    oracle().orc_rodrigues_vec2mat(_d(w), _d(R))
    return R.reshape(3, 3)


def rodrigues_mat2vec(R):
    R = np.ascontiguousarray(R, dtype=np.float64).reshape(9)
    w = np.zeros(3)
    oracle().orc_rodrigues_mat2vec(_d(R), _d(w))
    return w


def undistort_points(K, dist, uv):
    """oracle restatement of cv::undistortPoints(.., K, dist, noArray(), P = K) (float in / out)"""
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    dist = np.ascontiguousarray(dist, dtype=np.float64).reshape(-1)
    uv = np.ascontiguousarray(uv, dtype=np.float32)
    out = np.empty_like(uv)
    fp = C.POINTER(C.c_float)
    oracle().orc_undistort_points(K.ctypes.data_as(_dp), dist.ctypes.data_as(_dp), len(dist), uv.size // 2,
                                  uv.ctypes.data_as(fp), out.ctypes.data_as(fp))
    return out


def distort_points(K, dist, uv):
    """forward distortion model, fp64 (ideal pixel -> distorted pixel)"""
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    dist = np.ascontiguousarray(dist, dtype=np.float64).reshape(-1)
    uv = np.ascontiguousarray(uv, dtype=np.float64)
    out = np.empty_like(uv)
    oracle().orc_distort_points(K.ctypes.data_as(_dp), dist.ctypes.data_as(_dp), len(dist), uv.size // 2,
                                uv.ctypes.data_as(_dp), out.ctypes.data_as(_dp))
    return out


# ---- Initializer / IPPE restatement (oracle/init_oracle.cpp; parity unpinned, see its header) ----
def ippe_square(marker_size, K, dist, uv):
    """aruco::solvePnP_ for n markers of one camera: (T1[n,4,4], e1[n], T2[n,4,4], e2[n])"""
    uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 8)
    n = len(uv)
    cams = cam_models([K], [dist])
    T1 = np.zeros((n, 16)); T2 = np.zeros((n, 16)); e1 = np.zeros(n); e2 = np.zeros(n)
    fp = C.POINTER(C.c_float)
    oracle().orc_ippe_square(float(marker_size), cams, n, uv.ctypes.data_as(fp), _d(T1), _d(e1), _d(T2), _d(e2))
    return T1.reshape(n, 4, 4), e1, T2.reshape(n, 4, 4), e2


def vote(marker_size, T, T1inv, T2inv):
    """Initializer::find_best_transformation on one set: (best index, weight, cost[n])"""
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(-1, 16)
    A = np.ascontiguousarray(T1inv, dtype=np.float64).reshape(-1, 16)
    B = np.ascontiguousarray(T2inv, dtype=np.float64).reshape(-1, 16)
    n = len(T)
    cost = np.zeros(max(n, 1)); w = C.c_double(0)
    best = oracle().orc_vote(float(marker_size), n, _d(T), _d(A), _d(B), _d(cost), C.byref(w))
    return int(best), w.value, cost[:n]


def inv4(A):
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(16)
    out = np.zeros(16)
    oracle().orc_inv4(_d(A), _d(out))
    return out.reshape(4, 4)


def init_run(num_cam_slots, num_frames, det_frame, det_cam, det_id, det_uv, marker_size, Ks, dists, excluded=(),
             threshold=2.0, min_detections=2, fixed=None):
    """Initializer(detections, marker_size, cam_configs, excluded_cams): dict of ids and 4x4 transforms.
    fixed = (cam_ids, T_cam[n,4,4], marker_ids, T_marker[n,4,4]): apps/track.cpp's use, the transforms are given and only
    obtain_pose_estimations + init_object_transforms run."""
    df = np.ascontiguousarray(det_frame, dtype=np.int32); dc = np.ascontiguousarray(det_cam, dtype=np.int32)
    di = np.ascontiguousarray(det_id, dtype=np.int32); uv = np.ascontiguousarray(det_uv, dtype=np.float32).reshape(-1, 8)
    ex = np.ascontiguousarray(list(excluded), dtype=np.int32)
    cams = cam_models(Ks, dists)
    L = oracle()
    ip = lambda a: a.ctypes.data_as(_ip)
    if fixed is None:
        h = L.orc_init_run(num_cam_slots, num_frames, len(df), ip(df), ip(dc), ip(di), uv.ctypes.data_as(C.POINTER(C.c_float)),
                           float(marker_size), cams, ip(ex), len(ex), float(threshold), int(min_detections))
    else:
        ci = np.ascontiguousarray(fixed[0], dtype=np.int32); Tc = np.ascontiguousarray(fixed[1], dtype=np.float64).reshape(-1, 16)
        mi = np.ascontiguousarray(fixed[2], dtype=np.int32); Tm = np.ascontiguousarray(fixed[3], dtype=np.float64).reshape(-1, 16)
        h = L.orc_init_object_poses(num_cam_slots, num_frames, len(df), ip(df), ip(dc), ip(di),
                                    uv.ctypes.data_as(C.POINTER(C.c_float)), float(marker_size), cams, len(ci), ip(ci), _d(Tc),
                                    len(mi), ip(mi), _d(Tm), float(threshold), int(min_detections))
    try:
        cnt = np.zeros(6, dtype=np.int32)
        L.orc_init_counts(h, ip(cnt))
        c, m, f, kept = (int(v) for v in cnt[:4])
        cam_ids = np.zeros(max(c, 1), dtype=np.int32); T_cam = np.zeros((max(c, 1), 16))
        marker_ids = np.zeros(max(m, 1), dtype=np.int32); T_marker = np.zeros((max(m, 1), 16))
        frame_ids = np.zeros(max(f, 1), dtype=np.int32); T_object = np.zeros((max(f, 1), 16))
        kept_ids = np.zeros(max(kept, 1), dtype=np.int32)
        L.orc_init_get(h, ip(cam_ids), _d(T_cam), ip(marker_ids), _d(T_marker), ip(frame_ids), _d(T_object), ip(kept_ids))
    finally:
        L.orc_init_free(h)
    return dict(cam_ids=cam_ids[:c], T_cam=T_cam[:c].reshape(c, 4, 4), marker_ids=marker_ids[:m],
                T_marker=T_marker[:m].reshape(m, 4, 4), frame_ids=frame_ids[:f], T_object=T_object[:f].reshape(f, 4, 4),
                kept_frame_ids=kept_ids[:kept], root_cam=int(cnt[4]), root_marker=int(cnt[5]))
