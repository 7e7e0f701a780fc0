"""The OpenCV arithmetic the oracle restates, pinned against an INDEPENDENT implementation.

libs/multicam_mapper.cpp leans on cv::Rodrigues (:470,478,910-911), cv::Mat::inv (:619), cv::Mat products (:619-640) and
cv::undistortPoints (:570).  OpenCV is not in the build container and the reference holds no vectors for these calls, so
tests/golden/make_primitives.py computes them with scipy.spatial.transform.Rotation, numpy.linalg.inv, a vectorised numpy
projection and the published distortion model (iterated as OpenCV 3.2 does, and inverted exactly by Newton), none of which
shares code with oracle/ or the product.  Here: the oracle (CPU) and the HIP kernels (GPU, through the C ABI) against that
fixture.  Tolerances are stated per branch of the algorithm; float outputs are compared bit for bit.
"""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import aar
import oracle_lib as ol
from conftest import load_golden

G = None


def fixture():
    global G
    if G is None:
        G = load_golden("g0_primitives")
    return G


def _second_dataset(g):
    ds = aar.Dataset()
    ds.num_cams, ds.num_markers, ds.num_frames, ds.root_cam, ds.root_marker = [int(v) for v in g["b_meta"]]
    ds.marker_size = float(g["marker_size"][0])
    ds.cam_ids = np.arange(ds.num_cams, dtype=np.int32)
    ds.marker_ids = np.arange(ds.num_markers, dtype=np.int32)
    ds.frame_ids = g["b_frame_ids"]
    ds.image_sizes = np.tile(np.array([1280, 720], dtype=np.int32), (ds.num_cams, 1))
    ds.cam_mats = g["b_cam_mats"]
    ds.dist_coeffs = np.zeros((ds.num_cams, 5))
    ds.obs_frame, ds.obs_cam, ds.obs_marker, ds.obs_uv = g["b_obs_frame"], g["b_obs_cam"], g["b_obs_marker"], g["b_obs_uv"]
    ds.x_full, ds.x_truth = g["b_x_full"], None
    ds.num_obs = len(ds.obs_frame)
    ds.optimize_cam_poses = ds.optimize_marker_poses = ds.optimize_object_poses = True
    ds.optimize_cam_intrinsics = False
    return ds


# ---------------------------------------------------------------------------------------------------------------------
# CPU: oracle restatement and the product's host-side SE(3) helpers against scipy / numpy
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("impl", ["oracle", "product_host"])
def test_rodrigues_vec2mat_equals_scipy(impl):
    _, g = fixture()
    f = ol.rodrigues_vec2mat if impl == "oracle" else aar.rodrigues_vec2mat     # oracle/ba_oracle.cpp vs automatic-ar_amd/host/se3.h
    R = np.array([f(w).reshape(9) for w in g["rv"]])
    # Rodrigues' formula against scipy's quaternion route: a few ulp of entries <= 1, every class of angle
    # (0, denormal, < DBL_EPSILON -> identity, tiny, generic, pi - 1e-10 .. pi)
    assert np.abs(R - g["rv_R"]).max() < 2e-15
    th = np.linalg.norm(g["rv"], axis=1)
    ident = np.array([f(w).reshape(9) for w in g["rv"][th < 2.2e-16]])
    assert np.array_equal(ident, np.tile(np.eye(3).reshape(9), (len(ident), 1)))     # theta < DBL_EPSILON: exactly I (cv::Rodrigues)


@pytest.mark.parametrize("impl", ["oracle", "product_host"])
def test_rodrigues_mat2vec_equals_scipy(impl):
    _, g = fixture()
    f = ol.rodrigues_mat2vec if impl == "oracle" else aar.rodrigues_mat2vec
    rv, cls = g["rv"], g["rv_class"]
    w = np.array([f(r.reshape(3, 3)) for r in g["rv_R"]])
    th = np.linalg.norm(rv, axis=1)
    s = np.abs(np.sin(th))
    dw = np.abs(w - g["m2v_w"]).max(axis=1)
    dR = np.abs(Rotation.from_rotvec(w).as_matrix().reshape(-1, 9) - g["rv_R"]).max(axis=1)
    assert np.all(np.linalg.norm(w, axis=1) <= np.pi + 1e-15)
    # generic angles: theta = acos((tr R - 1) / 2), omega = rho theta / (2 s): relative accuracy ~ eps / s
    gen = s >= 1.001e-5                  # (vectors within 0.1 % of the branch threshold s = 1e-5 may fall on either side)
    assert np.all(dw[gen & (cls == 0)] < 1e-13)
    assert np.all(dR[gen] < 1e-15 / s[gen] + 1e-13)
    # s < 1e-5, theta near 0: cv::Rodrigues returns the zero vector (the documented branch), i.e. it is off by theta itself
    small = (s < 0.999e-5) & (th < 1)
    assert np.all(w[small] == 0) and small.sum() >= 60
    # s < 1e-5, theta near pi: axis from the diagonal with OpenCV's sign rules; the sign of the axis is not recoverable from
    # the diagonal, so the result is a rotation by pi - d or pi + d about the right axis (2 d), a zero axis component comes
    # out as d / 2 (sqrt of (R_ii + 1) / 2 = d^2 / 4: another pi d / 2), and theta = acos(c) near c = -1 is good to sqrt(eps)
    flip = (s < 0.999e-5) & (th > 1)
    assert flip.sum() >= 100
    assert np.all(dR[flip] <= 4.0 * np.abs(np.pi - th[flip]) + 3e-8)
    # float-rounded (slightly non-orthogonal) matrices: the rotation vector of the nearest orthogonal matrix U V^T
    wf = np.array([f(r.reshape(3, 3)) for r in g["m2v_Rf"]])
    assert np.abs(wf - g["m2v_wf"]).max() < 1e-13


def test_inverse_4x4_equals_numpy():
    _, g = fixture()
    iv = np.array([ol.inv4(a.reshape(4, 4)).reshape(16) for a in g["inv_A"]])
    rel = np.abs(iv - g["inv_Ainv"]).max(axis=1) / np.abs(g["inv_Ainv"]).max(axis=1)
    assert rel[:800].max() < 2e-15           # rigid transforms, exact and float-rounded
    assert rel[800:].max() < 1e-14           # general matrices (condition numbers up to ~50)


def test_projection_rows_equal_numpy():
    ds, g = fixture()
    o = ol.Oracle(ds)
    assert np.abs(o.residuals(ds.x_full, res_mode=ol.RES_F64) - g["proj_r64"]).max() < 5e-12       # px (values up to 1280)
    assert np.array_equal(o.residuals(ds.x_full, res_mode=ol.RES_F32), g["proj_r32"])               # cv::Point2f store: bit for bit
    ds2 = _second_dataset(g)                                                                         # skewed K
    o2 = ol.Oracle(ds2)
    assert np.abs(o2.residuals(ds2.x_full, res_mode=ol.RES_F64) - g["b_proj_r64"]).max() < 5e-12
    r32 = o2.residuals(ds2.x_full, res_mode=ol.RES_F32)
    assert np.mean(r32 == g["b_proj_r32"]) > 0.999 and np.abs(r32 - g["b_proj_r32"]).max() < 2e-4    # <= 1 float ulp at 1280 px


def test_undistort_points_equals_numpy():
    _, g = fixture()
    K, uv = g["und_K"], g["und_uv"]
    for i in range(4):
        d = g["und_dist%d" % i]
        u = ol.undistort_points(K, d, uv)
        assert np.array_equal(u, g["und_fixed5_%d" % i]), i                 # OpenCV 3.2's five iterations, float out: bit for bit
        # ... and those five iterations sit on the exact inverse of the forward model up to their own convergence error
        err = np.abs(u.astype(np.float64) - g["und_newton_%d" % i])
        assert np.median(err) < 5e-5 and err.max() < (0.05 if i == 1 else 2e-4), i


# ---------------------------------------------------------------------------------------------------------------------
# GPU: the HIP kernels through the C ABI against the same independent vectors
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_residual_rows_equal_numpy():
    if aar.device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the GPU box (the product has no CPU path)")
    ds, g = fixture()
    with aar.Problem(ds, residual_mode=aar.RES_F32) as p:
        r, ss = p.eval_residuals(ds.x_full)
        assert np.array_equal(r, g["proj_r32"])                              # k_residual: float-faithful rows, bit for bit
        np.testing.assert_allclose(ss, float((g["proj_r32"] ** 2).sum()), rtol=1e-13)
    with aar.Problem(ds, residual_mode=aar.RES_F64) as p:
        r, _ = p.eval_residuals(ds.x_full)
        assert np.abs(r - g["proj_r64"]).max() < 1e-9                        # rigid inverse instead of LU: another rounding order
    ds2 = _second_dataset(g)
    with aar.Problem(ds2, residual_mode=aar.RES_F64) as p:
        r, _ = p.eval_residuals(ds2.x_full)
        assert np.abs(r - g["b_proj_r64"]).max() < 1e-9
    with aar.Problem(ds2, residual_mode=aar.RES_F32) as p:
        r, _ = p.eval_residuals(ds2.x_full)
        assert np.mean(r == g["b_proj_r32"]) > 0.999 and np.abs(r - g["b_proj_r32"]).max() < 2e-4


@pytest.mark.gpu
def test_hip_undistort_equals_numpy():
    if aar.device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the GPU box (the product has no CPU path)")
    _, g = fixture()
    for i in range(4):
        u = aar.undistort_points(g["und_K"], g["und_dist%d" % i], g["und_uv"])
        assert np.array_equal(u, g["und_fixed5_%d" % i]), i
