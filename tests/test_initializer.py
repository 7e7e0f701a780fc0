"""Initializer (libs/initializer.cpp) and square-marker IPPE (3rdparty/aruco/aruco/ippe.cpp): SURVEY.md section 8f, next row 1.

The reference files need OpenCV and cannot be built here, and the reference holds no fixtures for them, so the oracle
(oracle/init_oracle.cpp) cannot be pinned against OpenCV / aruco itself.  It is checked first by properties and, for IPPE, against
the two equations that DEFINE the method, evaluated with independent numpy (CPU tests below); the GPU tests then compare the HIP
kernels and the host pipeline with it on the same inputs.
"""
import os
import struct

import numpy as np
import pytest

import aar
import oracle_lib as O
from test_host_logic import CALIB_XML, CALIB_YAML

MS = 0.05


def scene(noise=0.1, frames=40, cams=8, markers=40, seed=None):
    over = dict(num_cams=cams, num_markers=markers, num_frames=frames, noise_px=noise)
    if seed is not None:
        over["seed"] = seed
    return aar.synth(3, **over)


def detections_of(ds, dists=None):
    """raw detections of a synthetic data set (frame / camera slot / marker id / corners), optionally re-distorted"""
    uv = ds.obs_uv.astype(np.float64).reshape(-1, 8)
    if dists is not None:
        K = ds.cam_mats.reshape(-1, 3, 3)
        out = uv.copy()
        for c in range(ds.num_cams):
            sel = ds.obs_cam == c
            out[sel] = O.distort_points(K[c], dists[c], uv[sel].reshape(-1, 2)).reshape(-1, 8)
        uv = out
    return aar.Detections(int(ds.cam_ids.max()) + 1, int(ds.frame_ids.max()) + 1, ds.frame_ids[ds.obs_frame],
                          ds.cam_ids[ds.obs_cam], ds.marker_ids[ds.obs_marker], uv.astype(np.float32))


def rigid(v):
    T = np.eye(4)
    T[:3, :3] = O.rodrigues_vec2mat(v[:3])
    T[:3, 3] = v[3:]
    return T


def truth_transforms(ds):
    C, M, F = ds.num_cams, ds.num_markers, ds.num_frames
    x = ds.x_truth
    cam = [np.eye(4) if c == ds.root_cam else rigid(x[6 * (c - (c > ds.root_cam)):][:6]) for c in range(C)]
    mk = [np.eye(4) if m == ds.root_marker else rigid(x[6 * (C - 1) + 6 * (m - (m > ds.root_marker)):][:6]) for m in range(M)]
    fr = [rigid(x[6 * (C - 1) + 6 * (M - 1) + 6 * f:][:6]) for f in range(F)]
    return cam, mk, fr


def oracle_init(det, Ks, dists, **kw):
    return O.init_run(det.num_cams, det.num_frames, det.det_frame, det.det_cam, det.det_id, det.det_uv, MS, Ks, dists, **kw)


def poses_of(Ts):
    return np.array([np.concatenate([O.rodrigues_mat2vec(T[:3, :3]), T[:3, 3]]) for T in Ts]).reshape(-1)


# ---------------------------------------------------------------------------------------------------------------------
# CPU: the oracle's properties and the host readers
# ---------------------------------------------------------------------------------------------------------------------
def test_oracle_ippe_recovers_exact_planar_poses():
    ds = scene(noise=0.0, frames=12)
    cam, mk, fr = truth_transforms(ds)
    K = ds.cam_mats.reshape(-1, 3, 3)
    for c in (0, 3):
        sel = np.nonzero(ds.obs_cam == c)[0]
        T1, e1, T2, e2 = O.ippe_square(MS, K[c], np.zeros(5), ds.obs_uv[sel])
        assert np.all(e1 <= e2) and np.all(np.isfinite(T1)) and np.all(np.isfinite(T2))
        assert np.median(e1) < 1e-6          # float corners: the true pose reprojects to ~1e-7 normalised units
        for k, o in enumerate(sel):
            Tt = np.linalg.inv(cam[c]) @ fr[ds.obs_frame[o]] @ mk[ds.obs_marker[o]]
            assert np.abs(T1[k] - Tt).max() < 2e-3, (c, k)
            # both solutions are proper rigid transforms, rounded to float (getRTMatrix CV_32F)
            for T in (T1[k], T2[k]):
                R = T[:3, :3]
                assert np.abs(R @ R.T - np.eye(3)).max() < 1e-6 and abs(np.linalg.det(R) - 1) < 1e-6
                np.testing.assert_array_equal(T, T.astype(np.float32).astype(np.float64))
                np.testing.assert_array_equal(T[3], [0, 0, 0, 1])


def test_oracle_ippe_second_solution_is_the_reflected_ambiguity():
    # the two IPPE rotations share the image of the marker centre and differ by the flip about the viewing ray;
    # under noise both reproject well, which is what the err2/err1 < threshold test keeps
    ds = scene(noise=0.3, frames=10)
    K = ds.cam_mats.reshape(-1, 3, 3)
    T1, e1, T2, e2 = O.ippe_square(MS, K[0], np.zeros(5), ds.obs_uv[ds.obs_cam == 0])
    assert np.all(e1 <= e2)
    c1 = T1[:, :3, 3] / T1[:, 2:3, 3]
    c2 = T2[:, :3, 3] / T2[:, 2:3, 3]
    assert np.abs(c1 - c2).max() < 5e-3     # same marker centre direction
    assert np.mean(e2 / e1 < 2.0) > 0.2     # a fair share of ambiguous detections at 0.3 px on 25-pixel markers


def test_oracle_ippe_satisfies_the_defining_equations_of_ippe():
    # An INDEPENDENT pin of the IPPE restatement (OpenCV / aruco cannot be built here): IPPE (Collins & Bartoli) is defined by two
    # equations, both checked with plain numpy that shares nothing with oracle/init_oracle.cpp.
    #  (1) rotation: with H the homography model plane -> normalised image (numpy DLT by SVD; exact for four corners), v = H(0, 0)
    #      and J the 2x2 Jacobian of H at the origin, BOTH returned rotations satisfy  [I2 | -v] R[:, :2] = gamma J, gamma > 0 --
    #      the pose's first-order behaviour at the marker centre equals the homography's;
    #  (2) translation: given R, t is the linear least-squares solution of the projection equations of the four corners.
    ds = scene(noise=0.3, frames=12)
    K = ds.cam_mats.reshape(-1, 3, 3)
    h = MS / 2
    model = np.array([[-h, h], [h, h], [h, -h], [-h, -h]])       # corner order of aruco::Marker (marker.cpp:358-367)
    checked = 0
    for c in (0, 2, 5):
        sel = np.nonzero(ds.obs_cam == c)[0][:60]
        uv = ds.obs_uv[sel].astype(np.float64).reshape(-1, 4, 2)
        T1, e1, T2, e2 = O.ippe_square(MS, K[c], np.zeros(5), ds.obs_uv[sel])
        Kinv = np.linalg.inv(K[c])
        for k in range(len(sel)):
            q = (Kinv @ np.c_[uv[k], np.ones(4)].T).T
            q = q[:, :2] / q[:, 2:3]
            A = []
            for (x, y), (a, b) in zip(model, q):     # DLT rows of  q ~ H (x, y, 1)
                A.append([x, y, 1, 0, 0, 0, -a * x, -a * y, -a])
                A.append([0, 0, 0, x, y, 1, -b * x, -b * y, -b])
            H = np.linalg.svd(np.array(A))[2][-1].reshape(3, 3)
            H /= H[2, 2]
            v = H[:2, 2]
            J = H[:2, :2] - np.outer(v, H[2, :2])
            for T in (T1[k], T2[k]):
                R, t = T[:3, :3], T[:3, 3]
                M = np.c_[np.eye(2), -v] @ R[:, :2]
                G = M @ np.linalg.inv(J)                # = gamma I2
                gamma = 0.5 * np.trace(G)
                assert gamma > 0 and np.abs(G - gamma * np.eye(2)).max() / gamma < 2e-5, (c, k, G)   # float-rounded R: ~1e-7
                rows, rhs = [], []
                for (x, y), (a, b) in zip(model, q):   # (R X + t)_xy = q (R X + t)_z
                    RX = R @ np.array([x, y, 0.0])
                    rows += [[1, 0, -a], [0, 1, -b]]
                    rhs += [a * RX[2] - RX[0], b * RX[2] - RX[1]]
                t_ls = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)[0]
                assert np.abs(t - t_ls).max() / np.abs(t_ls).max() < 2e-5, (c, k, t, t_ls)
                checked += 1
    assert checked >= 200


def _random_rigid(rng, n, rot=0.3, trans=0.2):
    out = np.zeros((n, 4, 4))
    for i in range(n):
        out[i] = rigid(np.concatenate([rng.normal(0, rot, 3), rng.normal(0, trans, 3)]))
    return out


def _vote_set(rng, n, n_good, sigma=1e-3):
    """candidates (T, T1inv, T2inv) around one true transform X: T2inv * X * T1inv = I for consistent ones"""
    X = rigid(np.array([0.3, -0.2, 0.5, 0.1, 0.2, 1.0]))
    T = np.zeros((n, 4, 4)); A = np.zeros((n, 4, 4)); B = np.zeros((n, 4, 4))
    P1 = _random_rigid(rng, n, 0.5, 0.5)
    for i in range(n):
        Xi = X @ rigid(np.concatenate([rng.normal(0, sigma, 3), rng.normal(0, sigma, 3)]))
        if i >= n_good:
            Xi = X @ rigid(np.concatenate([rng.normal(0, 0.8, 3), rng.normal(0, 0.3, 3)]))
        P2 = Xi @ P1[i]
        T[i] = P2 @ O.inv4(P1[i]); A[i] = P1[i]; B[i] = O.inv4(P2)
    return T, A, B


def test_oracle_vote_prefers_the_consistent_candidate():
    rng = np.random.default_rng(3)
    T, A, B = _vote_set(rng, 30, 22)
    best, weight, cost = O.vote(MS, T, A, B)
    assert 0 <= best < 22 and weight == cost[best] == cost.min()
    assert cost[22:].min() > 2 * cost[:22].max()   # every sum carries the 8 outliers on its j-side
    # first minimum wins ties; an empty set gives -1
    T2 = np.concatenate([T[best:best + 1], T]); A2 = np.concatenate([A[best:best + 1], A]); B2 = np.concatenate([B[best:best + 1], B])
    assert O.vote(MS, T2, A2, B2)[0] == 0
    assert O.vote(MS, T[:0], A[:0], B[:0])[0] == -1


def test_oracle_vote_costs_equal_the_formula_in_numpy():
    # Initializer::find_best_transformation (libs/initializer.cpp:151-193) in four lines of numpy, nothing shared with the oracle:
    # cost_i = sum_j sum_corners || p - T2inv_j T_i T1inv_j p ||, the winner is the FIRST minimum
    rng = np.random.default_rng(11)
    T, A, B = _vote_set(rng, 24, 17, sigma=5e-3)
    best, weight, cost = O.vote(MS, T, A, B)
    h = MS / 2
    P = np.array([[-h, h, 0, 1], [h, h, 0, 1], [h, -h, 0, 1], [-h, -h, 0, 1]]).T          # 4 x corners
    ref = np.array([sum(np.linalg.norm((P - B[j] @ T[i] @ A[j] @ P)[:3], axis=0).sum() for j in range(len(T))) for i in range(len(T))])
    np.testing.assert_allclose(cost, ref, rtol=1e-11)
    assert best == int(np.argmin(ref))


def test_oracle_inverse_is_general_not_rigid():
    rng = np.random.default_rng(5)
    M = _random_rigid(rng, 1)[0].astype(np.float32).astype(np.float64)   # float-rounded: R^T is not the inverse any more
    assert np.abs(O.inv4(M) @ M - np.eye(4)).max() < 1e-15
    np.testing.assert_allclose(O.inv4(M), np.linalg.inv(M), rtol=0, atol=1e-15)


def test_oracle_initializer_recovers_the_scene_without_noise():
    ds = scene(noise=0.0, frames=30)
    K = ds.cam_mats.reshape(-1, 3, 3)
    det = detections_of(ds)
    r = oracle_init(det, K, [np.zeros(5)] * ds.num_cams)
    cam, mk, fr = truth_transforms(ds)
    assert r["root_cam"] == 0 and r["root_marker"] == 0
    np.testing.assert_array_equal(r["cam_ids"], np.arange(ds.num_cams))
    np.testing.assert_array_equal(r["frame_ids"], r["kept_frame_ids"])
    seen = np.unique(ds.marker_ids[ds.obs_marker])
    np.testing.assert_array_equal(r["marker_ids"], seen)
    for i, c in enumerate(r["cam_ids"]):
        assert np.abs(r["T_cam"][i] - cam[c]).max() < 1e-4
    for i, m in enumerate(r["marker_ids"]):
        assert np.abs(r["T_marker"][i] - mk[m]).max() < 1e-4
    for i, f in enumerate(r["frame_ids"]):
        assert np.abs(r["T_object"][i] - fr[list(ds.frame_ids).index(f)]).max() < 1e-4


def test_oracle_initializer_degrades_gracefully_and_honours_exclusions():
    ds = scene(noise=0.05, frames=40)
    K = ds.cam_mats.reshape(-1, 3, 3)
    det = detections_of(ds)
    cam, mk, fr = truth_transforms(ds)
    r = oracle_init(det, K, [np.zeros(5)] * ds.num_cams)
    assert max(np.abs(r["T_cam"][i] - cam[c]).max() for i, c in enumerate(r["cam_ids"])) < 0.08
    # excluded cameras vanish from the ids, and frames they alone covered are dropped (libs/initializer.cpp:373-380)
    r2 = oracle_init(det, K, [np.zeros(5)] * ds.num_cams, excluded=(0, 5))
    assert 0 not in r2["cam_ids"] and 5 not in r2["cam_ids"] and r2["root_cam"] == 1
    np.testing.assert_array_equal(r2["T_cam"][0], np.eye(4))
    # min_detections: with a very high bar nothing is left
    r3 = oracle_init(det, K, [np.zeros(5)] * ds.num_cams, min_detections=10 ** 6)
    assert len(r3["frame_ids"]) == 0 and len(r3["cam_ids"]) == 0


def test_detections_file_round_trip_truncation_and_subseqs(tmp_path):
    ds = scene(noise=0.3, frames=12)
    p = str(tmp_path / "aruco.detections")
    aar.detections_write(p, ds)
    det = aar.detections_read(p)
    ref = detections_of(ds)
    assert det.num_cams == ref.num_cams and det.num_frames == ref.num_frames
    for k in ("det_frame", "det_cam", "det_id", "det_uv"):
        np.testing.assert_array_equal(getattr(det, k), getattr(ref, k))
    # a record cut in the middle drops the whole last frame (libs/initializer.cpp:331-346)
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[:-10])
    cut = aar.detections_read(p)
    assert cut.num_frames == det.num_frames - 1 and np.all(cut.det_frame < det.num_frames - 1)
    n = len(cut.det_frame)
    np.testing.assert_array_equal(cut.det_uv, det.det_uv[:n])
    # sub-sequences [2,4] [8,9]: frames 0-1 and 5-7 are emptied, frames after the last range are left alone (:350-359)
    open(p, "wb").write(raw)
    ss = tmp_path / "subseqs.txt"
    ss.write_text("2 4\n8 9\n")
    sub = aar.detections_read(p, aar.subseqs_read(str(ss)))
    assert sub.num_frames == det.num_frames
    keep = ~np.isin(det.det_frame, [0, 1, 5, 6, 7])
    np.testing.assert_array_equal(sub.det_frame, det.det_frame[keep])
    np.testing.assert_array_equal(sub.det_uv, det.det_uv[keep])
    # an empty file has no frames; a missing one is an error
    open(p, "wb").write(struct.pack("<Q", 3))
    e = aar.detections_read(p)
    assert e.num_cams == 3 and e.num_frames == 0 and len(e.det_frame) == 0
    with pytest.raises(aar.AarError):
        aar.detections_read(str(tmp_path / "nope"))


def test_cam_configs_are_read_in_directory_name_order(tmp_path):
    for name, text, ext in (("cam_b", CALIB_YAML, "yml"), ("cam_a", CALIB_XML, "xml"), ("notes", None, None)):
        (tmp_path / name).mkdir()
        if text:
            (tmp_path / name / ("calib." + ext)).write_text(text)
    (tmp_path / "aruco.detections").write_bytes(b"")
    cams = aar.cam_configs_read(str(tmp_path))
    assert len(cams) == 2
    assert cams[0][2] == (1920, 1080) and len(cams[0][1]) == 5
    assert cams[1][2] == (1280, 720) and len(cams[1][1]) == 8


def test_initializer_fails_loudly_without_a_gpu(have_gpu):
    if have_gpu:
        pytest.skip("GPU present")
    ds = scene(noise=0.1, frames=6)
    K = ds.cam_mats.reshape(-1, 3, 3)
    with pytest.raises(aar.AarError) as e:
        aar.initializer_run(detections_of(ds), K, [np.zeros(5)] * ds.num_cams, MS)
    assert e.value.code == aar.AAR_ERR_NO_DEVICE
    with pytest.raises(aar.AarError) as e:
        aar.ippe_square(MS, K[0], np.zeros(5), ds.obs_uv[:4])
    assert e.value.code == aar.AAR_ERR_NO_DEVICE
    with pytest.raises(aar.AarError) as e:
        aar.vote_transforms(MS, [0, 1], np.eye(4)[None], np.eye(4)[None], np.eye(4)[None])
    assert e.value.code == aar.AAR_ERR_NO_DEVICE


# ---------------------------------------------------------------------------------------------------------------------
# GPU: HIP kernels and the host pipeline against the oracle
# ---------------------------------------------------------------------------------------------------------------------
DIST5 = np.array([-0.11, 0.085, 0.0012, -0.0007, -0.019])
DIST8 = np.array([0.2, -0.1, 0.001, 0.002, 0.01, 0.3, -0.05, 0.004])


@pytest.mark.gpu
@pytest.mark.parametrize("dist", [np.zeros(5), DIST5, DIST8], ids=["nodist", "dist5", "dist8"])
def test_ippe_kernel_equals_oracle(dist):
    ds = scene(noise=0.3, frames=30)
    K = ds.cam_mats.reshape(-1, 3, 3)[2]
    uv = ds.obs_uv[ds.obs_cam == 2].astype(np.float64)
    raw = O.distort_points(K, dist, uv.reshape(-1, 2)).reshape(-1, 8).astype(np.float32)
    g = aar.ippe_square(MS, K, dist, raw)
    o = O.ippe_square(MS, K, dist, raw)
    assert len(g[1]) == len(raw) > 50
    # float-rounded matrices: equal up to the last float digit (device libm vs glibc in acos / sin / cos)
    np.testing.assert_allclose(g[0], o[0], rtol=0, atol=2e-6)
    np.testing.assert_allclose(g[2], o[2], rtol=0, atol=2e-6)
    np.testing.assert_allclose(g[1], o[1], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(g[3], o[3], rtol=1e-4, atol=1e-9)
    assert np.mean(g[0] == o[0]) > 0.99 and np.mean(g[1] == o[1]) > 0.9
    assert np.all(g[1] <= g[3])


@pytest.mark.gpu
def test_vote_kernel_equals_oracle():
    rng = np.random.default_rng(11)
    sizes = [30, 0, 1, 64, 65, 200, 7]
    parts = [_vote_set(rng, n, max(1, (2 * n) // 3)) if n else (np.zeros((0, 4, 4)),) * 3 for n in sizes]
    T = np.concatenate([p[0] for p in parts]); A = np.concatenate([p[1] for p in parts]); B = np.concatenate([p[2] for p in parts])
    begin = np.concatenate([[0], np.cumsum(sizes)])
    best, weight, cost = aar.vote_transforms(MS, begin, T, A, B)
    for s, n in enumerate(sizes):
        sl = slice(begin[s], begin[s + 1])
        ob, ow, oc = O.vote(MS, T[sl], A[sl], B[sl])
        assert best[s] == ob, s
        if n:
            np.testing.assert_allclose(cost[sl], oc, rtol=1e-11, atol=1e-14)
            np.testing.assert_allclose(weight[s], ow, rtol=1e-11, atol=1e-14)
    # a NaN candidate never wins (curr_error < min_error is false); a NaN on the j-side poisons every sum: no winner
    Tn = T[:30].copy(); Tn[best[0], 0, 0] = np.nan
    b2, _, c2 = aar.vote_transforms(MS, [0, 30], Tn, A[:30], B[:30])
    assert np.isnan(c2[best[0]]) and np.isfinite(np.delete(c2, best[0])).all()
    assert b2[0] == O.vote(MS, Tn, A[:30], B[:30])[0] != best[0]
    Bn = B[:30].copy(); Bn[7, 1, 1] = np.nan
    b3, _, c3 = aar.vote_transforms(MS, [0, 30], T[:30], A[:30], Bn)
    assert np.all(np.isnan(c3)) and b3[0] == -1 == O.vote(MS, T[:30], A[:30], Bn)[0]


def _compare_with_oracle(ds_out, r, det, K, dists):
    np.testing.assert_array_equal(ds_out.cam_ids, r["cam_ids"])
    np.testing.assert_array_equal(ds_out.marker_ids, r["marker_ids"])
    np.testing.assert_array_equal(ds_out.frame_ids, r["frame_ids"])
    assert ds_out.root_cam == 0 and ds_out.root_marker == 0
    C, M, F = ds_out.num_cams, ds_out.num_markers, ds_out.num_frames
    x = ds_out.x_full
    np.testing.assert_allclose(x[:6 * (C - 1)], poses_of(r["T_cam"][1:]), rtol=0, atol=2e-6)
    np.testing.assert_allclose(x[6 * (C - 1):6 * (C - 1) + 6 * (M - 1)], poses_of(r["T_marker"][1:]), rtol=0, atol=2e-6)
    np.testing.assert_allclose(x[6 * (C - 1) + 6 * (M - 1):], poses_of(r["T_object"]), rtol=0, atol=2e-6)
    # observations: the detections of the kept frames in file order, corners undistorted with P = K
    keep = np.isin(det.det_frame, r["kept_frame_ids"]) & np.isin(det.det_cam, r["cam_ids"])
    np.testing.assert_array_equal(ds_out.frame_ids[ds_out.obs_frame], det.det_frame[keep])
    np.testing.assert_array_equal(ds_out.cam_ids[ds_out.obs_cam], det.det_cam[keep])
    np.testing.assert_array_equal(ds_out.marker_ids[ds_out.obs_marker], det.det_id[keep])
    for i, c in enumerate(ds_out.cam_ids):
        sel = ds_out.obs_cam == i
        want = O.undistort_points(K[c], dists[c], det.det_uv[keep][sel].reshape(-1, 2)).reshape(-1, 8)
        np.testing.assert_allclose(ds_out.obs_uv[sel], want, rtol=0, atol=1e-4)
        np.testing.assert_array_equal(ds_out.cam_mats.reshape(-1, 9)[i], np.asarray(K[c]).reshape(9))


@pytest.mark.gpu
@pytest.mark.parametrize("noise,distorted", [(0.3, False), (0.1, True)], ids=["noise0.3", "distorted"])
def test_initializer_pipeline_equals_oracle(noise, distorted):
    ds = scene(noise=noise, frames=40)
    K = ds.cam_mats.reshape(-1, 3, 3)
    dists = [DIST5 * (1 + 0.1 * c) if distorted else np.zeros(5) for c in range(ds.num_cams)]
    det = detections_of(ds, dists if distorted else None)
    out = aar.initializer_run(det, K, dists, MS, sizes=[(1280, 720)] * ds.num_cams)
    r = oracle_init(det, K, dists)
    _compare_with_oracle(out, r, det, K, dists)
    assert out.num_obs == ds.num_obs and out.marker_size == np.float32(MS)
    np.testing.assert_array_equal(out.image_sizes.reshape(-1, 2), [[1280, 720]] * out.num_cams)


@pytest.mark.gpu
def test_initializer_exclusions_threshold_and_min_detections_equal_oracle():
    ds = scene(noise=0.2, frames=30)
    K = ds.cam_mats.reshape(-1, 3, 3)
    dists = [np.zeros(5)] * ds.num_cams
    det = detections_of(ds)
    median = int(np.median(np.bincount(det.det_frame)))   # about half of the frames survive this bar
    for kw in (dict(excluded=(3,)), dict(threshold=1.2), dict(threshold=50.0), dict(min_detections=median)):
        out = aar.initializer_run(det, K, dists, MS, **kw)
        r = oracle_init(det, K, dists, **kw)
        _compare_with_oracle(out, r, det, K, dists)
    assert 3 not in aar.initializer_run(det, K, dists, MS, excluded=(3,)).cam_ids
    # cameras 0 and 5 out: the ring of cameras falls apart, which is reported (the reference runs into std::map::at)
    with pytest.raises(aar.AarError) as e:
        aar.initializer_run(det, K, dists, MS, excluded=(0, 5))
    assert "not connected" in str(e.value)
    with pytest.raises(aar.AarError):   # nothing left: the reference would build an empty mapper
        aar.initializer_run(det, K, dists, MS, min_detections=10 ** 6)


@pytest.mark.gpu
def test_initializer_then_lm_reaches_the_ground_truth():
    # the path find_solution runs: detections -> Initializer -> MultiCamMapper::solve (apps/find_solution.cpp:113-160)
    ds = scene(noise=0.1, frames=60)
    K = ds.cam_mats.reshape(-1, 3, 3)
    det = detections_of(ds)
    init = aar.initializer_run(det, K, [np.zeros(5)] * ds.num_cams, MS)
    with aar.Problem(init) as p:
        rmse0 = p.reproj_stats(init.x_full)[0]
        x, rep = p.lm_solve(init.x_full)
        rmse1 = p.reproj_stats(x)[0]
    assert rmse0 < 20 and rmse1 < 0.2 and rmse1 < rmse0
    # camera poses end up at the truth (same ids, same gauge: root camera / root marker 0)
    C = ds.num_cams
    np.testing.assert_allclose(x[:6 * (C - 1)], ds.x_truth[:6 * (C - 1)], rtol=0, atol=2e-2)


@pytest.mark.gpu
def test_initializer_reports_a_disconnected_camera_graph():
    # four cameras 90 degrees apart never share a marker: the reference would run into std::map::at
    ds = aar.synth(2, num_frames=20)
    K = ds.cam_mats.reshape(-1, 3, 3)
    with pytest.raises(aar.AarError) as e:
        aar.initializer_run(detections_of(ds), K, [np.zeros(5)] * ds.num_cams, MS)
    assert e.value.code == aar.AAR_ERR_INVALID and "not connected" in str(e.value)
    # and a camera slot without a calibration is refused
    ds = scene(noise=0.1, frames=6)
    K = ds.cam_mats.reshape(-1, 3, 3)
    with pytest.raises(aar.AarError):
        aar.initializer_run(detections_of(ds), K[:3], [np.zeros(5)] * 3, MS)


@pytest.mark.gpu
def test_find_solution_driver_runs_the_initializer_then_the_lm(tmp_path):
    # apps/find_solution.cpp:101-163 end to end through the C++ driver: calib folders + aruco.detections in, initial*.solution
    # and final*.solution out; -from-initial restarts from the file instead
    import subprocess
    from conftest import PKG
    exe = os.path.join(PKG, "aar_find_solution")
    folder = str(tmp_path / "seq")
    assert subprocess.run([exe, "--synth", "3", folder], capture_output=True, text=True).returncode == 0
    os.remove(os.path.join(folder, "initial.solution"))
    run = subprocess.run([exe, folder, "0.05"], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr + run.stdout
    assert "Initializer:" in run.stdout and "The algorithm took:" in run.stdout
    init = aar.solution_read(os.path.join(folder, "initial.solution"))
    fin = aar.solution_read(os.path.join(folder, "final.solution"))
    assert os.path.exists(os.path.join(folder, "initial.solution.yaml")) and os.path.exists(os.path.join(folder, "final.solution.yaml"))
    assert (init.num_cams, init.num_frames) == (8, 500) and init.num_obs == fin.num_obs
    np.testing.assert_array_equal(init.obs_uv, fin.obs_uv)
    with aar.Problem(init) as p:
        assert p.reproj_stats(fin.x_full)[0] < p.reproj_stats(init.x_full)[0]
    # the same Initializer result through the C ABI
    det = aar.detections_read(os.path.join(folder, "aruco.detections"))
    cams = aar.cam_configs_read(folder)
    again = aar.initializer_run(det, [c[0] for c in cams], [c[1] for c in cams], 0.05, sizes=[c[2] for c in cams])
    np.testing.assert_allclose(again.x_full, init.x_full, rtol=0, atol=1e-9)   # the file stores matrices: vec -> mat -> vec
    run2 = subprocess.run([exe, folder, "0.05", "x", "-from-initial"], capture_output=True, text=True)
    assert run2.returncode == 0 and "Initializer:" not in run2.stdout
    # the reference's options (parsed from argv[4] on, apps/find_solution.cpp:47) name the files the same way (:74-97) and
    # reach the Initializer: sub-sequences empty the frames outside them, excluded cameras vanish from the solution
    with open(os.path.join(folder, "subseqs.txt"), "w") as f:
        f.write("100 299\n")
    run3 = subprocess.run([exe, folder, "0.05", "x", "-subseqs", "-exclude-cams", "3", "-with-huber", "-thresh", "2.5"],
                          capture_output=True, text=True)
    assert run3.returncode == 0, run3.stderr + run3.stdout
    name = "_subseqs_with_huber_excluded_cams_3_thresh_2.5.solution"
    sub = aar.solution_read(os.path.join(folder, "initial" + name))
    assert os.path.exists(os.path.join(folder, "final" + name)) and os.path.exists(os.path.join(folder, "final" + name + ".yaml"))
    assert 3 not in sub.cam_ids and sub.num_cams == 7
    assert sub.frame_ids.min() >= 100 and sub.num_frames < init.num_frames   # frames after the last range are kept (:350-359)


@pytest.mark.gpu
def test_track_app_flow_initial_object_poses_then_track():
    # apps/track.cpp:68-123 for a whole recording: a solved map (here: the ground truth), NEW detections, the Initializer with
    # the map's transforms fixed, then MultiCamMapper::track() on every frame
    ds = scene(noise=0.2, frames=50)
    K = ds.cam_mats.reshape(-1, 3, 3)
    dists = [np.zeros(5)] * ds.num_cams
    det = detections_of(ds)
    sol = aar.Dataset()
    sol.__dict__.update(ds.__dict__)
    sol.x_full = ds.x_truth.copy()
    out = aar.initializer_run(det, K, dists, MS, solution=sol)
    C, M = ds.num_cams, ds.num_markers
    ns = 6 * (C - 1) + 6 * (M - 1)
    # the map is untouched: all of the solution's cameras / markers (seen or not), their poses and intrinsics
    np.testing.assert_array_equal(out.cam_ids, ds.cam_ids)
    np.testing.assert_array_equal(out.marker_ids, ds.marker_ids)
    np.testing.assert_array_equal(out.x_full[:ns], ds.x_truth[:ns])
    np.testing.assert_array_equal(out.cam_mats, ds.cam_mats)
    assert (out.optimize_cam_poses, out.optimize_marker_poses, out.optimize_object_poses) == (False, False, True)
    np.testing.assert_array_equal(out.frame_ids, ds.frame_ids)
    # object poses equal the oracle's, given the same fixed transforms
    cam, mk, fr = truth_transforms(ds)
    r = oracle_init(det, K, dists, fixed=(ds.cam_ids, np.array(cam), ds.marker_ids, np.array(mk)))
    np.testing.assert_array_equal(r["frame_ids"], out.frame_ids)
    np.testing.assert_allclose(out.x_full[ns:], poses_of(r["T_object"]), rtol=0, atol=2e-6)
    # track(): every frame's 6-DoF LM from those poses lands on the true object pose (0.2 px noise)
    with aar.Problem(out, optimize=(False, False, True)) as p:
        x, it, err = p.track(out.x_full)
    assert np.array_equal(x[:ns], ds.x_truth[:ns]) and np.all(it >= 1)
    T_true = np.array(fr)
    T_got = np.array([rigid(x[ns + 6 * f:][:6]) for f in range(out.num_frames)])
    T_init = np.array([rigid(out.x_full[ns + 6 * f:][:6]) for f in range(out.num_frames)])
    assert np.abs(T_got - T_true).max() < 0.02
    assert np.abs(T_got - T_true).mean() < np.abs(T_init - T_true).mean()
    # detections of a marker the map does not hold are dropped, as are frames that fall below min_detections because of it
    det2 = detections_of(ds)
    det2.det_id[det2.det_id == det2.det_id[0]] = 999
    out2 = aar.initializer_run(det2, K, dists, MS, solution=sol)
    assert out2.num_obs == int(np.sum(det2.det_id != 999)) and 999 not in out2.marker_ids
