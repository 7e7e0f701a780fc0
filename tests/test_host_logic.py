"""Host-side pieces of the product that need no GPU: synthetic generator, file formats, SE(3) helpers, shard
planning, the aar_find_solution driver's file contract.  CPU only."""
import os
import struct
import subprocess

import numpy as np
import pytest

import aar
import oracle_lib as ol
from conftest import PKG, load_golden


def test_generator_is_deterministic_and_matches_committed_inputs():
    # the golden fixtures hold the generator's output as of the commit that made them: bit-for-bit
    for name in ("g1_cfg2", "g1_cfg3_cut", "g1_cfg2_far"):
        gds, g = load_golden(name)
        cfg, C, M, F, scale = g["synth_args"]
        ds = aar.synth(int(cfg), num_cams=int(C), num_markers=int(M), num_frames=int(F), init_scale=float(scale))
        assert (ds.num_cams, ds.num_markers, ds.num_frames, ds.num_obs) == (gds.num_cams, gds.num_markers, gds.num_frames, gds.num_obs)
        for k in aar.Dataset.FIELDS:
            assert np.array_equal(getattr(ds, k), getattr(gds, k)), k


def test_generator_statistics():
    ds = aar.synth(3)
    assert (ds.num_cams, ds.num_markers) == (8, 40) and ds.num_frames == 500
    assert 10000 < ds.num_obs < 20000           # ~10 % visibility
    assert np.all(np.diff(ds.obs_frame) >= 0)   # reference residual order: frame-major ...
    same = np.diff(ds.obs_frame) == 0
    assert np.all(np.diff(ds.obs_cam)[same] >= 0)  # ... then camera
    assert np.bincount(ds.obs_frame, minlength=ds.num_frames).min() >= 2  # libs/initializer.cpp:379
    o = ol.Oracle(ds)
    st = o.reproj_stats(ds.x_truth)
    assert abs(st["rmse"] - 0.3 * np.sqrt(2)) < 0.01   # noise sigma 0.3 px per coordinate
    assert st["rmse"] < o.reproj_stats(ds.x_full)["rmse"] / 20
    assert ds.marker_size == float(np.float32(0.05))


def test_solution_file_round_trip_and_layout(tmp_path):
    ds, _ = load_golden("g2_small")
    path = str(tmp_path / "initial.solution")
    aar.solution_write(path, ds)
    back = aar.solution_read(path)
    for k in ("cam_ids", "marker_ids", "frame_ids", "image_sizes", "cam_mats", "dist_coeffs", "obs_frame", "obs_cam", "obs_marker", "obs_uv"):
        assert np.array_equal(getattr(back, k), getattr(ds, k)), k
    assert (back.root_cam, back.root_marker, back.marker_size) == (ds.root_cam, ds.root_marker, ds.marker_size)
    # poses are stored after a vec -> mat -> vec round trip (libs/multicam_mapper.cpp:1054,1086)
    np.testing.assert_allclose(back.x_full, ds.x_full, atol=1e-12)
    # byte layout of the header: size_t C, int32 ids[C], size_t root id, C x (int32 w, int32 h), size_t M ...
    raw = open(path, "rb").read()
    C, M, F = ds.num_cams, ds.num_markers, ds.num_frames
    off = 0
    assert struct.unpack_from("<Q", raw, off)[0] == C; off += 8
    assert list(struct.unpack_from("<%di" % C, raw, off)) == list(ds.cam_ids); off += 4 * C
    assert struct.unpack_from("<Q", raw, off)[0] == ds.cam_ids[ds.root_cam]; off += 8
    assert list(struct.unpack_from("<%di" % (2 * C), raw, off)) == list(ds.image_sizes.reshape(-1)); off += 8 * C
    assert struct.unpack_from("<Q", raw, off)[0] == M; off += 8 + 4 * M + 8
    assert struct.unpack_from("<d", raw, off)[0] == ds.marker_size; off += 8
    assert struct.unpack_from("<Q", raw, off)[0] == F; off += 8 + 4 * F
    nvec = 6 * (C - 1) + 6 * (M - 1) + 6 * F + 9 * C   # always the full default-Config vector (:1085-1089)
    vec = np.frombuffer(raw, dtype="<f8", count=nvec, offset=off); off += 8 * nvec
    np.testing.assert_allclose(vec[: ds.full_len], ds.x_full, atol=1e-12)
    assert list(vec[ds.full_len: ds.full_len + 4]) == [1000.0, 640.0, 1000.0, 360.0]  # fx, cx, fy, cy
    assert struct.unpack_from("<Q", raw, off)[0] == F     # serialize_frame_cam_markers
    assert raw[-4:] == bytes([1, 1, 1, 0])                # Config flags as 4 bools, intrinsics off
    n_marker_records = ds.num_obs
    # every frame carries one record per camera, empty ones included (what the reference writes after read_detections_file,
    # libs/initializer.cpp:333-347, and what its loop-counter reader needs, libs/multicam_mapper.cpp:1117-1119)
    assert len(raw) == off + 8 + F * (4 + 8) + F * C * (4 + 8) + n_marker_records * 36 + 4


def test_solution_reader_honours_ids_and_drops_unknown(tmp_path):
    ds, _ = load_golden("g2_small")
    ds.cam_ids = ds.cam_ids * 3 + 1          # non-contiguous ids
    ds.marker_ids = ds.marker_ids * 7 + 5
    path = str(tmp_path / "ids.solution")
    aar.solution_write(path, ds)
    back = aar.solution_read(path)
    assert np.array_equal(back.cam_ids, ds.cam_ids) and np.array_equal(back.marker_ids, ds.marker_ids)
    assert np.array_equal(back.obs_cam, ds.obs_cam) and np.array_equal(back.obs_marker, ds.obs_marker)
    with pytest.raises(aar.AarError) as e:
        aar.solution_read(str(tmp_path / "missing.solution"))
    assert e.value.code == aar.AAR_ERR_IO
    open(str(tmp_path / "short.solution"), "wb").write(open(path, "rb").read()[:100])
    with pytest.raises(aar.AarError):
        aar.solution_read(str(tmp_path / "short.solution"))


def test_detections_file_layout(tmp_path):
    # aruco.detections: size_t num_cams, then per frame, per camera: size_t n, n x (int32 id, 8 floats)
    ds, _ = load_golden("g2_small")
    path = str(tmp_path / "aruco.detections")
    aar.detections_write(path, ds)
    raw = open(path, "rb").read()
    ncam = struct.unpack_from("<Q", raw, 0)[0]
    assert ncam == ds.cam_ids.max() + 1
    off, frame, seen = 8, 0, []
    while off < len(raw):
        for c in range(ncam):
            n = struct.unpack_from("<Q", raw, off)[0]; off += 8
            for _ in range(n):
                mid = struct.unpack_from("<i", raw, off)[0]
                uv = struct.unpack_from("<8f", raw, off + 4)
                seen.append((frame, c, mid, uv)); off += 36
        frame += 1
    assert off == len(raw) and len(seen) == ds.num_obs
    assert frame == ds.frame_ids.max() + 1     # frame index = position in the file
    f0 = [(ds.frame_ids[f], ds.cam_ids[c], ds.marker_ids[m]) for f, c, m in zip(ds.obs_frame, ds.obs_cam, ds.obs_marker)]
    assert [(a, b, c) for a, b, c, _ in seen] == f0
    assert np.array_equal(np.array([s[3] for s in seen], dtype=np.float32), ds.obs_uv)


def test_yaml_solution_is_opencv_filestorage_shaped(tmp_path):
    ds, _ = load_golden("g2_small")
    path = str(tmp_path / "final.solution.yaml")
    aar.solution_write_yaml(path, ds)
    txt = open(path).read()
    assert txt.startswith("%YAML:1.0\n---\n")
    assert txt.count("!!opencv-matrix") == ds.num_cams + ds.num_markers + ds.num_frames
    for key in ("marker_size:", "transforms_to_root_cam:", "transforms_to_root_marker:", "root_marker_to_root_cam:"):
        assert key in txt
    assert txt.count("cam_id:") == ds.num_cams and txt.count("marker_id:") == ds.num_markers and txt.count("frame_id:") == ds.num_frames
    # root camera: identity, written the FileStorage way ("1." / "0.")
    first = txt.split("transforms_to_root_cam:")[1].split("}")[0]
    vals = first.split("data:[")[1].split("]")[0].replace("\n", " ").split(",")
    assert [v.strip() for v in vals] == ["1.", "0.", "0.", "0.", "0.", "1.", "0.", "0.", "0.", "0.", "1.", "0.", "0.", "0.", "0.", "1."]


def test_rodrigues_product_vs_oracle():
    rng = np.random.default_rng(3)
    for _ in range(200):
        w = rng.normal(size=3)
        w *= rng.uniform(0, 3.0) / np.linalg.norm(w)   # |w| < pi: the range cv::Rodrigues returns
        R = aar.rodrigues_vec2mat(w)
        np.testing.assert_allclose(R, ol.rodrigues_vec2mat(w), atol=1e-15)
        np.testing.assert_allclose(aar.rodrigues_mat2vec(R), w, atol=1e-12)
        np.testing.assert_allclose(aar.rodrigues_mat2vec(R * 1.00001), ol.rodrigues_mat2vec(R * 1.00001), atol=1e-9)
    for axis in (np.array([0, 0, 1.0]), np.array([2.0, -1.0, 2.0]) / 3):
        R = aar.rodrigues_vec2mat(axis * np.pi)     # theta = pi branch
        w = aar.rodrigues_mat2vec(R)
        np.testing.assert_allclose(aar.rodrigues_vec2mat(w), R, atol=1e-9)


def test_plan_shards_balances_observations():
    rng = np.random.default_rng(0)
    counts = rng.integers(2, 60, size=500)
    for world in (1, 2, 3, 4, 8):
        b = aar.plan_shards(counts, world)
        assert b[0] == 0 and b[-1] == 500 and np.all(np.diff(b) >= 0)
        per = np.array([counts[b[r]:b[r + 1]].sum() for r in range(world)])
        assert per.sum() == counts.sum()
        assert per.max() - per.min() <= 2 * counts.max()      # balanced by observation count, not frame count
    assert list(aar.plan_shards([5, 5], 4)) in ([0, 0, 1, 1, 2], [0, 1, 1, 2, 2], [0, 0, 1, 2, 2], [0, 1, 1, 1, 2])  # more ranks than frames
    assert list(aar.plan_shards([], 2)) == [0, 0, 0]


def test_find_solution_driver_file_contract(tmp_path):
    exe = os.path.join(PKG, "aar_find_solution")
    folder = str(tmp_path / "box")
    out = subprocess.run([exe, "--synth", "1", folder], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    for f in ("aruco.detections", "initial.solution", "initial.solution.yaml"):
        assert os.path.exists(os.path.join(folder, f))
    ds = aar.solution_read(os.path.join(folder, "initial.solution"))
    assert (ds.num_cams, ds.num_markers) == (3, 6)          # box-like stand-in for BASELINE.json configs[0]
    if aar.device_count() == 0:
        # no GPU here: the product must fail loudly, not fall back to a CPU path
        run = subprocess.run([exe, folder, "0.05"], capture_output=True, text=True)
        assert run.returncode != 0 and "no HIP device" in (run.stderr + run.stdout)
        assert not os.path.exists(os.path.join(folder, "final.solution"))


# ---- calibration files and remove_distortions ----
CALIB_XML = """<?xml version="1.0"?>
<opencv_storage>
<calibration_time>"Mon Feb 18 10:00:00 2019"</calibration_time>
<image_width>1920</image_width>
<image_height>1080</image_height>
<camera_matrix type_id="opencv-matrix">
  <rows>3</rows>
  <cols>3</cols>
  <dt>d</dt>
  <data>
    1.4321e+03 0. 9.61e+02 0. 1.4298e+03 5.395e+02 0. 0. 1.</data></camera_matrix>
<distortion_coefficients type_id="opencv-matrix">
  <rows>1</rows>
  <cols>5</cols>
  <dt>d</dt>
  <data>
    -1.1e-01 8.5e-02 1.2e-03 -7.e-04 -1.9e-02</data></distortion_coefficients>
<avg_reprojection_error>3.1e-01</avg_reprojection_error>
</opencv_storage>
"""

CALIB_YAML = """%YAML:1.0
---
image_width: 1280
image_height: 720
camera_matrix: !!opencv-matrix
   rows: 3
   cols: 3
   dt: d
   data: [ 9.5e+02, 0., 6.4e+02, 0., 9.52e+02,
       3.6e+02, 0., 0., 1. ]
distortion_coefficients: !!opencv-matrix
   rows: 8
   cols: 1
   dt: d
   data: [ 0.2, -0.1, 0.001, 0.002, 0.01, 0.3, -0.05, 0.004 ]
"""


def test_cam_config_reader_xml_and_yaml(tmp_path):
    # the four keys CamConfig::read_from_file takes (libs/cam_config.cpp:52-80), both cv::FileStorage dialects
    px = tmp_path / "calib.xml"
    px.write_text(CALIB_XML)
    K, dist, size = aar.cam_config_read(str(px))
    assert size == (1920, 1080)
    np.testing.assert_array_equal(K, [[1432.1, 0, 961.0], [0, 1429.8, 539.5], [0, 0, 1]])
    np.testing.assert_array_equal(dist, [-0.11, 0.085, 0.0012, -0.0007, -0.019])
    py = tmp_path / "calib.yml"
    py.write_text(CALIB_YAML)
    K, dist, size = aar.cam_config_read(str(py))
    assert size == (1280, 720)
    np.testing.assert_array_equal(K, [[950.0, 0, 640.0], [0, 952.0, 360.0], [0, 0, 1]])
    np.testing.assert_array_equal(dist, [0.2, -0.1, 0.001, 0.002, 0.01, 0.3, -0.05, 0.004])
    # a file that lacks one of the keys is refused, as the reference does (:57-77); so is a missing file
    bad = tmp_path / "bad.yml"
    bad.write_text(CALIB_YAML.replace("image_height", "image_heigth"))
    with pytest.raises(aar.AarError):
        aar.cam_config_read(str(bad))
    with pytest.raises(aar.AarError):
        aar.cam_config_read(str(tmp_path / "nope.xml"))


def test_oracle_undistort_inverts_the_distortion_model():
    # distort(undistort(p)) = p: the fixed-point inversion against the forward model, 5- and 8-coefficient vectors
    rng = np.random.default_rng(7)
    K = np.array([[1432.1, 0, 961.0], [0, 1429.8, 539.5], [0, 0, 1]])
    for dist in ([-0.11, 0.085, 0.0012, -0.0007, -0.019], [0.2, -0.1, 0.001, 0.002, 0.01, 0.3, -0.05, 0.004], []):
        uv = np.stack([rng.uniform(100, 1820, 4000), rng.uniform(60, 1020, 4000)], axis=1).astype(np.float32)
        und = ol.undistort_points(K, dist, uv)
        back = ol.distort_points(K, dist, und.astype(np.float64))
        # float output of undistortPoints (~6e-5 px at 1000 px) + what five iterations of the contraction leave at the image
        # corners (2.5e-3 px for the strong rational model)
        assert np.abs(back - uv).max() < (5e-3 if len(dist) > 5 else 2e-3), (dist, np.abs(back - uv).max())
        if len(dist) == 0:
            np.testing.assert_allclose(und, uv, atol=1.3e-4)   # no distortion: K^-1 then K, rounded to float


def test_solution_reader_reference_indexing(tmp_path):
    # libs/multicam_mapper.cpp:1101-1122: the reference files the f-th frame record under frame id f and the c-th camera record
    # of a frame under camera id c (loop counters).  Default reader: the stored ids.  The writer emits one record per camera
    # and frame (empty ones included) as the reference does after its Initializer, so the two views agree whenever camera and
    # frame ids are 0..n-1 -- which is what lets the reference's own track / overlay read a final.solution written here.
    ds = aar.synth(2)
    assert np.array_equal(ds.frame_ids, np.arange(ds.num_frames)) and np.array_equal(ds.cam_ids, np.arange(ds.num_cams))
    p = str(tmp_path / "a.solution")
    aar.solution_write(p, ds)
    a, b = aar.solution_read(p), aar.solution_read(p, reference_indexing=True)
    for k in ("obs_frame", "obs_cam", "obs_marker", "obs_uv", "frame_ids", "cam_ids"):
        assert np.array_equal(getattr(a, k), getattr(ds, k)) and np.array_equal(getattr(b, k), getattr(ds, k)), k
    # a solution without camera 1 (-exclude-cams): the c-th record is no longer camera c; the reference's view shifts camera 2's
    # detections to "camera 1" (unknown: dropped by fill_iteration_arrays, :356-360) and camera 3's to camera 2
    ex = aar.Dataset()
    ex.__dict__.update(ds.__dict__)
    keep = ds.obs_cam != 1
    ex.num_cams, ex.cam_ids = 3, np.array([0, 2, 3], dtype=np.int32)
    ex.image_sizes, ex.cam_mats, ex.dist_coeffs = ds.image_sizes[[0, 2, 3]], ds.cam_mats[[0, 2, 3]], ds.dist_coeffs[[0, 2, 3]]
    remap = np.array([0, -1, 1, 2])
    ex.obs_frame, ex.obs_marker, ex.obs_uv = ds.obs_frame[keep], ds.obs_marker[keep], ds.obs_uv[keep]
    ex.obs_cam = remap[ds.obs_cam[keep]].astype(np.int32)
    ex.num_obs = int(keep.sum())
    ex.x_full = np.concatenate([ds.x_full[6:18], ds.x_full[18:]])
    q = str(tmp_path / "ex.solution")
    aar.solution_write(q, ex)
    a, b = aar.solution_read(q), aar.solution_read(q, reference_indexing=True)
    assert np.array_equal(a.obs_cam, ex.obs_cam) and a.num_obs == ex.num_obs
    n0, n2, n3 = (int(np.sum(ex.cam_ids[ex.obs_cam] == c)) for c in (0, 2, 3))
    assert b.num_obs == n0 + n3                                    # records 0, 1, 2 = "cameras 0, 1, 2": 1 is unknown, 3's land on 2
    assert int(np.sum(b.cam_ids[b.obs_cam] == 2)) == n3 and int(np.sum(b.cam_ids[b.obs_cam] == 3)) == 0
    # a solution whose frame ids do not start at 0 (the -subseqs case): frame record 0 would be frame id 0, which does not exist
    sub = aar.Dataset()
    sub.__dict__.update(ds.__dict__)
    sub.frame_ids = ds.frame_ids + 100
    q = str(tmp_path / "b.solution")
    aar.solution_write(q, sub)
    assert np.array_equal(aar.solution_read(q).frame_ids, sub.frame_ids)
    with pytest.raises(aar.AarError) as e:
        aar.solution_read(q, reference_indexing=True)
    assert e.value.code == aar.AAR_ERR_INVALID and "re-indexing" in str(e.value)


def test_cam_configs_in_readdir_order(tmp_path):
    # libs/cam_config.cpp:80-95 takes the sub-directories in readdir order; the default here is ascending name
    import ctypes
    folder = tmp_path / "seq"
    for name, w in (("cam_b", 1920), ("cam_a", 1280), ("cam_c", 640)):
        (folder / name).mkdir(parents=True)
        (folder / name / "calib.yml").write_text(CALIB_YAML.replace("image_width: 1280", "image_width: %d" % w))
    by_name = [c[2][0] for c in aar.cam_configs_read(str(folder))]
    assert by_name == [1280, 1920, 640]
    listed = [n for n in os.listdir(str(folder))]                    # os.listdir = readdir order minus "." and ".."
    by_readdir = [c[2][0] for c in aar.cam_configs_read(str(folder), readdir_order=True)]
    assert by_readdir == [{"cam_a": 1280, "cam_b": 1920, "cam_c": 640}[n] for n in listed]
    assert sorted(by_readdir) == sorted(by_name)


def test_solution_writer_gathers_interleaved_cameras(tmp_path):
    # a frame whose observations of one camera are NOT contiguous in the caller's arrays (camera 0, 1, 0 ...): every detection is
    # written under its camera (the round trip comes back grouped by camera, nothing dropped)
    ds = aar.synth(3, num_frames=6)
    inter, nf0 = None, 0
    for fr in range(ds.num_frames):           # the first frame in which ordering by marker breaks up the camera runs
        f0 = np.nonzero(ds.obs_frame == fr)[0]
        order = np.argsort(ds.obs_marker[f0], kind="stable")
        cams = ds.obs_cam[f0][order]
        if np.count_nonzero(np.diff(cams) != 0) + 1 > len(set(cams.tolist())):
            inter = np.concatenate([np.arange(0, f0[0]), f0[order], np.arange(f0[-1] + 1, ds.num_obs)])
            nf0, first = len(f0), f0[0]
            break
    assert inter is not None
    mixed = aar.Dataset.__new__(aar.Dataset)
    mixed.__dict__.update(ds.__dict__)
    for k in ("obs_frame", "obs_cam", "obs_marker", "obs_uv"):
        setattr(mixed, k, np.ascontiguousarray(getattr(ds, k)[inter]))
    path = str(tmp_path / "mixed.solution")
    aar.solution_write(path, mixed)
    back = aar.solution_read(path)
    assert back.num_obs == ds.num_obs
    key = lambda d: sorted(zip(d.obs_frame.tolist(), d.obs_cam.tolist(), d.obs_marker.tolist(), map(tuple, d.obs_uv.tolist())))
    assert key(back) == key(ds)
    assert np.all(np.diff(back.obs_cam[first: first + nf0]) >= 0)        # grouped by camera again


def test_host_code_under_address_and_ub_sanitizers(tmp_path):
    # file formats, shard plan, detections / calibration readers (truncated and malformed inputs included) compiled with
    # -fsanitize=address,undefined against stubs of the device entry points: the HOST code of the product, no GPU needed
    import subprocess
    from conftest import PKG, ROOT
    exe = str(tmp_path / "asan_host_main")
    host = [os.path.join(PKG, "host", f) for f in ("dataset.cpp", "synth.cpp", "solution_io.cpp", "cam_config.cpp", "initializer.cpp")]
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                         os.path.join(ROOT, "tests", "tools", "asan_host_main.cpp")] + host + ["-o", exe], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr and "LeakSanitizer" not in run.stderr
    assert "obs " in run.stdout and "subseqs" in run.stdout


def test_reference_shaped_solver_code_compiles_against_the_mirror(tmp_path):
    # tests/tools/solver_seam_main.cpp holds caller code in the shape of libs/multicam_mapper.cpp:419-443 (solver.solve(io_vec,
    # bind(&MultiCamMapper::error_function, ...), bind(&MultiCamMapper::jacobian_function, ...)), init(z, f), step(f, J), step(f),
    # solve(z, bind(&MultiCamMapper::error_function_tracking, ...))): it must COMPILE and link against the mirror of
    # automatic-ar_amd/host/multicam_mapper.h anywhere; what it computes is checked under -m gpu.  Without a device it stops at
    # the first device call with the library's message.
    import subprocess
    from conftest import PKG, ROOT
    exe = str(tmp_path / "solver_seam_main")
    cc = subprocess.run(["g++", "-O0", "-std=c++17", "-Wall", os.path.join(ROOT, "tests", "tools", "solver_seam_main.cpp"), "-o", exe, "-L" + PKG, "-laar",
                         "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    if aar.device_count() == 0:
        run = subprocess.run([exe, "2"], capture_output=True, text=True, timeout=120)
        assert run.returncode == 2 and "no CPU path" in run.stderr        # exception: no HIP device ...; this library has no CPU path
