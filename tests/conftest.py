import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "automatic-ar_amd")
for p in (PKG, os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # build what is missing (the driver normally runs __graft_entry__.build() first)
    if not os.path.exists(os.path.join(PKG, "libaar.so")):
        subprocess.check_call(["make", "-s", "-j8", "-C", PKG, "all"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])


GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    """A golden fixture as (Dataset, dict of expected arrays)."""
    import aar

    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    ds = aar.Dataset()
    m = g["meta"]
    ds.num_cams, ds.num_markers, ds.num_frames, ds.root_cam, ds.root_marker = [int(v) for v in m]
    ds.marker_size = float(g["marker_size"][0])
    for k in aar.Dataset.FIELDS:
        setattr(ds, k, g[k])
    ds.num_obs = len(ds.obs_frame)
    ds.optimize_cam_poses = ds.optimize_marker_poses = ds.optimize_object_poses = True
    ds.optimize_cam_intrinsics = False
    return ds, g


@pytest.fixture(scope="session")
def have_gpu():
    import aar

    return aar.device_count() > 0
