"""Differences between two pose vectors AS TRANSFORMS (test infrastructure).

The reference's result is a set of 4x4 matrices (`.solution.yaml`: transforms_to_root_cam, transforms_to_root_marker, root_marker_to_root_cam,
libs/multicam_mapper.cpp:1233-1268); its pose vector holds them as (Rodrigues vector, translation) 6-tuples in the order cameras without the root,
markers without the root, frames (fill_io_vec_*, :500-522).  Two runs are compared entry by entry on the rotation MATRICES and the translations --
Rodrigues vectors themselves are not comparable near theta = pi.
"""
import numpy as np
from scipy.spatial.transform import Rotation


def _groups(ds):
    C, M, F = int(ds.num_cams), int(ds.num_markers), int(ds.num_frames)
    a, b = 6 * (C - 1), 6 * (C - 1) + 6 * (M - 1)
    return {"cams": (0, a), "markers": (a, b), "frames": (b, b + 6 * F)}


def pose_delta(ds, xa, xb):
    """{"cams" | "markers" | "frames": (max |R_a - R_b| over matrix entries, max |t_a - t_b| in metres)} for two pose vectors of data set `ds`
    (entries behind the poses -- an intrinsics tail -- are ignored)"""
    out = {}
    for name, (lo, hi) in _groups(ds).items():
        if hi == lo:
            out[name] = (0.0, 0.0)
            continue
        pa, pb = np.asarray(xa[lo:hi], dtype=np.float64).reshape(-1, 6), np.asarray(xb[lo:hi], dtype=np.float64).reshape(-1, 6)
        Ra, Rb = Rotation.from_rotvec(pa[:, :3]).as_matrix(), Rotation.from_rotvec(pb[:, :3]).as_matrix()
        out[name] = (float(np.abs(Ra - Rb).max()), float(np.abs(pa[:, 3:] - pb[:, 3:]).max()))
    return out


def pose_delta_max(ds, xa, xb):
    """(largest rotation-matrix entry difference, largest translation difference) over all groups"""
    d = pose_delta(ds, xa, xb)
    return max(v[0] for v in d.values()), max(v[1] for v in d.values())
