"""Which entities carry the distance between the direct device path and the reference-faithful CPU run?

The widest pose bar of tests/test_gpu_solvers.py (POSE_BAR_FAITHFUL = 3e-4) exists because the direct path itself ends 1.5e-4 / 6e-5 m from the
reference-faithful run (analytic Jacobian against the reference's float-quantised central differences, libs/multicam_mapper.cpp:976-994).  This script prints,
for config 3 (or a fixture), every camera's and marker's pose distance (largest rotation-matrix entry, largest translation component) between
    the device path with solver = direct               (what the product computes)
    the oracle's LM with the faithful numeric Jacobian  (oracle/ba_oracle.cpp; pinned to the real SparseLevMarq by tests/test_oracle_golden.py)
next to the number of marker observations the entity has, sorted by that count -- and the same per frame, binned by observations per frame.

    python scripts/dev/pose_delta_by_entity.py [config | fixture]        (profiles/r06_pose_delta_by_entity.txt holds the committed runs: 3, g1_cfg2)
"""
import os
import sys

import numpy as np
from scipy.spatial.transform import Rotation

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import aar  # noqa: E402
import oracle_lib as ol  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
if cfg.isdigit():
    ds = aar.synth(int(cfg))
    o = ol.Oracle(ds)
    xf, repf = o.lm_solve(ds.x_full, jac_mode=ol.JAC_NUMERIC_F32, res_mode=ol.RES_F32)
else:   # a golden fixture: its faithful run is stored (tests/golden/make_golden.py: the REAL SparseLevMarq driving the faithful callbacks)
    from conftest import load_golden
    ds, g = load_golden(cfg)
    o = ol.Oracle(ds)
    xf, repf = np.array(g["faithful_x"]), {"iterations": int(g["faithful_iterations"][0]) if "faithful_iterations" in g else -1}
C, M, F = int(ds.num_cams), int(ds.num_markers), int(ds.num_frames)
with aar.Problem(ds, solver="direct") as p:
    xd, repd = p.lm_solve(ds.x_full)
    rd, _ = p.reproj_stats(xd)
with aar.Problem(ds) as p:
    xa, repa = p.lm_solve(ds.x_full)
    st = p.solver_stats()
rf = o.reproj_stats(xf)["rmse"]
print("config %s: faithful CPU run %d LM steps, RMSE %.9f px | device direct %d steps, RMSE %.9f px (delta %.1e) | device default (%s) %d steps" %
      (cfg, repf["iterations"], rf, repd["iterations"], rd, abs(rd - rf), st["solver"], repa["iterations"]))


def dist(xa_, xb_, lo, n):
    pa, pb = np.asarray(xa_[lo:lo + 6 * n]).reshape(-1, 6), np.asarray(xb_[lo:lo + 6 * n]).reshape(-1, 6)
    Ra, Rb = Rotation.from_rotvec(pa[:, :3]).as_matrix(), Rotation.from_rotvec(pb[:, :3]).as_matrix()
    return np.abs(Ra - Rb).reshape(n, -1).max(axis=1), np.abs(pa[:, 3:] - pb[:, 3:]).max(axis=1)


obs_cam = np.bincount(np.asarray(ds.obs_cam), minlength=C)
obs_mk = np.bincount(np.asarray(ds.obs_marker), minlength=M)
obs_fr = np.bincount(np.asarray(ds.obs_frame), minlength=F)
rc, rm = int(ds.root_cam), int(ds.root_marker)
cams = [c for c in range(C) if c != rc]
mks = [m for m in range(M) if m != rm]
a, b = 6 * (C - 1), 6 * (C - 1) + 6 * (M - 1)
for title, other in (("direct device path vs reference-faithful CPU run", xd), ("default device path (%s) vs reference-faithful CPU run" % st["solver"], xa)):
    print("\n== %s ==" % title)
    dRc, dTc = dist(other, xf, 0, C - 1)
    dRm, dTm = dist(other, xf, a, M - 1)
    dRf, dTf = dist(other, xf, b, F)
    print("cameras (id, observations, |dR|max, |dt|max m), by observations:")
    for k in np.argsort([obs_cam[c] for c in cams]):
        print("   cam %2d  %6d  %.2e  %.2e" % (cams[k], obs_cam[cams[k]], dRc[k], dTc[k]))
    print("markers (id, observations, |dR|max, |dt|max m), by observations:")
    for k in np.argsort([obs_mk[m] for m in mks]):
        print("   marker %3d  %6d  %.2e  %.2e" % (mks[k], obs_mk[mks[k]], dRm[k], dTm[k]))
    print("frames, binned by observations per frame (frames, max |dR|, max |dt|, median |dR|, median |dt|):")
    edges = [0, 5, 10, 15, 20, 25, 30, 40, 1000]
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = (obs_fr >= lo) & (obs_fr < hi)
        if sel.any():
            print("   %3d..%-4d %5d  %.2e  %.2e  %.2e  %.2e" % (lo, hi - 1, sel.sum(), dRf[sel].max(), dTf[sel].max(), np.median(dRf[sel]), np.median(dTf[sel])))
    print("largest: cameras R %.2e t %.2e | markers R %.2e t %.2e | frames R %.2e t %.2e" % (dRc.max(), dTc.max(), dRm.max(), dTm.max(), dRf.max(), dTf.max()))
    few = np.array([obs_mk[m] for m in mks])
    order = np.argsort(few)
    q = max(1, len(mks) // 4)
    print("markers: the quarter with the FEWEST observations (<= %d) carries R %.2e t %.2e; the quarter with the MOST (>= %d) R %.2e t %.2e" %
          (few[order[q - 1]], dRm[order[:q]].max(), dTm[order[:q]].max(), few[order[-q]], dRm[order[-q:]].max(), dTm[order[-q:]].max()))
