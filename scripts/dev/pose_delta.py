"""Final POSES of the inexact solvers (SPCG / PCG / AUTO) against the direct solver's, as transforms.

north_star asks for final.solution POSES within a stated tolerance, not only for the reprojection error: a marker seen once or twice sits in a flat
valley of the cost where an inexact LM step can move its pose without moving the RMSE.  For every workload (synthetic configs at full size, the golden
fixtures) this prints, per entity group (cameras, markers, frames): the largest entry-wise difference of the rotation matrices and of the translations
between the run of a solver and the direct run, and -- fixtures -- against the reference-faithful CPU run's final vector stored in the fixture.

    python scripts/dev/pose_delta.py [2 3 4 5 g1_cfg2 ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import aar  # noqa: E402
from conftest import load_golden  # noqa: E402
from pose_metrics import pose_delta  # noqa: E402


def main():
    names = sys.argv[1:] or ["2", "3", "4", "5", "g1_cfg2", "g1_cfg2_retry", "g1_cfg2_far", "g1_cfg2_huber", "g1_cfg2_huber_retry", "g1_cfg3_cut"]
    for name in names:
        if name.isdigit():
            ds, g = aar.synth(int(name)), {}
        else:
            ds, g = load_golden(name)
        huber = bool(g.get("with_huber", [0])[0]) if "with_huber" in g else ("huber" in name)
        prm = aar.lm_default_params(tau=float(g["tau"][0])) if "tau" in g else None
        runs = {}
        for s in ("direct", "auto", "spcg", "pcg"):
            try:
                with aar.Problem(ds, solver=s, with_huber=huber) as p:
                    x, rep = p.lm_solve(ds.x_full, params=prm, trace_cap=600)
                    runs[s] = (x, rep["iterations"], p.reproj_stats(x)[0], p.solver_stats())
            except aar.AarError as e:
                print("  %s %s: %s" % (name, s, e))
        xd = runs["direct"][0]
        print("%s  (C %d M %d F %d, N %d)%s" % (name, ds.num_cams, ds.num_markers, ds.num_frames, ds.num_obs, " huber" if huber else ""))
        for s, (x, its, rmse, st) in runs.items():
            d = pose_delta(ds, x, xd)
            line = "  %-6s -> %-6s its %3d cg %5d fb %d rmse %.9f d_rmse %+.2e | vs direct: cams R %.1e t %.1e  markers R %.1e t %.1e  frames R %.1e t %.1e" % (
                s, st["solver"], its, st["total_iterations"], st["fallbacks"], rmse, rmse - runs["direct"][2],
                d["cams"][0], d["cams"][1], d["markers"][0], d["markers"][1], d["frames"][0], d["frames"][1])
            if "faithful_x" in g:
                f = pose_delta(ds, x, g["faithful_x"])
                line += " | vs faithful: cams %.1e %.1e markers %.1e %.1e frames %.1e %.1e" % (f["cams"] + f["markers"] + f["frames"])
            print(line)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
