"""Forcing-sequence sweep of the inexact solvers against the FINAL POSES of the direct solver (GPU box).

For every workload and every (pcg_eta_loose, pcg_eta) pair: LM steps, CG iterations per damped solve, LM it/s over repeated solves, |RMSE - direct|,
and the largest rotation-matrix-entry / translation difference of the final poses against the direct run (tests/pose_metrics.py).  The direct run's
own last LM step is printed as the scale: how far the poses still moved in the step after which the reference's stopping rule fired.

    python scripts/dev/eta_pose_sweep.py [3 4 5 g1_cfg3_cut ...]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")]
import aar  # noqa: E402
from conftest import load_golden  # noqa: E402
from pose_metrics import pose_delta_max  # noqa: E402

SPCG = [(0.1, 0.02), (0.0, 0.02), (0.0, 0.01), (0.0, 3e-3), (0.0, 1e-3), (0.0, 3e-4), (0.0, 1e-4), (0.02, 1e-3), (0.01, 1e-3), (0.01, 1e-4), (3e-3, 1e-4)]
PCG = [(0.3, 0.1), (0.0, 0.1), (0.0, 0.03), (0.0, 0.01), (0.0, 3e-3), (0.0, 1e-3), (0.0, 3e-4), (0.0, 1e-4), (0.03, 1e-3), (0.01, 1e-3), (0.01, 1e-4)]
if os.environ.get("SWEEP_GRID") == "a":      # the first pass (profiles/r05_eta_pose_sweep.txt, part a): loose early / tight late
    SPCG = [(0.1, 0.02), (0.1, 5e-3), (0.1, 1e-3), (0.1, 1e-4), (0.1, 1e-5), (0.1, 1e-6), (0.02, 1e-4), (0.0, 1e-4), (0.0, 1e-6)]
    PCG = [(0.3, 0.1), (0.3, 1e-2), (0.3, 1e-3), (0.3, 1e-4), (0.3, 1e-5), (0.1, 1e-4), (0.0, 1e-4), (0.0, 1e-6)]


def rate(p, x0, prm, reps):
    n = 0
    aar.lib().aar_device_synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        _, r = p.lm_solve(x0, params=prm, trace_cap=1)
        n += r["iterations"]
    aar.lib().aar_device_synchronize()
    return n / (time.perf_counter() - t0)


def main():
    names = sys.argv[1:] or ["3", "4", "5", "g1_cfg3_cut", "g1_cfg2_retry", "g1_cfg2_far"]
    for name in names:
        ds, g = (aar.synth(int(name)), {}) if name.isdigit() else load_golden(name)
        prm = aar.lm_default_params(tau=float(g["tau"][0])) if "tau" in g else None
        reps = 3 if name == "5" else 20
        with aar.Problem(ds, solver="direct") as p:
            xd, rep = p.lm_solve(ds.x_full, params=prm, trace_cap=600)
            rm0 = p.reproj_stats(xd)[0]
            it0 = rep["iterations"]
            # the poses one LM step before the stop: the resolution of the reference's own stopping rule
            xprev, _ = p.lm_solve(ds.x_full, params=aar.lm_default_params(max_iters=it0 - 1, **({"tau": float(g["tau"][0])} if "tau" in g else {})), trace_cap=1)
            last = pose_delta_max(ds, xd, xprev)
            r0 = rate(p, ds.x_full, prm, reps)
        print("%s: direct %d LM steps, RMSE %.9f, %.0f it/s; its last step moved the poses by R %.1e t %.1e; last delta_norm %.2e"
              % (name, it0, rm0, r0, last[0], last[1], rep["trace"][-1]["delta_norm"]), flush=True)
        with aar.Problem(ds, solver="auto") as p:
            auto = p.solver_stats()["solver"]
        for solver, grid in (("spcg", SPCG), ("pcg", PCG)):
            if (name in ("4", "5") or not name.isdigit()) and solver != auto and not (solver == "spcg" and auto == "direct"):
                continue
            for loose, tight in grid:
                try:
                    with aar.Problem(ds, solver=solver, pcg_eta=tight, pcg_eta_loose=loose if loose > 0 else tight, pcg_abs_tol=float(os.environ.get("SWEEP_ABS_TOL", "1.0"))) as p:
                        x, rep = p.lm_solve(ds.x_full, params=prm, trace_cap=600)
                        rmse = p.reproj_stats(x)[0]
                        st = p.solver_stats()
                        r = rate(p, ds.x_full, prm, reps)
                        dR, dt = pose_delta_max(ds, x, xd)
                        print("  %-4s%s loose %-5g tight %-6g: %3d LM steps, CG/solve %6.2f, fb %d, %6.0f it/s (%.2fx direct), dRMSE %+.1e, poses vs direct R %.1e t %.1e"
                              % (solver, "*" if solver == auto else " ", loose, tight, rep["iterations"], st["total_iterations"] / max(1, st["solves"]), st["fallbacks"],
                                 r, r / r0, rmse - rm0, dR, dt), flush=True)
                except aar.AarError as e:
                    print("  %s loose %g tight %g: %s" % (solver, loose, tight, e), flush=True)


if __name__ == "__main__":
    main()
