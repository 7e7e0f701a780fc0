"""Absolute tolerance of the inner solves (aar_solver_options.pcg_abs_tol) against rate and final poses: configs 3, 4, 5 with the default relative forcing term and
pcg_abs_tol in {off, 5e-5, 2e-5 (default), 1e-5}; LM it/s over repeated solves, CG iterations per LM step, dRMSE and pose difference against the direct solver."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")]
import aar
from pose_metrics import pose_delta_max
for cfg in (3, 4, 5):
    ds = aar.synth(cfg)
    with aar.Problem(ds, solver="direct") as p:
        xd, rd = p.lm_solve(ds.x_full)
        rmd = p.reproj_stats(xd)[0]
    for tol in (1.0, 5e-5, 2e-5, 1e-5):
        with aar.Problem(ds, pcg_abs_tol=tol) as p:
            x, r = p.lm_solve(ds.x_full)
            rm = p.reproj_stats(x)[0]
            st0 = p.solver_stats()
            reps = 3 if cfg == 5 else 20
            aar.lib().aar_device_synchronize(); t0 = time.perf_counter(); n = 0
            for _ in range(reps):
                _, rr = p.lm_solve(ds.x_full, trace_cap=1); n += rr["iterations"]
            aar.lib().aar_device_synchronize(); dt = time.perf_counter() - t0
            st = p.solver_stats()
        print("config %d (%s) abs_tol %-6g: %2d LM steps, %7.0f it/s, CG/step %5.2f, fb %d, dRMSE %+.1e, poses R %.1e t %.1e" % (
            cfg, st["solver"], tol, r["iterations"], n / dt, (st["total_iterations"] - st0["total_iterations"]) / n, st["fallbacks"], rm - rmd, *pose_delta_max(ds, x, xd)), flush=True)
