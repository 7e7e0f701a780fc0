"""Forcing-term sweep of the spcg solver on a GPU box: LM steps, CG iterations per solve, final RMSE against the direct path, LM it/s."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")]
import aar

cfgs = [int(a) for a in sys.argv[1:]] or [2, 3, 4]
for cfg in cfgs:
    ds = aar.synth(cfg)
    with aar.Problem(ds, solver="direct") as p:
        x, rep = p.lm_solve(ds.x_full)
        rm0, _ = p.reproj_stats(x)
        it0 = rep["iterations"]
    print("cfg %d direct: %d LM iterations, RMSE %.9f" % (cfg, it0, rm0), flush=True)
    for eta in (0.3, 0.1, 0.05, 0.03, 0.02, 0.01, 0.003, 0.001):
        with aar.Problem(ds, solver="spcg", pcg_eta=eta) as p:
            x, rep = p.lm_solve(ds.x_full)
            rmse, _ = p.reproj_stats(x)
            s0 = p.solver_stats()
            n = 0
            aar.lib().aar_device_synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                _, r = p.lm_solve(ds.x_full, trace_cap=1)
                n += r["iterations"]
            aar.lib().aar_device_synchronize()
            dt = time.perf_counter() - t0
            st = p.solver_stats()
            print("cfg %d eta %-6g: %2d LM iterations, |RMSE - direct| %.2e px, CG its/solve %5.2f, fallbacks %d, %.0f it/s (%.1f us/step)"
                  % (cfg, eta, rep["iterations"], abs(rmse - rm0), st["total_iterations"] / max(1, st["solves"]), st["fallbacks"], n / dt, 1e6 * dt / n), flush=True)
