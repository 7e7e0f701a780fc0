"""Bring-up check of the spcg solver (csrc/spcg_kernels.hip) on a GPU box: the damped step against the direct chain at a tight
forcing term, then whole LM solves (iterations, final error, CG iterations, wall time) for direct / spcg at the default eta."""
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
    with aar.Problem(ds, solver="direct") as pd, aar.Problem(ds, solver="spcg", pcg_eta=1e-12) as pt:
        H, B, ss = pd.eval_normal_equations(ds.x_full)
        for mu in (H.diagonal().max(), 1e-3 * H.diagonal().max(), 1e-6 * H.diagonal().max()):
            d0 = pd.eval_damped_step(ds.x_full, mu)
            d1 = pt.eval_damped_step(ds.x_full, mu)
            st = pt.solver_stats()
            print("cfg %d mu %.3e: |d_spcg - d_direct| / |d_direct| = %.3e   (CG iterations %d, fallbacks %d)"
                  % (cfg, mu, np.linalg.norm(d1 - d0) / np.linalg.norm(d0), st["last_iterations"], st["fallbacks"]), flush=True)
    for solver in ("direct", "spcg"):
        with aar.Problem(ds, solver=solver) as p:
            x, rep = p.lm_solve(ds.x_full)
            rmse, _ = p.reproj_stats(x)
            st0 = p.solver_stats()
            n = 0
            aar.lib().aar_device_synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                _, r = p.lm_solve(ds.x_full, trace_cap=1)
                n += r["iterations"]
            aar.lib().aar_device_synchronize()
            dt = time.perf_counter() - t0
            st = p.solver_stats()
            print("cfg %d %-6s: %2d LM iterations, RMSE %.9f px, %.1f it/s (%.1f us/step), CG its/solve %.2f, fallbacks %d, tries %s"
                  % (cfg, solver, rep["iterations"], rmse, n / dt, 1e6 * dt / n, st["total_iterations"] / max(1, st["solves"]), st["fallbacks"],
                     [t["tries"] for t in rep["trace"]]), flush=True)
