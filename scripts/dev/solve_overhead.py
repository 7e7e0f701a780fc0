"""Where a solve's wall time goes outside the LM loop: Python binding, upload / first evaluation / download -- per-solve wall time against
the library's own solve_seconds (loop only) for 1-iteration and whole solves."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")]
import aar
ds = aar.synth(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
for solver in ("direct", "auto"):
    with aar.Problem(ds, solver=solver) as p:
        for mi in (1, 2, 10000):
            prm = aar.lm_default_params(max_iters=mi)
            for _ in range(20):
                p.lm_solve(ds.x_full, params=prm, trace_cap=1)
            aar.lib().aar_device_synchronize()
            n, its, inner = 300, 0, 0.0
            t0 = time.perf_counter()
            for _ in range(n):
                x, rep = p.lm_solve(ds.x_full, params=prm, trace_cap=1)
                its += rep["iterations"]; inner += rep["solve_seconds"]
            aar.lib().aar_device_synchronize()
            dt = time.perf_counter() - t0
            print("%-6s max_iters %5d: %.1f us per solve wall (%.1f its), %.1f us inside the LM loop, %.1f us around it; %.1f us per step overall"
                  % (solver, mi, 1e6 * dt / n, its / n, 1e6 * inner / n, 1e6 * (dt - inner) / n, 1e6 * dt / its))
