"""Scratch (GPU): wall time of lm_solve against its iteration cap -> fixed cost per solve and marginal cost per LM step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd")]
import numpy as np, aar
ds = aar.synth(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
with aar.Problem(ds) as p:
    for i in range(30): p.lm_solve(ds.x_full, params=aar.lm_default_params(), trace_cap=1)
    res = []
    for cap in (1, 2, 4, 8, 15):
        prm = aar.lm_default_params(max_iters=cap)
        for i in range(20): p.lm_solve(ds.x_full, params=prm, trace_cap=1)
        aar.lib().aar_device_synchronize()
        t0 = time.perf_counter()
        n = 300
        for i in range(n): x, rep = p.lm_solve(ds.x_full, params=prm, trace_cap=1)
        aar.lib().aar_device_synchronize()
        dt = (time.perf_counter() - t0) / n
        res.append((rep["iterations"], dt * 1e6))
        print("cap %2d: %2d iterations, %.1f us per solve" % (cap, rep["iterations"], dt * 1e6))
    it = np.array([r[0] for r in res], float); us = np.array([r[1] for r in res])
    A = np.vstack([np.ones_like(it), it]).T
    c = np.linalg.lstsq(A, us, rcond=None)[0]
    print("fit: %.1f us fixed per solve + %.1f us per LM step" % (c[0], c[1]))
