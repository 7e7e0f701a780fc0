"""Scratch (GPU): AAR_SOLVER=pcg, threads per workgroup at configs 4 / 5"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import aar
os.environ["AAR_SOLVER"] = "pcg"
for cfg, steps in ((4, 150), (5, 30)):
    ds = aar.synth(cfg)
    for th in (256, 512):
        os.environ["AAR_PCG_THREADS"] = str(th)
        with aar.Problem(ds) as p:
            x, rep = p.lm_solve(ds.x_full)
            rm, _ = p.reproj_stats(x)
            it0 = p.pcg_iterations()[1]
            done, t0 = 0, time.perf_counter()
            while done < steps:
                x2, rep2 = p.lm_solve(ds.x_full, params=aar.lm_default_params(max_iters=min(15, steps - done)))
                done += rep2["iterations"]
            aar.lib().aar_device_synchronize()
            dt = time.perf_counter() - t0
            its = p.pcg_iterations()[1] - it0
        print("config %d threads %4d: %8.1f it/s (%.3f ms per LM step, %.1f CG its per step) rmse %.9f" % (cfg, th, done / dt, 1e3 * dt / done, its / done, rm), flush=True)
