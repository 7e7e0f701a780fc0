"""Scratch (GPU): AAR_SOLVER=pcg, grid size sweep"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import aar
os.environ["AAR_SOLVER"] = "pcg"
for cfg, steps, grids in ((3, 300, (8, 16, 32, 48, 64, 128, 256)), (2, 300, (8, 16, 32, 64)), (5, 30, (128, 256, 512, 1024))):
    ds = aar.synth(cfg)
    for g in grids:
        os.environ["AAR_PCG_GRID"] = str(g)
        with aar.Problem(ds) as p:
            p.lm_solve(ds.x_full)
            it0 = p.pcg_iterations()[1]
            done, t0 = 0, time.perf_counter()
            while done < steps:
                x2, rep2 = p.lm_solve(ds.x_full, params=aar.lm_default_params(max_iters=min(15, steps - done)))
                done += rep2["iterations"]
            aar.lib().aar_device_synchronize()
            dt = time.perf_counter() - t0
            its = p.pcg_iterations()[1] - it0
        print("config %d grid %4d: %8.1f it/s (%.3f ms per LM step, %.1f CG its per step)" % (cfg, g, done / dt, 1e3 * dt / done, its / done), flush=True)
