"""Scratch (GPU): AAR_SOLVER=pcg against the direct path -- one damped step at a tight tolerance, whole solves at the default eta, timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import aar

def run(cfg, steps, pcg, eta=None):
    os.environ["AAR_SOLVER"] = "pcg" if pcg else "direct"
    if eta is not None: os.environ["AAR_PCG_ETA"] = str(eta)
    else: os.environ.pop("AAR_PCG_ETA", None)
    ds = aar.synth(cfg)
    with aar.Problem(ds) as p:
        d = p.eval_damped_step(ds.x_full, 1e3)
        x, rep = p.lm_solve(ds.x_full)
        rmse, _ = p.reproj_stats(x)
        it0 = p.pcg_iterations()[1]
        done, t0 = 0, time.perf_counter()
        while done < steps:
            x2, rep2 = p.lm_solve(ds.x_full, params=aar.lm_default_params(max_iters=min(15, steps - done)))
            done += rep2["iterations"]
        aar.lib().aar_device_synchronize()
        dt = time.perf_counter() - t0
        its = p.pcg_iterations()[1] - it0
    return d, rmse, rep["iterations"], done / dt, 1e3 * dt / done, its / max(1, done)

for cfg, steps in ((2, 300), (3, 300), (4, 150), (5, 30)):
    d0, rm0, it0, r0, ms0, _ = run(cfg, steps, False)
    dt, _, _, _, _, cg = run(cfg, 15, True, eta=1e-10)
    print("config %d: damped step PCG(eta 1e-10) vs direct: rel %.2e (%.1f CG its per solve)" % (cfg, np.abs(dt - d0).max() / np.abs(d0).max(), cg), flush=True)
    for eta in (0.1, 0.01):
        d1, rm1, it1, r1, ms1, cg = run(cfg, steps, True, eta=eta)
        print("   eta %-5g direct %8.1f it/s (%.3f ms, %d LM steps, RMSE %.9f) | pcg %8.1f it/s (%.3f ms, %d LM steps, RMSE delta %.2e px, %.1f CG its per LM step)"
              % (eta, r0, ms0, it0, rm0, r1, ms1, it1, abs(rm1 - rm0), cg), flush=True)
