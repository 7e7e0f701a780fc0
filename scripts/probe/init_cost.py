# Diagnostic: fixed cost of one solve (upload, start-point evaluation, first non-speculative step, download, wrapper) against a step
import cProfile, pstats, sys, time, numpy as np
sys.path.insert(0, "automatic-ar_amd")
import aar
ds = aar.synth(3)
p = aar.Problem(ds, residual_mode=aar.RES_F32)
for mi in (1, 2, 5, 15):
    prm = aar.lm_default_params(max_iters=mi)
    for _ in range(20): p.lm_solve(ds.x_full, params=prm, trace_cap=1)
    aar.lib().aar_device_synchronize()
    t0 = time.perf_counter(); n = 200
    its = 0; csec = 0.0
    for _ in range(n):
        x, rep = p.lm_solve(ds.x_full, params=prm, trace_cap=1); its += rep["iterations"]; csec += rep["solve_seconds"]
    aar.lib().aar_device_synchronize()
    dt = time.perf_counter() - t0
    print("max_iters", mi, "iterations/solve", its / n, "us per solve", 1e6 * dt / n, "of which inside aar_lm_solve after init", 1e6 * csec / n)
prm = aar.lm_default_params(max_iters=1)
pr = cProfile.Profile(); pr.enable()
for _ in range(500): p.lm_solve(ds.x_full, params=prm, trace_cap=1)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
