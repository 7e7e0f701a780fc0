import sys, os
sys.path[:0] = ["automatic-ar_amd", "tests"]
import numpy as np, aar, oracle_lib as ol
from conftest import load_golden
for name, huber in (("g1_cfg2_retry", False), ("g1_cfg2_far", False), ("g1_cfg2_huber_retry", True), ("g1_cfg3_cut", False)):
    ds, g = load_golden(name)
    prm = aar.lm_default_params(tau=float(g["tau"][0])) if "tau" in g else None
    for rep_i in range(3):
        with aar.Problem(ds, with_huber=huber) as p:
            x_d, rep_d = p.lm_solve(ds.x_full, params=prm, trace_cap=600); rmse_d, _ = p.reproj_stats(x_d)
        with aar.Problem(ds, with_huber=huber, solver="spcg") as p:
            x, rep = p.lm_solve(ds.x_full, params=prm, trace_cap=600); rmse, _ = p.reproj_stats(x); st = p.solver_stats()
        print(name, "its", rep["iterations"], rep_d["iterations"], "drmse_faithful %.2e drmse_direct %.2e" % (abs(rmse - g["faithful_rmse"][0]), abs(rmse - rmse_d)), "fallbacks", st["fallbacks"], "solves", st["solves"], flush=True)
