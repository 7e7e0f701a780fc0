import sys, numpy as np
sys.path[:0] = ["/root/repo/automatic-ar_amd", "/root/repo/tests"]
import aar
from test_gpu_parity import load_golden
ds, g = load_golden("g1_cfg2_huber_retry")
prm = aar.lm_default_params(tau=float(g["tau"][0]))
for det in (True, False):
    with aar.Problem(ds, with_huber=True, deterministic=det) as p:
        x, rep = p.lm_solve(ds.x_full, params=prm, trace_cap=600)
    mu = np.array([t["mu"] for t in rep["trace"]]); k = 60
    print("deterministic", det, "max rel dev of mu over 60 steps: %.3e" % np.max(np.abs(mu[:k] - g["analytic_mu"][:k]) / g["analytic_mu"][:k]))
