"""The default path (solver AUTO) from FAR starts: initial perturbation x1, x3, x8 (the reference takes 16, 16, 17 iterations there) at config 3, x3 at
config 4, tau = 1 and tau = 1e-6 (tiny initial damping: rejected tries): LM steps, fall-backs, final RMSE and poses against the direct solver."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")]
import aar
from pose_metrics import pose_delta_max
for cfg, scale, tau in ((3, 1.0, 1.0), (3, 3.0, 1.0), (3, 8.0, 1.0), (3, 1.0, 1e-6), (3, 3.0, 1e-6), (4, 3.0, 1.0), (4, 1.0, 1e-6)):
    ds = aar.synth(cfg, init_scale=scale)
    prm = aar.lm_default_params(tau=tau)
    out = {}
    off = "--abs-tol-off" in sys.argv      # the relative forcing term alone (pcg_abs_tol = 1: never binding), as before the absolute tolerance existed
    for s in ("direct", None):
        with aar.Problem(ds, solver=s, pcg_abs_tol=(1.0 if off and s is None else None)) as p:
            x, rep = p.lm_solve(ds.x_full, params=prm, trace_cap=600)
            st = p.solver_stats()
            out[s] = (x, rep, p.reproj_stats(x)[0], st)
    xd, rd, rmd, _ = out["direct"]
    x, r, rm, st = out[None]
    print("config %d, start x%g, tau %g: direct %d steps (max tries %d) RMSE %.9f | default (%s) %d steps (max tries %d), %d CG its, %d fall-backs, dRMSE %+.1e, poses R %.1e t %.1e" % (
        cfg, scale, tau, rd["iterations"], max(t["tries"] for t in rd["trace"]), rmd, st["solver"], r["iterations"], max(t["tries"] for t in r["trace"]),
        st["total_iterations"], st["fallbacks"], rm - rmd, *pose_delta_max(ds, x, xd)), flush=True)
