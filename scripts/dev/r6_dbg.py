import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")]
import aar
comm = aar.Comm(aar.Comm.make_id(), 1, 0, 0)
for cfg in (3, 5):
    ds = aar.synth(cfg)
    try:
        with aar.Problem(ds, solver="pcg", comm=comm) as p:
            x, r = p.lm_solve(ds.x_full)
            print(cfg, "ok", r["iterations"], p.solver_stats()["total_iterations"], flush=True)
    except aar.AarError as e:
        print(cfg, "FAILED", e, flush=True)
