"""Bring-up check of k_pcgf's coarse space (GPU): repeated LM runs and partial runs with solver = pcg, with and without the coarse space."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import aar
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ds = aar.synth(cfg)
with aar.Problem(ds, solver="pcg") as p:
    for rep in range(6):
        for mi in (15, 7, 8, 3):
            try:
                x, r = p.lm_solve(ds.x_full, params=aar.lm_default_params(max_iters=mi), trace_cap=64)
                st = p.solver_stats()
                print("rep %d max_iters %2d: %2d iterations, final err %.9g, cg total %d, last %d, tries %s" % (rep, mi, r["iterations"], r["final_err"], st["total_iterations"], st["last_iterations"], [t["tries"] for t in r["trace"]][:16]), flush=True)
            except aar.AarError as e:
                print("rep %d max_iters %d: FAILED %s" % (rep, mi, e), flush=True)
