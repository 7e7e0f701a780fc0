"""k_spcg's cost model on a GPU box: average kernel time (HIP events on the library's stream) against average CG iterations per solve,
varied through the forcing term -- time = set-up + iterations x hand-over."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")]
import aar

cfgs = [int(a) for a in sys.argv[1:]] or [2, 3]
for cfg in cfgs:
    ds = aar.synth(cfg)
    pts = []
    for eta in (0.9, 0.3, 0.1, 1e-2, 1e-4, 1e-8):
        with aar.Problem(ds, solver="spcg", pcg_eta=eta) as p:
            p.lm_solve(ds.x_full, trace_cap=1)
            s0 = p.solver_stats()
            p.set_kernel_profiling(True)
            for _ in range(5):
                p.lm_solve(ds.x_full, trace_cap=1)
            kt = p.kernel_times()
            p.set_kernel_profiling(False)
            s1 = p.solver_stats()
            sec, cnt = kt["k_spcg"]
            its = (s1["total_iterations"] - s0["total_iterations"]) / max(1, s1["solves"] - s0["solves"])
            pts.append((its, 1e6 * sec / cnt))
            print("cfg %d eta %-6g: k_spcg %.2f us avg over %d launches, %.2f CG iterations per solve, fallbacks %d; others: %s"
                  % (cfg, eta, 1e6 * sec / cnt, cnt, its, s1["fallbacks"],
                     " ".join("%s %.1f" % (k, 1e6 * v[0] / v[1]) for k, v in kt.items() if v[1] and k != "k_spcg")), flush=True)
    a = np.array(pts)
    slope, icpt = np.polyfit(a[:, 0], a[:, 1], 1)
    print("cfg %d: k_spcg ~ %.2f us + %.3f us x iterations" % (cfg, icpt, slope), flush=True)
