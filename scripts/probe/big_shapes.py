import sys, time, numpy as np
sys.path.insert(0, "automatic-ar_amd")
import aar
for (C, M, F) in ((16, 400, 1000), (40, 100, 600), (3, 500, 300)):
    t0 = time.time()
    ds = aar.synth(5, num_cams=C, num_markers=M, num_frames=F)
    t1 = time.time()
    with aar.Problem(ds, residual_mode=aar.RES_F32) as p:
        x, rep = p.lm_solve(ds.x_full)
        rmse, _ = p.reproj_stats(x)
    print("C/M/F", C, M, F, "obs", ds.num_obs, "synth %.1fs" % (t1 - t0), "iters", rep["iterations"], "stop", rep["stop_code"], "rmse %.6f" % rmse,
          "ms/iter %.3f" % (1e3 * rep["solve_seconds"] / max(1, rep["iterations"])))
