"""Scratch (GPU): the damped step with and without the look-ahead of the dense factorisation (AAR_LDL_LOOKAHEAD), at 5-, 7- and 14-tile systems."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd")]
CASES = {"5t": dict(cfg=3, num_cams=4, num_markers=62, num_frames=40), "7t": dict(cfg=3, num_cams=8, num_markers=100, num_frames=120), "cfg5": dict(cfg=5),
         "4t-intr": dict(cfg=3, intr=True)}
if len(sys.argv) > 1:
    import numpy as np, aar
    kw = dict(CASES[sys.argv[1]]); cfg = kw.pop("cfg"); intr = kw.pop("intr", False)
    ds = aar.synth(cfg, **kw)
    with aar.Problem(ds, intrinsics=intr) as p:
        x0 = p.x_with_intrinsics(ds.x_full) if intr else ds.x_full
        d = p.eval_damped_step(x0, 1e-3)
        np.save("/tmp/la_%s_%s.npy" % (os.environ.get("AAR_LDL_LOOKAHEAD", "1"), sys.argv[1]), d)
        x, rep = p.lm_solve(x0)
        print(sys.argv[1], "lookahead", os.environ.get("AAR_LDL_LOOKAHEAD", "1"), "iterations", rep["iterations"], "final_err %.12g" % rep["final_err"], "n =", len(d))
else:
    import numpy as np
    for c in CASES:
        for m in ("0", "1"):
            subprocess.run([sys.executable, __file__, c], env=dict(os.environ, AAR_LDL_LOOKAHEAD=m))
        a, b = np.load("/tmp/la_0_%s.npy" % c), np.load("/tmp/la_1_%s.npy" % c)
        print(c, "max |delta_on - delta_off| / max |delta_off| =", float(np.abs(a - b).max() / np.abs(a).max()))
