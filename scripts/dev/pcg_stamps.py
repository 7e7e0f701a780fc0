"""Scratch (GPU): cycle stamps of k_pcgf's CG iteration (diagnostic build: scripts/dev/pcg_stamps.sh; AAR_LIB=.../libaar_st.so).  s_memtime ticks at 100 MHz on
gfx950 (10 ns): phases of an iteration in us for three workgroups (first, middle, last), averaged over the iterations of the last solve."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import aar
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ds = aar.synth(cfg)
with aar.Problem(ds, solver="pcg") as p:
    for _ in range(2):
        p.lm_solve(ds.x_full)
    x, rep = p.lm_solve(ds.x_full, params=aar.lm_default_params(max_iters=12))
    aar.lib().aar_device_synchronize()
    buf = (C.c_ulonglong * (3 * 32 * 16))()
    assert aar.lib().aar_debug_pcg_stamps(buf) == 0
    st = np.array(buf[:], dtype=np.float64).reshape(3, 32, 16)
    its = p.pcg_iterations()[0]
    print("last solve: %d CG iterations" % its)
    tick = float(os.environ.get("TICK_NS", "10"))
    names = ["top", "yacc0+sync", "U p (wave 0)"] + ["frames w%d" % w for w in range(8)] + ["end sync", "flush issued", "hop", "updates"]
    for sel, label in enumerate(("wg 0", "wg G/2", "wg G-1")):
        rows = []
        for it in range(1, min(its, 30)):
            t0 = st[sel, it, 0]
            rows.append([(st[sel, it, k] - t0) * tick * 1e-3 for k in range(1, 15)] + [(st[sel, it + 1, 0] - t0) * tick * 1e-3 if it + 1 < its else np.nan])
        r = np.nanmean(np.array(rows), axis=0)
        su = [(st[sel, 31, k] - st[sel, 31, 0]) * tick * 1e-3 for k in range(1, 6)] + [(st[sel, 0, 0] - st[sel, 31, 0]) * tick * 1e-3]
        print(label, "SET-UP:", " ".join("%s %.1f" % (n, v) for n, v in zip(["slots pass", "flush issued", "hop", "inverses", "z p init", "first iteration top"], su)))
        print(label, " ".join("%s %.1f" % (n, v) for n, v in zip(names[1:] + ["next top"], r)))
