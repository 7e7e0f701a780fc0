# Diagnostic: which call of the first two solves of a process stalls for ~10 ms?
import sys, time, numpy as np
sys.path.insert(0, "automatic-ar_amd")
import aar
ds = aar.synth(3)
p = aar.Problem(ds, residual_mode=aar.RES_F32)
ev = []
def T(name, f):
    t0 = time.perf_counter(); r = f(); dt = 1e3 * (time.perf_counter() - t0)
    if dt > 1.0: ev.append("%s %.2f ms" % (name, dt))
    return r
for s in range(3):
    T("solve%d init" % s, lambda: p.lm_init(ds.x_full, params=aar.lm_default_params()))
    for k in range(8):
        T("solve%d step%d" % (s, k), lambda: p.lm_step())
    T("solve%d sync" % s, lambda: aar.lib().aar_device_synchronize())
    T("solve%d get" % s, lambda: p.lm_get_solution())
print("EV", "; ".join(ev))
