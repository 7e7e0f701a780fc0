# Diagnostic: where does a short timed region (the driver's --steps 20 --warmup 5) lose milliseconds?
import sys, time, numpy as np
sys.path.insert(0, "automatic-ar_amd")
import aar
ds = aar.synth(3)
p = aar.Problem(ds, residual_mode=aar.RES_F32)
def solve(mi):
    t0 = time.perf_counter()
    x, rep = p.lm_solve(ds.x_full, params=aar.lm_default_params(max_iters=mi), trace_cap=1)
    return rep["iterations"], 1e3 * (time.perf_counter() - t0), 1e3 * rep["solve_seconds"]
out = [solve(5)]
aar.lib().aar_device_synchronize()
t0 = time.perf_counter()
out.append(solve(20)); out.append(solve(5))
aar.lib().aar_device_synchronize()
tot = 1e3 * (time.perf_counter() - t0)
print("OUT", " | ".join("%d it %.2f ms (inner %.2f)" % o for o in out), "timed %.2f ms" % tot)
