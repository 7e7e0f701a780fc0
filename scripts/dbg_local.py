import sys, threading
sys.path.insert(0, "automatic-ar_amd"); sys.path.insert(0, "tests")
import numpy as np, aar
from conftest import load_golden
ds, g = load_golden("g1_cfg2")
with aar.Problem(ds) as p:
    x1, rep1 = p.lm_solve(ds.x_full)
print("single: initial", rep1["initial_err"], [round(t["err"], 1) for t in rep1["trace"][:4]])
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
group = aar.LocalGroup(world)
out = [None] * world
def run(r):
    comm = aar.Comm.local(group, r, 0)
    with aar.Problem(ds, comm=comm) as p:
        x, rep = p.lm_solve(ds.x_full)
        out[r] = (x, rep, p.local_obs)
    comm.close()
th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in th]; [t.join() for t in th]
for r in range(world):
    x, rep, n = out[r]
    print("rank", r, "local obs", n, "initial", rep["initial_err"], [round(t["err"], 1) for t in rep["trace"][:4]], [round(t["mu"], 3) for t in rep["trace"][:3]])
print("single mu", [round(t["mu"], 3) for t in rep1["trace"][:3]])
