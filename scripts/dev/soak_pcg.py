"""GPU soak of the PCG kernels' barrier tree (grid_hop_tree): thousands of solves through k_pcgf (fp32 and fp64 blocks), the deterministic k_pcg, and the sharded kernels
behind in-process rank groups; every solve must land on the same LM step count and (to the atomics' noise) the same final error; no launch may time out."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")]
import aar
bad = 0
X = int(os.environ.get("SOAK_X", "1"))
for label, cfg, kw, env, reps in (("k_pcgf fp32, config 3", 3, {}, {}, 1500), ("k_pcgf fp64, config 3", 3, {}, {"AAR_PCG_W32": "0"}, 800), ("k_pcg deterministic, config 3", 3, dict(deterministic=True), {}, 500),
                                   ("k_pcgf fp32, config 4", 4, {}, {}, 400), ("k_pcgf fp32, config 5", 5, {}, {}, 40), ("k_pcg deterministic, config 5", 5, dict(deterministic=True), {}, 8)):
    os.environ.update(env)
    ds = aar.synth(cfg)
    t0 = time.time()
    with aar.Problem(ds, solver="pcg", **kw) as p:
        x, rep = p.lm_solve(ds.x_full)
        e0, n0, worst = rep["final_err"], rep["iterations"], 0.0
        for i in range(reps * X):
            x, rep = p.lm_solve(ds.x_full)
            dev = abs(rep["final_err"] - e0) / e0
            worst = max(worst, dev)
            if rep["iterations"] != n0 or not dev < 2e-5:
                bad += 1; print("  DEVIATION", label, i, rep["iterations"], n0, dev, flush=True)
        its = p.pcg_iterations()[1]
    for k in env: os.environ.pop(k, None)
    print("%-34s %5d solves x %d LM steps, %7d CG iterations: worst relative deviation of the final error %.1e  (%.1f s)" % (label, reps * X, n0, its, worst, time.time() - t0), flush=True)
# sharded kernels: two in-process ranks
ds = aar.synth(3)
grp = aar.LocalGroup(2)
res = [None, None]
def run(rank):
    comm = aar.Comm.local(grp, rank)
    with aar.Problem(ds, comm=comm, solver="pcg") as q:
        errs = []
        for i in range(150 * X):
            x, rep = q.lm_solve(ds.x_full); errs.append((rep["iterations"], rep["final_err"]))
        res[rank] = errs
    comm.close()
th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
t0 = time.time()
[t.start() for t in th]; [t.join() for t in th]
grp.close()
same = res[0] == res[1] and len(set(n for n, _ in res[0])) == 1
print("sharded PCG, two in-process ranks: %d solves each, identical on both ranks: %s  (%.1f s)" % (len(res[0]), same, time.time() - t0))
print("BAD", bad + (0 if same else 1))
