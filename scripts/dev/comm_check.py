import os, sys, json
sys.path[:0] = ["/root/repo/automatic-ar_amd", "/root/repo/tests"]
import aar
ds = aar.synth(3)
uid = aar.Comm.make_id()
comm = aar.Comm(uid, 1, 0, 0)
for solver in (None, "direct"):
    for comm_on in (False, True):
        with aar.Problem(ds, comm=comm if comm_on else None, solver=solver) as p:
            x, rep = p.lm_solve(ds.x_full)
            st0 = p.solver_stats()
            import time
            aar.lib().aar_device_synchronize(); t0 = time.perf_counter(); n = 0
            for _ in range(20):
                _, r = p.lm_solve(ds.x_full, trace_cap=1); n += r["iterations"]
            aar.lib().aar_device_synchronize(); dt = time.perf_counter() - t0
            st = p.solver_stats()
            print("solver %s comm %s: %d LM steps, %.1f us/step, CG solves per LM step %.2f, CG its per solve %.2f, per LM step %.2f, fallbacks %d" % (
                st["solver"], comm_on, rep["iterations"], 1e6 * dt / n, (st["solves"] - st0["solves"]) / n, (st["total_iterations"] - st0["total_iterations"]) / max(1, st["solves"] - st0["solves"]),
                (st["total_iterations"] - st0["total_iterations"]) / n, st["fallbacks"]), flush=True)
comm.close()
