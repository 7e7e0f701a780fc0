"""Which solver AUTO should pick: DIRECT / SPCG / PCG at the library's default forcing terms over the number of shared entities (GPU box).

For every (cameras, markers) pair x 500 frames: LM it/s over repeated solves, CG iterations per LM step, LM steps, final RMSE and pose difference against the
direct solver -- on one rank and, with --comm, behind a single-rank RCCL communicator (the multi-GPU code path: fused all-reduce, speculative next solve).

    python scripts/dev/auto_crossover.py [--comm] > profiles/r05_auto_crossover.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")]
import aar  # noqa: E402
from pose_metrics import pose_delta_max  # noqa: E402

SHAPES = [(8, 40), (8, 56), (8, 72), (8, 88), (12, 100), (16, 112), (16, 144), (16, 200)]


def main():
    comm = None
    if "--comm" in sys.argv:
        comm = aar.Comm(aar.Comm.make_id(), 1, 0, 0)
    shapes = [(C, M, 500) for C, M in SHAPES]
    if "--frames" in sys.argv:      # the other axis: many shared entities, longer sequences (where CG through the frame blocks overtakes the explicit Schur complement)
        shapes = [(12, 100, 2000), (12, 100, 5000), (16, 144, 1500), (16, 144, 4000), (16, 200, 1000), (16, 200, 2000), (16, 200, 5000)]
    print("# LM it/s over repeated solves from the same start; %s" % ("single-rank RCCL communicator" if comm else "no communicator"))
    print("%5s %7s %8s %6s %8s | %-6s %9s %8s %9s %10s %10s" % ("cams", "markers", "entities", "frames", "slots", "solver", "LM it/s", "LM steps", "CG/step", "dRMSE", "dpose"))
    for C, M, frames in shapes:
        ds = aar.synth(3, num_cams=C, num_markers=M, num_frames=frames)
        import numpy as np
        slots = len(set(zip(ds.obs_frame.tolist(), ds.obs_cam.tolist()))) + len(set(zip(ds.obs_frame.tolist(), ds.obs_marker.tolist())))
        with aar.Problem(ds, comm=comm) as p:
            auto = p.solver_stats()["solver"]
        xd = rm_d = None
        for s in ("direct", "spcg", "pcg"):
            try:
                with aar.Problem(ds, solver=s, comm=comm) as p:
                    x, rep = p.lm_solve(ds.x_full)
                    rmse = p.reproj_stats(x)[0]
                    st0 = p.solver_stats()
                    reps = max(2, int(20 * 48 * 500 / ((C + M) * frames)))
                    aar.lib().aar_device_synchronize()
                    t0 = time.perf_counter()
                    n = 0
                    for _ in range(reps):
                        _, r = p.lm_solve(ds.x_full, trace_cap=1)
                        n += r["iterations"]
                    aar.lib().aar_device_synchronize()
                    dt = time.perf_counter() - t0
                    st = p.solver_stats()
                if s == "direct":
                    xd, rm_d = x, rmse
                print("%5d %7d %8d %6d %8d | %-6s %9.0f %8d %9.2f %10.1e %10.1e%s" % (C, M, C + M, frames, slots, s, n / dt, rep["iterations"], (st["total_iterations"] - st0["total_iterations"]) / max(1, n),
                                                                            rmse - rm_d, max(pose_delta_max(ds, x, xd)), "   <- AUTO" if s == auto else ""), flush=True)
            except aar.AarError as e:
                print("%5d %7d %8d %6d %8d | %-6s refused: %s" % (C, M, C + M, frames, slots, s, str(e)[:90]), flush=True)
    if comm:
        comm.close()


if __name__ == "__main__":
    main()
