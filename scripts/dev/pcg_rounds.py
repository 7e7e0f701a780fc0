"""Scratch (GPU): does k_pcgf's time follow the ROUNDS of whole frames per wavefront (2048 wavefronts: 4096 frames = 2 rounds, 5000 and 6144 = 3) rather than the frame count?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")]
import aar
for F in (3072, 4096, 4200, 5000, 6144, 6300):
    ds = aar.synth(5, num_frames=F)
    with aar.Problem(ds, solver="pcg") as p:
        for _ in range(2): p.lm_solve(ds.x_full)
        it0 = p.pcg_iterations()[1]
        p.set_kernel_profiling(True)
        n = 0
        for _ in range(3):
            x, rep = p.lm_solve(ds.x_full); n += rep["iterations"]
        kt = p.kernel_times()
        p.set_kernel_profiling(False)
        its = p.pcg_iterations()[1] - it0
        sec, cnt = kt["k_pcg"]
        print("F %5d: k_pcg %7.1f us per solve, %5.2f CG its per solve -> %5.1f us per (iteration + set-up share); per frame and iteration %.2f ns" % (F, 1e6 * sec / cnt, its / cnt, 1e6 * sec / (its + cnt), 1e9 * sec / (its + cnt) / F), flush=True)
