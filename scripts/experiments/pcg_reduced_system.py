"""Experiment (CPU, numpy; uses the oracle as test infrastructure, nothing here is product code):
how many iterations would a matrix-free PCG on the reduced system need, along a real LM run?

VERDICT r2 item 2 proposed the reduced camera/marker system solved by PCG THROUGH the frame blocks,
    y = (U + mu I) p - sum_f W_f (V_f + mu I)^-1 W_f^T p,
as the one formulation in which nothing O(n^3) is replicated across GPUs (the frame sum shards, one 8 n-byte all-reduce per
iteration, no S, no dense LDL^T).  Its cost is iterations x (one pass over the W blocks + one small all-reduce), so the iteration
count decides.  This script runs an inexact LM -- each damped system solved by block-Jacobi-preconditioned CG on the Schur operator to a
relative residual eta -- next to the exact LM on the same synthetic problems and reports, per eta: CG iterations per LM step
(mean / max), LM steps, and the final RMSE difference to the exact run (the north star's bar: 1e-4 px).

    python scripts/experiments/pcg_reduced_system.py [config ...]        (default: 2 3)
Output: a table per config; profiles/r03_pcg_experiment.txt holds the committed run.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")]
import aar
import oracle_lib as ol


def split(H, B, ns):
    """dense normal equations -> shared block U, coupling W (ns x nf), frame blocks V (6x6 each), gradients"""
    U = H[:ns, :ns]
    W = H[:ns, ns:]
    nf = (H.shape[0] - ns) // 6
    V = np.stack([H[ns + 6 * f: ns + 6 * f + 6, ns + 6 * f: ns + 6 * f + 6] for f in range(nf)])
    return U, W, V, B[:ns], B[ns:]


def schur_pcg(U, W, V, bs, bf, mu, eta, max_it=2000):
    ns, nf = U.shape[0], V.shape[0]
    Vi = np.linalg.inv(V + mu * np.eye(6))                       # per-frame 6x6 inverses (what k_frame_inv / pass A produce)
    Wb = W.reshape(ns, nf, 6)
    rhs = bs - np.einsum("sfi,fij,fj->s", Wb, Vi, bf.reshape(nf, 6))

    def apply(p):                                                 # the two passes over W of one PCG iteration
        t = np.einsum("fij,fj->fi", Vi, np.einsum("sfi,s->fi", Wb, p))
        return U @ p + mu * p - np.einsum("sfi,fi->s", Wb, t)

    # block-Jacobi preconditioner: the 6x6 diagonal blocks of the Schur complement (what a device version would keep per entity)
    S_diag = np.zeros((ns // 6, 6, 6))
    for a in range(ns // 6):
        Wa = Wb[6 * a: 6 * a + 6]                                 # 6 x nf x 6
        S_diag[a] = U[6 * a: 6 * a + 6, 6 * a: 6 * a + 6] + mu * np.eye(6) - np.einsum("ifk,fkl,jfl->ij", Wa, Vi, Wa)
    Mi = np.linalg.inv(S_diag)
    prec = lambda r: np.einsum("aij,aj->ai", Mi, r.reshape(-1, 6)).reshape(-1)
    x = np.zeros(ns)
    r = rhs.copy()
    z = prec(r)
    p = z.copy()
    rz = r @ z
    r0 = np.linalg.norm(rhs)
    it = 0
    while it < max_it and np.linalg.norm(r) > eta * r0:
        Ap = apply(p)
        alpha = rz / (p @ Ap)
        x += alpha * p
        r -= alpha * Ap
        z = prec(r)
        rz_new = r @ z
        p = z + (rz_new / rz) * p
        rz = rz_new
        it += 1
    df = np.einsum("fij,fj->fi", Vi, bf.reshape(nf, 6) - np.einsum("sfi,s->fi", Wb, x)).reshape(-1)
    return np.concatenate([x, df]), it


def lm(o, x0, ns, eta, max_steps=60):
    """the reference's LM rules (libs/sparselevmarq.h:349-472) with the damped system solved exactly (eta None) or by PCG"""
    x = x0.copy()
    z = o.extract_z(x)
    H, B = o.normal_equations(x, z=z, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F32)
    err = float(np.sum(o.residuals(x, z=z, res_mode=ol.RES_F32) ** 2))
    mu, v, prev = H.diagonal().max(), 2.0, err
    cg_its, rows = [], 8.0 * o.N
    for step in range(max_steps):
        accepted = False
        for tries in range(6):
            if eta is None:
                d = np.linalg.solve(H + mu * np.eye(H.shape[0]), B)
            else:
                d, it = schur_pcg(*split(H, B, ns), mu, eta)
                cg_its.append(it)
            zt = z + d
            et = float(np.sum(o.residuals(x, z=zt, res_mode=ol.RES_F32) ** 2))
            L = 0.5 * d @ (mu * d - B)
            gain = (et - prev) / L
            if gain > 0 and et - prev < 0:
                mu *= max(0.33, 1 - (2 * gain - 1) ** 3)
                v = 2.0
                z, err, accepted = zt, et, True
                break
            mu *= v
            v *= 5
        if accepted:
            H, B = o.normal_equations(x, z=z, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F32)
        stop = abs(prev - err) / rows <= 1e-4 or not accepted or err > prev
        prev = err
        if stop:
            break
    return np.sqrt(err / (4.0 * o.N)), step + 1, cg_its


def main():
    cfgs = [int(a) for a in sys.argv[1:]] or [2, 3]
    for cfg in cfgs:
        ds = aar.synth(cfg) if cfg <= 3 else aar.synth(5, num_frames=150)      # config 5's cameras / markers, a 150-frame cut (dense H must fit)
        o = ol.Oracle(ds)
        ns = 6 * (ds.num_cams - 1 + ds.num_markers - 1)
        t0 = time.time()
        rm0, st0, _ = lm(o, ds.x_full, ns, None)
        print("config %d (%d cams / %d markers / %d frames, reduced system n = %d): exact LM %d steps, RMSE %.9f px  [%.0f s]"
              % (cfg, ds.num_cams, ds.num_markers, ds.num_frames, ns, st0, rm0, time.time() - t0), flush=True)
        print("   eta      LM steps   CG its / solve (mean, max)   total CG its   |RMSE - exact| px")
        for eta in (1e-1, 1e-2, 1e-4, 1e-8):
            rm, st, its = lm(o, ds.x_full, ns, eta)
            print("   %-8g %5d      %8.1f %6d            %8d       %.2e" % (eta, st, np.mean(its), np.max(its), np.sum(its), abs(rm - rm0)), flush=True)


if __name__ == "__main__":
    main()
