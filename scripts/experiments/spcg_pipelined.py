"""Experiment (CPU, numpy; uses the oracle as test infrastructure, nothing here is product code):
the recurrences of csrc/spcg_kernels.hip written out in numpy -- block-Jacobi-preconditioned PIPELINED conjugate gradients
(Ghysels & Vanroose 2014, Alg. 3: one matrix-vector product and ONE global reduction per iteration, both dot products taken
on vectors that exist before the product) on the EXPLICIT reduced system  S = U + mu I - sum_f W_f (V_f + mu I)^-1 W_f^T --
next to the textbook PCG of pcg_reduced_system.py, inside the same inexact LM loop.  Question: does the pipelined form need the
same number of iterations and end the LM run at the same error?

    python scripts/experiments/spcg_pipelined.py [config ...]        (default: 2 3)
Output: profiles/r04_spcg_experiment.txt holds the committed run.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests"), os.path.dirname(os.path.abspath(__file__))]
import aar
import oracle_lib as ol
import pcg_reduced_system as base


def reduced(U, W, V, bs, bf, mu):
    ns, nf = U.shape[0], V.shape[0]
    Vi = np.linalg.inv(V + mu * np.eye(6))
    Wb = W.reshape(ns, nf, 6)
    S = U + mu * np.eye(ns) - np.einsum("sfi,fij,tfj->st", Wb, Vi, Wb)
    rhs = bs - np.einsum("sfi,fij,fj->s", Wb, Vi, bf.reshape(nf, 6))
    return S, rhs, Vi, Wb


def pipelined(S, b, eta, max_it=2000, crit="r"):
    ns = S.shape[0]
    Mi = np.linalg.inv(np.stack([S[6 * a: 6 * a + 6, 6 * a: 6 * a + 6] for a in range(ns // 6)]))
    prec = lambda r: np.einsum("aij,aj->ai", Mi, r.reshape(-1, 6)).reshape(-1)
    x = np.zeros(ns)
    r = b.copy()
    u = prec(r)
    w = S @ u
    bb = r @ r if crit == "r" else r @ u
    z = q = s = p = np.zeros(ns)
    g_old = a_old = 0.0
    it = 0
    while True:
        m = prec(w)
        gam, dlt, rho = r @ u, w @ u, r @ r            # ONE reduction: all three exist before the product
        if (rho if crit == "r" else gam) <= eta * eta * bb or it >= max_it:
            break
        n = S @ m
        beta = gam / g_old if it else 0.0
        alpha = gam / (dlt - beta * gam / a_old) if it else gam / dlt
        z = n + beta * z
        q = m + beta * q
        s = w + beta * s
        p = u + beta * p
        x = x + alpha * p
        r = r - alpha * s
        u = u - alpha * q
        w = w - alpha * z
        g_old, a_old = gam, alpha
        it += 1
    return x, it


def textbook(S, b, eta, max_it=2000):
    ns = S.shape[0]
    Mi = np.linalg.inv(np.stack([S[6 * a: 6 * a + 6, 6 * a: 6 * a + 6] for a in range(ns // 6)]))
    prec = lambda r: np.einsum("aij,aj->ai", Mi, r.reshape(-1, 6)).reshape(-1)
    x = np.zeros(ns)
    r = b.copy()
    z = prec(r)
    p = z.copy()
    rz = r @ z
    r0 = np.linalg.norm(b)
    it = 0
    while it < max_it and np.linalg.norm(r) > eta * r0:
        Ap = S @ p
        alpha = rz / (p @ Ap)
        x += alpha * p
        r -= alpha * Ap
        z = prec(r)
        rzn = r @ z
        p = z + (rzn / rz) * p
        rz = rzn
        it += 1
    return x, it


def lm(o, x0, ns, eta, solver, max_steps=60):
    x = x0.copy()
    z = o.extract_z(x)
    H, B = o.normal_equations(x, z=z, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F32)
    err = float(np.sum(o.residuals(x, z=z, res_mode=ol.RES_F32) ** 2))
    mu, v, prev = H.diagonal().max(), 2.0, err
    its, rows = [], 8.0 * o.N
    for step in range(max_steps):
        accepted = False
        for _ in range(6):
            U, W, V, bs, bf = base.split(H, B, ns)
            S, rhs, Vi, Wb = reduced(U, W, V, bs, bf, mu)
            if solver is None:
                xs = np.linalg.solve(S, rhs)
            else:
                xs, it = solver(S, rhs, eta)
                its.append(it)
            df = np.einsum("fij,fj->fi", Vi, bf.reshape(-1, 6) - np.einsum("sfi,s->fi", Wb, xs)).reshape(-1)
            d = np.concatenate([xs, df])
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
    return np.sqrt(err / (4.0 * o.N)), step + 1, its


def main():
    cfgs = [int(a) for a in sys.argv[1:]] or [2, 3]
    for cfg in cfgs:
        ds = aar.synth(cfg) if cfg <= 3 else aar.synth(5, num_frames=150)
        o = ol.Oracle(ds)
        ns = 6 * (ds.num_cams - 1 + ds.num_markers - 1)
        t0 = time.time()
        rm0, st0, _ = lm(o, ds.x_full, ns, None, None)
        print("config %d (n = %d): exact LM %d steps, RMSE %.9f px  [%.0f s]" % (cfg, ns, st0, rm0, time.time() - t0), flush=True)
        print("   eta     solver      LM steps   CG its / solve (mean, max)   per step                          |RMSE - exact| px")
        for eta in (1e-1, 1e-2):
            for name, fn in (("textbook", textbook), ("pipelined", pipelined), ("pipel. r'M^-1r", lambda S, b, eta: pipelined(S, b, eta, crit="g"))):
                rm, st, its = lm(o, ds.x_full, ns, eta, fn)
                print("   %-7g %-14s %5d      %8.1f %6d            %-40s %.2e" % (eta, name, st, np.mean(its), np.max(its), " ".join(map(str, its)), abs(rm - rm0)), flush=True)


if __name__ == "__main__":
    main()
