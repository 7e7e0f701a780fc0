"""Experiment (CPU, numpy; the oracle is test infrastructure, nothing here is product code): would the CG through the frame blocks (k_pcgf) keep the final POSES if its
operator read the W blocks in fp32 (half the bytes of a pass over W: the frame pass of an iteration is HBM-bound at config 5)?  The right-hand side, the preconditioner
and the back-substitution keep fp64 W; only  y = U p - W (V + mu)^-1 W^T p  sees the rounded copy.  Inexact LM by the reference's rules along its own trajectory, final
pose vector against the exact run's.
    python scripts/experiments/pcg_w_float.py [config ...]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "scripts", "experiments")]
import aar, oracle_lib as ol
import pcg_reduced_system as E


def schur_pcg(U, W, V, bs, bf, mu, eta, abs_tol, wf32, max_it=2000):
    ns, nf = U.shape[0], V.shape[0]
    Vi = np.linalg.inv(V + mu * np.eye(6))
    Wb = W.reshape(ns, nf, 6)
    Wo = Wb.astype(np.float32).astype(np.float64) if wf32 else Wb          # what the operator reads
    if wf32 == "all":                                                       # ... and the right-hand side, the preconditioner and the back-substitution too
        Wb = Wo
    rhs = bs - np.einsum("sfi,fij,fj->s", Wb, Vi, bf.reshape(nf, 6))
    apply = lambda p: U @ p + mu * p - np.einsum("sfi,fi->s", Wo, np.einsum("fij,fj->fi", Vi, np.einsum("sfi,s->fi", Wo, p)))
    S_diag = np.zeros((ns // 6, 6, 6))
    for a in range(ns // 6):
        Wa = Wb[6 * a: 6 * a + 6]
        S_diag[a] = U[6 * a: 6 * a + 6, 6 * a: 6 * a + 6] + mu * np.eye(6) - np.einsum("ifk,fkl,jfl->ij", Wa, Vi, Wa)
    Mi = np.linalg.inv(S_diag)
    prec = lambda r: np.einsum("aij,aj->ai", Mi, r.reshape(-1, 6)).reshape(-1)
    x = np.zeros(ns); r = rhs.copy(); z = prec(r); p = z.copy(); rz = r @ z; bb = rhs @ rhs; it = 0
    while it < max_it and (r @ r > eta * eta * bb or rz > abs_tol * abs_tol * mu):       # csrc/pcg_kernels.hip: both tests must hold
        Ap = apply(p); alpha = rz / (p @ Ap); x += alpha * p; r -= alpha * Ap
        z = prec(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; it += 1
    df = np.einsum("fij,fj->fi", Vi, bf.reshape(nf, 6) - np.einsum("sfi,s->fi", Wb, x)).reshape(-1)
    return np.concatenate([x, df]), it


def lm(o, x0, ns, eta, abs_tol=5e-5, wf32=False, max_steps=60):
    x = x0.copy(); z = o.extract_z(x)
    H, B = o.normal_equations(x, z=z, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F32)
    err = float(np.sum(o.residuals(x, z=z, res_mode=ol.RES_F32) ** 2))
    mu, v, prev = H.diagonal().max(), 2.0, err
    its, rows = [], 8.0 * o.N
    for step in range(max_steps):
        accepted = False
        for tries in range(6):
            if eta is None:
                d = np.linalg.solve(H + mu * np.eye(H.shape[0]), B)
            else:
                d, it = schur_pcg(*E.split(H, B, ns), mu, eta, abs_tol, wf32); its.append(it)
            zt = z + d
            et = float(np.sum(o.residuals(x, z=zt, res_mode=ol.RES_F32) ** 2))
            gain = (et - prev) / (0.5 * d @ (mu * d - B))
            if gain > 0 and et - prev < 0:
                mu *= max(0.33, 1 - (2 * gain - 1) ** 3); v = 2.0; z, err, accepted = zt, et, True
                break
            mu *= v; v *= 5
        if accepted:
            H, B = o.normal_equations(x, z=z, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F32)
        stop = abs(prev - err) / rows <= 1e-4 or not accepted or err > prev
        prev = err
        if stop:
            break
    return z, np.sqrt(err / (4.0 * o.N)), step + 1, its


for cfg in ([int(a) for a in sys.argv[1:]] or [3, 5]):
    ds = aar.synth(cfg) if cfg <= 3 else aar.synth(5, num_frames=150)
    o = ol.Oracle(ds); ns = 6 * (ds.num_cams - 1 + ds.num_markers - 1)
    t0 = time.time()
    z0, rm0, st0, _ = lm(o, ds.x_full, ns, None)
    print("config %d (%d cams / %d markers / %d frames): exact LM %d steps, RMSE %.9f px  [%.0f s]" % (cfg, ds.num_cams, ds.num_markers, ds.num_frames, st0, rm0, time.time() - t0), flush=True)
    for eta in (5e-3, 1e-3):
        for wf32 in (False, True, "all"):
            z, rm, st, its = lm(o, ds.x_full, ns, eta, wf32=wf32)
            print("   eta %-6g W in %s: %2d LM steps, CG %5.1f per solve, dRMSE %+.1e px, max |pose vector - exact run's| shared %.1e frames %.1e" % (
                eta, {False: "fp64", True: "fp32 (operator)", "all": "fp32 (everywhere)"}[wf32], st, np.mean(its), rm - rm0, np.abs(z - z0)[:ns].max(), np.abs(z - z0)[ns:].max()), flush=True)
