"""Experiment (CPU, numpy; the oracle is test infrastructure, nothing here is product code): the coarse space of scripts/experiments/spcg_coarse.py in the
PCG through the frame blocks (csrc/pcg_kernels.hip, k_pcgf): textbook PCG, stopping rule |r| <= eta |b| and r^T M^-1 r <= eps^2 mu (eta 5e-3, eps 5e-5),
block-Jacobi against block-Jacobi + Z E^-1 Z^T (E = Z^T A Z: full 12 x 12, or its two diagonal 6 x 6 blocks), along an inexact LM run.
    python scripts/experiments/pcg_coarse.py [config] [frames]
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "scripts", "experiments")]
import aar, oracle_lib as ol
import pcg_reduced_system as E
import spcg_coarse as SC   # (runs its own experiment on import when executed as a script only)


def pcg(S, b, mu, eta, eps, Z, blockdiag, max_it=400):
    ns = S.shape[0]
    Mi = np.linalg.inv(np.stack([S[6 * a: 6 * a + 6, 6 * a: 6 * a + 6] for a in range(ns // 6)]))
    bj = lambda r: np.einsum("aij,aj->ai", Mi, r.reshape(-1, 6)).reshape(-1)
    if Z is not None:
        Em = Z.T @ S @ Z
        if blockdiag:
            for a in range(0, Em.shape[0], 6): Em[a:a + 6, :a] = 0; Em[a:a + 6, a + 6:] = 0
        Ei = np.linalg.inv(Em)
        prec = lambda r: bj(r) + Z @ (Ei @ (Z.T @ r))
    else:
        prec = bj
    x = np.zeros(ns); r = b.copy(); z = prec(r); p = z.copy(); rz = r @ z; bb = b @ b; it = 0
    while it < max_it and (r @ r > eta * eta * bb or rz > eps * eps * mu):
        Ap = S @ p; al = rz / (p @ Ap); x += al * p; r -= al * Ap; z = prec(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; it += 1
    return x, it


def lm(o, x0, ns, nc, mode, blockdiag=False, eta=5e-3, eps=5e-5, max_steps=60):
    x = x0.copy(); z = o.extract_z(x)
    H, B = o.normal_equations(x, z=z, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F32)
    err = float(np.sum(o.residuals(x, z=z, res_mode=ol.RES_F32) ** 2))
    mu, v, prev = H.diagonal().max(), 2.0, err
    its, rows = [], 8.0 * o.N
    for step in range(max_steps):
        accepted = False
        for _ in range(6):
            U, W, V, bs, bf = E.split(H, B, ns); nf = V.shape[0]
            Vi = np.linalg.inv(V + mu * np.eye(6)); Wb = W.reshape(ns, nf, 6)
            S = U + mu * np.eye(ns) - np.einsum("sfi,fij,tfj->st", Wb, Vi, Wb)
            rhs = bs - np.einsum("sfi,fij,fj->s", Wb, Vi, bf.reshape(nf, 6))
            if mode is None:
                xs = np.linalg.solve(S, rhs)
            else:
                xs, it = pcg(S, rhs, mu, eta, eps, SC.coarse(z, ns, nc, mode) if mode else None, blockdiag); its.append(it)
            df = np.einsum("fij,fj->fi", Vi, bf.reshape(-1, 6) - np.einsum("sfi,s->fi", Wb, xs)).reshape(-1)
            d = np.concatenate([xs, df]); zt = z + d
            et = float(np.sum(o.residuals(x, z=zt, res_mode=ol.RES_F32) ** 2))
            L = 0.5 * d @ (mu * d - B); gain = (et - prev) / L
            if gain > 0 and et - prev < 0:
                mu *= max(0.33, 1 - (2 * gain - 1) ** 3); v = 2.0; z, err, accepted = zt, et, True
                break
            mu *= v; v *= 5
        if accepted:
            H, B = o.normal_equations(x, z=z, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F32)
        stop = abs(prev - err) / rows <= 1e-4 or not accepted or err > prev
        prev = err
        if stop: break
    return np.sqrt(err / (4.0 * o.N)), step + 1, its, z


if __name__ == "__main__":
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    ds = aar.synth(cfg) if len(sys.argv) <= 2 else aar.synth(cfg, num_frames=int(sys.argv[2]))
    o = ol.Oracle(ds); nc = 6 * (ds.num_cams - 1); ns = nc + 6 * (ds.num_markers - 1)
    rm0, st0, _, z0 = lm(o, ds.x_full, ns, nc, None)
    print("config %d (%d frames, n = %d): exact LM %d steps, RMSE %.9f px" % (cfg, ds.num_frames, ns, st0, rm0), flush=True)
    for name, mode, bd in (("block-Jacobi", "", False), ("+ cm, E 12 x 12", "cm", False), ("+ cm, E block-diagonal", "cm", True)):
        rm, st, its, z = lm(o, ds.x_full, ns, nc, mode, bd)
        print("  %-26s LM steps %2d  CG its total %4d  per step %-50s |RMSE - exact| %.1e px  max |z - z_exact| shared %.1e frames %.1e" %
              (name, st, sum(its), " ".join(map(str, its)), abs(rm - rm0), np.abs(z - z0)[:ns].max(), np.abs(z - z0)[ns:].max()), flush=True)
