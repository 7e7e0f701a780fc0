"""Experiment (CPU, numpy; the oracle is test infrastructure, nothing here is product code): can information carried from the PREVIOUS LM steps shorten the
block-Jacobi CG on the explicit reduced system S + mu I (csrc/spcg_kernels.hip), at the same ABSOLUTE stopping threshold (r^T M^-1 r <= eta^2 b^T M^-1 b)?
    x0      cold start (what k_spcg does)
    warm1   x0 = alpha delta_prev, alpha = Galerkin
    warmk   x0 = Galerkin solution over the last k steps' delta_s
    defl    deflated CG over a recycled space: the last k deltas (+ optionally harmonic Ritz-ish: the CG search directions' span is not kept)
    python scripts/experiments/spcg_recycling.py [config] [eta]
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "scripts", "experiments")]
import aar, oracle_lib as ol
import pcg_reduced_system as E


def bj(S):
    ns = S.shape[0]; Mi = np.zeros((ns, ns))
    for a in range(ns // 6):
        sl = slice(6 * a, 6 * a + 6); Mi[sl, sl] = np.linalg.inv(S[sl, sl])
    return Mi


def pcg(S, b, Mi, eta, x0=None, max_it=400, P=None):
    """P: deflation basis (columns); deflated PCG (Saad et al. 2000)"""
    bb = b @ (Mi @ b)
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - S @ x
    if P is not None:
        AP = S @ P; G = np.linalg.inv(P.T @ AP)
        x = x + P @ (G @ (P.T @ r)); r = b - S @ x
    z = Mi @ r
    p = z.copy()
    if P is not None: p = p - P @ (G @ (AP.T @ z))
    rz = r @ z; it = 0
    while it < max_it and rz > eta * eta * bb:
        Ap = S @ p; al = rz / (p @ Ap); x += al * p; r -= al * Ap; z = Mi @ r; rzn = r @ z
        p = z + (rzn / rz) * p
        if P is not None: p = p - P @ (G @ (AP.T @ z))
        rz = rzn; it += 1
    return x, it


cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
eta = float(sys.argv[2]) if len(sys.argv) > 2 else 3e-4
ds = aar.synth(cfg) if cfg <= 3 else aar.synth(cfg, num_frames=300)
o = ol.Oracle(ds); ns = 6 * (ds.num_cams - 1 + ds.num_markers - 1)
x = ds.x_full; z = o.extract_z(x)
H, B = o.normal_equations(x, z=z, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F32)
err = float(np.sum(o.residuals(x, z=z, res_mode=ol.RES_F32) ** 2)); mu = H.diagonal().max(); v = 2.0; prev = err
tot = {}
hist = []
for step in range(20):
    U, W, V, bs, bf = E.split(H, B, ns)
    nf = V.shape[0]
    Vi = np.linalg.inv(V + mu * np.eye(6)); Wb = W.reshape(ns, nf, 6)
    S = U + mu * np.eye(ns) - np.einsum("sfi,fij,tfj->st", Wb, Vi, Wb)
    rhs = bs - np.einsum("sfi,fij,fj->s", Wb, Vi, bf.reshape(nf, 6))
    xe = np.linalg.solve(S, rhs)
    Mi = bj(S)
    line = "step %2d mu %.3e" % (step, mu)
    res = {}
    res["cold"] = pcg(S, rhs, Mi, eta)
    for k in (1, 2, 4, 8):
        if len(hist) >= 1:
            P = np.stack(hist[-k:], axis=1)
            Q, _ = np.linalg.qr(P)
            res["warm%d" % k] = pcg(S, rhs, Mi, eta, x0=Q @ np.linalg.solve(Q.T @ S @ Q, Q.T @ rhs))
            res["defl%d" % k] = pcg(S, rhs, Mi, eta, P=Q)
        else:
            res["warm%d" % k] = res["defl%d" % k] = res["cold"]
    for kname, (xs, it) in res.items():
        tot[kname] = tot.get(kname, 0) + it
        line += "  %s %3d" % (kname, it)
    print(line, flush=True)
    hist.append(xe.copy())
    d = np.linalg.solve(H + mu * np.eye(H.shape[0]), B)
    zt = z + d; et = float(np.sum(o.residuals(x, z=zt, res_mode=ol.RES_F32) ** 2))
    L = 0.5 * d @ (mu * d - B); gain = (et - prev) / L
    if gain > 0 and et < prev:
        mu *= max(0.33, 1 - (2 * gain - 1) ** 3); z = zt; err = et
        H, B = o.normal_equations(x, z=z, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F32)
    else:
        mu *= v; v *= 5
    if abs(prev - err) / (8.0 * o.N) <= 1e-4:
        break
    prev = err
print("totals", tot)
