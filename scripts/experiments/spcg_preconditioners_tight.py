"""Experiment (CPU, numpy; the oracle is test infrastructure, nothing here is product code): CG iterations per damped solve on the EXPLICIT reduced system S + mu I
along the exact LM trajectory, at the TIGHT forcing terms the final poses need (round 5: DESIGN.md section 12), stopping rule in the preconditioner's norm as k_spcg
(r^T M^-1 r <= eta^2 b^T M^-1 b), for
    bj      block-Jacobi, the 6x6 diagonal blocks of S              (what csrc/spcg_kernels.hip builds)
    cam     the camera-camera block of S exactly + 6x6 marker blocks
    arrow   [[S_cc, S_cm], [S_mc, blockdiag(S_mm)]] solved exactly
    python scripts/experiments/spcg_preconditioners_tight.py [config] [eta ...]
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "scripts", "experiments")]
import aar, oracle_lib as ol
import pcg_reduced_system as E


def build(S, nc, mode):
    ns = S.shape[0]
    if mode == "bj":
        Mi = np.zeros((ns, ns))
        for a in range(ns // 6):
            sl = slice(6 * a, 6 * a + 6); Mi[sl, sl] = np.linalg.inv(S[sl, sl])
    elif mode == "cam":
        Mi = np.zeros((ns, ns)); Mi[:nc, :nc] = np.linalg.inv(S[:nc, :nc])
        for a in range(nc // 6, ns // 6):
            sl = slice(6 * a, 6 * a + 6); Mi[sl, sl] = np.linalg.inv(S[sl, sl])
    else:
        M = S.copy(); D = np.zeros((ns - nc, ns - nc))
        for a in range((ns - nc) // 6):
            sl = slice(6 * a, 6 * a + 6); D[sl, sl] = S[nc + 6 * a: nc + 6 * a + 6, nc + 6 * a: nc + 6 * a + 6]
        M[nc:, nc:] = D; Mi = np.linalg.inv(M)
    return Mi


def pcg(S, b, Mi, eta, max_it=400):
    x = np.zeros_like(b); r = b.copy(); z = Mi @ r; p = z.copy(); rz = r @ z; bb = rz; it = 0
    while it < max_it and rz > eta * eta * bb:
        Ap = S @ p; al = rz / (p @ Ap); x += al * p; r -= al * Ap; z = Mi @ r; rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; it += 1
    return x, it


cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
etas = [float(a) for a in sys.argv[2:]] or [0.02, 3e-4]
ds = aar.synth(cfg) if cfg <= 3 else aar.synth(cfg, num_frames=300)
o = ol.Oracle(ds); ns = 6 * (ds.num_cams - 1 + ds.num_markers - 1); nc = 6 * (ds.num_cams - 1)
x = ds.x_full; z = o.extract_z(x)
H, B = o.normal_equations(x, z=z, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F32)
err = float(np.sum(o.residuals(x, z=z, res_mode=ol.RES_F32) ** 2)); mu = H.diagonal().max(); v = 2.0; prev = err
tot = {}
for step in range(20):
    U, W, V, bs, bf = E.split(H, B, ns)
    nf = V.shape[0]
    Vi = np.linalg.inv(V + mu * np.eye(6)); Wb = W.reshape(ns, nf, 6)
    S = U + mu * np.eye(ns) - np.einsum("sfi,fij,tfj->st", Wb, Vi, Wb)
    rhs = bs - np.einsum("sfi,fij,fj->s", Wb, Vi, bf.reshape(nf, 6))
    xe = np.linalg.solve(S, rhs)
    line = "step %2d mu %.3e cond %.1e" % (step, mu, np.linalg.cond(S))
    for mode in ("bj", "cam", "arrow"):
        Mi = build(S, nc, mode)
        for eta in etas:
            xs, it = pcg(S, rhs, Mi, eta)
            tot[(mode, eta)] = tot.get((mode, eta), 0) + it
            line += "  %s@%g: %3d (err %.0e)" % (mode, eta, it, np.abs(xs - xe).max() / np.abs(xe).max())
    print(line, flush=True)
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
print("totals", {"%s@%g" % k: v for k, v in tot.items()})
